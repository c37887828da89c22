// engine_impl.h — what the translation units of the engine share (not installed): the context and the plan, the locks and the
// handle registry, error macros, and the prototypes of the helpers that cross a file boundary.
//   engine.hip      contexts, options, plan creation / destruction, lane and gang planning
//   slice_loop.hip  one configuration: atoms of (k, j), potential, the six passes of the fused slice loop, hipGraph replay,
//                   incoming wave, exit-wave post-processing, detector chain
//   gangs.hip       gangs of configurations in one launch, lanes, progress
//   engine_abi.hip  the measurement entry points of include/fdes_abi.h, RCCL, taps and micro-benchmark hooks
#ifndef FDES_ENGINE_IMPL_H_
#define FDES_ENGINE_IMPL_H_
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <dlfcn.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "fdes_internal.h"
#include "fft.h"
#include "fft_lds.h"
#include "gen_jit.h"
#include "geometry.h"
#include "kernels.h"

using namespace fdes;


struct fdes_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int opt_fft = 0;   // 0 auto, 1 rocFFT, 2 hand-written
    int opt_graph = 1;    // replay the fused slice loop of a configuration as a hipGraph (one instantiated graph per empty-slice pattern)
    uint32_t seed = 1; // src/crystalMaker.cu:292
    bool is_lane_ctx = false;
    int bench_band = 0;   // fdes_bench_pass only
    int bench_alt = -1;   // fdes_bench_pass only: >= 0: odd streams run pass (alt / 10000, alt / 100 % 100, alt % 100) instead
    int bench_tall = 1;   // fdes_bench_pass only: rows = bench_tall * n (emulates a batch of configurations in one launch)
    int bench_pitch = 0;  // fdes_bench_pass only: rows of every scratch grid are padded by this many elements
    int bench_serial = 0; // fdes_bench_pass only: the `streams` buffer sets are used round-robin on ONE stream (a footprint beyond the Infinity Cache without concurrency)
    float2* share_PT = nullptr; // lane contexts: tables owned by the parent plan (PT: separable propagator, px[m1] | py[m2])
    float* share_GT = nullptr;
    int band_skip = 1;    // do not move / transform the rows and columns the 2/3 band limit zeroes anyway
    int skip_empty = 1;   // slices without atoms: t = 1, only the Fresnel step is applied (fused loop)
    int lanes_active = 0; // > 0: run_config only deals to the first n lanes (bench: time a kernel without a co-running lane)
    int lanes = 0;        // configurations in flight at once (own stream + buffers each) in the fused slice loop; 0: by grid size
    int pass_threads = 0; // 0 auto: 256-thread pass workgroups (two per CU) when lanes > 1 and the grid allows, else 512
    int split = -1;       // potential / transmission passes (P1'..P4) on a stream of their own, one slice pair ahead of the
                          // wave's passes (P5, P6): concurrency inside ONE configuration; -1 auto, 0 off, 1 on (issued
                          // directly, never captured into a graph)
    int batch = -1;       // slice pairs per launch of the potential chain of a one-lane plan: -1 auto (by grid size), 0 / 1 off, 2 ... 8
    int gang = -1;        // configurations of one measurement whose slice loops run in lockstep on one lane, every pass ONE launch
                          // (grid z = configuration): -1 auto (by grid size), 0 / 1 off, 2 ... 16
    int stagger = 0;      // one-wave-per-row passes: start delay between the waves of a CU, in units of 64 cycles (0: none)
    int walk = 1;         // every pass is launched in this many parts (2: a part takes half of the workgroup slots, two lanes' passes share every CU)
    int pitch_pad = -1;   // elements added to every row of the fused loop's grids; -1 auto: 32 for 2048-point rows, 64 from 4096 on
    int deterministic = 1; // the deposit of the generic (rocFFT) path and of the potential output adds the atoms in sorted order through LDS (bit-reproducible); 0: global float atomics as the reference
    int jit = -1;         // mixed-radix grid lengths without compiled-in kernels get theirs compiled by hipRTC at plan creation (gen_jit.cpp): -1 = unless FDES_JIT=0, 0 off, 1 on
    int peer_copy = 1;    // 0: fdes_plan_accumulate_from stages partial sums through host memory instead of a peer copy (the fallback path, forced)
    int probe_stride = 0; // > 0: bracket every probe_stride-th 2-D FFT with HIP events (bench roofline)
    int probe_pass = 5;   // fused loop: the pass class that is bracketed (1 = P1' ... 6 = P6; bench.py's per-pass table)
    // plans are expensive to create: one per grid size AND requested back-end (option "fft" may change between plans)
    std::map<std::tuple<int, int, int>, Fft2D*> fft_cache; // (third entry: option fft + 4 * (run-time compilation asked for))
    // plans created on this context and not yet destroyed: fdes_destroy takes them down first, so a host that forgets
    // fdes_plan_destroy (or a Python finaliser that runs late) cannot leave a plan pointing at a dead context
    std::vector<fdes_plan*> plans;
    // progress report (the reference prints a percentage from inside its slice loop, src/crystalMaker.cu:341 ->
    // src/optimFunctions.cu:257): called on the host between configurations, never from a captured graph
    fdes_progress_fn progress = nullptr;
    void* progress_user = nullptr;
    int progress_min_ms = 200;
    std::chrono::steady_clock::time_point progress_last{};
    int64_t progress_total = 0; // slice-propagations of the whole job (0: unknown)
    int64_t progress_done = 0;
};

struct EvPair { hipEvent_t a, b; int slices; int configs = 1; };

struct fdes_plan {
    fdes_ctx* ctx = nullptr;
    fdes_params p0{};  // as given (before sub-slicing), own arrays
    fdes_params p{};   // sub-sliced, shares p0's arrays
    int ratio = 1;
    KP kp{};
    int nAt = 0, nZ = 0;
    int Zlist[103];
    std::vector<Kirk> kz;
    // atoms
    int32_t* Z_d = nullptr;
    uint8_t* spec_d = nullptr;
    float *xyz0_d = nullptr, *xyzTO_d = nullptr, *xyzK_d = nullptr, *xyzFP_d = nullptr, *dwf_d = nullptr, *occ_d = nullptr;
    int cur_k = -1;
    AtomBins bins;
    int bins_cap_keys = 0;
    int deposit_blocks = 1;
    // grids
    size_t m12 = 0;
    float2 *D = nullptr, *VH = nullptr, *T = nullptr, *PSI = nullptr, *P = nullptr, *I = nullptr, *EW = nullptr;
    float* J = nullptr;
    float* scal = nullptr;
    Fft2D* fft = nullptr; // owned by the context's cache
    // fused LDS-pass slice loop (power-of-two grids): spectra in transposed ("T", [kx][y|ky]) and mixed
    // ("N", [y][kx]) layouts, tables in T layout
    bool fused = false;
    int wg = 512;                       // threads per pass workgroup
    std::vector<fdes_plan*> lanes;      // extra lanes (own context/stream/buffers); this plan is lane 0
    std::vector<fdes_ctx*> lane_ctx;
    std::vector<hipEvent_t> lane_ev;
    bool is_lane = false;
    fdes_ctx* parent_ctx = nullptr;    // lanes follow the runtime options (probe_stride) of the context that owns the plan
    fdes_plan* top = nullptr;          // lanes: the plan they belong to
    // skip_empty bookkeeping of a (top-level) plan: configurations in a row in which no slice was empty, configurations
    // seen, questions asked.  A dense specimen (a crystal that fills the box) never has an empty slice: after kDenseAfter
    // such configurations the per-configuration question (one D2H of the segment table and one host wait on the lane's
    // stream) is only asked every kDenseRecheck-th configuration; meanwhile every slice takes the full sequence, which is
    // always correct.
    int dense_streak = 0;
    int64_t cfg_seen = 0, empty_queries = 0;
    unsigned rr = 0;                    // round-robin lane selector
    bool lanes_dirty = false;           // lanes hold partial sums not yet folded into lane 0
    std::vector<int> seg_h;             // per-slice occupancy of the current configuration as a monotone table [m3 * nZ + 1] (slice q is empty iff
                                        // seg_h[(q + 1) nZ] == seg_h[q nZ]); empty: not asked
    // "which slices are empty" is answered on a stream of its own (empty_query): the host never waits for a lane's slice loops
    hipStream_t qs = nullptr;
    int* slice_occ_d = nullptr;         // [gang][m3] occupancy flags
    int* slice_occ_h = nullptr;         // pinned host copy
    int64_t slices_skipped = 0;
    float2 *A = nullptr, *B = nullptr, *C = nullptr, *C2 = nullptr, *E = nullptr, *F = nullptr, *PSIH = nullptr, *PT = nullptr;
    float* GT = nullptr;
    bool tables_shared = false; // PT / GT belong to the parent plan (lanes)
    // Row pitches of the fused loop's grids (elements): "N" grids [y][kx] have m2 rows of pitchN >= m1, "T" grids [kx][y|ky]
    // have m1 rows of pitchT >= m2.  A transposed store writes one short segment into each of several thousand rows: with
    // rows a power of two apart these segments pile up on a few memory channels once the working set leaves the
    // Infinity Cache (4096^2, two streams: 95.6 us for a transposing copy against 60-67 us with 64 elements of padding).
    int pitchN = 0, pitchT = 0;
    size_t gsz = 0;             // elements of one fused grid (either layout)
    // P^n tables (separable: m1 + m2 complex numbers) for runs of n empty slices (skip_empty): built on first use on this
    // plan's stream, least recently used of 16 replaced
    struct PowTab { int n; float2* tab; uint64_t used; };
    std::vector<PowTab> pow_tabs;
    uint64_t pow_tick = 0;
    // hipGraph replay of the fused slice loop (option "graph"): the launch sequence of a configuration depends only on
    // the number of slices and on which slices are empty, so an instantiated graph is kept per such pattern
    struct LoopGraph { uint64_t key; std::vector<uint8_t> pattern; hipGraphExec_t exec; int64_t skipped; uint64_t used; std::vector<std::pair<int, float2*>> pow; };
    std::vector<std::pair<int, float2*>>* capture_pow = nullptr; // P^n tables of the graph being captured (built by its own nodes)
    std::vector<LoopGraph> graphs;
    uint64_t graph_tick = 0;
    bool capturing = false;
    std::vector<EvPair> probe;
    size_t probe_used = 0;
    uint64_t fft_calls = 0;
    bool want_ew = false;
    // split slice loop: the potential chain runs on `vs`, the wave chain on the context's stream
    bool split = false, tap_mode = false;
    // the incoming wave of the current configuration is band-limited in kx (set by incoming_wave): every case except a
    // CBED probe with a beam tilt, whose phase ramp comes after the band limit (src/multisliceSimulation.cu:583-590)
    bool wave_bl = true;
    hipStream_t vs = nullptr;
    float2* Eb[2] = {nullptr, nullptr};  // band-limited transmission spectra of the pair's two slices (split: two buffers)
    hipEvent_t evE[2] = {nullptr, nullptr}, evP5[2] = {nullptr, nullptr}, evFork = nullptr, evJoin = nullptr;
    bool p5_seen[2] = {false, false};
    // batched potential chain (one-lane plans up to 2^20 pixels): `nb` slice pairs per launch of the potential /
    // transmission passes (grid.z), their band-limited transmission spectra in two sets of 2 nb grids that the wave chain
    // consumes one batch behind
    int nb = 1;
    float2 *bA = nullptr, *bB = nullptr, *bCC = nullptr, *bE[2] = {nullptr, nullptr};
    hipEvent_t evReady[2] = {nullptr, nullptr}, evDone[2] = {nullptr, nullptr};
    // gang of configurations: run_config only queues; `gang` queued configurations of one measurement are
    // then issued together - atoms and incoming wave per member, ONE slice loop whose passes carry the members as grid z.
    // The buffers the passes touch hold `gang` members back to back (member 0 = the plan's own pointers).
    int gang = 1;                         // members (1: off)
    int gn = 1;                           // members of the gang being issued (pass launches: nbatch)
    struct GangCfg { int k, j; float w; int slot; };
    // gangs ACROSS measurements (a tilt / defocus series without frozen phonons has ONE configuration per measurement):
    // the members then belong to different k - own incoming wave, own tilt - and add into intensity slots of their own
    // (I holds `gang` slots back to back); only fdes_build_measurements drives it, the plan API stays one k at a time
    bool one_shot_few = false;  // fdes_build_measurements: this plan lives for one job that replays its slice loop fewer than four
                                // times per lane - capturing and instantiating a graph (2.3 ms for 600 nodes) costs more than it saves
    bool gang_k = false;
    std::vector<std::pair<int, int>> gfinal; // (k, slot) whose detector chain waits for the members of k to be issued
    float* Jout = nullptr;                   // where finished images go: this plan's J, or the top plan's (lanes)
    std::vector<GangCfg> gq;              // queued configurations (all of one measurement k)
    std::vector<AtomBins> gbins;          // member views of the binning buffers
    std::vector<std::vector<int>> gseg;   // members' (slice, species) segment tables (skip_empty)
    float* gxyzFP = nullptr;              // [gang][3 nAt] jittered coordinates (member 0 = xyzFP_d)
    float2* gscr = nullptr;               // [gang][m12] scratch of the members' 2-D transforms outside the slice loop
    std::vector<void*> gang_owned;        // per-member binning arrays of members >= 1
    size_t recs_stride = 0, rowstart_stride = 0, seg_stride = 0;
    float2* peer_stage = nullptr;   // landing buffer for another GPU's partial sum (fdes_plan_accumulate_from; receive buffer of fdes_plan_reduce_intensity)
    float* real_send = nullptr;     // real view of this plan's intensity sum, packed for the way to another GPU (16 MiB instead of 32 at 2048^2)
    bool peer_host_only = false;    // the peer copy was refused once: partial sums are staged through host memory (option "peer_copy" 0 forces it)
    std::vector<float2> peer_host;
    hipEvent_t peer_ev = nullptr;
    float* span_stage = nullptr; size_t span_stage_n = 0; // fdes_plan_reduce_intensity_span: landing zone of the root (peers side by side)
    float* span_send = nullptr; size_t span_send_n = 0;   // ... and what a peer sends: float view of I [+ EW]
    // timing
    std::vector<EvPair> evs;
    size_t ev_used = 0;
    size_t ev_done = 0; // events of one stream complete in order: evs[i].b has been seen complete for i < ev_done
    int64_t cfg_done = 0; // configurations behind those events (a gang's pair of events stands for all its members)
    int64_t slices_done = 0;
};

namespace fdes_engine {


extern std::once_flag g_rocfft_once;
// Host threads driving different GPUs (or several plans on one GPU) share the process.  A stream capture begun in the
// default (global / relaxed) mode is invalidated by synchronising runtime calls (hipMalloc, hipFree, blocking hipMemcpy) made
// by ANY thread meanwhile ("operation failed due to a previous error during capture"); until round 4 one process-wide mutex
// therefore serialised every plan creation / destruction with every capture, so that eight workers set their plans up one
// after the other.  Round 5: the slice loop is captured in hipStreamCaptureModeThreadLocal - calls of OTHER threads, on this
// or another device, do not touch the capture; the capturing thread's own hipMalloc (the table of a power of the propagator
// that a run of empty slices needs) exchanges the mode for the duration of that call - and what is left is one lock PER
// DEVICE around the capture itself and around the allocations a thread makes while another thread of the same device may be
// capturing (FDES_CAPTURE_LOCK=0 drops even that: the test of the capture mode, tests/test_gpu_r5.py).
struct DeviceLocks {
    std::recursive_mutex m[65];
    bool enabled = true;
    DeviceLocks() { const char* e = std::getenv("FDES_CAPTURE_LOCK"); enabled = !(e && e[0] == '0'); }
};
extern DeviceLocks g_dev_locks;
struct DeviceGuard { // lock of one device (index 64: devices beyond 63)
    std::unique_lock<std::recursive_mutex> lk;
    explicit DeviceGuard(int device) { if (g_dev_locks.enabled) lk = std::unique_lock<std::recursive_mutex>(g_dev_locks.m[(device >= 0 && device < 64) ? device : 64]); }
};
// a synchronising allocation made by a thread that may itself be capturing (thread-local capture mode forbids it otherwise)
struct RelaxCapture {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    RelaxCapture() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
    ~RelaxCapture() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
};

// Live handles.  Every entry point that destroys checks its handle here first, so destroying twice, destroying a plan
// after its context, or a finaliser that runs after the at-exit sweep below are refused (FDES_EINVAL) instead of
// touching freed memory.
extern std::mutex g_live_mutex;
extern std::set<const void*> g_live_ctx, g_live_plan;
bool live_ctx(const fdes_ctx* c);
bool live_plan(const fdes_plan* p);
extern std::once_flag g_atexit_once;
void shutdown_all();
// Destroyed handles are not handed back to the allocator at once: a stale handle (a late finaliser, a host bug) whose
// address the allocator had given to a NEW context or plan would pass the registry check and hit the wrong object.  The
// emptied shells (a few hundred bytes each; every GPU resource and vector is released before) wait in a graveyard of
// 1024 entries, so an address is reused only after 1024 later destructions.
template <class T> void bury(T* obj)
{
    static std::mutex m;
    static std::vector<T*> graveyard;
    static size_t next = 0;
    std::lock_guard<std::mutex> g(m);
    if (graveyard.size() < 1024) { graveyard.push_back(obj); return; }
    delete graveyard[next];
    graveyard[next] = obj;
    next = (next + 1) % graveyard.size();
}

// run-time options of a lane are those of the context that owns the plan (the lane contexts are private copies made at
// plan creation: only what shapes the allocation - fft, lanes, pass_threads - is frozen there)
inline const fdes_ctx* owner_ctx(const fdes_plan* pl) { return pl->parent_ctx ? pl->parent_ctx : pl->ctx; }

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_) + " @" + __FILE__ + ":" + std::to_string(__LINE__); \
            return FDES_EGPU;                                                                     \
        }                                                                                         \
    } while (0)
#define RC(expr)                \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != FDES_OK) return rc_; \
    } while (0)


template <class T> int dmalloc(fdes_ctx* c, T** p, size_t n)
{
    HIPCHK(c, hipMalloc((void**)p, sizeof(T) * (n > 0 ? n : 1)));
    return FDES_OK;
}

hipError_t fft_exec(fdes_plan* pl, float2* data, bool inverse, hipStream_t st);
KP make_kp(const fdes_params& p);
int tilt_coordinates(fdes_plan* pl, float* xyz, float t_0, float t_1, float t_2);
int ensure_tilt(fdes_plan* pl, int k);
int empty_query(fdes_plan* pl, int n, const int* ks, const int* js);
int config_atoms(fdes_plan* pl, int k, int j, bool query = true, float* xyz = nullptr, AtomBins* bins_p = nullptr);
int bandwidth_limit(fdes_plan* pl, float2* f);
int phase_grating(fdes_plan* pl, const float* xyz, const BinGeom& g, int s);
int phase_grating_pair(fdes_plan* pl, const float* xyz, const BinGeom& g, int s0);
int forward_propagation(fdes_plan* pl, int comp = -1);
void gang_strides(const fdes_plan* pl, PassArgs& a, size_t in0, size_t out, size_t out2 = 0, size_t in1 = 0);
int probe_bracket(fdes_plan* pl, PassArgs& a, int cls);
int fused_potential_pair(fdes_plan* pl, int s0);
int propagator_pow(fdes_plan* pl, int n, float2** out);
int fused_empty_run(fdes_plan* pl, int s, int nslices, int* consumed);
int fused_wave_step(fdes_plan* pl, int s, const float2* E, int ei);
int fused_slice(fdes_plan* pl, int s, int nslices, int* consumed);
int batched_loop(fdes_plan* pl, int nslices);
int split_fork(fdes_plan* pl);
int split_join(fdes_plan* pl);
int fused_enter(fdes_plan* pl);
int fused_leave(fdes_plan* pl, bool propagated);
int incoming_wave(fdes_plan* pl, int k, float2* psi = nullptr);
int slice_loop(fdes_plan* pl, int nslices);
int exit_wave_post(fdes_plan* pl, int k, float weight, float2* psi = nullptr, float2* acc = nullptr);
int finalize_measurement(fdes_plan* pl, int k, float2* acc = nullptr);
int fft_gang(fdes_plan* pl, float2* data, int n, size_t stride, bool inverse);
int bandwidth_limit_gang(fdes_plan* pl, float2* f, int n);
int incoming_wave_gang(fdes_plan* pl, int n);
int exit_wave_post_gang(fdes_plan* pl, int n);
int finalize_gang(fdes_plan* pl);
int gang_flush(fdes_plan* pl);
int gang_flush_all(fdes_plan* pl);
int fold_lanes(fdes_plan* pl);
int64_t configs_finished(fdes_plan* pl, hipEvent_t* oldest_pending);
void report_progress(fdes_plan* pl, int64_t issued, int64_t total_configs, bool final);
bool gang_one_lane(const fdes_ctx* c, const fdes_plan* pl);
int plan_gang(const fdes_ctx* c, const fdes_plan* pl, bool* across_k = nullptr);
int plan_lanes(const fdes_ctx* c, const fdes_plan* pl);
int check_params(fdes_ctx* ctx, const fdes_params* p, const fdes_atoms* a);
int create_ctx(fdes_ctx** out, int gpu_index, int prio_class);
PassArgs pass_x(fdes_plan* pl);
PassArgs pass_y(fdes_plan* pl);
hipStream_t vstream(fdes_plan* pl);

} // namespace fdes_engine
using namespace fdes_engine;

#endif
