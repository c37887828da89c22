// engine.hip — host driver of the MI355X FDES engine: context, plan, slice loop, C-ABI.
//
// Restates the control flow of buildMeasurements / phaseGrating / forwardPropagation /
// incomingWave / applyLensFunction / diffractionPattern / addNoiseAndMtf
// (src/crystalMaker.cu:227-424, 507-536, 579-613, 700-718; src/multisliceSimulation.cu:538-622)
// on one HIP stream, with all scratch allocated once per plan (the reference cudaMalloc/cudaFree's
// two grids per slice, :516-517,534-535, and prints to stderr inside the slice loop, :341).
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
    int peer_copy = 1;    // 0: fdes_plan_accumulate_from stages partial sums through host memory instead of a peer copy (the fallback path, forced)
    int probe_stride = 0; // > 0: bracket every probe_stride-th 2-D FFT with HIP events (bench roofline)
    int probe_pass = 5;   // fused loop: the pass class that is bracketed (1 = P1' ... 6 = P6; bench.py's per-pass table)
    // plans are expensive to create: one per grid size AND requested back-end (option "fft" may change between plans)
    std::map<std::tuple<int, int, int>, Fft2D*> fft_cache;
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

namespace {

std::once_flag g_rocfft_once;
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
DeviceLocks g_dev_locks;
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
std::mutex g_live_mutex;
std::set<const void*> g_live_ctx, g_live_plan;
bool live_ctx(const fdes_ctx* c) { if (!c) return false; std::lock_guard<std::mutex> g(g_live_mutex); return g_live_ctx.count(c) != 0; }
bool live_plan(const fdes_plan* p) { if (!p) return false; std::lock_guard<std::mutex> g(g_live_mutex); return g_live_plan.count(p) != 0; }
std::once_flag g_atexit_once;
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
const fdes_ctx* owner_ctx(const fdes_plan* pl) { return pl->parent_ctx ? pl->parent_ctx : pl->ctx; }

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

// 2-D FFT of one grid, optionally bracketed by events (sampled) for the roofline measurement.
hipError_t fft_exec(fdes_plan* pl, float2* data, bool inverse, hipStream_t st)
{
    fdes_ctx* c = pl->ctx;
    const bool probe = c->probe_stride > 0 && (pl->fft_calls++ % (uint64_t)c->probe_stride) == 0;
    EvPair* ev = nullptr;
    if (probe) {
        if (pl->probe_used == pl->probe.size()) {
            EvPair e{};
            hipError_t r = hipEventCreate(&e.a);
            if (r != hipSuccess) return r;
            r = hipEventCreate(&e.b);
            if (r != hipSuccess) return r;
            pl->probe.push_back(e);
        }
        ev = &pl->probe[pl->probe_used++];
        hipError_t r = hipEventRecord(ev->a, st);
        if (r != hipSuccess) return r;
    }
    hipError_t r = pl->fft->exec(data, inverse, st);
    if (r != hipSuccess) return r;
    if (ev) r = hipEventRecord(ev->b, st);
    return r;
}

KP make_kp(const fdes_params& p)
{
    KP k{};
    k.m1 = p.m1; k.m2 = p.m2; k.m3 = p.m3; k.n1 = p.n1; k.n2 = p.n2; k.dn1 = p.dn1; k.dn2 = p.dn2; k.mode = p.mode;
    k.d1 = p.d1; k.d2 = p.d2; k.d3 = p.d3; k.lambda = p.lambda; k.sigma = p.sigma; k.imPot = p.imPot;
    k.defocspread = p.defocspread; k.illangle = p.illangle; k.mtfa = p.mtfa; k.mtfb = p.mtfb; k.mtfc = p.mtfc;
    k.mtfd = p.mtfd; k.ObjAp = p.ObjAp; k.ab = p.ab;
    return k;
}

template <class T> int dmalloc(fdes_ctx* c, T** p, size_t n)
{
    HIPCHK(c, hipMalloc((void**)p, sizeof(T) * (n > 0 ? n : 1)));
    return FDES_OK;
}

// tiltCoordinates, src/crystalMaker.cu:427-454.  cos/sin on the host, as the reference.
int tilt_coordinates(fdes_plan* pl, float* xyz, float t_0, float t_1, float t_2)
{
    hipStream_t st = pl->ctx->stream;
    if (fabsf(t_2) > FLT_EPSILON) HIPCHK(pl->ctx, geom_srot(xyz, pl->nAt, 0, 1, cosf(t_2), -sinf(t_2), st));
    if (fabsf(t_1) > FLT_EPSILON) HIPCHK(pl->ctx, geom_srot(xyz, pl->nAt, 0, 2, cosf(t_1), -sinf(t_1), st));
    if (fabsf(t_0) > FLT_EPSILON) HIPCHK(pl->ctx, geom_srot(xyz, pl->nAt, 1, 2, cosf(t_0), -sinf(t_0), st));
    return FDES_OK;
}

// src/crystalMaker.cu:330-331
int ensure_tilt(fdes_plan* pl, int k)
{
    if (pl->cur_k == k) return FDES_OK;
    HIPCHK(pl->ctx, hipMemcpyAsync(pl->xyzK_d, pl->xyzTO_d, sizeof(float) * 3 * (size_t)pl->nAt, hipMemcpyDeviceToDevice, pl->ctx->stream));
    RC(tilt_coordinates(pl, pl->xyzK_d, pl->p.tiltspec[2 * k], pl->p.tiltspec[2 * k + 1], 0.f));
    pl->cur_k = k;
    return FDES_OK;
}

// Option skip_empty: which slices of the configurations (ks[g], js[g]), g < n, hold atoms -> pl->seg_h (slice q counts as
// occupied when it is occupied in ANY of them: a gang skips a slice only when it is empty in every member).  Asked on the
// plan's query stream from the constant tilt-offset coordinates (geom_slice_occupancy recomputes tilt, jitter and the
// binning's slice test), so the one host wait per question covers a few microseconds of work of its own and NOT the slice
// loops queued on the lane's stream (until round 3 the question read the binning's segment table behind them).
// pl->gseg[g] receives member g's own table when n > 1.
int empty_query(fdes_plan* pl, int n, const int* ks, const int* js)
{
    fdes_ctx* c = pl->ctx;
    const int m3 = pl->p.m3, nZ = pl->nZ;
    const int cap = pl->gang > 1 ? pl->gang : 1;
    if (n > cap) return FDES_EINVAL;
    if (!pl->qs) {
        DeviceGuard guard(c->device); // stream creation vs a capture in another thread of this device
        int least = 0, greatest = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(c, hipStreamCreateWithPriority(&pl->qs, hipStreamNonBlocking, greatest)); // a queue of its own, ahead of the lanes' kernels
        RC(dmalloc(c, &pl->slice_occ_d, (size_t)cap * m3));
        HIPCHK(c, hipHostMalloc((void**)&pl->slice_occ_h, sizeof(int) * (size_t)cap * m3, hipHostMallocDefault));
    }
    BinGeom g{pl->p.m1, pl->p.m2, m3, nZ, pl->p.d1, pl->p.d2, pl->p.d3};
    for (int q = 0; q < n; q++)
        HIPCHK(c, geom_slice_occupancy(pl->slice_occ_d + (size_t)q * m3, pl->xyzTO_d, pl->dwf_d, pl->nAt, g, pl->p.tiltspec[2 * ks[q]], pl->p.tiltspec[2 * ks[q] + 1],
                                       pl->p.frPh > 0, owner_ctx(pl)->seed, ks[q], js[q], pl->qs));
    HIPCHK(c, hipMemcpyAsync(pl->slice_occ_h, pl->slice_occ_d, sizeof(int) * (size_t)n * m3, hipMemcpyDeviceToHost, pl->qs));
    HIPCHK(c, hipStreamSynchronize(pl->qs));
    const size_t len = (size_t)m3 * nZ + 1;
    auto table = [&](std::vector<int>& t, auto occupied) { // monotone, one step per occupied slice
        t.assign(len, 0);
        int cum = 0;
        for (int q = 0; q < m3; q++) {
            for (int z = 0; z < nZ; z++) t[(size_t)q * nZ + z] = cum;
            cum += occupied(q) ? 1 : 0;
        }
        t[len - 1] = cum;
    };
    if (n > 1)
        for (int q = 0; q < n; q++) {
            if (pl->gseg.size() < (size_t)n) pl->gseg.resize((size_t)n);
            table(pl->gseg[(size_t)q], [&](int s) { return pl->slice_occ_h[(size_t)q * m3 + s] != 0; });
        }
    table(pl->seg_h, [&](int s) { for (int q = 0; q < n; q++) if (pl->slice_occ_h[(size_t)q * m3 + s]) return true; return false; });
    return FDES_OK;
}

// src/crystalMaker.cu:335-337 + the per-configuration (slice, species) binning
int config_atoms(fdes_plan* pl, int k, int j, bool query = true, float* xyz = nullptr, AtomBins* bins_p = nullptr) // xyz / bins_p: a gang member's coordinates and binning buffers
{
    float* const xyzFP = xyz ? xyz : pl->xyzFP_d;
    AtomBins& bins = bins_p ? *bins_p : pl->bins;
    fdes_ctx* c = pl->ctx;
    RC(ensure_tilt(pl, k));
    if (pl->p.frPh > 0)
        HIPCHK(c, geom_jitter(xyzFP, pl->xyzK_d, pl->dwf_d, pl->nAt, owner_ctx(pl)->seed, k, j, c->stream));
    else
        HIPCHK(c, hipMemcpyAsync(xyzFP, pl->xyzK_d, sizeof(float) * 3 * (size_t)pl->nAt, hipMemcpyDeviceToDevice, c->stream));
    BinGeom g{pl->p.m1, pl->p.m2, pl->p.m3, pl->nZ, pl->p.d1, pl->p.d2, pl->p.d3};
    HIPCHK(c, geom_bin_atoms(xyzFP, pl->spec_d, pl->occ_d, pl->nAt, g, bins, pl->fused || owner_ctx(pl)->deterministic, c->stream));
    if (!query) return FDES_OK; // (a gang asks once for all its members, gang_flush)
    fdes_plan* tp = pl->top ? pl->top : pl;
    constexpr int kDenseAfter = 8, kDenseRecheck = 64;
    bool ask = pl->fused && owner_ctx(pl)->skip_empty;
    if (ask && tp->dense_streak >= kDenseAfter && (tp->cfg_seen % kDenseRecheck) != 0) ask = false;
    tp->cfg_seen++;
    if (ask) {
        // which slices hold atoms decides the launch sequence
        const int kk = k, jj = j;
        RC(empty_query(pl, 1, &kk, &jj));
        tp->empty_queries++;
        bool any_empty = false;
        for (int q = 0; q < pl->p.m3 && !any_empty; q++) any_empty = pl->seg_h[(size_t)(q + 1) * pl->nZ] == pl->seg_h[(size_t)q * pl->nZ];
        tp->dense_streak = any_empty ? 0 : tp->dense_streak + 1;
    } else {
        pl->seg_h.clear();
    }
    return FDES_OK;
}

// bandwidthLimit, src/multisliceSimulation.cu:552-560
int bandwidth_limit(fdes_plan* pl, float2* f)
{
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, fft_exec(pl,f, false, c->stream));
    HIPCHK(c, k_mask_scale(f, pl->p.m1, pl->p.m2, 1.f / ((float)pl->m12), c->stream));
    HIPCHK(c, fft_exec(pl,f, true, c->stream));
    return FDES_OK;
}

// phaseGrating, src/crystalMaker.cu:507-536 -> VH (real space potential of sub-slice s)
int phase_grating(fdes_plan* pl, const float* xyz, const BinGeom& g, int s)
{
    fdes_ctx* c = pl->ctx;
    KP kp = pl->kp;
    kp.m3 = g.m3;
    kp.d3 = g.d3;
    for (int z = 0; z < pl->nZ; z++) {
        if (owner_ctx(pl)->deterministic && geom_deposit_tile_fits(g.m1)) HIPCHK(c, geom_deposit_tile(pl->D, pl->bins, s * pl->nZ + z, -1, true, pl->p.imPot, g, c->stream));
        else
        HIPCHK(c, geom_deposit(pl->D, xyz, pl->occ_d, pl->bins, s * pl->nZ + z, g, pl->p.imPot, pl->deposit_blocks, c->stream));
        HIPCHK(c, fft_exec(pl,pl->D, false, c->stream));
        HIPCHK(c, k_filter_accum(pl->VH, pl->D, kp, pl->kz[z], z == 0, c->stream));
    }
    HIPCHK(c, fft_exec(pl,pl->VH, true, c->stream));
    return FDES_OK;
}

// Packed potential of the slice pair (s0, s0 + 1) on the generic path: W = V_s0 + i V_(s0+1) in VH from one forward
// transform per species and one inverse transform for two slices (the deposits are real, the filter is real and
// even), with the filter read from the per-plan table instead of being re-evaluated (3 expf + 2 sinf per pixel).
int phase_grating_pair(fdes_plan* pl, const float* xyz, const BinGeom& g, int s0)
{
    fdes_ctx* c = pl->ctx;
    for (int z = 0; z < pl->nZ; z++) {
        const int k0 = s0 * pl->nZ + z, k1 = (s0 + 1 < g.m3) ? (s0 + 1) * pl->nZ + z : -1;
        if (owner_ctx(pl)->deterministic && geom_deposit_tile_fits(g.m1)) HIPCHK(c, geom_deposit_tile(pl->D, pl->bins, k0, k1, false, 0.f, g, c->stream));
        else
        HIPCHK(c, geom_deposit_pair(pl->D, xyz, pl->occ_d, pl->bins, k0, k1, g, pl->deposit_blocks, c->stream));
        HIPCHK(c, fft_exec(pl, pl->D, false, c->stream));
        HIPCHK(c, k_filter_accum_tab(pl->VH, pl->D, pl->GT + (size_t)z * pl->m12, pl->m12, z == 0, c->stream));
    }
    HIPCHK(c, fft_exec(pl, pl->VH, true, c->stream));
    return FDES_OK;
}

// forwardPropagation, src/multisliceSimulation.cu:538-549 (V in VH; comp >= 0: component of the packed pair potential)
int forward_propagation(fdes_plan* pl, int comp = -1)
{
    fdes_ctx* c = pl->ctx;
    if (comp >= 0) HIPCHK(c, k_transmit_comp(pl->T, pl->VH, pl->m12, comp, pl->p.imPot, c->stream));
    else HIPCHK(c, k_transmit(pl->T, pl->VH, pl->m12, c->stream));
    RC(bandwidth_limit(pl, pl->T));
    HIPCHK(c, k_mul(pl->PSI, pl->T, pl->PSI, pl->m12, c->stream));      // multiplyElementwise(t, psi)
    HIPCHK(c, fft_exec(pl,pl->PSI, false, c->stream));                  // convolveWithFrProp
    HIPCHK(c, k_mul(pl->PSI, pl->PSI, pl->P, pl->m12, c->stream));
    HIPCHK(c, fft_exec(pl,pl->PSI, true, c->stream));
    return FDES_OK;
}

// ---- fused slice loop: six LDS row passes per slice, no stand-alone point-wise kernel.
//   P1  D[y][x]        -FFT_x->                          A_z[kx][y]      (per species; clears D)
//   P2  A_z[kx][y]     -FFT_y, * G_z, sum_z, IFFT_y->    B[y][kx]
//   P3  B[y][kx]       -IFFT_x, exp(iV), FFT_x->         C[kx][y]
//   P4  C[kx][y]       -FFT_y, band limit / m12, IFFT_y-> E[y][kx]
//   P5  E, PSIH[y][kx] -IFFT_x both, t * psi, FFT_x->    F[kx][y]
//   P6  F[kx][y]       -FFT_y, * P, IFFT_y->             PSIH[y][kx]
// pass over the rows of an "N" grid ([y][kx], row length m1) with a transposed store into a "T" grid, and the reverse;
// callers whose input or output is a dense natural grid (PSI, T, user buffers) override the pitch with 0
PassArgs pass_x(fdes_plan* pl) { PassArgs a; a.tw0 = pl->fft->tw0x; a.tw1 = pl->fft->tw1x; a.nrows = pl->p.m2; a.wg = pl->wg; a.pitch_in = pl->pitchN; a.pitch_out = pl->pitchT; a.walk = owner_ctx(pl)->walk; a.stagger = owner_ctx(pl)->stagger; return a; }
PassArgs pass_y(fdes_plan* pl) { PassArgs a; a.tw0 = pl->fft->tw0y; a.tw1 = pl->fft->tw1y; a.nrows = pl->p.m1; a.wg = pl->wg; a.pitch_in = pl->pitchT; a.pitch_out = pl->pitchN; a.walk = owner_ctx(pl)->walk; a.stagger = owner_ctx(pl)->stagger; return a; }

// stream of the potential / transmission passes
hipStream_t vstream(fdes_plan* pl) { return (pl->split && !pl->tap_mode) ? pl->vs : pl->ctx->stream; }

// gang (fdes_plan::gn members in one launch, grid z = member): element strides of the operands between members
void gang_strides(const fdes_plan* pl, PassArgs& a, size_t in0, size_t out, size_t out2 = 0, size_t in1 = 0)
{
    if (pl->gn <= 1) return;
    a.nbatch = pl->gn;
    a.bstride_in0 = in0; a.bstride_out = out; a.bstride_out2 = out2; a.bstride_in1 = in1;
}

// roofline probe (bench.py): every probe_stride-th launch of the pass class the context's probe_pass names (1 = P1', 2 = P2,
// 3 = P3, 4 = P4, 5 = P5, 6 = P6) is bracketed by the start / stop events of the dispatch itself (hipExtLaunchKernelGGL)
int probe_bracket(fdes_plan* pl, PassArgs& a, int cls)
{
    fdes_ctx* c = pl->ctx;
    const fdes_ctx* oc = owner_ctx(pl);
    if (pl->capturing || oc->probe_stride <= 0 || oc->probe_pass != cls) return FDES_OK;
    if ((pl->fft_calls++ % (uint64_t)oc->probe_stride) != 0) return FDES_OK;
    if (pl->probe_used == pl->probe.size()) {
        EvPair e{};
        HIPCHK(c, hipEventCreate(&e.a));
        HIPCHK(c, hipEventCreate(&e.b));
        pl->probe.push_back(e);
    }
    EvPair* ev = &pl->probe[pl->probe_used++];
    a.ev_start = ev->a;
    a.ev_stop = ev->b;
    return FDES_OK;
}

// Potential of the slice PAIR (s0, s0 + 1), s0 even: W = V_s0 + i V_(s0+1) (the deposits are real and the filter
// G is real and even, so one complex transform carries two slices).  P1' builds the x-spectra of the deposit rows
// straight from the sorted atom records (no deposit grid), P2 applies the filter in (kx, ky) and sums the species.
int fused_potential_pair(fdes_plan* pl, int s0)
{
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2;
    {   // one launch, grid.y = species
        PassArgs a = pass_x(pl);
        a.out = pl->A;
        a.nspecies = pl->nZ; a.species_stride = pl->gsz;
        a.recs = pl->bins.recs_sorted; a.rowstart = pl->bins.rowstart;
        a.q0 = s0 * pl->nZ;
        a.q1 = (s0 + 1 < pl->p.m3) ? (s0 + 1) * pl->nZ : -1;
        gang_strides(pl, a, 0, pl->gsz * (size_t)pl->nZ);
        if (pl->gn > 1) { // every member deposits the same slice pair from its own records
            a.bstride_recs = pl->recs_stride; a.bstride_rowstart = pl->rowstart_stride;
            for (int g = 0; g < pl->gn; g++) { a.zq0[g] = a.q0; a.zq1[g] = a.q1; }
        }
        RC(probe_bracket(pl, a, 1));
        HIPCHK(c, lds_pass(m1, XF_FWD, MID_ATOMS, XF_NONE, true, a, vstream(pl)));
    }
    PassArgs b = pass_y(pl);
    b.in0 = pl->A; b.gtab = pl->GT; b.out = pl->B; b.nspecies = pl->nZ; b.species_stride = pl->gsz;
    gang_strides(pl, b, pl->gsz * (size_t)pl->nZ, pl->gsz);
    RC(probe_bracket(pl, b, 2));
    HIPCHK(c, lds_pass(m2, XF_FWD, pl->nZ == 1 ? MID_GTAB : MID_GTABN, XF_INV, true, b, vstream(pl)));
    return FDES_OK;
}

// separable table of P^n (n >= 2) for a run of empty slices: px^n[m1] | py^n[m2]
int propagator_pow(fdes_plan* pl, int n, float2** out)
{
    fdes_ctx* c = pl->ctx;
    const size_t len = (size_t)pl->p.m1 + (size_t)pl->p.m2;
    if (pl->capture_pow) {
        // inside a graph capture the table belongs to that graph: its build kernel is one of the nodes, so every replay
        // refreshes it and nothing depends on what other patterns did to a shared cache in between
        for (auto& e : *pl->capture_pow)
            if (e.first == n) { *out = e.second; return FDES_OK; }
        float2* t = nullptr;
        {
            RelaxCapture relax; // hipMalloc from the capturing thread
            RC(dmalloc(c, &t, len));
        }
        pl->capture_pow->push_back({n, t});
        HIPCHK(c, k_build_propagator_1d(t, t + pl->p.m1, pl->kp, n, c->stream));
        *out = t;
        return FDES_OK;
    }
    for (auto& e : pl->pow_tabs)
        if (e.n == n) { e.used = ++pl->pow_tick; *out = e.tab; return FDES_OK; }
    DeviceGuard guard(c->device); // hipMalloc vs a capture in another thread of this device
    float2* tab = nullptr;
    if (pl->pow_tabs.size() >= 16) {
        size_t lru = 0;
        for (size_t i = 1; i < pl->pow_tabs.size(); i++) if (pl->pow_tabs[i].used < pl->pow_tabs[lru].used) lru = i;
        tab = pl->pow_tabs[lru].tab; // stream order makes the overwrite safe: its last reader was enqueued earlier
        pl->pow_tabs.erase(pl->pow_tabs.begin() + (long)lru);
    } else {
        RC(dmalloc(c, &tab, len));
    }
    HIPCHK(c, k_build_propagator_1d(tab, tab + pl->p.m1, pl->kp, n, c->stream));
    pl->pow_tabs.push_back({n, tab, ++pl->pow_tick});
    *out = tab;
    return FDES_OK;
}

// a run of slices without atoms, starting at s: V = 0, t = BL(1) = 1; psi <- F^-1[P^n F[psi]] as one Fresnel step.
// PSIH is [y][kx]: a transposing copy gives the y-pass its rows, then the usual propagator pass.
int fused_empty_run(fdes_plan* pl, int s, int nslices, int* consumed)
{
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2;
    const int md = m1 < m2 ? m1 : m2, band = md * md;
    const int bs = (owner_ctx(pl)->band_skip && m1 == m2) ? 1 : 0;
    auto empty = [&](int q) { return q >= pl->p.m3 || pl->seg_h[(size_t)(q + 1) * pl->nZ] == pl->seg_h[(size_t)q * pl->nZ]; };
    PassArgs a5 = pass_x(pl);
    a5.in0 = pl->PSIH; a5.out = pl->F;
    a5.scale = (float)m1; // P5 hands m1 * FFT_x(t psi) to P6 (unnormalised x round trip); exact power of two
    // (an incoming wave that is not band-limited in kx needs no special case here: the dead kx rows this copy drops are
    // zeroed by the masked propagator whatever they held)
    a5.band = band; a5.skip_dead_loads = bs; a5.skip_dead_stores = bs;
    gang_strides(pl, a5, pl->gsz, pl->gsz);
    HIPCHK(c, lds_pass(m1, XF_NONE, MID_SCALE, XF_NONE, true, a5, c->stream));
    int run = 1;
    while (s + run < nslices && empty(s + run)) run++;
    float2* ptab = pl->PT;
    if (run > 1) RC(propagator_pow(pl, run, &ptab));
    PassArgs a6 = pass_y(pl);
    a6.in0 = pl->F; a6.prow = ptab; a6.pcol = ptab + m1; a6.mindim = md; a6.out = pl->PSIH;
    a6.band = band; a6.live_rows_only = bs;
    gang_strides(pl, a6, pl->gsz, pl->gsz);
    HIPCHK(c, lds_pass(m2, XF_FWD, MID_PTAB, XF_INV, true, a6, c->stream));
    pl->slices_skipped += (int64_t)run * pl->gn;
    *consumed = run;
    return FDES_OK;
}

// the wave's two passes of slice s: P5 (t psi from the band-limited transmission spectrum E and psi-hat) and P6
// (Fresnel propagator); ei >= 0: the split loop's "E consumed" event of that buffer is recorded behind P5
int fused_wave_step(fdes_plan* pl, int s, const float2* E, int ei)
{
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2;
    const int md = m1 < m2 ? m1 : m2, band = md * md;
    const int bs = (owner_ctx(pl)->band_skip && m1 == m2) ? 1 : 0;
    PassArgs a5 = pass_x(pl);
    a5.in0 = E; a5.in1 = pl->PSIH; a5.out = pl->F;
    // Dead kx columns: E's are never written by P4 (they may hold the pair potential's stale values: B aliases E), so
    // they are always skipped; psi-hat's are exact zeros after any masked propagator, but the FIRST product of a
    // configuration sees the incoming wave, which the reference multiplies by t in full (src/multisliceSimulation.cu:546)
    // and which is not band-limited in kx when a tilted CBED probe leaves the band (:583-590) - then all of it is read.
    a5.band = band; a5.skip_dead_loads = bs ? ((s == 0 && !pl->wave_bl) ? 1 : 3) : 0; a5.skip_dead_stores = bs;
    gang_strides(pl, a5, pl->gsz, pl->gsz, 0, pl->gsz);
    RC(probe_bracket(pl, a5, 5));
    HIPCHK(c, lds_pass(m1, XF_INV, MID_MULPSI, XF_FWD, true, a5, c->stream));
    if (ei >= 0) {
        HIPCHK(c, hipEventRecord(pl->evP5[ei], c->stream));
        pl->p5_seen[ei] = true;
    }
    PassArgs a6 = pass_y(pl);
    a6.in0 = pl->F; a6.prow = pl->PT; a6.pcol = pl->PT + m1; a6.mindim = md; a6.out = pl->PSIH;
    a6.band = band; a6.live_rows_only = bs;
    gang_strides(pl, a6, pl->gsz, pl->gsz);
    RC(probe_bracket(pl, a6, 6));
    HIPCHK(c, lds_pass(m2, XF_FWD, MID_PTAB, XF_INV, true, a6, c->stream));
    return FDES_OK;
}

// one slice of the fused loop; *consumed = slices advanced (a run of empty slices is one Fresnel step with P^n)
int fused_slice(fdes_plan* pl, int s, int nslices, int* consumed)
{
    *consumed = 1;
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2;
    // 2/3 band limit: rows/columns whose own frequency index already fails 9 i^2 <= mindim^2 are exact zeros after
    // P4 (mask) and P6 (masked propagator): P4/P6 run only their live row groups, P3/P5 do not store the rows those
    // never read, P5 does not load the columns they never write (pre-zeroed at plan creation).
    const int md = m1 < m2 ? m1 : m2, band = md * md;
    const int bs = (owner_ctx(pl)->band_skip && m1 == m2) ? 1 : 0; // the column classes of the passes assume the band of a square grid
    auto empty = [&](int q) { return q >= pl->p.m3 || pl->seg_h[(size_t)(q + 1) * pl->nZ] == pl->seg_h[(size_t)q * pl->nZ]; };
    const bool have_seg = !pl->seg_h.empty();
    if (have_seg && empty(s)) return fused_empty_run(pl, s, nslices, consumed);
    // the pair's potential and both transmission functions are built at the pair's first non-empty slice:
    // C <- F_x[t_s0], C2 <- F_x[t_(s0+1)] from one read and one inverse transform of W = V_s0 + i V_(s0+1)
    if ((s & 1) == 0 || (have_seg && empty(s - 1))) {
        RC(fused_potential_pair(pl, s & ~1));
        PassArgs a3 = pass_x(pl);
        a3.in0 = pl->B; a3.out = pl->C; a3.out2 = pl->C2; a3.scale = pl->p.imPot;
        a3.band = band; a3.skip_dead_stores = bs;
        gang_strides(pl, a3, pl->gsz, pl->gsz, pl->gsz);
        RC(probe_bracket(pl, a3, 3));
        HIPCHK(c, lds_pass(m1, XF_INV, MID_EXPIV_PAIR, XF_FWD, true, a3, vstream(pl)));
    }
    const bool split = pl->split && !pl->tap_mode;
    const int ei = s & 1;
    PassArgs a4 = pass_y(pl);
    a4.in0 = (s & 1) ? pl->C2 : pl->C; a4.out = pl->Eb[ei]; a4.scale = 1.f / ((float)pl->m12); a4.mindim = md;
    a4.band = band; a4.live_rows_only = bs;
    gang_strides(pl, a4, pl->gsz, pl->gsz);
    if (split && pl->p5_seen[ei]) HIPCHK(c, hipStreamWaitEvent(pl->vs, pl->evP5[ei], 0)); // the wave chain has consumed this buffer
    RC(probe_bracket(pl, a4, 4));
    HIPCHK(c, lds_pass(m2, XF_FWD, MID_MASK, XF_INV, true, a4, vstream(pl)));
    if (split) {
        HIPCHK(c, hipEventRecord(pl->evE[ei], pl->vs));
        HIPCHK(c, hipStreamWaitEvent(c->stream, pl->evE[ei], 0));
    }
    return fused_wave_step(pl, s, pl->Eb[ei], split ? ei : -1);
}

// The slices [0, nslices) of one configuration with the potential chain in BATCHES (plans with nb > 1): the potential does
// not depend on the wave (src/crystalMaker.cu:339-343: phaseGrating takes the atoms and the slice index only), so P1',
// P2, P3 of nb slice pairs and P4 of their 2 nb slices are one launch each (grid.z = pair resp. slice) on the potential
// stream - a single slice's rows cannot fill the chip at 1024^2 and below, and a single image has no second
// configuration to run beside it - while the wave stream runs P5 / P6 of the previous batch.  Empty slices take no
// part in the batch (skip_empty); the wave chain handles their runs as fused_slice does.
int batched_loop(fdes_plan* pl, int nslices)
{
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2, nZ = pl->nZ, nb = pl->nb;
    const int md = m1 < m2 ? m1 : m2, band = md * md;
    const int bs = (owner_ctx(pl)->band_skip && m1 == m2) ? 1 : 0;
    const bool have_seg = !pl->seg_h.empty();
    auto empty = [&](int q) { return q >= pl->p.m3 || pl->seg_h[(size_t)(q + 1) * nZ] == pl->seg_h[(size_t)q * nZ]; };
    bool used[2] = {false, false};
    int sw = 0; // next slice of the wave chain
    int set = 0;
    for (int s0 = 0; s0 < nslices; s0 += 2 * nb, set ^= 1) {
        const int s1 = (s0 + 2 * nb < nslices) ? s0 + 2 * nb : nslices;
        // ---- potential chain of the batch
        int np = 0, ns = 0, eidx[16], zq0[16], zq1[16], zin[16];
        for (int i = 0; i < 16; i++) eidx[i] = -1;
        for (int sp = s0; sp < s1; sp += 2) {
            const bool e0 = have_seg && empty(sp), e1 = (sp + 1 >= s1) || (have_seg && empty(sp + 1));
            if (e0 && e1) continue;
            zq0[np] = sp * nZ;
            zq1[np] = (sp + 1 < pl->p.m3) ? (sp + 1) * nZ : -1;
            if (!e0) { zin[ns] = 2 * np; eidx[sp - s0] = ns++; }
            if (!e1) { zin[ns] = 2 * np + 1; eidx[sp + 1 - s0] = ns++; }
            np++;
        }
        if (used[set]) HIPCHK(c, hipStreamWaitEvent(pl->vs, pl->evDone[set], 0)); // the wave chain has consumed this set
        if (np > 0) {
            PassArgs a1 = pass_x(pl); // P1': records -> x spectra, [pair][species] grids
            a1.out = pl->bA; a1.nspecies = nZ; a1.species_stride = pl->gsz;
            a1.recs = pl->bins.recs_sorted; a1.rowstart = pl->bins.rowstart;
            a1.q0 = zq0[0]; a1.q1 = zq1[0];
            a1.nbatch = np; a1.bstride_out = pl->gsz * (size_t)nZ;
            for (int i = 0; i < np; i++) { a1.zq0[i] = zq0[i]; a1.zq1[i] = zq1[i]; }
            HIPCHK(c, lds_pass(m1, XF_FWD, MID_ATOMS, XF_NONE, true, a1, pl->vs));
            PassArgs a2 = pass_y(pl); // P2: filter, species sum -> packed pair potentials
            a2.in0 = pl->bA; a2.gtab = pl->GT; a2.out = pl->bB; a2.nspecies = nZ; a2.species_stride = pl->gsz;
            a2.nbatch = np; a2.bstride_in0 = pl->gsz * (size_t)nZ; a2.bstride_out = pl->gsz;
            HIPCHK(c, lds_pass(m2, XF_FWD, nZ == 1 ? MID_GTAB : MID_GTABN, XF_INV, true, a2, pl->vs));
            PassArgs a3 = pass_x(pl); // P3: both transmission functions of every pair -> bCC[2 pair], bCC[2 pair + 1]
            a3.in0 = pl->bB; a3.out = pl->bCC; a3.out2 = pl->bCC + pl->gsz; a3.scale = pl->p.imPot;
            a3.band = band; a3.skip_dead_stores = bs;
            a3.nbatch = np; a3.bstride_in0 = pl->gsz; a3.bstride_out = 2 * pl->gsz; a3.bstride_out2 = 2 * pl->gsz;
            HIPCHK(c, lds_pass(m1, XF_INV, MID_EXPIV_PAIR, XF_FWD, true, a3, pl->vs));
        }
        if (ns > 0) {
            PassArgs a4 = pass_y(pl); // P4: band limit of the non-empty slices' transmission functions -> bE[set][slice]
            a4.in0 = pl->bCC; a4.out = pl->bE[set]; a4.scale = 1.f / ((float)pl->m12); a4.mindim = md;
            a4.band = band; a4.live_rows_only = bs;
            a4.nbatch = ns; a4.use_zin = 1; a4.bstride_in0 = pl->gsz; a4.bstride_out = pl->gsz;
            for (int i = 0; i < ns; i++) a4.zin[i] = zin[i];
            if (ns == 1) a4.in0 = pl->bCC + (size_t)zin[0] * pl->gsz; // a batch of one is launched without the batch offsets
            HIPCHK(c, lds_pass(m2, XF_FWD, MID_MASK, XF_INV, true, a4, pl->vs));
        }
        HIPCHK(c, hipEventRecord(pl->evReady[set], pl->vs));
        HIPCHK(c, hipStreamWaitEvent(c->stream, pl->evReady[set], 0));
        // ---- wave chain of the batch
        int s = sw > s0 ? sw : s0;
        while (s < s1) {
            if (have_seg && empty(s)) {
                int run = 1;
                RC(fused_empty_run(pl, s, nslices, &run));
                s += run;
                continue;
            }
            RC(fused_wave_step(pl, s, pl->bE[set] + (size_t)eidx[s - s0] * pl->gsz, -1));
            s++;
        }
        sw = s;
        HIPCHK(c, hipEventRecord(pl->evDone[set], c->stream));
        used[set] = true;
    }
    return FDES_OK;
}

// fork / join of the potential stream around the slices of one configuration (also inside a stream capture, where the
// event edges become graph dependencies)
int split_fork(fdes_plan* pl)
{
    if (!(pl->split && !pl->tap_mode)) return FDES_OK;
    fdes_ctx* c = pl->ctx;
    pl->p5_seen[0] = pl->p5_seen[1] = false;
    HIPCHK(c, hipEventRecord(pl->evFork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(pl->vs, pl->evFork, 0));
    return FDES_OK;
}
int split_join(fdes_plan* pl)
{
    if (!(pl->split && !pl->tap_mode)) return FDES_OK;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipEventRecord(pl->evJoin, pl->vs));
    HIPCHK(c, hipStreamWaitEvent(c->stream, pl->evJoin, 0));
    return FDES_OK;
}

// real-space wave <-> mixed (y, kx) representation the fused loop carries between slices
int fused_enter(fdes_plan* pl)
{
    PassArgs a = pass_x(pl);
    a.in0 = pl->PSI; a.out = pl->PSIH;
    a.pitch_in = 0; a.pitch_out = pl->pitchN; // dense real-space wave -> padded mixed representation, natural store
    gang_strides(pl, a, pl->m12, pl->gsz);
    HIPCHK(pl->ctx, lds_pass(pl->p.m1, XF_FWD, MID_NONE, XF_NONE, false, a, pl->ctx->stream));
    return FDES_OK;
}
int fused_leave(fdes_plan* pl, bool propagated)
{
    PassArgs a = pass_x(pl);
    a.in0 = pl->PSIH; a.out = pl->PSI;
    if (owner_ctx(pl)->band_skip && pl->p.m1 == pl->p.m2 && propagated) { // the dead columns were last written by fused_enter: they count as zero
        const int md = pl->p.m1 < pl->p.m2 ? pl->p.m1 : pl->p.m2;
        a.band = md * md;
        a.skip_dead_loads = 1;
    }
    a.pitch_out = 0; // dense
    a.scale = 1.f / (float)pl->p.m1; // PSIH = FFT_x(psi), unnormalised transforms (m1 is a power of two: exact)
    gang_strides(pl, a, pl->gsz, pl->m12);
    HIPCHK(pl->ctx, lds_pass(pl->p.m1, XF_INV, MID_SCALE, XF_NONE, false, a, pl->ctx->stream));
    return FDES_OK;
}

// incomingWave, src/multisliceSimulation.cu:563-591
int incoming_wave(fdes_plan* pl, int k, float2* psi = nullptr) // psi: where the wave goes (default: the plan's PSI)
{
    float2* const PSI = psi ? psi : pl->PSI;
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    HIPCHK(c, k_fill(PSI, pl->m12, 1.f, 0.f, c->stream));
    pl->wave_bl = !(p.mode == 2 && p.doBeamTilt);
    if (p.mode == 2) {
        HIPCHK(c, k_lens(PSI, pl->kp, p.defoci[k], c->stream));
        HIPCHK(c, fft_exec(pl,PSI, true, c->stream));
        HIPCHK(c, k_fftshift(pl->T, PSI, p.m1, p.m2, c->stream));
        HIPCHK(c, hipMemcpyAsync(PSI, pl->T, sizeof(float2) * pl->m12, hipMemcpyDeviceToDevice, c->stream));
        RC(bandwidth_limit(pl, PSI));
        HIPCHK(c, k_normalize_to(PSI, pl->m12, sqrtf((float)(p.n1 * p.n2)), pl->scal, c->stream));
    }
    if (p.doBeamTilt) HIPCHK(c, k_tilt_beam(PSI, pl->kp, p.tiltbeam[2 * k], p.tiltbeam[2 * k + 1], 1, c->stream));
    if (p.doBeamTilt && (p.mode == 0 || p.mode == 1)) {
        HIPCHK(c, k_tukey(PSI, pl->kp, c->stream));
        RC(bandwidth_limit(pl, PSI));
    }
    return FDES_OK;
}

// slice loop of one configuration up to nslices (src/crystalMaker.cu:339-344)
int slice_loop(fdes_plan* pl, int nslices)
{
    BinGeom g{pl->p.m1, pl->p.m2, pl->p.m3, pl->nZ, pl->p.d1, pl->p.d2, pl->p.d3};
    fdes_ctx* c = pl->ctx;
    const fdes_ctx* oc = owner_ctx(pl); // lanes follow the owner's runtime options
    // the launch sequence of one configuration: fused LDS passes, or rocFFT + point-wise kernels for the other grid sizes
    auto issue = [&]() -> int {
        if (pl->fused) {
            RC(fused_enter(pl));
            RC(split_fork(pl));
            if (pl->nb > 1 && !pl->tap_mode) RC(batched_loop(pl, nslices));
            else
            for (int s = 0, adv = 1; s < nslices; s += adv) RC(fused_slice(pl, s, nslices, &adv));
            RC(split_join(pl));
            return fused_leave(pl, nslices > 0);
        }
        for (int s = 0; s < nslices; s++) {
            if ((s & 1) == 0) RC(phase_grating_pair(pl, pl->xyzFP_d, g, s));
            RC(forward_propagation(pl, s & 1));
        }
        return FDES_OK;
    };
    const bool timing_probe = (oc->probe_stride > 0);
    // the two-stream loop is issued directly: captured, its cross-stream edges cost 5 % (12.2 k against 12.85 k)
    if (!oc->opt_graph || timing_probe || nslices < 1 || (pl->fused && pl->split) || (pl->top ? pl->top : pl)->one_shot_few) return issue();
    // key: slice count, band option and the empty-slice pattern (FNV-1a over one bit per slice)
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    // the pattern itself (slice count, band option, one byte per slice) is kept beside its hash and compared on a
    // hit: a colliding hash must not replay another pattern's launch sequence
    std::vector<uint8_t> pattern;
    pattern.reserve((size_t)pl->p.m3 + 8);
    for (int b = 0; b < 4; b++) pattern.push_back((uint8_t)((unsigned)nslices >> (8 * b)));
    pattern.push_back((uint8_t)oc->band_skip);
    pattern.push_back((uint8_t)oc->walk);
    pattern.push_back((uint8_t)(oc->stagger & 255));
    pattern.push_back((uint8_t)(oc->stagger >> 8));
    pattern.push_back((uint8_t)(pl->split ? 1 : 0));
    pattern.push_back((uint8_t)(pl->wave_bl ? 1 : 0));
    pattern.push_back((uint8_t)pl->gn);
    pattern.push_back(pl->seg_h.empty() ? 0 : 1);
    if (!pl->seg_h.empty())
        for (int q = 0; q < pl->p.m3; q++) pattern.push_back(pl->seg_h[(size_t)(q + 1) * pl->nZ] == pl->seg_h[(size_t)q * pl->nZ] ? 2 : 3);
    for (uint8_t b : pattern) mix(b);
    fdes_plan::LoopGraph* gr = nullptr;
    for (auto& e : pl->graphs) if (e.key == key && e.pattern == pattern) gr = &e;
    if (!gr) {
        DeviceGuard guard(c->device);
        const int64_t skipped0 = pl->slices_skipped;
        std::vector<std::pair<int, float2*>> pow_owned;
        pl->capture_pow = &pow_owned;
        pl->capturing = true;
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        int rc = FDES_OK;
        if (e == hipSuccess) rc = issue();
        hipGraph_t graph = nullptr;
        hipError_t e2 = (e == hipSuccess) ? hipStreamEndCapture(c->stream, &graph) : e;
        pl->capturing = false;
        pl->capture_pow = nullptr;
        const int64_t skipped = pl->slices_skipped - skipped0;
        pl->slices_skipped = skipped0;
        hipGraphExec_t exec = nullptr;
        hipError_t e3 = hipSuccess;
        if (rc == FDES_OK && e2 == hipSuccess) e3 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc != FDES_OK || e2 != hipSuccess || e3 != hipSuccess) {
            for (auto& e : pow_owned) (void)hipFree(e.second);
            if (rc != FDES_OK) return rc;
            HIPCHK(c, e2);
            HIPCHK(c, e3);
        }
        if (pl->graphs.size() >= 8) { // drop the least recently used pattern
            size_t lru = 0;
            for (size_t i = 1; i < pl->graphs.size(); i++) if (pl->graphs[i].used < pl->graphs[lru].used) lru = i;
            (void)hipStreamSynchronize(c->stream); // a replay of the evicted graph may still be reading its tables
            (void)hipGraphExecDestroy(pl->graphs[lru].exec);
            for (auto& e : pl->graphs[lru].pow) (void)hipFree(e.second);
            pl->graphs.erase(pl->graphs.begin() + (long)lru);
        }
        pl->graphs.push_back({key, pattern, exec, skipped, 0, pow_owned});
        gr = &pl->graphs.back();
    }
    gr->used = ++pl->graph_tick;
    HIPCHK(c, hipGraphLaunch(gr->exec, c->stream));
    pl->slices_skipped += gr->skipped;
    return FDES_OK;
}

// exit-wave post-processing + accumulation (src/crystalMaker.cu:346-366)
int exit_wave_post(fdes_plan* pl, int k, float weight, float2* psi = nullptr, float2* acc = nullptr) // psi: the exit wave; acc: the intensity sum it is added to
{
    float2* const PSI = psi ? psi : pl->PSI;
    float2* const I = acc ? acc : pl->I;
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    if (pl->want_ew) HIPCHK(c, k_axpy(pl->EW, PSI, pl->m12, weight, c->stream));
    if (p.mode == 0) {
        // applyLensFunction (src/multisliceSimulation.cu:614-622) + intensityValues + Caxpy
        HIPCHK(c, fft_exec(pl,PSI, false, c->stream));
        HIPCHK(c, k_lens(PSI, pl->kp, p.defoci[k], c->stream));
        HIPCHK(c, fft_exec(pl,PSI, true, c->stream));
        HIPCHK(c, k_intensity_axpy(I, PSI, pl->m12, 1.f / ((float)pl->m12), weight, c->stream));
    } else {
        // diffractionPattern (src/crystalMaker.cu:700-718)
        if (p.doBeamTilt) HIPCHK(c, k_tilt_beam(PSI, pl->kp, p.tiltbeam[2 * k], p.tiltbeam[2 * k + 1], -1, c->stream));
        if (p.mode == 1) {
            HIPCHK(c, k_mask_filter(PSI, pl->kp, c->stream));
            RC(bandwidth_limit(pl, PSI));
        }
        HIPCHK(c, fft_exec(pl,PSI, false, c->stream));
        HIPCHK(c, k_fftshift(pl->T, PSI, p.m1, p.m2, c->stream));
        HIPCHK(c, k_intensity_axpy(I, pl->T, pl->m12, sqrtf(1.f / ((float)pl->m12)), weight, c->stream));
    }
    return FDES_OK;
}

// addNoiseAndMtf, src/crystalMaker.cu:579-613: the summed intensity in pl->I -> image k
int finalize_measurement(fdes_plan* pl, int k, float2* acc = nullptr) // acc: the summed intensity (default: the plan's I)
{
    float2* const I = acc ? acc : pl->I;
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    const float alpha = 1.f / ((float)(p.m1 * p.m2));
    HIPCHK(c, fft_exec(pl,I, false, c->stream));
    if (fabsf(p.illangle) > FLT_EPSILON) {
        if (p.mode == 0) HIPCHK(c, k_spatial_incoherence(I, pl->kp, p.defoci[k], 0, c->stream));
        if (p.mode == 1 || p.mode == 2) HIPCHK(c, k_spatial_incoherence(I, pl->kp, p.defoci[k], 1, c->stream));
    }
    if (p.pD > FLT_EPSILON) {
        HIPCHK(c, k_scale(I, pl->m12, alpha, c->stream));
        HIPCHK(c, fft_exec(pl,I, true, c->stream));
        HIPCHK(c, k_noise(I, pl->m12, p.pD, (uint32_t)(1 + p.n3), k, c->stream)); // seed 1 + n3, :295
        HIPCHK(c, fft_exec(pl,I, false, c->stream));
    }
    HIPCHK(c, k_mtf(I, pl->kp, alpha, c->stream));
    HIPCHK(c, fft_exec(pl,I, true, c->stream));
    HIPCHK(c, k_crop(pl->Jout + (size_t)k * p.n1 * p.n2, I, pl->kp, c->stream));
    return FDES_OK;
}

// 2-D transforms of n grids `stride` elements apart: with the hand-written back-end the two passes take the grids as grid z
int fft_gang(fdes_plan* pl, float2* data, int n, size_t stride, bool inverse)
{
    fdes_ctx* c = pl->ctx;
    if (n <= 1 || pl->fft->backend != 2 || !pl->gscr) {
        for (int g = 0; g < n; g++) HIPCHK(c, fft_exec(pl, data + (size_t)g * stride, inverse, c->stream));
        return FDES_OK;
    }
    const int xf = inverse ? XF_INV : XF_FWD;
    PassArgs a;
    a.in0 = data; a.out = pl->gscr; a.tw0 = pl->fft->tw0x; a.tw1 = pl->fft->tw1x; a.nrows = pl->p.m2; a.wg = pl->fft->wg;
    a.nbatch = n; a.bstride_in0 = stride; a.bstride_out = pl->m12;
    HIPCHK(c, lds_pass(pl->p.m1, xf, MID_NONE, XF_NONE, true, a, c->stream));
    PassArgs b;
    b.in0 = pl->gscr; b.out = data; b.tw0 = pl->fft->tw0y; b.tw1 = pl->fft->tw1y; b.nrows = pl->p.m1; b.wg = pl->fft->wg;
    b.nbatch = n; b.bstride_in0 = pl->m12; b.bstride_out = stride;
    HIPCHK(c, lds_pass(pl->p.m2, xf, MID_NONE, XF_NONE, true, b, c->stream));
    return FDES_OK;
}

// bandwidthLimit of n grids m12 apart
int bandwidth_limit_gang(fdes_plan* pl, float2* f, int n)
{
    RC(fft_gang(pl, f, n, pl->m12, false));
    HIPCHK(pl->ctx, k_mask_scale_gang(f, pl->m12, n, pl->p.m1, pl->p.m2, 1.f / ((float)pl->m12), pl->ctx->stream));
    return fft_gang(pl, f, n, pl->m12, true);
}

// incoming_wave of the n members (PSI, m12 apart).  Members of one k share their wave: built once, copied.  Members with a k
// of their own (gangs across measurements): plane waves with a beam tilt take one launch per step for all members; a
// CBED probe (own lens, own norm) is built member by member.
int incoming_wave_gang(fdes_plan* pl, int n)
{
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    float2* const psi0 = pl->PSI;
    bool same_k = true;
    for (int g = 1; g < n; g++) same_k = same_k && pl->gq[(size_t)g].k == pl->gq[0].k;
    if (n > 1 && !same_k && p.mode != 2 && pl->gscr) {
        pl->wave_bl = true;
        HIPCHK(c, k_fill(psi0, pl->m12 * (size_t)n, 1.f, 0.f, c->stream));
        if (p.doBeamTilt) {
            GangPar tb0, tb1;
            tb0.n = tb1.n = n;
            for (int g = 0; g < n; g++) { tb0.f[g] = p.tiltbeam[2 * pl->gq[(size_t)g].k]; tb1.f[g] = p.tiltbeam[2 * pl->gq[(size_t)g].k + 1]; }
            HIPCHK(c, k_tilt_beam_gang(psi0, pl->m12, pl->kp, tb0, tb1, 1, c->stream));
            HIPCHK(c, k_tukey_gang(psi0, pl->m12, n, pl->kp, c->stream));
            RC(bandwidth_limit_gang(pl, psi0, n));
        }
        return FDES_OK;
    }
    int rcw = FDES_OK;
    for (int g = 0; g < n && rcw == FDES_OK; g++) {
        float2* const mine = psi0 + (size_t)g * pl->m12;
        if (g > 0 && pl->gq[(size_t)g].k == pl->gq[(size_t)g - 1].k) {
            if (hipMemcpyAsync(mine, mine - pl->m12, sizeof(float2) * pl->m12, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rcw = FDES_EGPU;
        } else {
            rcw = incoming_wave(pl, pl->gq[(size_t)g].k, mine);
        }
    }
    return rcw;
}

// exit_wave_post of the n members of a gang (waves back to back in PSI) in a handful of launches (no exit-wave output:
// that case goes member by member)
int exit_wave_post_gang(fdes_plan* pl, int n)
{
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    if (n <= 1 || pl->want_ew || !pl->gscr) {
        int rce = FDES_OK;
        for (int g = 0; g < n && rce == FDES_OK; g++)
            rce = exit_wave_post(pl, pl->gq[(size_t)g].k, pl->gq[(size_t)g].w, pl->PSI + (size_t)g * pl->m12, pl->I + (size_t)pl->gq[(size_t)g].slot * pl->m12);
        return rce;
    }
    GangPar dk, wt;
    dk.n = wt.n = n;
    for (int g = 0; g < n; g++) {
        dk.f[g] = p.defoci[pl->gq[(size_t)g].k];
        wt.f[g] = pl->gq[(size_t)g].w;
        wt.k[g] = pl->gq[(size_t)g].slot;
    }
    if (p.mode == 0) {
        RC(fft_gang(pl, pl->PSI, n, pl->m12, false));
        HIPCHK(c, k_lens_gang(pl->PSI, pl->m12, pl->kp, dk, c->stream));
        RC(fft_gang(pl, pl->PSI, n, pl->m12, true));
        HIPCHK(c, k_intensity_gang(pl->I, pl->PSI, pl->m12, 1.f / ((float)pl->m12), wt, c->stream));
        return FDES_OK;
    }
    // diffractionPattern (src/crystalMaker.cu:700-718), as exit_wave_post
    if (p.doBeamTilt) {
        GangPar tb0, tb1;
        tb0.n = tb1.n = n;
        for (int g = 0; g < n; g++) { tb0.f[g] = p.tiltbeam[2 * pl->gq[(size_t)g].k]; tb1.f[g] = p.tiltbeam[2 * pl->gq[(size_t)g].k + 1]; }
        HIPCHK(c, k_tilt_beam_gang(pl->PSI, pl->m12, pl->kp, tb0, tb1, -1, c->stream));
    }
    if (p.mode == 1) {
        HIPCHK(c, k_mask_filter_gang(pl->PSI, pl->m12, n, pl->kp, c->stream));
        RC(bandwidth_limit_gang(pl, pl->PSI, n));
    }
    RC(fft_gang(pl, pl->PSI, n, pl->m12, false));
    HIPCHK(c, k_fftshift_gang(pl->gscr, pl->PSI, pl->m12, n, p.m1, p.m2, c->stream)); // (the transforms are done with their scratch)
    HIPCHK(c, k_intensity_gang(pl->I, pl->gscr, pl->m12, sqrtf(1.f / ((float)pl->m12)), wt, c->stream));
    return FDES_OK;
}

// finalize_measurement of the measurements in gfinal when their slots are 0, 1, 2 ... in order (what
// fdes_build_measurements queues): the detector chain over all slots per launch
int finalize_gang(fdes_plan* pl)
{
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    const int n = (int)pl->gfinal.size();
    bool in_order = n > 1 && n <= 16 && pl->gscr != nullptr;
    for (int q = 0; q < n && in_order; q++) in_order = pl->gfinal[(size_t)q].second == q;
    if (!in_order) {
        int rcf = FDES_OK;
        for (size_t q = 0; q < pl->gfinal.size() && rcf == FDES_OK; q++)
            rcf = finalize_measurement(pl, pl->gfinal[q].first, pl->I + (size_t)pl->gfinal[q].second * pl->m12);
        return rcf;
    }
    GangPar dk, kk;
    dk.n = kk.n = n;
    for (int q = 0; q < n; q++) { dk.f[q] = p.defoci[pl->gfinal[(size_t)q].first]; kk.k[q] = pl->gfinal[(size_t)q].first; }
    const float alpha = 1.f / ((float)(p.m1 * p.m2));
    RC(fft_gang(pl, pl->I, n, pl->m12, false));
    if (fabsf(p.illangle) > FLT_EPSILON) HIPCHK(c, k_spatial_incoherence_gang(pl->I, pl->m12, pl->kp, p.mode == 0 ? 0 : 1, dk, c->stream));
    if (p.pD > FLT_EPSILON) {
        HIPCHK(c, k_scale(pl->I, pl->m12 * (size_t)n, alpha, c->stream));
        RC(fft_gang(pl, pl->I, n, pl->m12, true));
        HIPCHK(c, k_noise_gang(pl->I, pl->m12, pl->m12, p.pD, (uint32_t)(1 + p.n3), kk, c->stream));
        RC(fft_gang(pl, pl->I, n, pl->m12, false));
    }
    HIPCHK(c, k_mtf_gang(pl->I, pl->m12, n, pl->kp, alpha, c->stream));
    RC(fft_gang(pl, pl->I, n, pl->m12, true));
    HIPCHK(c, k_crop_gang(pl->Jout, pl->I, pl->m12, pl->kp, kk, c->stream));
    return FDES_OK;
}

// The queued configurations of this plan as ONE gang: the incoming wave once per measurement k (it depends on k only;
// members of the same k get a copy), atoms / records per member, one slice loop with the members as grid z, the detector
// chain per member into the member's intensity slot.  The members are the configurations of one measurement - or, for a
// series with one configuration per measurement (gang_k), measurements.  A slice counts as empty (skip_empty) only when
// it is empty in every member: the others run the full sequence on it, which is always correct (t = band-limited 1).
int gang_flush(fdes_plan* pl)
{
    const int n = (int)pl->gq.size();
    fdes_ctx* c = pl->ctx;
    if (n > 0) {
        RC(incoming_wave_gang(pl, n));
        if (pl->nAt > 0) {
            // tilt, jitter and binning of all members in one launch each (geometry.hip, *_gang): what config_atoms does
            // member by member, to the bit
            int ks[16], js[16];
            float t0[16], t1[16];
            bool same_k = true;
            for (int g = 0; g < n; g++) {
                ks[g] = pl->gq[(size_t)g].k; js[g] = pl->gq[(size_t)g].j;
                t0[g] = pl->p.tiltspec[2 * ks[g]]; t1[g] = pl->p.tiltspec[2 * ks[g] + 1];
                same_k = same_k && ks[g] == ks[0];
            }
            const size_t n3f = 3 * (size_t)pl->nAt;
            if (same_k) {
                RC(ensure_tilt(pl, ks[0]));
                if (pl->p.frPh > 0) HIPCHK(c, geom_jitter_gang(pl->gxyzFP, pl->xyzK_d, 0, pl->dwf_d, pl->nAt, n, owner_ctx(pl)->seed, ks, js, c->stream));
                else for (int g = 0; g < n; g++) HIPCHK(c, hipMemcpyAsync(pl->gxyzFP + (size_t)g * n3f, pl->xyzK_d, sizeof(float) * n3f, hipMemcpyDeviceToDevice, c->stream));
            } else {
                HIPCHK(c, geom_tilt_gang(pl->gxyzFP, pl->xyzTO_d, pl->nAt, n, t0, t1, c->stream));
                if (pl->p.frPh > 0) HIPCHK(c, geom_jitter_gang(pl->gxyzFP, pl->gxyzFP, n3f, pl->dwf_d, pl->nAt, n, owner_ctx(pl)->seed, ks, js, c->stream));
            }
            BinGeom bg{pl->p.m1, pl->p.m2, pl->p.m3, pl->nZ, pl->p.d1, pl->p.d2, pl->p.d3};
            HIPCHK(c, geom_bin_atoms_gang(pl->gxyzFP, pl->spec_d, pl->occ_d, pl->nAt, n, bg, pl->bins, pl->seg_stride, pl->rowstart_stride, c->stream));
        } else
        for (int g = 0; g < n; g++)
            RC(config_atoms(pl, pl->gq[(size_t)g].k, pl->gq[(size_t)g].j, false, pl->gxyzFP + (size_t)g * 3 * (size_t)pl->nAt, &pl->gbins[(size_t)g]));
        // which slices hold atoms: one question (n small copies, ONE host wait) for the whole gang; the rules of
        // config_atoms for when a dense specimen is no longer asked, counted per member
        bool have_all = false;
        {
            fdes_plan* tp = pl->top ? pl->top : pl;
            constexpr int kDenseAfter = 8, kDenseRecheck = 64;
            bool ask = owner_ctx(pl)->skip_empty != 0;
            if (ask && tp->dense_streak >= kDenseAfter && (tp->cfg_seen % kDenseRecheck) >= n) ask = false;
            tp->cfg_seen += n;
            if (ask) {
                std::vector<int> qk((size_t)n), qj((size_t)n);
                for (int g = 0; g < n; g++) { qk[(size_t)g] = pl->gq[(size_t)g].k; qj[(size_t)g] = pl->gq[(size_t)g].j; }
                RC(empty_query(pl, n, qk.data(), qj.data())); // fills seg_h (occupied in ANY member) and, for n > 1, gseg[g]
                tp->empty_queries++;
                for (int g = 0; g < n; g++) {
                    const std::vector<int>& t = n > 1 ? pl->gseg[(size_t)g] : pl->seg_h;
                    bool any_empty = false;
                    for (int q = 0; q < pl->p.m3 && !any_empty; q++) any_empty = t[(size_t)(q + 1) * pl->nZ] == t[(size_t)q * pl->nZ];
                    tp->dense_streak = any_empty ? 0 : tp->dense_streak + 1;
                }
                have_all = true;
            }
        }
        if (!have_all) pl->seg_h.clear();
        if (pl->ev_used == pl->evs.size()) {
            EvPair e{};
            HIPCHK(c, hipEventCreate(&e.a));
            HIPCHK(c, hipEventCreate(&e.b));
            pl->evs.push_back(e);
        }
        EvPair& ev = pl->evs[pl->ev_used++];
        ev.slices = pl->p.m3 * n;
        ev.configs = n;
        HIPCHK(c, hipEventRecord(ev.a, c->stream));
        pl->gn = n;
        const int rcl = slice_loop(pl, pl->p.m3);
        pl->gn = 1;
        RC(rcl);
        HIPCHK(c, hipEventRecord(ev.b, c->stream));
        pl->slices_done += (int64_t)pl->p.m3 * n;
        const int rce = exit_wave_post_gang(pl, n);
        pl->gq.clear();
        RC(rce);
    }
    // measurements whose last member has just been issued: detector chain on their slot
    if (!pl->gfinal.empty()) {
        const int rcf = finalize_gang(pl);
        pl->gfinal.clear();
        RC(rcf);
    }
    return FDES_OK;
}

// everything queued on this plan and its lanes is issued (before anything reads or resets the sums)
int gang_flush_all(fdes_plan* pl)
{
    RC(gang_flush(pl));
    for (fdes_plan* l : pl->lanes) {
        const int rc = gang_flush(l);
        if (rc != FDES_OK) { pl->ctx->err = "lane: " + l->ctx->err; return rc; }
    }
    return FDES_OK;
}

// Lane 0 takes over the partial sums of the other lanes: I += I_lane (and the exit-wave sum), ordered by
// events in both directions (lane stream -> lane 0 before the read, lane 0 -> lane stream before the lane
// reuses its accumulators).
int fold_lanes(fdes_plan* pl)
{
    fdes_ctx* c = pl->ctx;
    RC(gang_flush_all(pl));
    if (!pl->lanes_dirty) return FDES_OK;
    for (size_t l = 0; l < pl->lanes.size(); l++) {
        fdes_plan* lp = pl->lanes[l];
        HIPCHK(c, hipEventRecord(pl->lane_ev[l], lp->ctx->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, pl->lane_ev[l], 0));
        HIPCHK(c, k_axpy(pl->I, lp->I, pl->m12, 1.f, c->stream));
        HIPCHK(c, k_fill(lp->I, pl->m12, 0.f, 0.f, c->stream));
        if (pl->want_ew) {
            HIPCHK(c, k_axpy(pl->EW, lp->EW, pl->m12, 1.f, c->stream));
            HIPCHK(c, k_fill(lp->EW, pl->m12, 0.f, 0.f, c->stream));
        }
        HIPCHK(c, hipEventRecord(pl->lane_ev[l], c->stream));
        HIPCHK(c, hipStreamWaitEvent(lp->ctx->stream, pl->lane_ev[l], 0));
    }
    pl->lanes_dirty = false;
    return FDES_OK;
}

// configurations whose slice loop has finished on the GPU (this plan and its lanes); oldest_pending = the end event of
// the oldest one still running, if any
int64_t configs_finished(fdes_plan* pl, hipEvent_t* oldest_pending)
{
    int64_t n = 0;
    auto scan = [&](fdes_plan* q) {
        while (q->ev_done < q->ev_used && hipEventQuery(q->evs[q->ev_done].b) == hipSuccess) q->cfg_done += q->evs[q->ev_done++].configs;
        (void)hipGetLastError(); // hipErrorNotReady is not an error
        n += q->cfg_done;
        if (oldest_pending && !*oldest_pending && q->ev_done < q->ev_used) *oldest_pending = q->evs[q->ev_done].b;
    };
    scan(pl);
    for (fdes_plan* l : pl->lanes) scan(l);
    return n;
}

// rate-limited progress report of fdes_build_measurements; also bounds the number of configurations in flight
void report_progress(fdes_plan* pl, int64_t issued, int64_t total_configs, bool final)
{
    fdes_ctx* c = pl->ctx;
    if (!c->progress) return;
    const int64_t depth = 2 * (int64_t)(pl->lanes.size() + 1);
    int64_t done = 0;
    for (;;) {
        hipEvent_t pending = nullptr;
        done = configs_finished(pl, &pending);
        if ((!final && issued - done <= depth) || !pending) break;
        (void)hipEventSynchronize(pending);
    }
    const auto now = std::chrono::steady_clock::now();
    if (!final && std::chrono::duration_cast<std::chrono::milliseconds>(now - c->progress_last).count() < c->progress_min_ms) return;
    c->progress_last = now;
    c->progress(c->progress_user, done * (int64_t)pl->p.m3, total_configs * (int64_t)pl->p.m3);
}

// Configurations a plan keeps in flight: lanes hide the gap between dependent kernels of one stream - three up to 1024^2,
// where the kernels are no longer than that gap, two above - but never more than the job has configurations (n3 x
// frozen-phonon configurations): a single image gets one lane, and its concurrency from the split slice loop instead.
// (A fourth lane at 1024^2 is 43 k slice-propagations/s on C4 against 38 k with three - or 17 k: which of the two depends
// on the hardware queues the runtime happens to hand out, i.e. on what the process created before; six / eight lanes
// 25 k / 20 k; GPU_MAX_HW_QUEUES changes nothing.  tools/exp/c4_job.py, tools/exp/stream_overlap.hip.)
// Members of a gang (configurations of ONE measurement in lockstep on a lane): only the fused loop on one stream takes
// them, and never more than a measurement has configurations
// a job of gangs that is too small for a second lane to pay (or told to use one)
bool gang_one_lane(const fdes_ctx* c, const fdes_plan* pl)
{
    if (c->lanes > 0) return c->lanes == 1;
    const double job = (double)pl->p.n3 * (double)(pl->p.frPh > 0 ? pl->p.frPh : 1) * (double)pl->p.m3 * (double)pl->p.m1 * (double)pl->p.m2;
    return job < 2e9;
}

int plan_gang(const fdes_ctx* c, const fdes_plan* pl, bool* across_k = nullptr)
{
    const int count = pl->p.frPh > 0 ? pl->p.frPh : 1;
    // what a gang is made of: the configurations of one measurement, or - a series without frozen phonons has one
    // configuration per measurement - the measurements themselves (fdes_build_measurements drives those)
    const bool ak = count < 2;
    const int units = ak ? pl->p.n3 : count;
    if (across_k) *across_k = false;
    if (c->gang == 0 || c->gang == 1 || units < 2) return 1;
    if (c->opt_fft == 1 || !Fft2D::lds_supported(pl->p.m1, pl->p.m2)) return 1;
    if (c->split > 0 || c->batch > 1 || c->pass_threads == 65 || c->walk > 1) return 1;
    // auto (tools/gang_tables.sh -> profiles/r03_gang_tables.txt: SrTiO3 tilt series, 16 configurations per tilt,
    // slice-propagations/s; without gangs on three lanes -> members x lanes): 256^2 76 k -> 302 k (16 x 1), 359 k (16 x 2), 279 k
    // (8 x 2); 512^2 53 k -> 134 k (16 x 1), 139 k (8 x 2 and 16 x 2); 800^2 22 k -> 27 k (16 x 1), 28 k (4 x 2); 1024^2 38 k -> 46 k
    // (8 x 1), 47-48 k (4 x 2), 43 k (8 x 2); from 2048^2 on one configuration's rows fill the chip.  Two lanes need two gangs
    // per measurement resp. series.  A second lane doubles the set-up (streams, plan, tables) and pays from about 2 x 10^9
    // pixel-slices on (whole boundary call: bin/dataFDES.cnf 150 ms ungrouped, 54 ms with 8 x 2 lanes, 23 ms with 16 members
    // on one lane; 512^2 x 256 tilts 100 ms against 96 ms): small jobs get one lane and larger gangs.
    int g = c->gang;
    if (g < 0) {
        const size_t m12 = (size_t)pl->p.m1 * (size_t)pl->p.m2;
        if (gang_one_lane(c, pl)) g = m12 <= ((size_t)1 << 18) ? 16 : (m12 <= ((size_t)1 << 20) ? 8 : 1);
        else {
            g = m12 <= ((size_t)1 << 16) ? 16 : (m12 <= ((size_t)1 << 18) ? 8 : (m12 <= ((size_t)1 << 20) ? 4 : 1));
            if (units >= 4 && g > units / 2) g = units / 2;
        }
    }
    if (g > 16) g = 16;
    g = g < units ? g : units;
    if (across_k) *across_k = ak && g > 1;
    return g;
}

int plan_lanes(const fdes_ctx* c, const fdes_plan* pl)
{
    if (c->is_lane_ctx) return 1;
    const int g = plan_gang(c, pl);
    long total = (long)pl->p.n3 * (long)(pl->p.frPh > 0 ? pl->p.frPh : 1);
    if (g > 1) total = pl->p.frPh >= 2 ? (long)pl->p.n3 * (((long)pl->p.frPh + g - 1) / g) : ((long)pl->p.n3 + g - 1) / g; // gangs in flight, not configurations
    if (c->lanes > 0) return g > 1 ? (int)(total < c->lanes ? total : c->lanes) : c->lanes;
    if (g > 1 && gang_one_lane(c, pl)) return 1;
    const int by_size = (pl->fused && pl->m12 <= (size_t)1024 * 1024 && g == 1) ? 3 : 2; // (gangs: two lanes measured equal to or better than three)
    return (int)(total < by_size ? total : by_size);
}

int check_params(fdes_ctx* ctx, const fdes_params* p, const fdes_atoms* a)
{
    if (!p || !a || !p->tiltspec || !p->tiltbeam || !p->defoci) { ctx->err = "null parameter / atom pointers"; return FDES_EINVAL; }
    if (p->n3 < 1 || p->cap < p->n3 || p->n1 < 1 || p->n2 < 1 || p->dn1 < 0 || p->dn2 < 0 || p->m3 < 1) { ctx->err = "bad sizes"; return FDES_EINVAL; }
    if (p->m1 != p->n1 + 2 * p->dn1 || p->m2 != p->n2 + 2 * p->dn2) { ctx->err = "m != n + 2*dn: call fdes_params_consistent first"; return FDES_EINVAL; }
    if (p->m1 < 4 || p->m2 < 4) { ctx->err = "grid too small"; return FDES_EINVAL; }
    if (p->mode < 0 || p->mode > 2) { ctx->err = "mode must be 0, 1 or 2"; return FDES_EINVAL; }
    if (!(p->d1 > 0) || !(p->d2 > 0) || !(p->d3 > 0)) { ctx->err = "pixel sizes must be positive"; return FDES_EINVAL; }
    if (a->nAt < 0 || (a->nAt > 0 && (!a->Z || !a->xyz || !a->dwf || !a->occ))) { ctx->err = "bad atom arrays"; return FDES_EINVAL; }
    return FDES_OK;
}

} // namespace

extern "C" {

int fdes_gpu_available(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n > 0;
}

namespace {
// prio_class: 0 = default priority; 1 / 2 = the greatest / least stream priority of the device.  HIP multiplexes
// streams of one priority onto a few hardware queues, and two lanes that land on the same queue run one after the
// other; streams of different priority use different queues (measured: three lanes at 1024^2, three species:
// 20.5 k/s with equal priorities, 30.9 k/s with distinct ones).
int create_ctx(fdes_ctx** out, int gpu_index, int prio_class)
{
    if (!out) return FDES_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || gpu_index < 0 || gpu_index >= n) { (void)hipGetLastError(); return FDES_EGPU; }
    fdes_ctx* c = new fdes_ctx();
    c->device = gpu_index;
    bool ok = hipSetDevice(gpu_index) == hipSuccess;
    if (ok && prio_class > 0) {
        int least = 0, greatest = 0;
        ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
             hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_class == 1 ? greatest : least) == hipSuccess;
    } else if (ok) {
        ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        delete c;
        return FDES_EGPU;
    }
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
    // Registered AFTER the HIP runtime and rocFFT have initialised, so at process exit it runs BEFORE their own
    // teardown (exit handlers run in reverse order of registration): whatever the host left open is closed while
    // the runtime is still alive, instead of by a finaliser that runs after it is gone.
    std::call_once(g_atexit_once, [] { std::atexit(shutdown_all); });
    { std::lock_guard<std::mutex> g(g_live_mutex); g_live_ctx.insert(c); }
    *out = c;
    return FDES_OK;
}

void shutdown_all()
{
    for (;;) {
        fdes_ctx* c = nullptr;
        {
            std::lock_guard<std::mutex> g(g_live_mutex);
            for (const void* q : g_live_ctx)
                if (!((const fdes_ctx*)q)->is_lane_ctx) { c = (fdes_ctx*)q; break; } // lane contexts go with their plan
        }
        if (!c) break;
        (void)fdes_destroy(c);
    }
}
} // namespace

int fdes_create(fdes_ctx** out, int gpu_index) { return create_ctx(out, gpu_index, 0); }

int fdes_destroy(fdes_ctx* c)
{
    if (!c || !live_ctx(c)) return FDES_EINVAL;
    (void)hipSetDevice(c->device);
    while (!c->plans.empty()) (void)fdes_plan_destroy(c->plans.back()); // each call removes itself from the list
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->fft_cache) { kv.second->destroy(); delete kv.second; }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    { std::lock_guard<std::mutex> g(g_live_mutex); g_live_ctx.erase(c); }
    c->fft_cache.clear();
    c->plans.clear();
    c->stream = nullptr;
    bury(c);
    return FDES_OK;
}

const char* fdes_last_error(const fdes_ctx* c) { return c ? c->err.c_str() : "null context"; }

int fdes_set_progress(fdes_ctx* c, fdes_progress_fn fn, void* user, int min_interval_ms)
{
    if (!c || min_interval_ms < 0) return FDES_EINVAL;
    c->progress = fn;
    c->progress_user = user;
    c->progress_min_ms = min_interval_ms;
    return FDES_OK;
}

int fdes_set_option(fdes_ctx* c, const char* key, int64_t value)
{
    if (!live_ctx(c) || !key) return FDES_EINVAL;
    if (!std::strcmp(key, "fft")) { if (value < 0 || value > 2) return FDES_EINVAL; c->opt_fft = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "graph")) { c->opt_graph = value != 0; return FDES_OK; }
    if (!std::strcmp(key, "seed")) { c->seed = (uint32_t)value; return FDES_OK; }
    if (!std::strcmp(key, "pass_threads")) { if (value != 0 && value != 1 && value != 64 && value != 65 && value != 128 && value != 256 && value != 512 && value != 513) return FDES_EINVAL; c->pass_threads = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "split")) { if (value < -1 || value > 1) return FDES_EINVAL; c->split = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "gang")) { if (value < -1 || value > 16) return FDES_EINVAL; c->gang = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "batch")) { if (value < -1 || value > 8) return FDES_EINVAL; c->batch = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "stagger")) { if (value < -2048 || value > 1024) return FDES_EINVAL; c->stagger = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "walk")) { if (value < 1 || value > 8) return FDES_EINVAL; c->walk = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "pitch_pad")) { if (value < -1 || value > 1024 || (value > 0 && value % 2)) return FDES_EINVAL; c->pitch_pad = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "band_skip")) { c->band_skip = value != 0; return FDES_OK; }
    if (!std::strcmp(key, "skip_empty")) { c->skip_empty = value != 0; return FDES_OK; }
    if (!std::strcmp(key, "lanes")) { if (value < 0 || value > 8) return FDES_EINVAL; c->lanes = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "deterministic")) { c->deterministic = value != 0; return FDES_OK; }
    if (!std::strcmp(key, "peer_copy")) { c->peer_copy = value != 0; return FDES_OK; }
#if FDES_TEST_HOOKS
    // keys of bench.py's roofline probe and of the fdes_bench_pass micro-benchmark: a TEST_HOOKS=0 build does not know them
    if (!std::strcmp(key, "lanes_active")) { c->lanes_active = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "bench_alt")) { c->bench_alt = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "bench_tall")) { if (value < 1 || value > 4) return FDES_EINVAL; c->bench_tall = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "bench_band")) { c->bench_band = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "bench_serial")) { c->bench_serial = value != 0; return FDES_OK; }
    if (!std::strcmp(key, "bench_pitch")) { if (value < 0 || value > 4096) return FDES_EINVAL; c->bench_pitch = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "probe_stride")) { c->probe_stride = (int)value; return FDES_OK; }
    if (!std::strcmp(key, "probe_pass")) { if (value < 1 || value > 6) return FDES_EINVAL; c->probe_pass = (int)value; return FDES_OK; }
#endif
    return FDES_EINVAL;
}

int fdes_plan_destroy(fdes_plan* pl)
{
    if (!pl || !live_plan(pl)) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    DeviceGuard guard(c->device);
    { std::lock_guard<std::mutex> g(g_live_mutex); g_live_plan.erase(pl); }
    c->plans.erase(std::remove(c->plans.begin(), c->plans.end(), pl), c->plans.end());
    (void)hipSetDevice(c->device);
    for (fdes_plan* l : pl->lanes) fdes_plan_destroy(l);
    for (fdes_ctx* lc : pl->lane_ctx) fdes_destroy(lc);
    for (hipEvent_t e : pl->lane_ev) (void)hipEventDestroy(e);
    pl->lanes.clear();
    (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = {pl->Z_d, pl->spec_d, pl->xyz0_d, pl->xyzTO_d, pl->xyzK_d, pl->xyzFP_d, pl->dwf_d, pl->occ_d, pl->bins.keys,
                    pl->bins.keys_sorted, pl->bins.vals, pl->bins.order, pl->bins.seg, pl->bins.tmp, pl->bins.recs, pl->bins.recs_sorted, pl->bins.rowstart, pl->D, pl->VH, pl->T, pl->PSI,
                    pl->P, pl->I, pl->EW, pl->J, pl->scal, pl->A == pl->C ? nullptr : pl->A, pl->C, pl->C2, pl->E, pl->PSIH,
                    pl->tables_shared ? nullptr : pl->PT, pl->tables_shared ? nullptr : pl->GT, pl->peer_stage, pl->real_send, pl->span_stage, pl->span_send}; // F aliases C
    for (void* q : ptrs) if (q) (void)hipFree(q);
    for (void* q : pl->gang_owned) if (q) (void)hipFree(q);
    pl->gang_owned = {}; pl->gbins = {}; pl->gseg = {}; pl->gq = {};
    for (auto& g : pl->graphs) { (void)hipGraphExecDestroy(g.exec); for (auto& e : g.pow) (void)hipFree(e.second); }
    for (auto& e : pl->pow_tabs) if (e.tab) (void)hipFree(e.tab);
    for (auto& e : pl->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : pl->probe) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (pl->peer_ev) (void)hipEventDestroy(pl->peer_ev);
    if (pl->qs) { (void)hipStreamSynchronize(pl->qs); (void)hipStreamDestroy(pl->qs); }
    if (pl->slice_occ_d) (void)hipFree(pl->slice_occ_d);
    if (pl->slice_occ_h) (void)hipHostFree(pl->slice_occ_h);
    if (pl->split) {
        if (pl->vs) { (void)hipStreamSynchronize(pl->vs); (void)hipStreamDestroy(pl->vs); }
        for (hipEvent_t e : {pl->evE[0], pl->evE[1], pl->evP5[0], pl->evP5[1], pl->evFork, pl->evJoin}) if (e) (void)hipEventDestroy(e);
        for (void* q : {(void*)pl->Eb[1], (void*)pl->B, (void*)pl->F}) if (q) (void)hipFree(q);
    }
    for (void* q : {(void*)pl->bA, (void*)pl->bB, (void*)pl->bCC, (void*)pl->bE[0], (void*)pl->bE[1]}) if (q) (void)hipFree(q);
    for (hipEvent_t e : {pl->evReady[0], pl->evReady[1], pl->evDone[0], pl->evDone[1]}) if (e) (void)hipEventDestroy(e);
    fdes_params_release(&pl->p0);
    // release what the shell owns on the host; the shell itself goes to the graveyard (see bury())
    pl->kz = {}; pl->lanes = {}; pl->lane_ctx = {}; pl->lane_ev = {}; pl->seg_h = {}; pl->pow_tabs = {}; pl->graphs = {}; pl->probe = {};
    pl->evs = {}; pl->peer_host = {};
    pl->ctx = nullptr;
    bury(pl);
    return FDES_OK;
}

int fdes_plan_create(fdes_ctx* c, const fdes_params* p_in, const fdes_atoms* a, fdes_plan** out)
{
    if (!live_ctx(c) || !out) return FDES_EINVAL;
    *out = nullptr;
    // (no lock: with thread-local captures the allocations below disturb nobody, and the workers of fdes_build_measurements_multi
    //  create their plans at the same time; g_plan_create_overlap counts creations that ran side by side - the harness under
    //  tests/host_cpp/ asserts on it)
    RC(check_params(c, p_in, a));
    HIPCHK(c, hipSetDevice(c->device));
    fdes_plan* pl = new fdes_plan();
    pl->ctx = c;
    int rc = fdes_params_clone(&pl->p0, p_in);
    if (rc) { delete pl; return rc; }
    { std::lock_guard<std::mutex> g(g_live_mutex); g_live_plan.insert(pl); }
    c->plans.push_back(pl);
    pl->p = pl->p0; // shares arrays
    pl->ratio = fdes_params_sub_slices(&pl->p); // src/crystalMaker.cu:246-247
    pl->kp = make_kp(pl->p);
    pl->m12 = (size_t)pl->p.m1 * pl->p.m2;
    pl->nAt = a->nAt;
    const int nAt = a->nAt;
    // species list in first-seen order (listOfElements, src/crystalMaker.cu:539-570)
    std::vector<uint8_t> spec((size_t)(nAt > 0 ? nAt : 1), 0);
    pl->nZ = 0;
    for (int i = 0; i < nAt; i++) {
        int f = -1;
        for (int q = 0; q < pl->nZ; q++) if (pl->Zlist[q] == a->Z[i]) { f = q; break; }
        if (f < 0) {
            if (pl->nZ >= 103) { c->err = "more than 103 species"; fdes_plan_destroy(pl); return FDES_EINVAL; }
            f = pl->nZ;
            pl->Zlist[pl->nZ++] = a->Z[i];
        }
        spec[i] = (uint8_t)f;
    }
    if (pl->nZ == 0) { pl->nZ = 1; pl->Zlist[0] = 0; }
    for (int q = 0; q < pl->nZ; q++) pl->kz.push_back(kirkland_params(pl->Zlist[q]));

#define PLCHK(expr) do { int r_ = (expr); if (r_ != FDES_OK) { fdes_plan_destroy(pl); return r_; } } while (0)
#define PLHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e_); fdes_plan_destroy(pl); return FDES_EGPU; } } while (0)
    const size_t n3f = 3 * (size_t)nAt;
    PLCHK(dmalloc(c, &pl->Z_d, (size_t)nAt));
    PLCHK(dmalloc(c, &pl->spec_d, (size_t)nAt));
    PLCHK(dmalloc(c, &pl->xyz0_d, n3f));
    PLCHK(dmalloc(c, &pl->xyzTO_d, n3f));
    PLCHK(dmalloc(c, &pl->xyzK_d, n3f));
    pl->gang = plan_gang(c, pl, &pl->gang_k);
    const size_t G = (size_t)pl->gang;
    PLCHK(dmalloc(c, &pl->xyzFP_d, n3f * G));
    pl->gxyzFP = pl->xyzFP_d;
    PLCHK(dmalloc(c, &pl->dwf_d, (size_t)nAt));
    PLCHK(dmalloc(c, &pl->occ_d, (size_t)nAt));
    if (nAt > 0) {
        PLHIP(hipMemcpyAsync(pl->Z_d, a->Z, sizeof(int32_t) * nAt, hipMemcpyHostToDevice, c->stream));
        PLHIP(hipMemcpyAsync(pl->spec_d, spec.data(), (size_t)nAt, hipMemcpyHostToDevice, c->stream));
        PLHIP(hipMemcpyAsync(pl->xyz0_d, a->xyz, sizeof(float) * n3f, hipMemcpyHostToDevice, c->stream));
        PLHIP(hipMemcpyAsync(pl->dwf_d, a->dwf, sizeof(float) * nAt, hipMemcpyHostToDevice, c->stream));
        PLHIP(hipMemcpyAsync(pl->occ_d, a->occ, sizeof(float) * nAt, hipMemcpyHostToDevice, c->stream));
        PLHIP(hipStreamSynchronize(c->stream)); // host vector `spec` goes out of scope below
    }
    // binning buffers sized for the larger of the sub-sliced and the original slicing
    const int m3max = pl->p.m3 > pl->p0.m3 ? pl->p.m3 : pl->p0.m3;
    pl->bins_cap_keys = m3max * pl->nZ;
    // (gang: every binning array holds the members back to back - one sort serves them all, geom_bin_atoms_gang - and
    //  member g works on its own range through the view gbins[g] when it is binned alone)
    const size_t nA = (size_t)(nAt > 0 ? nAt : 1);
    pl->recs_stride = nA;
    pl->seg_stride = (size_t)pl->bins_cap_keys + 2;
    pl->rowstart_stride = (size_t)m3max * pl->nZ * (size_t)(pl->p.m2 + 1);
    PLCHK(dmalloc(c, &pl->bins.keys, nA * G));
    PLCHK(dmalloc(c, &pl->bins.keys_sorted, nA * G));
    PLCHK(dmalloc(c, &pl->bins.vals, nA * G));
    PLCHK(dmalloc(c, &pl->bins.order, nA * G));
    PLCHK(dmalloc(c, &pl->bins.seg, pl->seg_stride * G));
    PLCHK(dmalloc(c, &pl->bins.recs, nA * G));
    PLCHK(dmalloc(c, &pl->bins.recs_sorted, nA * G));
    PLCHK(dmalloc(c, &pl->bins.rowstart, pl->rowstart_stride * G)); // first sorted position of every (slice, species, row)
    pl->bins.tmp_bytes = geom_sort_temp_bytes((int)(nA * G));
    PLHIP(hipMalloc(&pl->bins.tmp, pl->bins.tmp_bytes > 0 ? pl->bins.tmp_bytes : 16));
    pl->gbins.assign(G, pl->bins);
    pl->gseg.assign(G, {});
    for (size_t g = 1; g < G; g++) {
        AtomBins& v = pl->gbins[g];
        v.keys += g * nA; v.keys_sorted += g * nA; v.vals += g * nA; v.order += g * nA; v.recs += g * nA; v.recs_sorted += g * nA;
        v.seg += g * pl->seg_stride;
        v.rowstart += g * pl->rowstart_stride;
    }
    {   // enough blocks for an average segment, capped; the kernel strides over the rest
        long avg = (long)nAt / (pl->p.m3 > 0 ? pl->p.m3 : 1) + 1;
        long b = (avg * 4 + 255) / 256;
        pl->deposit_blocks = (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
    }
    PLCHK(dmalloc(c, &pl->D, pl->m12));
    PLCHK(dmalloc(c, &pl->VH, pl->m12));
    PLCHK(dmalloc(c, &pl->T, pl->m12));
    PLCHK(dmalloc(c, &pl->PSI, pl->m12 * G));
    if (G > 1) { PLCHK(dmalloc(c, &pl->gscr, pl->m12 * G)); pl->gang_owned.push_back(pl->gscr); }
    PLCHK(dmalloc(c, &pl->P, pl->m12));
    PLCHK(dmalloc(c, &pl->I, pl->m12 * (pl->gang_k ? G : (size_t)1)));
    PLCHK(dmalloc(c, &pl->EW, pl->m12));
    PLCHK(dmalloc(c, &pl->J, (size_t)pl->p.n1 * pl->p.n2 * pl->p.n3));
    PLCHK(dmalloc(c, &pl->scal, (size_t)1056)); // k_normalize_to: the sum + 1024 block partials
    PLHIP(hipMemsetAsync(pl->D, 0, sizeof(float2) * pl->m12, c->stream));
    PLHIP(hipMemsetAsync(pl->I, 0, sizeof(float2) * pl->m12 * (pl->gang_k ? G : (size_t)1), c->stream));
    pl->Jout = pl->J;
    PLHIP(hipMemsetAsync(pl->EW, 0, sizeof(float2) * pl->m12, c->stream));
    PLHIP(hipMemsetAsync(pl->J, 0, sizeof(float) * (size_t)pl->p.n1 * pl->p.n2 * pl->p.n3, c->stream));
    {
        auto key = std::make_tuple(pl->p.m1, pl->p.m2, c->opt_fft);
        auto it = c->fft_cache.find(key);
        if (it == c->fft_cache.end()) {
            std::string ferr;
            Fft2D* f = new Fft2D();
            if (f->create(pl->p.m1, pl->p.m2, c->opt_fft, c->stream, &ferr) != 0) {
                f->destroy();
                delete f;
                c->err = "FFT plan: " + ferr;
                fdes_plan_destroy(pl);
                return FDES_EGPU;
            }
            it = c->fft_cache.emplace(key, f).first;
        }
        pl->fft = it->second;
    }
    PLHIP(k_build_propagator(pl->P, pl->kp, 0, c->stream));
    pl->fused = pl->fft->backend == 2;
    if (!pl->fused) { // filter table of the generic path (natural layout)
        PLCHK(dmalloc(c, &pl->GT, pl->m12 * (size_t)pl->nZ));
        for (int z = 0; z < pl->nZ; z++) PLHIP(k_build_gtab(pl->GT + (size_t)z * pl->m12, pl->kp, pl->kz[z], 0, 0, c->stream));
    }
    if (pl->fused) {
        const int m1 = pl->p.m1, m2 = pl->p.m2;
        const bool ok256 = (m2 % lds_fft_rows_per_block(m1, 256) == 0) && (m1 % lds_fft_rows_per_block(m2, 256) == 0) &&
                           lds_fft_rows_per_block(m1, 256) >= 4 && lds_fft_rows_per_block(m2, 256) >= 4; // >= 32-byte transposed segments
        // 256-thread workgroups (two per CU) measured faster or equal for every pass up to 2048-point rows, with one
        // or two lanes; 4096-point rows keep 512 threads (256 would cut the transposed-store segments to 16 bytes)
        // up to 1024^2 a pass is as long as its slowest workgroup: one row per thread, four rows per workgroup
        const bool small = m1 <= 1024 && m2 <= 1024;
        const bool wave_len = m1 == 2048 || m2 == 2048 || m1 == 1024 || m2 == 1024;
        if ((c->pass_threads == 64 || c->pass_threads == 65 || c->pass_threads == 128) && (wave_pass_supported_len(m1) || wave_pass_supported_len(m2))) pl->wg = c->pass_threads; // one wave per row where the row length has such a kernel
        else if (c->pass_threads == 512) pl->wg = 512;
        // 2048- and 1024-point rows: one wave per row (fft_wave.hip: one LDS exchange per transform, no barrier inside it; headline
        // +4 ... +6 % over 256 threads x 2 rows, 1024^2 +5 % over one row per thread, A/B on one box; shorter rows of a mixed
        // grid fall back to one row per thread); 4096-point rows: measured equal to 512 threads, which stay
        else if (c->pass_threads == 0 && wave_len && m1 <= 2048 && m2 <= 2048) pl->wg = 64;
        else if ((c->pass_threads == 1 || c->pass_threads == 513 || (c->pass_threads == 0 && small)) && m1 <= 2048 && m2 <= 2048) pl->wg = 1;
        else pl->wg = ok256 ? 256 : 512;
        // (mixed grids, e.g. 1000 x 512: the rows per workgroup of one axis must divide the other axis)
        if (m2 % lds_fft_rows_per_block(m1, pl->wg) != 0 || m1 % lds_fft_rows_per_block(m2, pl->wg) != 0) pl->wg = pl->fft->wg;
        // Slice-loop working set: the transient grids share buffers (A -> [P2] -> B; B -> [P3] -> C, C2; C | C2 -> [P4] -> E;
        // E, PSIH -> [P5] -> F; F -> [P6] -> PSIH: A, C and F are never live together, nor are B and E), and the
        // lanes share the read-only tables PT / GT: 4 grids per lane + the tables instead of 7.5 per lane, so that two
        // lanes at 2048^2 (304 MiB) mostly stay inside the 256 MiB Infinity Cache.  Dead (band-limited) rows of C / F
        // may hold stale data of the other tenant: P4 / P6 never read them.
        {
            // measured: 4096^2 +30 % with 64 elements (C5 2231 -> 2898 slice-propagations/s; 96 ... 320 equal, 32 as bad as
            // none); 2048^2 +1.4 % with 64, +2.7 % with 32 (six of seven A/B rounds); 8 or 16 elements are worse than none
            const int big = m1 > m2 ? m1 : m2;
            const int pad = c->pitch_pad >= 0 ? c->pitch_pad : (big >= 4096 ? 64 : (big >= 2048 ? 32 : 0));
            pl->pitchN = m1 + pad;
            pl->pitchT = m2 + pad;
            const size_t gn = (size_t)pl->pitchN * (size_t)m2, gt = (size_t)pl->pitchT * (size_t)m1;
            pl->gsz = gn > gt ? gn : gt;
        }
        // split: the potential chain of the next slice pair runs beside the wave chain of this one, so the buffers that
        // one stream shares between its own consecutive passes stay aliased (A and C) and those that cross streams do
        // not (B, two E, F): 7 grids per lane instead of 4
        // (auto: one-lane plans from 2^20 pixels on; below that the single-stream hipGraph wins: 256^2 x 32 slices 35.5 k
        // against 27.4 k slice-propagations/s, 512^2 x 32 27.6 k against 23.4 k, 1024^2 x 32 17.6 k against 19.8 k)
        pl->split = c->split > 0 || (c->split < 0 && plan_lanes(c, pl) == 1 && pl->m12 >= ((size_t)1 << 20));
        // batched potential chain: one-lane plans whose slices cannot fill the chip by themselves (up to 2^20 pixels; a
        // any fused grid; not with the pipelined one-wave-per-row kernels, which take no batches)
        {
            int nb = 1;
            if ((plan_lanes(c, pl) == 1 || c->batch > 1) && c->split != 0 && c->pass_threads != 65) { // (an explicit batch also applies to the lanes of a multi-configuration plan: measured)
                if (c->batch > 1) nb = c->batch;
                else if (c->batch < 0) nb = pl->m12 <= ((size_t)1 << 18) ? 8 : (pl->m12 <= ((size_t)1 << 20) ? 4 : 1); // measured (tools/bench_single.py): 512^2 8 > 4 > 2; 1024^2 4 >= 2, 8 lower
            }
            if (pl->gang > 1) { nb = 1; pl->split = false; } // a gang fills the launches with configurations instead
            pl->nb = nb;
            if (nb > 1) pl->split = true;
        }
        PLCHK(dmalloc(c, &pl->C, pl->gsz * G));
        if (pl->split) PLCHK(dmalloc(c, &pl->F, pl->gsz)); else pl->F = pl->C;
        if (pl->nZ == 1) pl->A = pl->C;
        else PLCHK(dmalloc(c, &pl->A, pl->gsz * (size_t)pl->nZ * G));
        PLCHK(dmalloc(c, &pl->C2, pl->gsz * G)); // x-spectrum of the transmission function of the pair's second slice
        PLCHK(dmalloc(c, &pl->E, pl->gsz * G));
        pl->Eb[0] = pl->Eb[1] = pl->E;
        if (pl->split) {
            PLCHK(dmalloc(c, &pl->Eb[1], pl->gsz));
            PLCHK(dmalloc(c, &pl->B, pl->gsz));
            PLHIP(hipMemsetAsync(pl->Eb[1], 0, sizeof(float2) * pl->gsz, c->stream));
            PLHIP(hipMemsetAsync(pl->F, 0, sizeof(float2) * pl->gsz, c->stream));
            // same priority as the wave stream: with a priority class of its own (greatest or least) the captured graph
            // of the two-stream loop runs at half the rate (6.2 k against 12.2 k slice-propagations/s at 2048^2)
            PLHIP(hipStreamCreateWithFlags(&pl->vs, hipStreamNonBlocking));
            for (hipEvent_t* e : {&pl->evE[0], &pl->evE[1], &pl->evP5[0], &pl->evP5[1], &pl->evFork, &pl->evJoin})
                PLHIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        } else {
            pl->B = pl->E; // the packed pair potential is consumed by P3 before P4 writes E
        }
        if (pl->nb > 1) {
            const size_t nbz = (size_t)pl->nb;
            PLCHK(dmalloc(c, &pl->bA, pl->gsz * nbz * (size_t)pl->nZ));
            PLCHK(dmalloc(c, &pl->bB, pl->gsz * nbz));
            PLCHK(dmalloc(c, &pl->bCC, pl->gsz * 2 * nbz));
            for (int q = 0; q < 2; q++) {
                PLCHK(dmalloc(c, &pl->bE[q], pl->gsz * 2 * nbz));
                PLHIP(hipMemsetAsync(pl->bE[q], 0, sizeof(float2) * pl->gsz * 2 * nbz, c->stream)); // dead kx columns read as zero
                PLHIP(hipEventCreateWithFlags(&pl->evReady[q], hipEventDisableTiming));
                PLHIP(hipEventCreateWithFlags(&pl->evDone[q], hipEventDisableTiming));
            }
        }
        PLCHK(dmalloc(c, &pl->PSIH, pl->gsz * G));
        if (c->share_PT) { pl->PT = c->share_PT; pl->GT = c->share_GT; pl->tables_shared = true; }
        else {
            PLCHK(dmalloc(c, &pl->PT, (size_t)pl->p.m1 + (size_t)pl->p.m2));
            PLCHK(dmalloc(c, &pl->GT, pl->gsz * (size_t)pl->nZ));
        }
        // dead (band-limited) rows / columns of these grids are never written again: they must read as zero
        for (float2* q : {pl->C, pl->C2, pl->E, pl->PSIH}) PLHIP(hipMemsetAsync(q, 0, sizeof(float2) * pl->gsz * G, c->stream));
        if (!pl->tables_shared) {
            PLHIP(k_build_propagator_1d(pl->PT, pl->PT + pl->p.m1, pl->kp, 1, c->stream));
            PLHIP(hipMemsetAsync(pl->GT, 0, sizeof(float) * pl->gsz * (size_t)pl->nZ, c->stream));
            for (int z = 0; z < pl->nZ; z++) PLHIP(k_build_gtab(pl->GT + (size_t)z * pl->gsz, pl->kp, pl->kz[z], 1, pl->pitchT, c->stream));
        }
    }
    // tilt offset (src/crystalMaker.cu:282-283)
    PLHIP(hipMemcpyAsync(pl->xyzTO_d, pl->xyz0_d, sizeof(float) * n3f, hipMemcpyDeviceToDevice, c->stream));
    PLCHK(tilt_coordinates(pl, pl->xyzTO_d, pl->p.tilt_offset_x, pl->p.tilt_offset_y, pl->p.tilt_offset_z));
    PLHIP(hipStreamSynchronize(c->stream));
    // lanes hide the gap between dependent kernels of one stream (about 8 us on this part): three up to 1024^2, where
    // the kernels are no longer than that gap (each lane on a stream of its own priority class, see create_ctx), two
    // above (a third lane only thrashes the Infinity Cache at 2048^2; equal priorities there: a high-priority lane
    // starves the other one of workgroup slots, -1.5 %).
    // (the rocFFT path stays at two: with three prioritised lanes it drops from 9.2 k to 3.6 k slices/s at 800^2)
    const int nlanes = plan_lanes(c, pl);
    if (nlanes > 1 && !c->is_lane_ctx) {
        for (int l = 1; l < nlanes; l++) {
            fdes_ctx* lc = nullptr;
            PLCHK(create_ctx(&lc, c->device, nlanes >= 3 ? l % 3 : 0)); // (every assignment of the three classes to a fourth lane measured the same)
            lc->is_lane_ctx = true;
            // frozen here: fft, lanes, pass_threads (they shape the lane plan); the others are read through owner_ctx()
            lc->opt_fft = c->opt_fft; lc->opt_graph = c->opt_graph; lc->seed = c->seed; lc->probe_stride = c->probe_stride; lc->probe_pass = c->probe_pass; lc->pass_threads = c->pass_threads; lc->lanes = c->lanes; lc->skip_empty = c->skip_empty; lc->band_skip = c->band_skip; lc->pitch_pad = c->pitch_pad; lc->split = pl->split ? 1 : 0; lc->batch = c->batch > 1 ? c->batch : 0; lc->gang = pl->gang; lc->walk = c->walk;
            lc->share_PT = pl->PT; lc->share_GT = pl->GT; // read-only tables of the parent plan (built and synchronised above)
            pl->lane_ctx.push_back(lc);
            fdes_plan* lp = nullptr;
            int lrc = fdes_plan_create(lc, p_in, a, &lp);
            if (lrc != FDES_OK) { c->err = "lane plan: " + lc->err; fdes_plan_destroy(pl); return lrc; }
            lp->is_lane = true;
            lp->parent_ctx = c;
            lp->top = pl;
            pl->lanes.push_back(lp);
            hipEvent_t ev;
            PLHIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            pl->lane_ev.push_back(ev);
        }
    }
#undef PLCHK
#undef PLHIP
    *out = pl;
    return FDES_OK;
}

int fdes_plan_begin_measurement(fdes_plan* pl, int k)
{
    if (!live_plan(pl) || k < 0 || k >= pl->p.n3) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!pl->is_lane) {
        RC(gang_flush_all(pl));
        if (pl->gang > 1) pl->rr = (pl->rr + (unsigned)pl->gang - 1) / (unsigned)pl->gang * (unsigned)pl->gang; // a measurement starts a new gang (on the next lane)
    }
    HIPCHK(c, k_fill(pl->I, pl->m12, 0.f, 0.f, c->stream));
    if (pl->want_ew) HIPCHK(c, k_fill(pl->EW, pl->m12, 0.f, 0.f, c->stream));
    for (fdes_plan* l : pl->lanes) { l->want_ew = pl->want_ew; RC(fdes_plan_begin_measurement(l, k)); }
    return ensure_tilt(pl, k);
}

int fdes_plan_run_config(fdes_plan* pl, int k, int j, float weight)
{
    if (!live_plan(pl) || k < 0 || k >= pl->p.n3 || j < 0) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!pl->lanes.empty()) {
        unsigned nl = (unsigned)(pl->lanes.size() + 1);
        if (c->lanes_active > 0 && (unsigned)c->lanes_active < nl) nl = (unsigned)c->lanes_active;
        // round robin; with gangs a lane is dealt configurations until its gang is full
        const unsigned lane = (pl->gang > 1 ? pl->rr++ / (unsigned)pl->gang : pl->rr++) % nl;
        if (lane > 0) {
            pl->lanes_dirty = true;
            int rcl = fdes_plan_run_config(pl->lanes[lane - 1], k, j, weight);
            if (rcl != FDES_OK) c->err = "lane: " + pl->lanes[lane - 1]->ctx->err;
            return rcl;
        }
    }
    if (pl->gang > 1 && pl->fused && !pl->tap_mode && owner_ctx(pl)->probe_stride <= 0) {
        if (!pl->gq.empty() && pl->gq[0].k != k) RC(gang_flush(pl));
        pl->gq.push_back({k, j, weight, 0});
        return (int)pl->gq.size() >= pl->gang ? gang_flush(pl) : FDES_OK;
    }
    RC(incoming_wave(pl, k));
    RC(config_atoms(pl, k, j));
    if (pl->ev_used == pl->evs.size()) {
        EvPair e{};
        HIPCHK(c, hipEventCreate(&e.a));
        HIPCHK(c, hipEventCreate(&e.b));
        pl->evs.push_back(e);
    }
    EvPair& ev = pl->evs[pl->ev_used++];
    ev.slices = pl->p.m3;
    HIPCHK(c, hipEventRecord(ev.a, c->stream));
    RC(slice_loop(pl, pl->p.m3));
    HIPCHK(c, hipEventRecord(ev.b, c->stream));
    pl->slices_done += pl->p.m3;
    return exit_wave_post(pl, k, weight);
}

int fdes_plan_end_measurement(fdes_plan* pl, int k)
{
    if (!live_plan(pl) || k < 0 || k >= pl->p.n3) return FDES_EINVAL;
    HIPCHK(pl->ctx, hipSetDevice(pl->ctx->device));
    RC(fold_lanes(pl));
    return finalize_measurement(pl, k);
}

int fdes_plan_intensity_ptr(fdes_plan* pl, void** dev_ptr, size_t* bytes)
{
    if (!live_plan(pl) || !dev_ptr) return FDES_EINVAL;
    // queued gang members are issued and the lanes' partial sums folded into I first (stream-ordered; the caller
    // synchronises with fdes_plan_sync before touching the memory): an in-place reduce through this pointer would
    // otherwise miss them, and end_measurement would add them AFTER the reduce
    HIPCHK(pl->ctx, hipSetDevice(pl->ctx->device));
    RC(fold_lanes(pl));
    *dev_ptr = pl->I;
    if (bytes) *bytes = sizeof(float2) * pl->m12;
    return FDES_OK;
}

int fdes_plan_copy_intensity(fdes_plan* pl, void* dev_buf, int to_plan)
{
    if (!live_plan(pl) || !dev_buf) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    RC(fold_lanes(pl));
    // hipMemcpyDefault: `dev_buf` may be device memory (RCCL buffers of the one-process-per-GPU launch) or host memory
    if (to_plan) HIPCHK(c, hipMemcpyAsync(pl->I, dev_buf, sizeof(float2) * pl->m12, hipMemcpyDefault, c->stream));
    else HIPCHK(c, hipMemcpyAsync(dev_buf, pl->I, sizeof(float2) * pl->m12, hipMemcpyDefault, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

int fdes_plan_copy_intensity_real(fdes_plan* pl, void* dev_buf, int to_plan)
{
    if (!live_plan(pl) || !dev_buf) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    RC(fold_lanes(pl));
    // dev_buf must be DEVICE memory here (a kernel reads / writes it); I.y is identically zero (k_intensity_axpy)
    if (to_plan) HIPCHK(c, k_real_unpack(pl->I, (const float*)dev_buf, pl->m12, c->stream));
    else HIPCHK(c, k_real_pack((float*)dev_buf, pl->I, pl->m12, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

int fdes_plan_images_ptr(fdes_plan* pl, void** dev_ptr, size_t* bytes)
{
    if (!live_plan(pl) || !dev_ptr) return FDES_EINVAL;
    *dev_ptr = pl->J;
    if (bytes) *bytes = sizeof(float) * (size_t)pl->p.n1 * pl->p.n2 * pl->p.n3;
    return FDES_OK;
}

int fdes_plan_sync(fdes_plan* pl)
{
    if (!live_plan(pl)) return FDES_EINVAL;
    HIPCHK(pl->ctx, hipSetDevice(pl->ctx->device));
    RC(gang_flush_all(pl));
    for (fdes_plan* l : pl->lanes) { if (l->vs) HIPCHK(pl->ctx, hipStreamSynchronize(l->vs)); HIPCHK(pl->ctx, hipStreamSynchronize(l->ctx->stream)); }
    if (pl->vs) HIPCHK(pl->ctx, hipStreamSynchronize(pl->vs));
    HIPCHK(pl->ctx, hipStreamSynchronize(pl->ctx->stream));
    return FDES_OK;
}

int fdes_plan_get_images(fdes_plan* pl, float* image)
{
    if (!live_plan(pl) || !image) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(image, pl->J, sizeof(float) * (size_t)pl->p.n1 * pl->p.n2 * pl->p.n3, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

int fdes_grid_backend(int m1, int m2, int fft_option)
{
    if (m1 < 1 || m2 < 1) return FDES_EINVAL;
    return (fft_option != 1 && Fft2D::lds_supported(m1, m2)) ? 2 : 1;
}
int fdes_plan_fft_backend(const fdes_plan* pl) { return live_plan(pl) ? pl->fft->backend : FDES_EINVAL; }
int fdes_plan_lanes(const fdes_plan* pl) { return live_plan(pl) ? (int)pl->lanes.size() + 1 : FDES_EINVAL; }
int fdes_plan_gang(const fdes_plan* pl) { return live_plan(pl) ? pl->gang : FDES_EINVAL; }
int fdes_plan_num_slices(const fdes_plan* pl) { return live_plan(pl) ? pl->p.m3 : FDES_EINVAL; }
int64_t fdes_plan_empty_queries(const fdes_plan* pl) { return live_plan(pl) ? (pl->top ? pl->top : pl)->empty_queries : 0; }

int64_t fdes_plan_slices_done(const fdes_plan* pl)
{
    if (!live_plan(pl)) return 0;
    int64_t n = pl->slices_done;
    for (const fdes_plan* l : pl->lanes) n += l->slices_done;
    return n;
}

int fdes_plan_slice_loop_ms(fdes_plan* pl, double* total_ms, int64_t* slices)
{
    if (!live_plan(pl)) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    RC(gang_flush(pl));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double t = 0;
    int64_t n = 0;
    for (size_t i = 0; i < pl->ev_used; i++) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, pl->evs[i].a, pl->evs[i].b));
        t += ms;
        n += pl->evs[i].slices;
    }
    pl->ev_used = 0;
    pl->ev_done = 0;
    pl->cfg_done = 0;
    for (fdes_plan* l : pl->lanes) {
        double tl = 0;
        int64_t nl = 0;
        RC(fdes_plan_slice_loop_ms(l, &tl, &nl));
        t += tl;
        n += nl;
    }
    if (total_ms) *total_ms = t;
    if (slices) *slices = n;
    return FDES_OK;
}

int fdes_plan_probe_ms(fdes_plan* pl, double* total_ms, int64_t* launches)
{
    if (!live_plan(pl)) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double t = 0;
    for (size_t i = 0; i < pl->probe_used; i++) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, pl->probe[i].a, pl->probe[i].b));
        t += ms;
    }
    int64_t nl = (int64_t)pl->probe_used;
    pl->probe_used = 0;
    for (fdes_plan* l : pl->lanes) {
        double tl = 0;
        int64_t ll = 0;
        RC(fdes_plan_probe_ms(l, &tl, &ll));
        t += tl;
        nl += ll;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = nl;
    return FDES_OK;
}

int fdes_plan_want_exitwave(fdes_plan* pl, int on)
{
    if (!live_plan(pl)) return FDES_EINVAL;
    pl->want_ew = on != 0;
    return FDES_OK;
}

int fdes_plan_get_exitwave(fdes_plan* pl, float* ew)
{
    if (!live_plan(pl) || !ew || !pl->want_ew) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    RC(fold_lanes(pl));
    HIPCHK(c, hipMemcpyAsync(ew, pl->EW, sizeof(float2) * pl->m12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

// The sum over configurations of src/crystalMaker.cu:347-365 when a measurement spans GPUs: dst.I += src.I (and the
// coherent exit-wave sum when wanted), device to device.  The source's partial sum crosses xGMI once
// (hipMemcpyPeerAsync into a landing buffer on dst's GPU, ordered behind src's stream by an event) and is added by one
// axpy kernel on dst's stream; on one GPU the axpy reads the source directly.  Synchronises dst's stream, so the
// caller may let src continue afterwards.
int fdes_plan_accumulate_from(fdes_plan* dst, fdes_plan* src)
{
    if (!live_plan(dst) || !live_plan(src) || dst == src || dst->m12 != src->m12) return FDES_EINVAL;
    fdes_ctx *dc = dst->ctx, *sc = src->ctx;
    const bool forced_host = owner_ctx(dst)->peer_copy == 0; // test option: take the host-staged path even on one device
    if (forced_host) dst->peer_host_only = true;
    const bool same = dc->device == sc->device && !forced_host;
    HIPCHK(sc, hipSetDevice(sc->device));
    RC(fold_lanes(src));
    if (!same) {
        // the intensity sum travels as its real view (I.y is identically zero): half the bytes over xGMI
        if (!src->real_send) {
            DeviceGuard guard(sc->device); // hipMalloc vs a capture in another thread of that device
            RC(dmalloc(sc, &src->real_send, src->m12));
        }
        HIPCHK(sc, k_real_pack(src->real_send, src->I, src->m12, sc->stream));
    }
    if (!src->peer_ev) HIPCHK(sc, hipEventCreateWithFlags(&src->peer_ev, hipEventDisableTiming));
    HIPCHK(sc, hipEventRecord(src->peer_ev, sc->stream));
    HIPCHK(dc, hipSetDevice(dc->device));
    RC(fold_lanes(dst));
    HIPCHK(dc, hipStreamWaitEvent(dc->stream, src->peer_ev, 0));
    if (!same && !dst->peer_stage) {
        DeviceGuard guard(dc->device); // hipMalloc vs a capture in another thread of that device
        RC(dmalloc(dc, &dst->peer_stage, dst->m12));
    }
    const int nsum = (dst->want_ew && src->want_ew) ? 2 : 1;
    for (int q = 0; q < nsum; q++) { // q = 0: intensity (float view between devices); q = 1: coherent exit-wave sum (complex)
        float2* acc = q ? dst->EW : dst->I;
        if (same) {
            HIPCHK(dc, k_axpy(acc, q ? src->EW : src->I, dst->m12, 1.f, dc->stream));
            continue;
        }
        const void* part = q ? (const void*)src->EW : (const void*)src->real_send;
        const size_t bytes = (q ? sizeof(float2) : sizeof(float)) * dst->m12;
        // xGMI peer copy; when the runtime refuses it (no peer access between the two devices, or the copy itself
        // fails) the partial sum is staged through host memory instead - slower, never wrong
        int can = 0;
        hipError_t pe = hipDeviceCanAccessPeer(&can, dc->device, sc->device);
        if (pe == hipSuccess && can && !dst->peer_host_only)
            pe = hipMemcpyPeerAsync(dst->peer_stage, dc->device, part, sc->device, bytes, dc->stream);
        else if (pe == hipSuccess) pe = hipErrorPeerAccessUnsupported;
        if (pe != hipSuccess) {
            (void)hipGetLastError();
            dst->peer_host_only = true;
            dst->peer_host.resize(dst->m12);
            HIPCHK(sc, hipSetDevice(sc->device));
            HIPCHK(sc, hipMemcpyAsync(dst->peer_host.data(), part, bytes, hipMemcpyDeviceToHost, sc->stream));
            HIPCHK(sc, hipStreamSynchronize(sc->stream));
            HIPCHK(dc, hipSetDevice(dc->device));
            HIPCHK(dc, hipMemcpyAsync(dst->peer_stage, dst->peer_host.data(), bytes, hipMemcpyHostToDevice, dc->stream));
            HIPCHK(dc, hipStreamSynchronize(dc->stream)); // peer_host is reused by the next sum
        }
        if (q) HIPCHK(dc, k_axpy(acc, dst->peer_stage, dst->m12, 1.f, dc->stream));
        else HIPCHK(dc, k_axpy_real(acc, reinterpret_cast<const float*>(dst->peer_stage), dst->m12, dc->stream));
    }
    HIPCHK(dc, hipStreamSynchronize(dc->stream));
    return FDES_OK;
}

// ---- RCCL: the reduction of SURVEY 8e / src/crystalMaker.cu:347-365 as ONE collective ------------------------------
// librccl.so is resolved at run time (dlopen, like libhdf5 in emd.cpp): the library has no link-time dependency on it, and
// a host that never creates a communicator never loads it.
namespace {
struct Rccl {
    void* so = nullptr;
    bool ok = false;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, fdes_comm_id, int) = nullptr; // ncclUniqueId is passed BY VALUE: a 128-byte struct
    int (*CommDestroy)(void*) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
        if (!r.so) return;
        *(void**)(&r.GetUniqueId) = dlsym(r.so, "ncclGetUniqueId");
        *(void**)(&r.CommInitRank) = dlsym(r.so, "ncclCommInitRank");
        *(void**)(&r.CommDestroy) = dlsym(r.so, "ncclCommDestroy");
        *(void**)(&r.Reduce) = dlsym(r.so, "ncclReduce");
        *(void**)(&r.GetErrorString) = dlsym(r.so, "ncclGetErrorString");
        *(void**)(&r.Send) = dlsym(r.so, "ncclSend");
        *(void**)(&r.Recv) = dlsym(r.so, "ncclRecv");
        *(void**)(&r.GroupStart) = dlsym(r.so, "ncclGroupStart");
        *(void**)(&r.GroupEnd) = dlsym(r.so, "ncclGroupEnd");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Reduce && r.GetErrorString && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    });
    return r;
}
} // namespace

struct fdes_comm {
    fdes_ctx* ctx = nullptr; // identity only (fdes_plan_reduce_intensity checks that plan and communicator belong together)
    void* comm = nullptr; // ncclComm_t
    int nranks = 0, rank = -1;
    int device = 0;       // what fdes_comm_destroy needs, kept here: the context may be gone by then
};

int fdes_comm_unique_id(fdes_comm_id* id)
{
    if (!id) return FDES_EINVAL;
    Rccl& r = rccl();
    if (!r.ok) return FDES_EUNSUPPORTED;
    return r.GetUniqueId(id) == 0 ? FDES_OK : FDES_EGPU;
}

int fdes_comm_create(fdes_ctx* c, int nranks, int rank, const fdes_comm_id* id, fdes_comm** out)
{
    if (!c || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return FDES_EINVAL;
    *out = nullptr;
    Rccl& r = rccl();
    if (!r.ok) { c->err = "librccl.so could not be loaded"; return FDES_EUNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    void* comm = nullptr;
    const int e = r.CommInitRank(&comm, nranks, *id, rank); // blocks until every rank has joined
    if (e != 0 || !comm) { c->err = std::string("ncclCommInitRank: ") + r.GetErrorString(e); return FDES_EGPU; }
    fdes_comm* k = new fdes_comm;
    k->ctx = c; k->comm = comm; k->nranks = nranks; k->rank = rank; k->device = c->device;
    *out = k;
    return FDES_OK;
}

int fdes_comm_destroy(fdes_comm* k)
{
    if (!k) return FDES_EINVAL;
    if (k->comm) {
        (void)hipSetDevice(k->device);
        (void)hipDeviceSynchronize(); // the collectives were enqueued on the context's stream; the context may have been destroyed already
        (void)rccl().CommDestroy(k->comm);
    }
    delete k;
    return FDES_OK;
}

int fdes_plan_reduce_intensity(fdes_plan* pl, fdes_comm* k, int root)
{
    if (!live_plan(pl) || !k || !k->comm || root < 0 || root >= k->nranks || pl->ctx != k->ctx) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    Rccl& r = rccl();
    HIPCHK(c, hipSetDevice(c->device));
    RC(fold_lanes(pl));
    if (!pl->real_send || (k->rank == root && !pl->peer_stage)) {
        DeviceGuard guard(c->device); // hipMalloc vs a capture in another thread of this device
        if (!pl->real_send) RC(dmalloc(c, &pl->real_send, pl->m12));
        if (k->rank == root && !pl->peer_stage) RC(dmalloc(c, &pl->peer_stage, pl->m12));
    }
    HIPCHK(c, k_real_pack(pl->real_send, pl->I, pl->m12, c->stream));
    float* recv = k->rank == root ? reinterpret_cast<float*>(pl->peer_stage) : pl->real_send; // (only the root's is written)
    const int e = r.Reduce(pl->real_send, recv, pl->m12, /* ncclFloat32 */ 7, /* ncclSum */ 0, root, k->comm, c->stream);
    if (e != 0) { c->err = std::string("ncclReduce: ") + r.GetErrorString(e); return FDES_EGPU; }
    if (k->rank == root) HIPCHK(c, k_real_unpack(pl->I, recv, pl->m12, c->stream));
    if (pl->want_ew) { // the coherent exit-wave sum of print_level 2 (src/crystalMaker.cu:347, 370) is complex: 2 m12 floats, in place
        const int e2 = r.Reduce(pl->EW, pl->EW, 2 * pl->m12, /* ncclFloat32 */ 7, /* ncclSum */ 0, root, k->comm, c->stream);
        if (e2 != 0) { c->err = std::string("ncclReduce (exit wave): ") + r.GetErrorString(e2); return FDES_EGPU; }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

// The same sum for a measurement whose configurations sit on the ranks lo .. hi of the communicator only (a series dealt over
// the GPUs: most measurements span two or three of them): no collective - the ranks outside the span take no part - but one
// group of point-to-point transfers, every rank of the span sending the float view of its sum (and its exit-wave sum) to
// `root`, which receives them side by side (each peer has an xGMI link of its own to the root) and adds them in rank order,
// so that the result does not depend on arrival order.
int fdes_plan_reduce_intensity_span(fdes_plan* pl, fdes_comm* k, int root, int lo, int hi)
{
    if (!live_plan(pl) || !k || !k->comm || lo < 0 || hi >= k->nranks || lo > hi || root < lo || root > hi || k->rank < lo || k->rank > hi || pl->ctx != k->ctx)
        return FDES_EINVAL;
    if (lo == 0 && hi == k->nranks - 1) return fdes_plan_reduce_intensity(pl, k, root);
    fdes_ctx* c = pl->ctx;
    Rccl& r = rccl();
    HIPCHK(c, hipSetDevice(c->device));
    RC(fold_lanes(pl));
    const size_t m12 = pl->m12, per = pl->want_ew ? 3 * m12 : m12; // floats per peer: intensity view [+ complex exit wave]
    const int npeer = hi - lo; // senders
    if (k->rank == root) {
        if (pl->span_stage_n < (size_t)npeer * per) {
            DeviceGuard guard(c->device);
            if (pl->span_stage) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(pl->span_stage); pl->span_stage = nullptr; }
            RC(dmalloc(c, &pl->span_stage, (size_t)npeer * per));
            pl->span_stage_n = (size_t)npeer * per;
        }
        int e = r.GroupStart();
        int slot = 0;
        for (int q = lo; q <= hi && e == 0; q++) {
            if (q == root) continue;
            e = r.Recv(pl->span_stage + (size_t)slot * per, per, /* ncclFloat32 */ 7, q, k->comm, c->stream);
            slot++;
        }
        const int e2 = r.GroupEnd();
        if (e != 0 || e2 != 0) { c->err = std::string("ncclRecv: ") + r.GetErrorString(e ? e : e2); return FDES_EGPU; }
        for (int i = 0; i < npeer; i++) { // fixed association order: ascending rank
            const float* part = pl->span_stage + (size_t)i * per;
            HIPCHK(c, k_axpy_real(pl->I, part, m12, c->stream));
            if (pl->want_ew) HIPCHK(c, k_axpy(pl->EW, reinterpret_cast<const float2*>(part + m12), m12, 1.f, c->stream));
        }
    } else {
        if (pl->span_send_n < per) {
            DeviceGuard guard(c->device);
            if (pl->span_send) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(pl->span_send); pl->span_send = nullptr; }
            RC(dmalloc(c, &pl->span_send, per));
            pl->span_send_n = per;
        }
        HIPCHK(c, k_real_pack(pl->span_send, pl->I, m12, c->stream));
        if (pl->want_ew) HIPCHK(c, hipMemcpyAsync(pl->span_send + m12, pl->EW, sizeof(float2) * m12, hipMemcpyDeviceToDevice, c->stream));
        const int e = r.Send(pl->span_send, per, /* ncclFloat32 */ 7, root, k->comm, c->stream);
        if (e != 0) { c->err = std::string("ncclSend: ") + r.GetErrorString(e); return FDES_EGPU; }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

// Potential output of print_level > 0 (src/crystalMaker.cu:381-397): tilt-offset-only, un-jittered potential of the
// ORIGINAL slices [s_lo, s_hi) (setSubSlices(1 / ratio)) into potential[(s - s_lo) * 2 m1 m2 ...].  Always computed
// (the reference leaves it uninitialised when ratio == 1, frPh == 0 and the last specimen tilt is zero).
int fdes_plan_potential(fdes_plan* pl, int s_lo, int s_hi, float* potential)
{
    if (!live_plan(pl) || !potential) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    const float inv = 1.f / (float)pl->ratio;
    BinGeom g{pl->p.m1, pl->p.m2, (int)(((float)pl->p.m3) * inv), pl->nZ, pl->p.d1, pl->p.d2, pl->p.d3 / inv};
    if (s_lo < 0 || s_hi > g.m3 || s_lo > s_hi) return FDES_EINVAL;
    HIPCHK(c, geom_bin_atoms(pl->xyzTO_d, pl->spec_d, pl->occ_d, pl->nAt, g, pl->bins, owner_ctx(pl)->deterministic != 0, c->stream));
    for (int s = s_lo; s < s_hi; s++) {
        RC(phase_grating(pl, pl->xyzTO_d, g, s));
        HIPCHK(c, hipMemcpyAsync(potential + 2 * pl->m12 * (size_t)(s - s_lo), pl->VH, sizeof(float2) * pl->m12, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return FDES_OK;
}

int fdes_plan_original_slices(const fdes_plan* pl) { return live_plan(pl) ? (int)(((float)pl->p.m3) * (1.f / (float)pl->ratio)) : FDES_EINVAL; }

// ------------------------------- stage taps (parity tests) -------------------------------------

int fdes_plan_tap_coords(fdes_plan* pl, int k, int j, float* xyz)
{
    if (!live_plan(pl) || !xyz || k >= pl->p.n3) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    const float* src = pl->xyzTO_d;
    if (k >= 0) {
        RC(ensure_tilt(pl, k));
        src = pl->xyzK_d;
        if (j >= 0) { RC(config_atoms(pl, k, j)); src = pl->xyzFP_d; }
    }
    HIPCHK(c, hipMemcpyAsync(xyz, src, sizeof(float) * 3 * (size_t)pl->nAt, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

int fdes_plan_tap_potential(fdes_plan* pl, int k, int j, int s, float* V)
{
    if (!live_plan(pl) || !V || k < 0 || k >= pl->p.n3 || s < 0 || s >= pl->p.m3) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    RC(config_atoms(pl, k, j < 0 ? 0 : j));
    BinGeom g{pl->p.m1, pl->p.m2, pl->p.m3, pl->nZ, pl->p.d1, pl->p.d2, pl->p.d3};
    if (pl->fused) {
        pl->tap_mode = true; // everything on the context's stream
        const int rcp = fused_potential_pair(pl, s & ~1);
        pl->tap_mode = false;
        RC(rcp);
        PassArgs a = pass_x(pl);
        a.in0 = pl->B; a.out = pl->T; a.pitch_out = 0;
        HIPCHK(c, lds_pass(pl->p.m1, XF_INV, MID_NONE, XF_NONE, false, a, c->stream));
        HIPCHK(c, k_pick_potential(pl->VH, pl->T, pl->m12, s & 1, pl->p.imPot, c->stream));
    } else {
        RC(phase_grating(pl, pl->xyzFP_d, g, s));
    }
    HIPCHK(c, hipMemcpyAsync(V, pl->VH, sizeof(float2) * pl->m12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

int fdes_plan_tap_wave(fdes_plan* pl, int k, int j, int nslices, float* psi)
{
    if (!live_plan(pl) || !psi || k < 0 || k >= pl->p.n3 || nslices < 0 || nslices > pl->p.m3) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    RC(incoming_wave(pl, k));
    RC(config_atoms(pl, k, j < 0 ? 0 : j));
    RC(slice_loop(pl, nslices));
    HIPCHK(c, hipMemcpyAsync(psi, pl->PSI, sizeof(float2) * pl->m12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

int fdes_plan_tap_propagator(fdes_plan* pl, float* P)
{
    if (!live_plan(pl) || !P) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(P, pl->P, sizeof(float2) * pl->m12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FDES_OK;
}

int fdes_plan_propagate_dev(fdes_plan* pl, void* psi_dev, const void* t_dev, int batch, int t_per_wave)
{
    if (!live_plan(pl) || !psi_dev || !t_dev || batch < 1) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (pl->fused) {
        // The propagation unit psi <- F^-1[P F[t psi]] as three row passes (SURVEY 8d's 80 B/px unit; here 24 + 24 + 16
        // = 64 B/px before band-limit bookkeeping): rows y: t psi, FFT_x -> [kx][y]; rows kx: FFT_y, P, IFFT_y -> [y][kx];
        // rows y: IFFT_x.  The x round trip is unnormalised (m1) and P carries 1 / (m1 m2): the last pass scales by 1.
        const int m1 = pl->p.m1, m2 = pl->p.m2;
        const int md = m1 < m2 ? m1 : m2, band = md * md;
        const int bs = (owner_ctx(pl)->band_skip && m1 == m2) ? 1 : 0;
        for (int b = 0; b < batch; b++) {
            float2* psi = (float2*)psi_dev + (size_t)b * pl->m12;
            const float2* t = (const float2*)t_dev + (t_per_wave ? (size_t)b * pl->m12 : 0);
            PassArgs a5 = pass_x(pl);
            a5.in0 = t; a5.in1 = psi; a5.out = pl->F; a5.pitch_in = 0; // caller's dense grids
            a5.band = band; a5.skip_dead_stores = bs;
            HIPCHK(c, lds_pass(m1, XF_NONE, MID_MULPSI, XF_FWD, true, a5, c->stream));
            PassArgs a6 = pass_y(pl);
            a6.in0 = pl->F; a6.prow = pl->PT; a6.pcol = pl->PT + m1; a6.mindim = md; a6.out = pl->E;
            a6.band = band; a6.live_rows_only = bs;
            HIPCHK(c, lds_pass(m2, XF_FWD, MID_PTAB, XF_INV, true, a6, c->stream));
            PassArgs a7 = pass_x(pl);
            a7.in0 = pl->E; a7.out = psi; a7.scale = 1.f; a7.pitch_out = 0;
            if (bs) { a7.band = band; a7.skip_dead_loads = 1; } // dead kx columns of E are never written: they count as zero
            HIPCHK(c, lds_pass(m1, XF_INV, MID_SCALE, XF_NONE, false, a7, c->stream));
        }
        return FDES_OK;
    }
    for (int b = 0; b < batch; b++) {
        float2* psi = (float2*)psi_dev + (size_t)b * pl->m12;
        const float2* t = (const float2*)t_dev + (t_per_wave ? (size_t)b * pl->m12 : 0);
        HIPCHK(c, k_mul(psi, t, psi, pl->m12, c->stream));
        HIPCHK(c, fft_exec(pl,psi, false, c->stream));
        HIPCHK(c, k_mul(psi, psi, pl->P, pl->m12, c->stream));
        HIPCHK(c, fft_exec(pl,psi, true, c->stream));
    }
    return FDES_OK;
}

// 2-D FFT of a host grid through the engine's FFT back-end (test hook for the FFT itself).
int fdes_fft2d_host(fdes_ctx* c, float* data, int m1, int m2, int inverse, int backend)
{
    if (!c || !data || m1 < 2 || m2 < 2) return FDES_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    Fft2D f;
    std::string ferr;
    if (f.create(m1, m2, backend, c->stream, &ferr) != 0) { f.destroy(); c->err = "FFT plan: " + ferr; return FDES_EGPU; }
    if (c->pass_threads == 64 || c->pass_threads == 65 || c->pass_threads == 128) f.wg = c->pass_threads;
    float2* d = nullptr;
    const size_t bytes = sizeof(float2) * (size_t)m1 * m2;
    hipError_t e = hipMalloc((void**)&d, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = f.exec(d, inverse != 0, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d) (void)hipFree(d);
    const int used = f.backend;
    f.destroy();
    if (e != hipSuccess) { c->err = std::string("fft2d_host: ") + hipGetErrorString(e); return FDES_EGPU; }
    return used; // 1 = rocFFT, 2 = LDS kernels
}

// Times one LDS row pass on scratch n x n grids (micro-benchmark hook): mean time per launch in us.
// streams > 1 issues the launches round-robin on that many HIP streams, each with its own grids
// (do concurrent kernels overlap their memory and compute phases?).
int fdes_bench_pass(fdes_ctx* c, int n, int pre, int mid, int post, int store_t, int iters, int streams, double* us)
{
    if (!c || !us || iters < 1 || streams < 1 || streams > 8 || !(lds_fft_supported_len(n) || gen_pass_supported_len(n))) return FDES_EINVAL;
    // passes whose operands this hook does not provide (atom records, second output grid, species loop) are refused:
    // launching them on the scratch arguments would write through null pointers
    if (mid == MID_ATOMS || mid == MID_GTABN) { c->err = "bench_pass: pass needs operands the hook does not provide"; return FDES_EINVAL; }
    if (c->bench_alt >= 0 && ((c->bench_alt / 100 % 100) == MID_ATOMS || (c->bench_alt / 100 % 100) == MID_GTABN)) return FDES_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    Fft2D f;
    std::string ferr;
    if (f.create(n, n, 2, c->stream, &ferr) != 0) { f.destroy(); c->err = ferr; return FDES_EGPU; }
    const size_t m12 = (size_t)(n + c->bench_pitch) * (n * (size_t)c->bench_tall + (size_t)c->bench_pitch);
    std::vector<void*> bufs;
    std::vector<hipStream_t> sts;
    std::vector<PassArgs> args;
    int rc = FDES_OK;
    for (int q = 0; q < streams && rc == FDES_OK; q++) {
        float2 *a = nullptr, *b = nullptr, *o = nullptr, *o2 = nullptr, *pt = nullptr;
        float* g = nullptr;
        hipStream_t st = nullptr;
        if (hipMalloc((void**)&a, 8 * m12) != hipSuccess || hipMalloc((void**)&b, 8 * m12) != hipSuccess || hipMalloc((void**)&o, 8 * m12) != hipSuccess ||
            (mid == MID_EXPIV_PAIR && hipMalloc((void**)&o2, 8 * m12) != hipSuccess) ||
            hipMalloc((void**)&pt, 8 * m12) != hipSuccess || hipMalloc((void**)&g, 4 * m12) != hipSuccess ||
            hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rc = FDES_ENOMEM; }
        bufs.insert(bufs.end(), {a, b, o, o2, pt, g});
        if (st) sts.push_back(st);
        if (rc != FDES_OK) break;
        // random operands: zero-filled grids let the chip hold a higher clock than real data does
        (void)k_fill_noise((float*)a, 2 * m12, 11u + q, c->stream); (void)k_fill_noise((float*)b, 2 * m12, 23u + q, c->stream);
        (void)k_fill_noise((float*)pt, 2 * m12, 37u + q, c->stream); (void)k_fill_noise(g, m12, 41u + q, c->stream);
        (void)hipMemsetAsync(o, 0, 8 * m12, c->stream);
        (void)hipStreamSynchronize(c->stream);
        PassArgs A;
        A.in0 = a; A.in1 = b; A.out = o; A.out2 = o2; A.zsrc = a; A.gtab = g; A.prow = pt; A.pcol = pt + n * c->bench_tall; A.tw0 = f.tw0x; A.tw1 = f.tw1x; A.nrows = n * c->bench_tall;
        A.nspecies = 1; A.species_stride = m12; A.scale = 1.f; A.mindim = n;
        A.walk = c->walk;
        if (c->bench_pitch) { A.pitch_in = n + c->bench_pitch; A.pitch_out = (store_t ? n * c->bench_tall : n) + c->bench_pitch; }
        A.wg = (c->pass_threads == 64 || c->pass_threads == 65 || c->pass_threads == 128) ? c->pass_threads : (c->pass_threads == 256 ? 256 : ((c->pass_threads == 513 || c->pass_threads == 1) && n <= 2048 ? 1 : 512));
        A.stagger = c->stagger;
        if (c->bench_band) { // micro-benchmark of the band-limit bookkeeping: bit 0 live rows only, bit 1 dead loads, bit 2 dead stores
            A.band = n * n;
            A.live_rows_only = (c->bench_band & 1) ? 1 : 0;
            A.skip_dead_loads = (c->bench_band & 2) ? 3 : 0;
            A.skip_dead_stores = (c->bench_band & 4) ? 1 : 0;
        }
        args.push_back(A);
    }
    // diagnostic build only (FDES_STAMP_FILE set, library built with -DFDES_STAMPS): the phase stamps of the LAST launch
    // on stream 0 are written to that file as raw uint64[blocks * waves * 16]
    unsigned long long* dbg = nullptr;
    const size_t dbg_n = (size_t)4096 * 8 * 16;
    const char* stamp_file = std::getenv("FDES_STAMP_FILE");
    if (rc == FDES_OK && stamp_file && hipMalloc((void**)&dbg, dbg_n * 8) == hipSuccess) {
        (void)hipMemset(dbg, 0, dbg_n * 8);
        args[0].dbg = dbg;
    }
    if (rc == FDES_OK) {
        hipError_t e = hipSuccess;
        auto go = [&](int q) {
            if (c->bench_alt >= 0 && (q & 1)) return lds_pass(n, c->bench_alt / 10000, c->bench_alt / 100 % 100, c->bench_alt % 100, store_t != 0, args[q], c->bench_serial ? sts[0] : sts[q]);
            return lds_pass(n, pre, mid, post, store_t != 0, args[q], c->bench_serial ? sts[0] : sts[q]);
        };
        for (int q = 0; q < streams && e == hipSuccess; q++) e = go(q);
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters && e == hipSuccess; i++)
            for (int q = 0; q < streams && e == hipSuccess; q++) e = go(q);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        if (e == hipSuccess) *us = std::chrono::duration<double, std::micro>(t1 - t0).count() / ((double)iters * streams);
        else { c->err = std::string("bench_pass: ") + hipGetErrorString(e); rc = FDES_EGPU; }
    }
    if (dbg) {
        std::vector<unsigned long long> h(dbg_n);
        if (hipMemcpy(h.data(), dbg, dbg_n * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* fp = std::fopen(stamp_file, "wb")) { std::fwrite(h.data(), 8, dbg_n, fp); std::fclose(fp); }
        }
        (void)hipFree(dbg);
    }
    for (void* q : bufs) if (q) (void)hipFree(q);
    for (hipStream_t st : sts) (void)hipStreamDestroy(st);
    f.destroy();
    return rc;
}

// ------------------------------- buildMeasurements ---------------------------------------------

// Complete measurements ks[0 .. n) - every configuration of each, detector chain included - with the images left in the
// plan's stack.  A series with one configuration per measurement is dealt to the lanes in blocks of `gang` measurements,
// each block one gang (own incoming wave, tilt and intensity slot per member), its images finished behind it on that
// lane; otherwise one measurement after the other through the plan API (whose gangs are the configurations of a k).
int fdes_plan_run_measurements(fdes_plan* pl, const int* ks, int n)
{
    if (!live_plan(pl) || pl->is_lane || n < 0 || (n > 0 && !ks)) return FDES_EINVAL;
    for (int i = 0; i < n; i++) if (ks[i] < 0 || ks[i] >= pl->p.n3) return FDES_EINVAL;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    const int count = pl->p.frPh > 0 ? pl->p.frPh : 1;
    const float alpha = 1.f / ((float)count); // src/crystalMaker.cu:302-304
    int rc = FDES_OK;
    if (!pl->gang_k || pl->want_ew) {
        for (int i = 0; i < n && rc == FDES_OK; i++) {
            rc = fdes_plan_begin_measurement(pl, ks[i]);
            for (int j = 0; j < count && rc == FDES_OK; j++) rc = fdes_plan_run_config(pl, ks[i], j, alpha);
            if (rc == FDES_OK) rc = fdes_plan_end_measurement(pl, ks[i]);
        }
        return rc;
    }
    RC(gang_flush_all(pl));
    const int G = pl->gang, nl = (int)pl->lanes.size() + 1;
    for (fdes_plan* l : pl->lanes) l->Jout = pl->J;
    for (int i0 = 0, b = 0; i0 < n && rc == FDES_OK; i0 += G, b++) {
        fdes_plan* lp = (b % nl) ? pl->lanes[(size_t)(b % nl) - 1] : pl;
        const int i1 = i0 + G < n ? i0 + G : n;
        for (int i = i0; i < i1 && rc == FDES_OK; i++) {
            if (k_fill(lp->I + (size_t)(i - i0) * lp->m12, lp->m12, 0.f, 0.f, lp->ctx->stream) != hipSuccess) { c->err = "k_fill"; rc = FDES_EGPU; }
            lp->gq.push_back({ks[i], 0, alpha, i - i0});
            lp->gfinal.push_back({ks[i], i - i0});
        }
        if (rc == FDES_OK) rc = gang_flush(lp);
        if (rc != FDES_OK && lp != pl) c->err = "lane: " + lp->ctx->err;
        if (rc == FDES_OK) report_progress(pl, (int64_t)i1, (int64_t)n, false);
    }
    if (rc == FDES_OK) rc = fdes_plan_sync(pl);
    return rc;
}

int fdes_build_measurements(fdes_ctx* c, const fdes_params* p, const fdes_atoms* a, float* image, float* potential, float* exitwave)
{
    if (!live_ctx(c) || !image) return FDES_EINVAL;
    fdes_plan* pl = nullptr;
    RC(fdes_plan_create(c, p, a, &pl));
    pl->want_ew = exitwave != nullptr;
    const int count = pl->p.frPh > 0 ? pl->p.frPh : 1;
    const float alpha = 1.f / ((float)count); // src/crystalMaker.cu:302-304
    {   // slice loops this job issues per lane: configurations, or gangs of them
        const long nl = (long)pl->lanes.size() + 1, g = pl->gang > 1 ? pl->gang : 1;
        const long loops = (pl->gang_k && !exitwave) ? ((long)pl->p.n3 + g - 1) / g : (long)pl->p.n3 * ((count + (pl->gang_k ? 1 : g) - 1) / (pl->gang_k ? 1 : g));
        pl->one_shot_few = (loops + nl - 1) / nl < 4;
    }
    int rc = FDES_OK;
    if (pl->gang_k && !exitwave) {
        std::vector<int> ks((size_t)pl->p.n3);
        for (int k = 0; k < pl->p.n3; k++) ks[(size_t)k] = k;
        rc = fdes_plan_run_measurements(pl, ks.data(), pl->p.n3);
    } else
    for (int k = 0; k < pl->p.n3 && rc == FDES_OK; k++) {
        rc = fdes_plan_begin_measurement(pl, k);
        for (int j = 0; j < count && rc == FDES_OK; j++) {
            rc = fdes_plan_run_config(pl, k, j, alpha);
            if (rc == FDES_OK) report_progress(pl, (int64_t)k * count + j + 1, (int64_t)pl->p.n3 * count, false);
        }
        if (rc == FDES_OK && exitwave) rc = fdes_plan_get_exitwave(pl, exitwave + 2 * pl->m12 * (size_t)k);
        if (rc == FDES_OK) rc = fdes_plan_end_measurement(pl, k);
    }
    if (rc == FDES_OK) rc = fdes_plan_get_images(pl, image);
    if (rc == FDES_OK) report_progress(pl, (int64_t)pl->p.n3 * count, (int64_t)pl->p.n3 * count, true);
    if (rc == FDES_OK && potential) rc = fdes_plan_potential(pl, 0, fdes_plan_original_slices(pl), potential);
    fdes_plan_destroy(pl);
    return rc;
}

} // extern "C"
