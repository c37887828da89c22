// engine.hip — host driver of the MI355X FDES engine: context, plan, slice loop, C-ABI.
//
// Restates the control flow of buildMeasurements / phaseGrating / forwardPropagation /
// incomingWave / applyLensFunction / diffractionPattern / addNoiseAndMtf
// (src/crystalMaker.cu:227-424, 507-536, 579-613, 700-718; src/multisliceSimulation.cu:538-622)
// on one HIP stream, with all scratch allocated once per plan (the reference cudaMalloc/cudaFree's
// two grids per slice, :516-517,534-535, and prints to stderr inside the slice loop, :341).
#include "engine_impl.h"

namespace fdes_engine {

std::once_flag g_rocfft_once;
DeviceLocks g_dev_locks;
std::mutex g_live_mutex;
std::set<const void*> g_live_ctx, g_live_plan;
bool live_ctx(const fdes_ctx* c) { if (!c) return false; std::lock_guard<std::mutex> g(g_live_mutex); return g_live_ctx.count(c) != 0; }
bool live_plan(const fdes_plan* p) { if (!p) return false; std::lock_guard<std::mutex> g(g_live_mutex); return g_live_plan.count(p) != 0; }
std::once_flag g_atexit_once;

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

int plan_gang(const fdes_ctx* c, const fdes_plan* pl, bool* across_k)
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

} // namespace fdes_engine

extern "C" {


int fdes_gpu_available(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n > 0;
}

} // extern "C"

namespace fdes_engine {
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
} // namespace fdes_engine

extern "C" {

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
    if (!std::strcmp(key, "jit")) { if (value < -1 || value > 1) return FDES_EINVAL; c->jit = (int)value; return FDES_OK; }
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
        const bool jit = c->jit < 0 ? gen_jit_default_on() : c->jit != 0;
        auto key = std::make_tuple(pl->p.m1, pl->p.m2, c->opt_fft + (jit ? 4 : 0));
        auto it = c->fft_cache.find(key);
        if (it == c->fft_cache.end()) {
            std::string ferr;
            Fft2D* f = new Fft2D();
            DeviceGuard guard(c->device); // allocations and a module load beside a possible capture in another thread of this device
            if (f->create(pl->p.m1, pl->p.m2, c->opt_fft, c->stream, &ferr, jit) != 0) {
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
        const bool ok256 = lds_fft_rows_per_block(m1, 256, m2) >= 4 && lds_fft_rows_per_block(m2, 256, m1) >= 4; // the rows per workgroup divide the other dimension; >= 32-byte transposed segments
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
        if (lds_fft_rows_per_block(m1, pl->wg, m2) <= 0 || lds_fft_rows_per_block(m2, pl->wg, m1) <= 0) pl->wg = pl->fft->wg;
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

} // extern "C"
