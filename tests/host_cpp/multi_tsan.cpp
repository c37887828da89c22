// multi_tsan.cpp — ThreadSanitizer harness for fdes_amd/csrc/multi.cpp (the one-process multi-GPU driver).
// TEST INFRASTRUCTURE.  multi.cpp is written against the public C-ABI only, so it links here against STUB plans that
// keep their "intensity" in host memory: every entry point multi.cpp calls is implemented below with plain loads and
// stores and NO locking, so any pair of calls that the driver's barriers fail to order shows up as a data race.  The
// result is also compared with the serial sum (partition + ownership logic).
// Build + run: tests/test_host_cpu.py::test_multi_gpu_driver_under_thread_sanitizer
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <map>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fdes_abi.h"

struct fdes_ctx { int device; };
struct fdes_plan {
    fdes_ctx* ctx;
    int n1, n2, n3, m3, count;
    size_t m12;
    std::vector<float> I, EW, J;
    bool want_ew = false;
    int cur_k = -1;
};

// stub communicator: the collective's own rendezvous is a real barrier (RCCL synchronises its ranks inside the call);
// the data it moves is read and written with plain loads and stores like everything else here
struct StubWorld {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, arrived = 0;
    unsigned gen = 0;
    std::vector<fdes_plan*> slot;
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned g = gen;
        if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return g != gen; });
    }
};
static StubWorld g_world;
// rendezvous of the ranks lo .. hi of a span (fdes_plan_reduce_intensity_span): only they take part
static std::mutex g_span_m;
static std::map<std::pair<int, int>, StubWorld*> g_span_worlds;
static StubWorld& span_world(int lo, int hi, int nranks)
{
    std::lock_guard<std::mutex> g(g_span_m);
    StubWorld*& w = g_span_worlds[{lo, hi}];
    if (!w) { w = new StubWorld; w->n = hi - lo + 1; w->slot.assign((size_t)nranks, nullptr); }
    return *w;
}
static int g_span_calls = 0; // (written by the roots under g_span_m)
// plan creation of the workers must overlap in time (round 5: no process-wide lock in front of fdes_plan_create): the stub
// takes 40 ms and records when it ran
static std::mutex g_create_m;
static std::vector<std::pair<double, double>> g_create_times;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct fdes_comm { int nranks, rank; };
static int g_reduce_calls = 0; // (written by rank 0 only)

static float contrib(int k, int j, size_t i) { return (float)((k * 131 + j * 17 + (int)(i % 97)) % 251) / 251.0f; }

extern "C" {
int fdes_create(fdes_ctx** c, int dev) { *c = new fdes_ctx{dev}; return dev < 0 ? FDES_EGPU : FDES_OK; }
int fdes_destroy(fdes_ctx* c) { delete c; return FDES_OK; }
const char* fdes_last_error(const fdes_ctx*) { return "stub"; }
int fdes_build_measurements(fdes_ctx*, const fdes_params*, const fdes_atoms*, float*, float*, float*) { return FDES_EUNSUPPORTED; }
int fdes_plan_create(fdes_ctx* c, const fdes_params* p, const fdes_atoms*, fdes_plan** out)
{
    const double t0 = now_ms();
    std::this_thread::sleep_for(std::chrono::milliseconds(40));
    { std::lock_guard<std::mutex> g(g_create_m); g_create_times.push_back({t0, now_ms()}); }
    fdes_plan* pl = new fdes_plan();
    pl->ctx = c; pl->n1 = p->n1; pl->n2 = p->n2; pl->n3 = p->n3; pl->m3 = p->m3; pl->count = p->frPh > 0 ? p->frPh : 1;
    pl->m12 = (size_t)p->m1 * p->m2;
    pl->I.assign(2 * pl->m12, 0.f); pl->EW.assign(2 * pl->m12, 0.f); pl->J.assign((size_t)p->n1 * p->n2 * p->n3, 0.f);
    *out = pl;
    return c->device == 99 ? FDES_ENOMEM : FDES_OK; // device 99: a worker that fails at plan creation
}
int fdes_plan_destroy(fdes_plan* pl) { delete pl; return FDES_OK; }
int fdes_plan_want_exitwave(fdes_plan* pl, int on) { pl->want_ew = on != 0; return FDES_OK; }
int fdes_plan_begin_measurement(fdes_plan* pl, int k)
{
    std::fill(pl->I.begin(), pl->I.end(), 0.f);
    std::fill(pl->EW.begin(), pl->EW.end(), 0.f);
    pl->cur_k = k;
    return FDES_OK;
}
int fdes_plan_run_config(fdes_plan* pl, int k, int j, float w)
{
    for (size_t i = 0; i < pl->I.size(); i++) { pl->I[i] += w * contrib(k, j, i); pl->EW[i] += w * contrib(k, j, i + 5); }
    return FDES_OK;
}
int fdes_plan_accumulate_from(fdes_plan* d, fdes_plan* s)
{
    if (d->cur_k != s->cur_k) return FDES_EINVAL; // the peer must still hold the same measurement
    for (size_t i = 0; i < d->I.size(); i++) { d->I[i] += s->I[i]; if (d->want_ew) d->EW[i] += s->EW[i]; }
    return FDES_OK;
}
int fdes_comm_unique_id(fdes_comm_id* id) { std::memset(id, 7, sizeof *id); return FDES_OK; }
int fdes_comm_create(fdes_ctx*, int nranks, int rank, const fdes_comm_id* id, fdes_comm** out)
{
    if (id->bytes[5] != 7) return FDES_EINVAL;
    if (rank == 0) { g_world.n = nranks; g_world.slot.assign((size_t)nranks, nullptr); } // (the driver's barriers order this before any reduce)
    *out = new fdes_comm{nranks, rank};
    return FDES_OK;
}
int fdes_comm_destroy(fdes_comm* c) { delete c; return FDES_OK; }
int fdes_plan_reduce_intensity(fdes_plan* pl, fdes_comm* c, int root)
{
    g_world.slot[(size_t)c->rank] = pl;
    g_world.wait();
    int rc = FDES_OK;
    if (c->rank == root) {
        if (c->rank == 0) g_reduce_calls++;
        for (int r = 0; r < c->nranks; r++) {
            const fdes_plan* s = g_world.slot[(size_t)r];
            if (r == root) continue;
            if (s->cur_k != pl->cur_k) rc = FDES_EINVAL; // every rank must hold the same measurement
            for (size_t i = 0; i < pl->I.size(); i += 2) pl->I[i] += s->I[i]; // the real view only
            if (pl->want_ew) for (size_t i = 0; i < pl->EW.size(); i++) pl->EW[i] += s->EW[i];
        }
    }
    g_world.wait();
    return rc;
}
int fdes_plan_reduce_intensity_span(fdes_plan* pl, fdes_comm* c, int root, int lo, int hi)
{
    if (lo == 0 && hi == c->nranks - 1) return fdes_plan_reduce_intensity(pl, c, root);
    if (c->rank < lo || c->rank > hi || root < lo || root > hi) return FDES_EINVAL;
    StubWorld& w = span_world(lo, hi, c->nranks);
    w.slot[(size_t)c->rank] = pl;
    w.wait();
    int rc = FDES_OK;
    if (c->rank == root) {
        { std::lock_guard<std::mutex> g(g_span_m); g_span_calls++; }
        for (int r = lo; r <= hi; r++) {
            const fdes_plan* s = w.slot[(size_t)r];
            if (r == root) continue;
            if (s->cur_k != pl->cur_k) rc = FDES_EINVAL; // every rank of the span must hold the same measurement
            for (size_t i = 0; i < pl->I.size(); i += 2) pl->I[i] += s->I[i];
            if (pl->want_ew) for (size_t i = 0; i < pl->EW.size(); i++) pl->EW[i] += s->EW[i];
        }
    }
    w.wait();
    return rc;
}
int fdes_plan_get_exitwave(fdes_plan* pl, float* ew) { std::memcpy(ew, pl->EW.data(), sizeof(float) * pl->EW.size()); return FDES_OK; }
int fdes_plan_end_measurement(fdes_plan* pl, int k)
{
    const size_t img = (size_t)pl->n1 * pl->n2;
    for (size_t i = 0; i < img; i++) pl->J[(size_t)k * img + i] = pl->I[2 * (i % pl->m12)];
    return FDES_OK;
}
int fdes_plan_run_measurements(fdes_plan* pl, const int* ks, int n)
{
    const int count = pl->count;
    for (int i = 0; i < n; i++) {
        int rc = fdes_plan_begin_measurement(pl, ks[i]);
        for (int j = 0; j < count && rc == FDES_OK; j++) rc = fdes_plan_run_config(pl, ks[i], j, 1.f / (float)count);
        if (rc == FDES_OK) rc = fdes_plan_end_measurement(pl, ks[i]);
        if (rc != FDES_OK) return rc;
    }
    return FDES_OK;
}
int fdes_plan_get_images(fdes_plan* pl, float* out) { std::memcpy(out, pl->J.data(), sizeof(float) * pl->J.size()); return FDES_OK; }
int fdes_plan_original_slices(const fdes_plan* pl) { return pl->m3; }
int fdes_plan_potential(fdes_plan* pl, int lo, int hi, float* pot)
{
    for (int s = lo; s < hi; s++)
        for (size_t i = 0; i < 2 * pl->m12; i++) pot[(size_t)(s - lo) * 2 * pl->m12 + i] = (float)s + 0.001f * (float)(i % 7);
    return FDES_OK;
}
}

static int run_case(int ngpu, int n3, int count, bool fail_one, bool rccl = false)
{
    if (rccl) setenv("FDES_REDUCE", "rccl", 1); else unsetenv("FDES_REDUCE");
    const int reduce_calls_before = g_reduce_calls;
    int span_calls_before = 0;
    { std::lock_guard<std::mutex> g(g_span_m); span_calls_before = g_span_calls; }
    { std::lock_guard<std::mutex> g(g_create_m); g_create_times.clear(); }
    fdes_params p;
    std::memset(&p, 0, sizeof p);
    p.n1 = 6; p.n2 = 5; p.m1 = 8; p.m2 = 8; p.n3 = n3; p.m3 = 7; p.frPh = count > 1 ? count : 0;
    fdes_atoms a = {0, nullptr, nullptr, nullptr, nullptr};
    std::vector<int> dev((size_t)ngpu);
    for (int r = 0; r < ngpu; r++) dev[(size_t)r] = r;
    if (fail_one) dev[(size_t)(ngpu / 2)] = 99;
    const size_t m12 = 64, img = 30;
    std::vector<float> image(img * (size_t)n3, -1.f), pot(2 * m12 * 7, -1.f), ew(2 * m12 * (size_t)n3, -1.f);
    const int rc = fdes_build_measurements_multi(ngpu, dev.data(), &p, &a, image.data(), pot.data(), ew.data());
    if (fail_one) return rc == FDES_OK ? 1 : 0; // must report the failure and must not hang
    if (rc != FDES_OK) return 1;
    if (rccl && n3 == 1 && count >= ngpu && g_reduce_calls != reduce_calls_before + 1) return 1; // the collective path was taken
    {   // every worker's plan creation ran while the others' did: the latest start lies before the earliest end
        std::lock_guard<std::mutex> g(g_create_m);
        if ((int)g_create_times.size() != ngpu) return 1;
        double latest_start = 0, earliest_end = 1e300;
        for (auto& t : g_create_times) { if (t.first > latest_start) latest_start = t.first; if (t.second < earliest_end) earliest_end = t.second; }
        if (ngpu > 1 && !(latest_start < earliest_end)) { std::printf("plan creations did not overlap\n"); return 1; }
    }
    if (rccl) { // measurements that span some but not all GPUs went through the span reduction, never through the tree
        int spans = 0;
        for (int k = 0; k < n3; k++) {
            const int cntk = count > 1 ? count : 1, total = n3 * cntk;
            auto rank_of = [&](int i) { const int base = total / ngpu, rem = total % ngpu; int lo = 0; for (int r = 0; r < ngpu; r++) { const int n = base + (r < rem ? 1 : 0); if (i < lo + n) return r; lo += n; } return ngpu - 1; };
            const int f = rank_of(k * cntk), l = rank_of(k * cntk + cntk - 1);
            if (l > f && !(f == 0 && l == ngpu - 1)) spans++;
        }
        std::lock_guard<std::mutex> g(g_span_m);
        if (g_span_calls != span_calls_before + spans) { std::printf("span reductions: %d, expected %d\n", g_span_calls - span_calls_before, spans); return 1; }
    }
    const int cnt = count > 1 ? count : 1;
    const float w = 1.f / (float)cnt;
    int bad = 0;
    for (int k = 0; k < n3; k++)
        for (size_t i = 0; i < img; i++) {
            double s = 0;
            for (int j = 0; j < cnt; j++) s += (double)w * contrib(k, j, 2 * (i % m12));
            if (std::abs((double)image[(size_t)k * img + i] - s) > 1e-5) bad++;
        }
    for (int k = 0; k < n3; k++)
        for (size_t i = 0; i < 2 * m12; i++) {
            double s = 0;
            for (int j = 0; j < cnt; j++) s += (double)w * contrib(k, j, i + 5);
            if (std::abs((double)ew[(size_t)k * 2 * m12 + i] - s) > 1e-5) bad++;
        }
    for (int s = 0; s < 7; s++)
        for (size_t i = 0; i < 2 * m12; i++)
            if (pot[(size_t)s * 2 * m12 + i] != (float)s + 0.001f * (float)(i % 7)) bad++;
    return bad;
}

int main()
{
    int bad = 0;
    const int cases[][3] = {{2, 1, 8}, {3, 2, 4}, {4, 5, 3}, {8, 1, 32}, {8, 64, 8}, {5, 3, 1}, {3, 7, 1}, {8, 3, 2}, {7, 1, 3}};
    for (auto& c : cases) {
        const int b = run_case(c[0], c[1], c[2], false);
        std::printf("gpus %d measurements %d configurations %d: %s\n", c[0], c[1], c[2], b ? "MISMATCH" : "ok");
        bad += b;
    }
    bad += run_case(4, 2, 4, true);
    // FDES_REDUCE=rccl: measurements that span all GPUs go through ONE collective (stub communicator above), the others
    // through the tree; a worker that fails must not leave its peers inside the collective
    for (auto& c : cases) {
        const int b = run_case(c[0], c[1], c[2], false, true);
        std::printf("rccl: gpus %d measurements %d configurations %d: %s\n", c[0], c[1], c[2], b ? "MISMATCH" : "ok");
        bad += b;
    }
    bad += run_case(4, 1, 8, true, true);
    std::printf(bad ? "FAILED\n" : "all ok\n");
    return bad ? 1 : 0;
}
