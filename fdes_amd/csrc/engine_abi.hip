// engine_abi.hip — the measurement entry points of include/fdes_abi.h (begin / run_config / end, images, sums, the reduction
// between GPUs: peer copies and RCCL), the potential output, and the taps / micro-benchmark hooks of include/fdes_abi_test.h.
// Split from engine.hip in round 5; shared declarations: engine_impl.h.
#include "engine_impl.h"

extern "C" {

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
    if (!gen_jit_default_on() && ((gen_pass_supported_len(m1) && gen_pass_needs_compiled(m1)) || (gen_pass_supported_len(m2) && gen_pass_needs_compiled(m2)))) return 1; // such rows run compiled-at-plan-creation kernels only
    return (fft_option != 1 && Fft2D::lds_supported(m1, m2)) ? 2 : 1;
}
int fdes_plan_fft_backend(const fdes_plan* pl) { return live_plan(pl) ? pl->fft->backend : FDES_EINVAL; }
int fdes_plan_jit_kernels(const fdes_plan* pl) { return live_plan(pl) ? ((pl->fft->backend == 2 && pl->fft->jit_x) ? 1 : 0) + ((pl->fft->backend == 2 && pl->fft->jit_y) ? 1 : 0) : FDES_EINVAL; }
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
    const size_t m12 = pl->m12, ew_off = (m12 + 1) & ~(size_t)1; // (the exit wave is read as float2: an even float offset)
    const size_t per = pl->want_ew ? ew_off + 2 * m12 : m12;      // floats per peer: intensity view [+ complex exit wave]
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
            if (pl->want_ew) HIPCHK(c, k_axpy(pl->EW, reinterpret_cast<const float2*>(part + ew_off), m12, 1.f, c->stream));
        }
    } else {
        if (pl->span_send_n < per) {
            DeviceGuard guard(c->device);
            if (pl->span_send) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(pl->span_send); pl->span_send = nullptr; }
            RC(dmalloc(c, &pl->span_send, per));
            pl->span_send_n = per;
        }
        HIPCHK(c, k_real_pack(pl->span_send, pl->I, m12, c->stream));
        if (pl->want_ew) HIPCHK(c, hipMemcpyAsync(pl->span_send + ew_off, pl->EW, sizeof(float2) * m12, hipMemcpyDeviceToDevice, c->stream));
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
    if (f.create(m1, m2, backend, c->stream, &ferr, c->jit < 0 ? gen_jit_default_on() : c->jit != 0) != 0) { f.destroy(); c->err = "FFT plan: " + ferr; return FDES_EGPU; }
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
