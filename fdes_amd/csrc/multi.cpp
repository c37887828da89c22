// multi.cpp — buildMeasurements over several GPUs of one node from ONE host process (one host thread per GPU), written
// against the public C-ABI only (what a maintainer of the reference's C++ host would write).  The (k, j) wave
// propagations of src/crystalMaker.cu:324-367 are independent; the flattened list is block-partitioned over the GPUs
// exactly as fdes_amd/shard.py does for the one-process-per-GPU launch.  The partial sums of a measurement k that spans
// GPUs (intensity; the coherent exit-wave sum when asked for; src/crystalMaker.cu:347-365 is the sum being distributed)
// end on the OWNER's GPU, which then applies addNoiseAndMtf.  Two ways:
//   * default: a binary TREE of fdes_plan_accumulate_from calls - in round s = 1, 2, 4 the GPU at offset i (a multiple of
//     2 s) of the span adds the sum of the GPU at offset i + s: log2(GPUs) rounds of concurrent peer copies over xGMI (the
//     intensity as its float view, 16 MiB at 2048^2) and add kernels instead of seven serial rounds through the owner; the
//     association order is fixed by the tree, so the images do not depend on timing;
//   * FDES_REDUCE=rccl: over a communicator created once per call, when every worker has a device of its own: ONE
//     ncclReduce(sum, float[m1 m2]) to the owner for a measurement that spans ALL GPUs (the frozen-phonon
//     configurations of one image dealt over the node; the exit-wave sum of print_level 2 rides along as a second reduce), one
//     group of ncclSend / ncclRecv to the owner for a measurement that spans only some of them (round 5:
//     fdes_plan_reduce_intensity_span - the ranks outside the span take no part); any failure to set the communicator up
//     takes the tree.  Not the default: creating a communicator costs more than a whole headline job (seconds against tens
//     of milliseconds), and in the collective RCCL picks the association order.
// The potential output (print_level > 0) does not depend on (k, j): its slices are dealt over the GPUs.  Random numbers are
// keyed on (k, j): the images do not depend on the partition beyond the association order of that one sum.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fdes_internal.h"

namespace {

class Barrier { // std::barrier is C++20
  public:
    explicit Barrier(int n) : n_(n) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        const unsigned gen = gen_;
        if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen != gen_; });
    }

  private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    unsigned gen_ = 0;
};

struct Part { int lo, hi; }; // flattened (k, j) indices [lo, hi)
Part part_of(int total, int world, int rank)
{
    const int base = total / world, rem = total % world;
    const int lo = rank * base + (rank < rem ? rank : rem);
    return {lo, lo + base + (rank < rem ? 1 : 0)};
}

} // namespace

extern "C" int fdes_build_measurements_multi(int ngpu, const int* devices, const fdes_params* p, const fdes_atoms* a, float* image,
                                             float* potential, float* exitwave)
{
    if (ngpu < 1 || !devices || !p || !a || !image) return FDES_EINVAL;
    if (ngpu == 1) {
        fdes_ctx* c = nullptr;
        int rc = fdes_create(&c, devices[0]);
        if (rc) return rc;
        rc = fdes_build_measurements(c, p, a, image, potential, exitwave);
        fdes_destroy(c);
        return rc;
    }
    const int n3 = p->n3, count = p->frPh > 0 ? p->frPh : 1, total = n3 * count;
    const float weight = 1.f / (float)count; // alpha of src/crystalMaker.cu:302-304
    const size_t m12 = (size_t)p->m1 * (size_t)p->m2, img = (size_t)p->n1 * (size_t)p->n2;
    // first[k] .. last[k] = GPUs holding any (k, j); the block partition is contiguous, so the owner of k (the GPU
    // that holds (k, 0)) is first[k]
    std::vector<int> first((size_t)n3, ngpu), last((size_t)n3, -1);
    for (int r = 0; r < ngpu; r++) {
        const Part q = part_of(total, ngpu, r);
        for (int i = q.lo; i < q.hi; i++) {
            const int k = i / count;
            if (r < first[(size_t)k]) first[(size_t)k] = r;
            if (r > last[(size_t)k]) last[(size_t)k] = r;
        }
    }
    std::vector<fdes_plan*> plans((size_t)ngpu, nullptr); // read by the adding GPU between the barriers of a split k
    std::vector<int> status((size_t)ngpu, FDES_OK);
    Barrier bar(ngpu);
    // FDES_REDUCE=rccl: one communicator for the call (one rank per worker; RCCL refuses two ranks on one device)
    bool want_rccl = false;
    fdes_comm_id comm_id;
    {
        const char* mode = std::getenv("FDES_REDUCE");
        bool distinct = true;
        for (int r = 0; r < ngpu; r++)
            for (int q = 0; q < r; q++) distinct = distinct && devices[r] != devices[q];
        bool any_split = false;
        for (int k = 0; k < n3; k++) any_split = any_split || last[(size_t)k] > first[(size_t)k];
        want_rccl = mode && !std::strcmp(mode, "rccl") && distinct && any_split && fdes_comm_unique_id(&comm_id) == FDES_OK;
    }
    std::vector<fdes_comm*> comms((size_t)ngpu, nullptr);
    const bool timing = std::getenv("FDES_TIMING") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    std::atomic<int> comm_failures{0};
    // per measurement: workers that arrived at k's collective in a failed state (written before k's barrier, read after it: every
    // rank decides from the same value, and a rank that fails later cannot change the decision a slow peer is still to read)
    std::vector<std::atomic<int>> failed_before_collective((size_t)n3);
    for (auto& f : failed_before_collective) f.store(0);
    auto worker = [&](int r) {
        int& rc = status[(size_t)r];
        fdes_ctx* ctx = nullptr;
        fdes_plan* pl = nullptr;
        rc = fdes_create(&ctx, devices[r]);
        bool rccl = false;
        if (want_rccl) { // every worker joins or none does: a rank that never calls ncclCommInitRank would block the others
            if (rc != FDES_OK) comm_failures++;
            bar.wait();
            if (comm_failures.load() == 0 && fdes_comm_create(ctx, ngpu, r, &comm_id, &comms[(size_t)r]) != FDES_OK) comm_failures++;
            bar.wait();
            rccl = comm_failures.load() == 0;
        }
        const auto tc0 = std::chrono::steady_clock::now();
        if (rc == FDES_OK) rc = fdes_plan_create(ctx, p, a, &pl);
        if (timing) { // FDES_TIMING=1: when each worker's plan creation ran (the workers create their plans side by side)
            const auto tc1 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "  FDES: worker %d (device %d): plan creation %.1f ms, started %.1f ms after the call\n", r, devices[r],
                         std::chrono::duration<double, std::milli>(tc1 - tc0).count(), std::chrono::duration<double, std::milli>(tc0 - t_call).count());
        }
        if (rc == FDES_OK && exitwave) rc = fdes_plan_want_exitwave(pl, 1);
        plans[(size_t)r] = rc == FDES_OK ? pl : nullptr;
        const Part q = part_of(total, ngpu, r);
        std::vector<float> mine;
        // one configuration per measurement (a tilt / defocus series without frozen phonons): no k is split, this GPU's
        // measurements are complete by themselves and go through the engine in one call (gangs of measurements)
        const bool whole = count == 1 && !exitwave;
        if (whole && rc == FDES_OK && q.hi > q.lo) {
            std::vector<int> ks;
            for (int i = q.lo; i < q.hi; i++) ks.push_back(i);
            rc = fdes_plan_run_measurements(pl, ks.data(), (int)ks.size());
        }
        for (int k = 0; k < n3 && !whole; k++) {
            const bool split = last[(size_t)k] > first[(size_t)k];
            const bool in_span = r >= first[(size_t)k] && r <= last[(size_t)k];
            const bool owner = first[(size_t)k] == r;
            const int jlo = (q.lo > k * count ? q.lo : k * count), jhi = (q.hi < (k + 1) * count ? q.hi : (k + 1) * count);
            if (rc == FDES_OK && in_span) {
                rc = fdes_plan_begin_measurement(pl, k);
                for (int i = jlo; i < jhi && rc == FDES_OK; i++) rc = fdes_plan_run_config(pl, k, i % count, weight);
            }
            if (split && rccl) {
                // one collective (span = every GPU) or one group of sends to the owner (a shorter span: only its ranks take
                // part); a rank that has failed must not leave the others inside either: ALL workers agree first
                if (rc != FDES_OK) failed_before_collective[(size_t)k]++;
                bar.wait();
                if (failed_before_collective[(size_t)k].load() != 0) { if (rc == FDES_OK) rc = FDES_EGPU; } // a peer failed
                else if (in_span) rc = fdes_plan_reduce_intensity_span(pl, comms[(size_t)r], first[(size_t)k], first[(size_t)k], last[(size_t)k]);
            } else if (split) { // every thread passes every barrier, whatever its state, so that nobody waits for ever
                if (rc != FDES_OK) plans[(size_t)r] = nullptr;
                const int f = first[(size_t)k], n = last[(size_t)k] - f + 1, i = r - f;
                for (int s = 1; s < n; s <<= 1) { // binary tree over the span, the owner (offset 0) at its root
                    bar.wait(); // the sources of this round are complete: their configurations resp. their own additions are enqueued / done
                    if (rc == FDES_OK && in_span && i % (2 * s) == 0 && i + s < n)
                        rc = plans[(size_t)(f + i + s)] ? fdes_plan_accumulate_from(pl, plans[(size_t)(f + i + s)]) : FDES_EGPU; // a peer failed
                    if (rc != FDES_OK && plans[(size_t)r]) plans[(size_t)r] = nullptr; // (only an adding GPU can get here: nobody reads its entry in this round)
                }
                bar.wait(); // the peers' sums have been read: they may start their next measurement
            }
            if (rc == FDES_OK && owner && exitwave) rc = fdes_plan_get_exitwave(pl, exitwave + 2 * m12 * (size_t)k);
            if (rc == FDES_OK && owner) rc = fdes_plan_end_measurement(pl, k);
        }
        if (rc == FDES_OK) {
            mine.resize(img * (size_t)n3);
            rc = fdes_plan_get_images(pl, mine.data());
            for (int k = 0; k < n3 && rc == FDES_OK; k++)
                if (first[(size_t)k] == r) std::memcpy(image + (size_t)k * img, mine.data() + (size_t)k * img, sizeof(float) * img);
        }
        if (rc == FDES_OK && potential) { // original slices dealt over the GPUs
            const Part ps = part_of(fdes_plan_original_slices(pl), ngpu, r);
            if (ps.hi > ps.lo) rc = fdes_plan_potential(pl, ps.lo, ps.hi, potential + 2 * m12 * (size_t)ps.lo);
        }
        if (rc != FDES_OK) std::fprintf(stderr, "  FDES: worker %d (device %d) failed with %d: %s\n", r, devices[r], rc, ctx ? fdes_last_error(ctx) : "no context");
        bar.wait(); // nobody destroys a plan that a late accumulate_from of another thread could still name
        if (comms[(size_t)r]) fdes_comm_destroy(comms[(size_t)r]);
        if (pl) fdes_plan_destroy(pl);
        if (ctx) fdes_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < ngpu; r++) th.emplace_back(worker, r);
    for (auto& t : th) t.join();
    for (int r = 0; r < ngpu; r++) if (status[(size_t)r] != FDES_OK) return status[(size_t)r];
    return FDES_OK;
}
