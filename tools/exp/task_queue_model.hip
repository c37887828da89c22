// task_queue_model.hip — round 5: would ONE persistent kernel whose workgroups pull row-group tasks of different passes
// of different chains from a device-side queue beat two streams of kernels?  (review of round 4, item 3: the bare-copy model
// that has to predict >= +10 % before such a slice loop is built.)
//
// Model of the 2048^2 slice loop: a CHAIN is a sequence of passes, each pass = 512 row-group tasks (4 rows of 2048 complex
// floats: coalesced row loads -> `spin` packed FMAs per register (the transforms' vector time: P5 issues ~1900 packed
// instructions per row) -> LDS tile -> transposed store in 32-byte segments, exactly the data movement of fft_wave.hip's
// passes, today's XCD remap and 32-element row padding).  Every task of pass n + 1 of a chain reads what EVERY task of
// pass n of that chain wrote (the transposition), so consecutive passes of one chain are separated by an all-to-all
// dependency; two chains (the engine's two lanes) are independent of each other.
//   A  streams   : each chain on its own stream, one kernel per pass (what the engine does today; kernel boundaries order a chain)
//   B  queue     : ONE launch of 512 workgroups (two per CU, as today's kernels).  Eight per-XCD task queues (workgroups
//                  of one XCD keep owning consecutive row groups: the 32-byte segments of their stores still meet in that
//                  L2) hold the tasks in the order pass 0 chain 0, pass 0 chain 1, pass 1 chain 0, ...; a workgroup takes the
//                  next task of its XCD's queue with one atomic add, waits until the counter of (chain, pass - 1) has
//                  reached 512 (one lane polls with sc1 loads + s_sleep; agent-scope acquire; barrier), runs the task, and
//                  publishes it (every wave's vmcnt(0), barrier, lane-0 agent-scope release, vmcnt(0), agent-scope add):
//                  the guide's "plain payload -> release fence -> counter" form.  While the tail of a pass of chain 0
//                  drains, the workgroups that are free already run tasks of chain 1: no kernel boundary, no generation.
//   C  queue, write-through: the same with sc1 stores of the payload (no release fence; the store segments then do not
//                  merge in L2).
// Every variant is checked word by word (a chain of transpositions of a known grid) before it is timed.
// Build: hipcc --offload-arch=gfx950 -O3 -o task_queue_model task_queue_model.hip ; run: ./task_queue_model [passes]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float cf __attribute__((ext_vector_type(2)));

constexpr int N = 2048, R = 4, P = N / 64, THR = 64 * R, ROWP = N + 32 / R, NG = N / R, PER = NG / 8;
constexpr int MAXPASS = 64, NCHAIN = 2, LINE = 32; // counters on 128-byte lines of their own

struct Queue {
    unsigned head[8 * LINE];
    unsigned done[NCHAIN * MAXPASS * LINE];
    unsigned err[LINE];
    unsigned xdone[NCHAIN * MAXPASS * 8 * LINE]; // variant D: tasks of (chain, pass) finished per XCD
};

__device__ __forceinline__ int group_of(int v, int nwg)
{
    const int q = nwg >> 3, rem = nwg & 7, xcd = v & 7, k = v >> 3;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
}

__device__ __forceinline__ void store_sc1(cf* p, cf v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// one row-group task: rows row0 .. row0 + 3 of `in` -> columns row0 .. row0 + 3 of `out`
template <bool SC1>
__device__ __forceinline__ void task(const cf* __restrict__ in, cf* __restrict__ out, int pitch, int row0, int spin, float one, float zero, cf* lds)
{
    const int tid = threadIdx.x, w = tid >> 6, t = tid & 63;
    const cf* src = in + (size_t)(row0 + w) * pitch;
    cf a[P];
#pragma unroll
    for (int l = 0; l < P; l++) {
        if constexpr (SC1) { // global_ sc1 loads to registers (L1 bypassed; never flat_): the consumer side of a write-through hand-off
            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(a[l]) : "v"(src + t + 64 * l) : "memory");
        } else a[l] = src[t + 64 * l];
    }
    if constexpr (SC1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int l = 0; l < P; l++) asm volatile("" : "+v"(a[l])); // uses stay behind the wait
    }
    const cf o2 = cf{one, one}, z2 = cf{zero, zero};
    for (int i = 0; i < spin; i++) {
#pragma unroll
        for (int l = 0; l < P; l++) a[l] = __builtin_elementwise_fma(a[l], o2, z2); // v_pk_fma_f32, values unchanged
    }
    cf* xr = lds + w * ROWP;
#pragma unroll
    for (int l = 0; l < P; l++) xr[t + 64 * l] = a[l];
    __syncthreads();
    const int rr = tid % R, c0 = tid / R;
    cf* dst = out + (size_t)c0 * pitch + row0 + rr;
#pragma unroll
    for (int it = 0; it < P; it++) {
        const cf v = lds[rr * ROWP + c0 + 64 * it];
        if constexpr (SC1) store_sc1(dst + (size_t)(64 * it) * pitch, v);
        else dst[(size_t)(64 * it) * pitch] = v;
    }
}

// A: one pass of one chain as a kernel
__global__ __launch_bounds__(THR, 2) void k_pass(const cf* __restrict__ in, cf* __restrict__ out, int pitch, int spin, float one, float zero)
{
    extern __shared__ cf lds[];
    task<false>(in, out, pitch, group_of((int)blockIdx.x, (int)gridDim.x) * R, spin, one, zero, lds);
}

// B / C: the whole job as one launch
template <int MODE> // 0 = B (release per task), 1 = C (write-through payload), 2 = D (one release per XCD and pass)
__global__ __launch_bounds__(THR, 2) void k_queue(cf* const* __restrict__ grids, Queue* q, int pitch, int npass, int nsets, int spin, float one, float zero)
{
    extern __shared__ cf lds[];
    __shared__ int s_task;
    const int tid = threadIdx.x;
    // wave 0 pulls, polls and publishes for the workgroup under WAVE-UNIFORM conditions (scalar branches): with per-lane
    // conditions (tid == 0) the compiler's structuriser moved lane 0's work into an outer loop and let the other lanes of its
    // wave run ahead into the next iteration's barrier - the first build of this model hung on exactly that
    constexpr bool SC1 = MODE == 1;
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    // B, C: blockIdx % 8 as the XCD label (speed only).  D: the XCD's real id - there one workgroup writes back the L2 that
    // the other workgroups of its XCD have stored into, so the label must be the truth
    const int xcd = MODE == 2 ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : ((int)blockIdx.x & 7);
    const unsigned need = MODE == 2 ? 8u : (unsigned)NG;
    const int ntask = npass * NCHAIN * PER;
    for (;;) {
        if (wave0) {
            unsigned v = 0;
            if (tid == 0) v = __hip_atomic_fetch_add(&q->head[xcd * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __builtin_amdgcn_readfirstlane(v);
            if (tid == 0) s_task = (int)v;
        }
        __syncthreads();
        const int i = __builtin_amdgcn_readfirstlane(s_task);
        if (i >= ntask) break; // every workgroup reaches this: the queue is finite
        const int n = i / (NCHAIN * PER), c = (i / PER) % NCHAIN, k = i % PER;
        if (n > 0 && wave0) {
            const unsigned* flag = &q->done[(c * MAXPASS + n - 1) * LINE];
            int guard = 0;
            // all lanes of the wave read the same word (one request); the loop condition is made scalar
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
                __builtin_amdgcn_s_sleep(2);
                if (++guard > (1 << 17) || __builtin_amdgcn_readfirstlane(__hip_atomic_load(&q->err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { // bounded: a wrong protocol shows as an error, not as a hang
                    __hip_atomic_store(&q->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            if constexpr (!SC1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        const cf* in = grids[c * nsets + n % nsets];
        cf* out = grids[c * nsets + (n + 1) % nsets];
        if constexpr (SC1) {
            // sc1 loads to registers in place of the acquire (the guide's consumer form for write-through payloads)
            task<true>(in, out, pitch, (xcd * PER + k) * R, spin, one, zero, lds);
        } else {
            task<false>(in, out, pitch, (xcd * PER + k) * R, spin, one, zero, lds);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // all stores of the workgroup have drained; the tile and s_task are free again
        if (wave0) {
            if constexpr (MODE == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr (MODE == 2) {
                // every store of this workgroup has been acknowledged by the XCD's L2 (the waits and the barrier above).  The
                // workgroup that finishes the XCD's last task of (chain, pass) writes that L2 back ONCE for all of them and
                // tells the chip; the partial lines of the others' transposed stores are left to merge until then
                unsigned old = 0;
                if (tid == 0) old = __hip_atomic_fetch_add(&q->xdone[((c * MAXPASS + n) * 8 + xcd) * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                old = __builtin_amdgcn_readfirstlane(old);
                if (old == (unsigned)(PER - 1)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (tid == 0) __hip_atomic_fetch_add(&q->done[(c * MAXPASS + n) * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (tid == 0) __hip_atomic_fetch_add(&q->done[(c * MAXPASS + n) * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int npass = argc > 1 ? atoi(argv[1]) : 24;
    const int vmask = argc > 2 ? atoi(argv[2]) : 15; // 1 = streams, 2 = queue B, 4 = queue C, 8 = queue D
    const bool verbose = getenv("TQ_VERBOSE") != nullptr; // passes per chain and job (even: a chain of transpositions ends on its input)
    const int pitch = N + 32;
    const size_t ne = (size_t)N * pitch;
    const size_t ldsb = sizeof(cf) * (size_t)ROWP * R;
    if (npass < 2 || npass > MAXPASS || (npass & 1)) { printf("passes: even, 2..%d\n", MAXPASS); return 1; }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pass), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_queue<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_queue<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_queue<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_queue<0>, THR, ldsb));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int slots = occ * prop.multiProcessorCount;
    const int grid = slots >= NG ? NG : (slots & ~7);
    printf("# task_queue_model: N=%d pitch=%d, %d passes per chain, %d chains; queue kernel: %d workgroups per CU admitted, grid %d\n", N, pitch, npass, NCHAIN, occ, grid);
    hipStream_t st[2];
    for (int q = 0; q < 2; q++) CK(hipStreamCreateWithFlags(&st[q], hipStreamNonBlocking));
    std::vector<float> hin(ne * 2), ho(ne * 2);
    unsigned s = 12345u;
    for (auto& v : hin) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8); }
    Queue* dq;
    CK(hipMalloc(&dq, sizeof(Queue)));

    for (int nsets : {2, 4}) { // grids a chain cycles through: 2 = cache-resident at 2048^2, 4 = the engine's working set per lane
        std::vector<cf*> g((size_t)NCHAIN * nsets);
        for (auto& p : g) { CK(hipMalloc(&p, ne * 8)); CK(hipMemset(p, 0, ne * 8)); }
        cf** dg;
        CK(hipMalloc(&dg, g.size() * sizeof(cf*)));
        CK(hipMemcpy(dg, g.data(), g.size() * sizeof(cf*), hipMemcpyHostToDevice));
        auto seed = [&]() {
            for (int c = 0; c < NCHAIN; c++) {
                CK(hipMemcpy(g[(size_t)c * nsets], hin.data(), ne * 8, hipMemcpyHostToDevice));
                for (int k = 1; k < nsets; k++) CK(hipMemset(g[(size_t)c * nsets + k], 0, ne * 8));
            }
            CK(hipDeviceSynchronize());
        };
        // the final grid of a chain: index npass % nsets; even number of transpositions: equal to the input (inside the N x N part)
        auto check = [&]() -> size_t {
            size_t bad = 0;
            for (int c = 0; c < NCHAIN; c++) {
                CK(hipMemcpy(ho.data(), g[(size_t)c * nsets + npass % nsets], ne * 8, hipMemcpyDeviceToHost));
                for (int r = 0; r < N; r++)
                    for (int col = 0; col < N; col++) {
                        const size_t i = ((size_t)r * pitch + col) * 2;
                        bad += (ho[i] != hin[i] || ho[i + 1] != hin[i + 1]);
                    }
            }
            return bad;
        };
        auto run_streams = [&](int nchain, int spin) {
            for (int n = 0; n < npass; n++)
                for (int c = 0; c < nchain; c++)
                    hipLaunchKernelGGL(k_pass, dim3(NG), dim3(THR), ldsb, st[c], g[(size_t)c * nsets + n % nsets], g[(size_t)c * nsets + (n + 1) % nsets], pitch, spin, 1.0f, 0.0f);
        };
        auto run_queue = [&](int mode, int spin) {
            CK(hipMemsetAsync(dq, 0, sizeof(Queue), st[0]));
            if (mode == 1) hipLaunchKernelGGL(k_queue<1>, dim3(grid), dim3(THR), ldsb, st[0], dg, dq, pitch, npass, nsets, spin, 1.0f, 0.0f);
            else if (mode == 2) hipLaunchKernelGGL(k_queue<2>, dim3(grid), dim3(THR), ldsb, st[0], dg, dq, pitch, npass, nsets, spin, 1.0f, 0.0f);
            else hipLaunchKernelGGL(k_queue<0>, dim3(grid), dim3(THR), ldsb, st[0], dg, dq, pitch, npass, nsets, spin, 1.0f, 0.0f);
        };
        auto timed = [&](auto&& f) -> double {
            const int reps = 10;
            for (int k = 0; k < 2; k++) f();
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < reps; k++) f();
            CK(hipDeviceSynchronize());
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        };
        for (int spin : {0, 30, 60}) {
            const double vec_us = spin * P * 4.0 / 2400.0; // packed FMA: 4 cycles per wave instruction at 2.4 GHz, per wave alone on its SIMD
            // correctness of each variant from a fresh seed
            unsigned herr[LINE] = {0}, herrC[LINE] = {0};
            size_t badA = 0, badB = 0, badC = 0;
            double a1 = 0, a2 = 0, b = 0, cq = 0;
            if (vmask & 1) {
                if (verbose) printf("   streams ...\n");
                seed(); run_streams(NCHAIN, spin); CK(hipDeviceSynchronize()); badA = check();
                a1 = timed([&] { run_streams(1, spin); }) / npass;
                a2 = timed([&] { run_streams(NCHAIN, spin); }) / (npass * NCHAIN);
                if (verbose) printf("   streams: wrong words %zu, %.2f / %.2f us per pass\n", badA, a1, a2);
            }
            if (vmask & 2) {
                if (verbose) printf("   queue ...\n");
                seed(); run_queue(0, spin); CK(hipDeviceSynchronize()); badB = check();
                CK(hipMemcpy(herr, dq->err, sizeof(herr), hipMemcpyDeviceToHost));
                if (verbose) printf("   queue: wrong words %zu, wait bound %u\n", badB, herr[0]);
                b = timed([&] { run_queue(0, spin); }) / (npass * NCHAIN);
            }
            if (vmask & 4) {
                if (verbose) printf("   queue, write-through ...\n");
                seed(); run_queue(1, spin); CK(hipDeviceSynchronize()); badC = check();
                CK(hipMemcpy(herrC, dq->err, sizeof(herrC), hipMemcpyDeviceToHost));
                if (verbose) printf("   queue, write-through: wrong words %zu, wait bound %u\n", badC, herrC[0]);
                cq = timed([&] { run_queue(1, spin); }) / (npass * NCHAIN);
            }
            unsigned herrD[LINE] = {0};
            size_t badD = 0;
            double dq_us = 0;
            if (vmask & 8) {
                if (verbose) printf("   queue, one release per XCD ...\n");
                seed(); run_queue(2, spin); CK(hipDeviceSynchronize()); badD = check();
                CK(hipMemcpy(herrD, dq->err, sizeof(herrD), hipMemcpyDeviceToHost));
                if (verbose) printf("   queue, one release per XCD: wrong words %zu, wait bound %u\n", badD, herrD[0]);
                dq_us = timed([&] { run_queue(2, spin); }) / (npass * NCHAIN);
            }
            herr[0] |= (herrC[0] << 1) | (herrD[0] << 2);
            badC += badD;
            printf("sets=%d spin=%2d (%4.1f us of packed FMA per wave) : A one stream %6.2f us/pass | A two streams %6.2f | B queue, release/acquire %6.2f (%+5.1f %%) | C queue, write-through %6.2f (%+5.1f %%) | D queue, one release per XCD and pass %6.2f (%+5.1f %%) %s%s\n",
                   nsets, spin, vec_us, a1, a2, b, 100.0 * (a2 / b - 1.0), cq, 100.0 * (a2 / cq - 1.0), dq_us, 100.0 * (a2 / dq_us - 1.0), (badA | badB | badC) ? "** WRONG **" : "", herr[0] ? " ** wait bound hit **" : "");
            if (badA | badB | badC) printf("   wrong words: streams %zu, queue %zu, queue write-through %zu\n", badA, badB, badC);
            fflush(stdout);
        }
        for (auto p : g) CK(hipFree(p));
        CK(hipFree(dg));
    }
    CK(hipFree(dq));
    return 0;
}
