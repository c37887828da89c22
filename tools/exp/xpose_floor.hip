// xpose_floor.hip — round 4: which side of a pass should carry the transposition, and how wide should its accesses be?
//
// Every pass of the slice loop reads whole rows and ends in a TRANSPOSED store (R rows per workgroup -> R * 8-byte
// segments per output row, staged through LDS); DESIGN 4.1 measured that store at 4.45-5.3 TB/s against 5.6-6.5 TB/s
// for natural stores.  This program measures the bare data movement of the alternatives, no transform:
//   mode 0  natural load  -> natural store                          (the floor)
//   mode 1  natural load  -> LDS tile -> transposed store           (what the passes do today; R * SW/8... segments)
//   mode 2  transposed load (R * 8-byte segments of every input row) -> LDS tile -> natural store
// for R = 4 / 8 rows per workgroup (one wave per row, as fft_wave.hip), 8- or 16-byte accesses per lane on either side,
// dense and padded rows, one and two streams, a cache-resident (2 grids per stream) and a cold (16 grids) working set.
// Every variant is checked against a host transposition before it is timed.
// Build: hipcc --offload-arch=gfx950 -O3 -o xpose_floor xpose_floor.hip ; run: ./xpose_floor [2048|4096]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float cf __attribute__((ext_vector_type(2)));
typedef float cf2 __attribute__((ext_vector_type(4)));

// workgroups that share an XCD (blockIdx % 8) own consecutive row groups
__device__ __forceinline__ int group_of(int v, int nwg)
{
    const int q = nwg >> 3, rem = nwg & 7, xcd = v & 7, k = v >> 3;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
}

template <int N, int R, int MODE> constexpr int rowp()
{
    // mode 1: the half-wave of a column-wise ds_read_b64 (R rows x 32 / R columns) must fall on 32 distinct 8-byte slots
    // mode 2: the 16-lane group of a column-wise ds_write_b64 (R rows x 16 / R columns) on 16 distinct slots
    return MODE == 2 ? N + 16 / R : N + 32 / R;
}

template <int N, int R, int MODE, int LW, int SW>
__global__ __launch_bounds__(64 * R) void k_move(const cf* __restrict__ in, cf* __restrict__ out, int pitch)
{
    constexpr int P = N / 64, THR = 64 * R, ROWP = rowp<N, R, MODE>();
    static_assert(ROWP % 2 == 0, "16-byte LDS accesses");
    extern __shared__ cf lds[];
    const int tid = threadIdx.x, w = tid >> 6, t = tid & 63;
    const int row0 = group_of((int)blockIdx.x, (int)gridDim.x) * R;
    if constexpr (MODE == 0) {
        const cf* src = in + (size_t)(row0 + w) * pitch;
        cf* dst = out + (size_t)(row0 + w) * pitch;
        if constexpr (LW == 8) {
            cf a[P];
#pragma unroll
            for (int l = 0; l < P; l++) a[l] = src[t + 64 * l];
#pragma unroll
            for (int l = 0; l < P; l++) dst[t + 64 * l] = a[l];
        } else {
            cf2 a[P / 2];
#pragma unroll
            for (int l = 0; l < P / 2; l++) a[l] = reinterpret_cast<const cf2*>(src)[t + 64 * l];
#pragma unroll
            for (int l = 0; l < P / 2; l++) reinterpret_cast<cf2*>(dst)[t + 64 * l] = a[l];
        }
        return;
    }
    if constexpr (MODE == 1) {
        const cf* src = in + (size_t)(row0 + w) * pitch;
        cf* xr = lds + w * ROWP;
        if constexpr (LW == 8) {
            cf a[P];
#pragma unroll
            for (int l = 0; l < P; l++) a[l] = src[t + 64 * l];
#pragma unroll
            for (int l = 0; l < P; l++) xr[t + 64 * l] = a[l];
        } else {
            cf2 a[P / 2];
#pragma unroll
            for (int l = 0; l < P / 2; l++) a[l] = reinterpret_cast<const cf2*>(src)[t + 64 * l];
#pragma unroll
            for (int l = 0; l < P / 2; l++) reinterpret_cast<cf2*>(xr)[t + 64 * l] = a[l];
        }
        __syncthreads();
        if constexpr (SW == 8) {
            const int rr = tid % R, c0 = tid / R; // c0 < 64
            cf* dst = out + (size_t)c0 * pitch + row0 + rr;
#pragma unroll
            for (int it = 0; it < P; it++) dst[(size_t)(64 * it) * pitch] = lds[rr * ROWP + c0 + 64 * it];
        } else {
            const int rp = tid % (R / 2), c0 = tid / (R / 2); // c0 < 128
            cf* dst = out + (size_t)c0 * pitch + row0 + 2 * rp;
#pragma unroll
            for (int it = 0; it < P / 2; it++) {
                const cf u = lds[(2 * rp) * ROWP + c0 + 128 * it], v = lds[(2 * rp + 1) * ROWP + c0 + 128 * it];
                *reinterpret_cast<cf2*>(dst + (size_t)(128 * it) * pitch) = cf2{u.x, u.y, v.x, v.y};
            }
        }
        return;
    }
    if constexpr (MODE == 2) {
        if constexpr (LW == 8) {
            const int rr = tid % R, r0 = tid / R; // r0 < 64
            const cf* src = in + (size_t)r0 * pitch + row0 + rr;
            cf a[P];
#pragma unroll
            for (int it = 0; it < P; it++) a[it] = src[(size_t)(64 * it) * pitch];
#pragma unroll
            for (int it = 0; it < P; it++) lds[rr * ROWP + r0 + 64 * it] = a[it];
        } else {
            const int rp = tid % (R / 2), r0 = tid / (R / 2); // r0 < 128
            const cf* src = in + (size_t)r0 * pitch + row0 + 2 * rp;
            cf2 a[P / 2];
#pragma unroll
            for (int it = 0; it < P / 2; it++) a[it] = *reinterpret_cast<const cf2*>(src + (size_t)(128 * it) * pitch);
#pragma unroll
            for (int it = 0; it < P / 2; it++) {
                lds[(2 * rp) * ROWP + r0 + 128 * it] = cf{a[it].x, a[it].y};
                lds[(2 * rp + 1) * ROWP + r0 + 128 * it] = cf{a[it].z, a[it].w};
            }
        }
        __syncthreads();
        cf* dst = out + (size_t)(row0 + w) * pitch;
        const cf* xr = lds + w * ROWP;
        if constexpr (SW == 8) {
#pragma unroll
            for (int l = 0; l < P; l++) dst[t + 64 * l] = xr[t + 64 * l];
        } else {
#pragma unroll
            for (int l = 0; l < P / 2; l++) reinterpret_cast<cf2*>(dst)[t + 64 * l] = reinterpret_cast<const cf2*>(xr)[t + 64 * l];
        }
    }
}

struct Bufs {
    std::vector<cf*> g; // grids
    hipStream_t st[2];
};

template <int N, int R, int MODE, int LW, int SW> void run(Bufs& B, int pitch, int nsets, const std::vector<float>& hin, const char* tag)
{
    constexpr int ROWP = rowp<N, R, MODE>();
    const size_t ldsb = MODE == 0 ? 0 : sizeof(cf) * (size_t)ROWP * R;
    auto kern = k_move<N, R, MODE, LW, SW>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const size_t ne = (size_t)N * pitch;
    // correctness
    CK(hipMemcpy(B.g[0], hin.data(), ne * 8, hipMemcpyHostToDevice));
    CK(hipMemset(B.g[1], 0, ne * 8));
    CK(hipDeviceSynchronize()); // the streams below are non-blocking: they do not wait for the null stream's memset
    hipLaunchKernelGGL(kern, dim3(N / R), dim3(64 * R), ldsb, B.st[0], B.g[0], B.g[1], pitch);
    CK(hipStreamSynchronize(B.st[0]));
    std::vector<float> ho(ne * 2);
    CK(hipMemcpy(ho.data(), B.g[1], ne * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int r = 0; r < N; r++)
        for (int c = 0; c < N; c++) {
            const size_t s = ((size_t)r * pitch + c) * 2, d = MODE == 0 ? s : ((size_t)c * pitch + r) * 2;
            bad += (ho[d] != hin[s] || ho[d + 1] != hin[s + 1]);
        }
    // timing: stream q ping-pongs over its own nsets grids
    double res[2] = {0, 0};
    for (int ns = 1; ns <= 2; ns++) {
        const int it = 300;
        auto go = [&](int q, int k) {
            cf* a = B.g[(size_t)q * nsets + (k % nsets)];
            cf* b = B.g[(size_t)q * nsets + ((k + 1) % nsets)];
            hipLaunchKernelGGL(kern, dim3(N / R), dim3(64 * R), ldsb, B.st[q], a, b, pitch);
        };
        for (int k = 0; k < 20; k++) for (int q = 0; q < ns; q++) go(q, k);
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < it; k++) for (int q = 0; q < ns; q++) go(q, k);
        CK(hipDeviceSynchronize());
        res[ns - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (it * ns);
    }
    const double mb = 2.0 * N * N * 8 / 1e6;
    printf("N=%d pitch=%d sets=%2d %-28s R=%d L%02d S%02d lds=%6zu : x1 %7.2f us %5.2f TB/s | x2 %7.2f us %5.2f TB/s %s\n", N, pitch, nsets, tag, R, LW, SW,
           ldsb, res[0], mb / res[0], res[1], mb / res[1], bad ? "** WRONG **" : "");
    fflush(stdout);
}

template <int N> void all(int pad, int nsets)
{
    const int pitch = N + pad;
    const size_t ne = (size_t)N * pitch;
    Bufs B;
    B.g.resize((size_t)2 * nsets);
    for (auto& p : B.g) { CK(hipMalloc(&p, ne * 8)); CK(hipMemset(p, 0, ne * 8)); }
    for (int q = 0; q < 2; q++) CK(hipStreamCreateWithFlags(&B.st[q], hipStreamNonBlocking));
    std::vector<float> hin(ne * 2);
    unsigned s = 12345u;
    for (auto& v : hin) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8); }
    run<N, 4, 0, 8, 8>(B, pitch, nsets, hin, "natural");
    run<N, 4, 0, 16, 16>(B, pitch, nsets, hin, "natural");
    run<N, 4, 1, 8, 8>(B, pitch, nsets, hin, "store-T (today)");
    run<N, 4, 1, 16, 8>(B, pitch, nsets, hin, "store-T");
    run<N, 4, 1, 8, 16>(B, pitch, nsets, hin, "store-T");
    run<N, 4, 1, 16, 16>(B, pitch, nsets, hin, "store-T");
    if constexpr (N <= 2048) {
        run<N, 8, 1, 8, 8>(B, pitch, nsets, hin, "store-T");
        run<N, 8, 1, 16, 16>(B, pitch, nsets, hin, "store-T");
    }
    run<N, 4, 2, 8, 8>(B, pitch, nsets, hin, "load-T");
    run<N, 4, 2, 16, 8>(B, pitch, nsets, hin, "load-T");
    run<N, 4, 2, 8, 16>(B, pitch, nsets, hin, "load-T");
    run<N, 4, 2, 16, 16>(B, pitch, nsets, hin, "load-T");
    if constexpr (N <= 2048) {
        run<N, 8, 2, 8, 8>(B, pitch, nsets, hin, "load-T");
        run<N, 8, 2, 16, 16>(B, pitch, nsets, hin, "load-T");
    }
    for (auto p : B.g) CK(hipFree(p));
    for (int q = 0; q < 2; q++) CK(hipStreamDestroy(B.st[q]));
}

// round 5: TWO rows per workgroup (16-byte segments; the row pairs 4g, 4g+1 and 4g+2, 4g+3 go to neighbouring workgroups of one
// XCD, so that their segments can meet in that L2 as 32-byte pairs) against four, with today's XCD remap and 64-element padding
template <int N> void two_rows(int pad, int nsets)
{
    const int pitch = N + pad;
    const size_t ne = (size_t)N * pitch;
    Bufs B;
    B.g.resize((size_t)2 * nsets);
    for (auto& p : B.g) { CK(hipMalloc(&p, ne * 8)); CK(hipMemset(p, 0, ne * 8)); }
    for (int q = 0; q < 2; q++) CK(hipStreamCreateWithFlags(&B.st[q], hipStreamNonBlocking));
    std::vector<float> hin(ne * 2);
    unsigned s = 12345u;
    for (auto& v : hin) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8); }
    run<N, 4, 0, 8, 8>(B, pitch, nsets, hin, "natural");
    run<N, 4, 1, 8, 8>(B, pitch, nsets, hin, "store-T, 4 rows (today)");
    run<N, 2, 1, 8, 8>(B, pitch, nsets, hin, "store-T, 2 rows");
    run<N, 2, 1, 8, 16>(B, pitch, nsets, hin, "store-T, 2 rows");
    run<N, 2, 1, 16, 16>(B, pitch, nsets, hin, "store-T, 2 rows");
    run<N, 2, 2, 8, 8>(B, pitch, nsets, hin, "load-T, 2 rows");
    for (auto p : B.g) CK(hipFree(p));
    for (int q = 0; q < 2; q++) CK(hipStreamDestroy(B.st[q]));
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 2048;
    if (argc > 2 && argv[2][0] == 'r') { // "r2": the two-row question only
        if (n == 2048) { two_rows<2048>(32, 2); two_rows<2048>(32, 16); }
        else { two_rows<4096>(64, 2); two_rows<4096>(64, 4); }
        return 0;
    }
    if (n == 2048) {
        all<2048>(32, 2);
        all<2048>(32, 16);
        all<2048>(0, 2);
    } else {
        all<4096>(64, 2);
        all<4096>(64, 4);
    }
    return 0;
}
