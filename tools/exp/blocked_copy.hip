// blocked_copy.hip — micro-experiment: memory side of a "2 x 2 blocked" grid layout for the row passes.
//   natural copy        : 16 B per lane, fully contiguous (the floor)
//   blocked transpose   : a workgroup owns ONE ROW PAIR (rows 2i, 2i+1 of an N x N float2 grid stored as 2 x 2 blocks:
//                         block (i, j) = [(2i,2j), (2i,2j+1), (2i+1,2j), (2i+1,2j+1)], 32 B, blocks of a row pair
//                         contiguous), loads it with 16-byte loads, swaps halves between even / odd lanes (DPP) so that
//                         a lane holds (row 2i, col c) and (row 2i+1, col c), and stores the TRANSPOSED blocks: for column
//                         pair j the 32-byte block [(2j,2i), (2j,2i+1), (2j+1,2i), (2j+1,2i+1)] at ((j * PB) + i) * 32 B.
// Build: hipcc --offload-arch=gfx950 -O3 -o blocked_copy blocked_copy.hip ; run: ./blocked_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_nat(const float4* __restrict__ in, float4* __restrict__ out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// N columns; T = N / 16 threads per row pair; thread t handles columns c = t + T l, l = 0..15
template <int N> __global__ __launch_bounds__(N / 16) void k_blk(const float4* __restrict__ in, float4* __restrict__ out, int pbi, int pbo)
{
    constexpr int T = N / 16;
    // XCD-aware order: blocks with equal blockIdx % 8 share an XCD and its L2: give them consecutive row pairs, so that
    // the four 32-byte blocks of a 128-byte output line are written through the same L2
    const int t = threadIdx.x, i = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    const float4* src = in + (size_t)i * (size_t)pbi * 2; // row pair i: pbi blocks of 2 float4
    float4 v[16];
#pragma unroll
    for (int l = 0; l < 16; l++) v[l] = src[t + T * l]; // lane t: float4 number c of the pair's stream = half (c & 1) of block c / 2
    // even lane holds (r0,c),(r0,c+1); odd lane holds (r1,c-1),(r1,c): exchange so that lane c holds (r0,c),(r1,c)
#pragma unroll
    for (int l = 0; l < 16; l++) {
        const bool odd = t & 1;
        // send what the partner needs: even lane gives (r0, c+1) = v.zw, odd lane gives (r1, c-1) = v.xy
        float sx = odd ? v[l].x : v[l].z, sy = odd ? v[l].y : v[l].w;
        float rx = __shfl_xor(sx, 1), ry = __shfl_xor(sy, 1);
        if (odd) { v[l].x = rx; v[l].y = ry; } else { v[l].z = rx; v[l].w = ry; }
        // now even: (r0,c),(r1,c)   odd: (r0,c),(r1,c) with x,y = r0 and z,w = r1
    }
    // transposed block store: output row pair = column pair j = c / 2, block index within it = i
#pragma unroll
    for (int l = 0; l < 16; l++) {
        const int c = t + T * l;
        out[((size_t)(c >> 1) * (size_t)pbo + (size_t)i) * 2 + (c & 1)] = v[l];
    }
}

template <int N> int run(int pad)
{
    const int pb = N / 2 + pad; // blocks per row pair (pitch)
    const size_t n4 = (size_t)(N / 2) * pb * 2;
    float4 *a[2], *b[2];
    hipStream_t st[2];
    for (int q = 0; q < 2; q++) {
        CK(hipMalloc(&a[q], n4 * 16)); CK(hipMalloc(&b[q], n4 * 16));
        CK(hipMemset(a[q], 1, n4 * 16)); CK(hipMemset(b[q], 0, n4 * 16));
        CK(hipStreamCreateWithFlags(&st[q], hipStreamNonBlocking));
    }
    for (int mode = 0; mode < 2; mode++)
        for (int ns = 1; ns <= 2; ns++) {
            auto go = [&](int q) {
                if (mode == 0) hipLaunchKernelGGL(k_nat, dim3(2048), dim3(256), 0, st[q], a[q], b[q], n4);
                else hipLaunchKernelGGL((k_blk<N>), dim3(N / 2), dim3(N / 16), 0, st[q], a[q], b[q], pb, pb);
            };
            for (int q = 0; q < ns; q++) go(q);
            CK(hipDeviceSynchronize());
            const int it = 200;
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < it; k++) for (int q = 0; q < ns; q++) go(q);
            CK(hipDeviceSynchronize());
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (it * ns);
            printf("N=%d pad=%d %s streams=%d: %7.2f us per launch  %.2f TB/s\n", N, pad, mode ? "blocked transpose" : "natural copy     ", ns, us,
                   2.0 * N * N * 8 / us / 1e6);
        }
    // correctness of the blocked transpose: out block (j, i) must equal the transpose of in block (i, j)
    std::vector<float> hin(n4 * 4), hout(n4 * 4);
    for (size_t k = 0; k < hin.size(); k++) hin[k] = (float)(k % 1000003);
    CK(hipMemcpy(a[0], hin.data(), n4 * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_blk<N>), dim3(N / 2), dim3(N / 16), 0, st[0], a[0], b[0], pb, pb);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hout.data(), b[0], n4 * 16, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int i = 0; i < N / 2; i++)
        for (int j = 0; j < N / 2; j++) {
            const float* bi = &hin[((size_t)i * pb + j) * 8];  // [(r0,c0),(r0,c1),(r1,c0),(r1,c1)] x float2
            const float* bo = &hout[((size_t)j * pb + i) * 8]; // [(c0,r0),(c0,r1),(c1,r0),(c1,r1)]
            const int map[4] = {0, 2, 1, 3};
            for (int e = 0; e < 4; e++) for (int k = 0; k < 2; k++) if (bo[2 * e + k] != bi[2 * map[e] + k]) bad++;
        }
    printf("N=%d pad=%d blocked transpose mismatches: %ld\n", N, pad, bad);
    for (int q = 0; q < 2; q++) { hipFree(a[q]); hipFree(b[q]); hipStreamDestroy(st[q]); }
    return bad ? 1 : 0;
}

int main()
{
    int rc = 0;
    for (int pad : {0, 32}) { rc |= run<2048>(pad); rc |= run<4096>(pad); }
    return rc;
}
