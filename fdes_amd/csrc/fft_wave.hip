// fft_wave.hip — LDS row-pass kernels, ONE WAVE PER ROW (gfx950), for 1024-, 2048- and 4096-point rows.
//
// The passes of fft_lds.hip give a row to T = N/16 threads (two or four waves above 1024 points): every row FFT is three
// radix stages with TWO exchanges through LDS and an s_barrier around each.  Here a row belongs to one wave, P = N/64
// points per lane (32 at N = 2048, 64 at N = 4096):
//   lane t holds x[t + 64 l], l < P                                    (coalesced 512-byte loads, as before)
//   S1  radix-P butterfly over the registers            Y[t][k1] = sum_l x[t + 64 l] W_P^(l k1)
//   S2  twiddle                                         Y[t][k1] *= W_N^(t k1)
//   S3  (P = 32) radix-2 across the two lane halves (t = t1 + 32 t0) by v_permlane32_swap_b32:
//                                                       Z[t1][k1][b] = W_64^(t1 b) (Y[t1][k1] + (-1)^b Y[t1 + 32][k1])
//       (P = 16) radix-4 across the four lane quarters (t = t1 + 16 t0) by v_permlane32_swap_b32 + v_permlane16_swap_b32:
//                                                       Z[t1][k1][b] = W_64^(t1 b) sum_t0 Y[t1 + 16 t0][k1] W_4^(t0 b)
//   S4  ONE exchange through this wave's own LDS region (no barrier: a wave's LDS instructions execute in order)
//   S5  radix-P butterfly over the registers (index t1 resp. t)
// and the result lands as register l of lane t = X[t + 64 l], the input pattern again, so FFT -> point-wise -> inverse
// FFT chain through registers exactly as in fft_lds.hip.  Half the LDS bytes per transform (the VGPR -> LDS store path,
// about 85 B/clk/CU, is the slow side of an exchange), no barrier inside a transform, and the four waves of a workgroup
// (four rows: 32-byte transposed-store segments, as before) run independently until the staged transposed store.
// Waves may be started staggered (PassArgs::stagger): each CU then has rows in different phases - loads in flight for
// one while another is in its butterflies - instead of all rows waiting for memory and then all computing at once.
//
// Stage twiddles W_N^(t k1): P - 1 per lane; built from base powers out of the existing tables (w^b, b < NB and
// w^(NB a), a < 8 with NB = P / 8) as ONE product each, rebuilt in every transform (fft_lds.hip, TWM_POW, for why).
#include "fft_lds.h"
#include "geometry.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "fft_wave.hip uses v_permlane32_swap / v_permlane16_swap (gfx950); this library is written for MI355X only"
#endif

#include <atomic>
#include <cmath>
#include <type_traits>

namespace fdes {

namespace {

#include "fft_dev.inc"

// cos / sin (2 pi m / 64); exact zeros and ones where the angle is a multiple of pi / 2
constexpr float kC64[64] = {
    1.0f, 9.9518472667e-01f, 9.8078528040e-01f, 9.5694033573e-01f, 9.2387953251e-01f, 8.8192126435e-01f, 8.3146961230e-01f, 7.7301045336e-01f,
    7.0710678119e-01f, 6.3439328416e-01f, 5.5557023302e-01f, 4.7139673683e-01f, 3.8268343237e-01f, 2.9028467725e-01f, 1.9509032202e-01f, 9.8017140330e-02f,
    0.0f, -9.8017140330e-02f, -1.9509032202e-01f, -2.9028467725e-01f, -3.8268343237e-01f, -4.7139673683e-01f, -5.5557023302e-01f, -6.3439328416e-01f,
    -7.0710678119e-01f, -7.7301045336e-01f, -8.3146961230e-01f, -8.8192126435e-01f, -9.2387953251e-01f, -9.5694033573e-01f, -9.8078528040e-01f, -9.9518472667e-01f,
    -1.0f, -9.9518472667e-01f, -9.8078528040e-01f, -9.5694033573e-01f, -9.2387953251e-01f, -8.8192126435e-01f, -8.3146961230e-01f, -7.7301045336e-01f,
    -7.0710678119e-01f, -6.3439328416e-01f, -5.5557023302e-01f, -4.7139673683e-01f, -3.8268343237e-01f, -2.9028467725e-01f, -1.9509032202e-01f, -9.8017140330e-02f,
    0.0f, 9.8017140330e-02f, 1.9509032202e-01f, 2.9028467725e-01f, 3.8268343237e-01f, 4.7139673683e-01f, 5.5557023302e-01f, 6.3439328416e-01f,
    7.0710678119e-01f, 7.7301045336e-01f, 8.3146961230e-01f, 8.8192126435e-01f, 9.2387953251e-01f, 9.5694033573e-01f, 9.8078528040e-01f, 9.9518472667e-01f};
// forward root of unity W_64^m = exp(-2 pi i m / 64)
__device__ __forceinline__ constexpr float w64re(int m) { return kC64[m & 63]; }
__device__ __forceinline__ constexpr float w64im(int m) { return -kC64[(m + 48) & 63]; } // -sin(x) = -cos(x - pi/2)

// multiply by the compile-time root W_64^m (forward) or its conjugate (inverse); the trivial ones cost no multiply
template <bool INV, int M> __device__ __forceinline__ cf mul_w64(cf a)
{
    constexpr int m = M & 63;
    if constexpr (m == 0) return a;
    else if constexpr (m == 16) return mul_mi<INV>(a);
    else if constexpr (m == 32) return -a;
    else if constexpr (m == 48) return -mul_mi<INV>(a);
    else {
        constexpr float re = w64re(m), im = w64im(m);
        return twmul<INV>(a, cf{re, im});
    }
}

// in-place 32-point DFT, natural order in and out: one radix-2 level (twiddles W_32^j) over two radix-16 butterflies
template <bool INV> __device__ __forceinline__ void r32(cf (&a)[32])
{
    cf e[16], o[16];
#define R32_STEP(J)                                  \
    {                                                \
        e[J] = a[J] + a[J + 16];                     \
        o[J] = mul_w64<INV, 2 * J>(a[J] - a[J + 16]); \
    }
    R32_STEP(0) R32_STEP(1) R32_STEP(2) R32_STEP(3) R32_STEP(4) R32_STEP(5) R32_STEP(6) R32_STEP(7)
    R32_STEP(8) R32_STEP(9) R32_STEP(10) R32_STEP(11) R32_STEP(12) R32_STEP(13) R32_STEP(14) R32_STEP(15)
#undef R32_STEP
    r16<INV>(e); // X[2 m]
    r16<INV>(o); // X[2 m + 1]
#pragma unroll
    for (int m = 0; m < 16; m++) {
        a[2 * m] = e[m];
        a[2 * m + 1] = o[m];
    }
}

// in-place 64-point DFT, natural order in and out: l = l0 + 8 l1, k = k1 + 8 k0; radix-8 over l1, twiddle W_64^(l0 k1),
// radix-8 over l0
template <bool INV, int L0> __device__ __forceinline__ void r64_twiddle_row(cf (&a)[64])
{
    // a[L0 + 8 k1] *= W_64^(L0 k1), k1 = 1..7
    a[L0 + 8] = mul_w64<INV, L0 * 1>(a[L0 + 8]);
    a[L0 + 16] = mul_w64<INV, L0 * 2>(a[L0 + 16]);
    a[L0 + 24] = mul_w64<INV, L0 * 3>(a[L0 + 24]);
    a[L0 + 32] = mul_w64<INV, L0 * 4>(a[L0 + 32]);
    a[L0 + 40] = mul_w64<INV, L0 * 5>(a[L0 + 40]);
    a[L0 + 48] = mul_w64<INV, L0 * 6>(a[L0 + 48]);
    a[L0 + 56] = mul_w64<INV, L0 * 7>(a[L0 + 56]);
}
template <bool INV> __device__ __forceinline__ void r64(cf (&a)[64])
{
#pragma unroll
    for (int l0 = 0; l0 < 8; l0++) r8<INV>(a[l0], a[l0 + 8], a[l0 + 16], a[l0 + 24], a[l0 + 32], a[l0 + 40], a[l0 + 48], a[l0 + 56]);
    r64_twiddle_row<INV, 1>(a); r64_twiddle_row<INV, 2>(a); r64_twiddle_row<INV, 3>(a); r64_twiddle_row<INV, 4>(a);
    r64_twiddle_row<INV, 5>(a); r64_twiddle_row<INV, 6>(a); r64_twiddle_row<INV, 7>(a);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++)
        r8<INV>(a[8 * k1], a[8 * k1 + 1], a[8 * k1 + 2], a[8 * k1 + 3], a[8 * k1 + 4], a[8 * k1 + 5], a[8 * k1 + 6], a[8 * k1 + 7]);
    // a[8 k1 + k0] = X[k1 + 8 k0]: transpose the register names
    cf t_;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = i + 1; j < 8; j++) { t_ = a[8 * i + j]; a[8 * i + j] = a[8 * j + i]; a[8 * j + i] = t_; }
}
template <int P, bool INV> __device__ __forceinline__ void rP(cf (&a)[P])
{
    if constexpr (P == 16) r16<INV>(a);
    else if constexpr (P == 32) r32<INV>(a);
    else r64<INV>(a);
}

// Geometry: four waves = four rows per workgroup.  A wave's LDS region holds either the exchange image of its row
// (P rows of 64 + 1 elements: the stride-65 writes of a 16-lane group fall on distinct banks, the reads are contiguous)
// or the row in natural order for the staged transposed store; regions are ROWP apart, ROWP = 8 (mod 32) elements, so
// that the 4 rows x 8 columns a half-wave reads for the transposed store fall on 64 distinct banks.
// RR = 8 (2048-point rows, PassArgs::wg = 128): eight waves = eight rows, 64-byte segments, one workgroup per CU; the
// half-wave of the staged store then reads 8 rows x 4 columns, regions 4 (mod 32) elements apart.
template <int N, int RR = 4> struct WaveGeo {
    static constexpr int P = N / 64;
    static constexpr int R = RR;
    static constexpr int THR = 64 * R;
    static constexpr int XROW = P * 65;
    static constexpr int ROWP = ((XROW + 31) / 32) * 32 + 32 / R;
    static constexpr int NB = P / 8;       // twiddle k = NB a + b
    static constexpr size_t LDS_BYTES = sizeof(float) * 2 * (size_t)ROWP * R + 64 + 512; // + the sine / cosine table (FDES_W_SINCOS_TAB)
    static_assert(XROW >= N && ROWP % 32 == 32 / R && (R == 4 || R == 8), "region layout");
};

// base powers of the stage twiddle w = W_N^t of lane t, and (P = 32) the radix-2 twiddle W_64^(t mod 32)
template <int P> struct TwWave {
    cf lo[P / 8 - 1]; // w^b, b = 1 .. NB - 1
    cf hi[7];         // w^(NB a), a = 1 .. 7
    cf w64[3];        // P = 32: [0] = W_64^(t1); P = 16: W_64^(t1 b), b = 1, 2, 3
};
// W_N^(t m) from the table tw0[k * (N/16) + t'] = W_N^(t' k) (k < 16, t' < N/16): m = k s with s a power of two, t' = s t
__host__ __device__ constexpr int tw_step(int m) { int s = 1; while (m / s >= 16) s *= 2; return s; }
template <int N> __device__ __forceinline__ cf tw_lookup(const cf* __restrict__ tw0, int m, int t)
{
    const int s = tw_step(m);
    return tw0[(m / s) * (N / 16) + s * t];
}
template <int N> __device__ __forceinline__ void tw_load(TwWave<N / 64>& tw, const cf* __restrict__ tw0, const int t)
{
    constexpr int P = N / 64, NB = P / 8;
#pragma unroll
    for (int b = 1; b < NB; b++) tw.lo[b - 1] = tw_lookup<N>(tw0, b, t);
#pragma unroll
    for (int a = 1; a < 8; a++) tw.hi[a - 1] = tw_lookup<N>(tw0, NB * a, t);
    if constexpr (P == 32) tw.w64[0] = tw_lookup<N>(tw0, 32, t & 31); // W_64^(t1) = W_2048^(32 t1)
    if constexpr (P == 16) {                                          // W_64^(t1 b) = W_1024^(16 b t1)
#pragma unroll
        for (int b = 1; b < 4; b++) tw.w64[b - 1] = tw_lookup<N>(tw0, 16 * b, t & 15);
    }
}

// makes a base twiddle opaque to the optimiser, so that the products built from it are rebuilt per transform instead of
// being computed once and kept alive across the pass (fft_lds.hip, TWM_POW)
__device__ __forceinline__ void tw_opaque(cf& x) { asm volatile("" : "+v"(x)); }

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("" ::: "memory");
}

// one row FFT of this wave: a[l] = x[t + 64 l] on entry, X[t + 64 l] on exit
// HX: the exchange goes through a window of HALF the size, real parts first, imaginary parts second (same LDS bytes in
// twice as many 32-bit accesses): a row then occupies 4 N bytes of LDS and twice as many rows fit on a CU
template <int N, bool INV, bool HX = false>
__device__ __forceinline__ void wave_fft(cf (&a)[N / 64], cf* __restrict__ xr, const int t, const TwWave<N / 64>& tw)
{
    constexpr int P = N / 64, NB = P / 8;
    rP<P, INV>(a);
    {   // S2: stage twiddles, each ONE product of two table values (or a table value)
        cf lo[NB - 1], hi[7];
#pragma unroll
        for (int b = 0; b < NB - 1; b++) { lo[b] = tw.lo[b]; tw_opaque(lo[b]); }
#pragma unroll
        for (int q = 0; q < 7; q++) { hi[q] = tw.hi[q]; tw_opaque(hi[q]); }
#pragma unroll
        for (int b = 1; b < NB; b++) a[b] = twmul_rt<INV>(a[b], lo[b - 1]);
#pragma unroll
        for (int q = 1; q < 8; q++) {
            a[NB * q] = twmul_rt<INV>(a[NB * q], hi[q - 1]);
#pragma unroll
            for (int b = 1; b < NB; b++) a[NB * q + b] = twmul_rt<INV>(a[NB * q + b], cmul_rt(hi[q - 1], lo[b - 1]));
        }
    }
    if constexpr (P == 32) {
        // S3: radix-2 over the lane halves.  v_permlane32_swap_b32 vdst, src swaps lanes 32-63 of vdst with lanes 0-31 of
        // src: afterwards a[c] holds the t0 = 0 value and a[c + 16] the t0 = 1 value of k1 = c + 16 h (h = this lane's half)
        cf w = tw.w64[0];
        tw_opaque(w);
#pragma unroll
        for (int c = 0; c < 16; c++) {
            typedef unsigned u2_ __attribute__((ext_vector_type(2)));
            const u2_ sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[c].x), __float_as_uint(a[c + 16].x), false, false);
            const u2_ sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[c].y), __float_as_uint(a[c + 16].y), false, false);
            const cf p = cf{__uint_as_float(sx.x), __uint_as_float(sy.x)}, q = cf{__uint_as_float(sx.y), __uint_as_float(sy.y)};
            a[c] = p + q;
            a[c + 16] = twmul_rt<INV>(p - q, w);
        }
        // S4: element (k1 = c + 16 h, b, t1) goes to lane k1 + 32 b, register t1 (below)
    } else if constexpr (P == 16) {
        // S3: radix-4 over the four lane quarters, t = t1 + 16 t0.  Two swap levels bring the four quarters' values of
        // k1 = c + 4 t0 (c < 4; t0 = this lane's quarter) into registers a[c + 4 g], g = quarter:
        //   v_permlane32_swap a[c], a[c + 8] (c < 8): lanes of half h keep k1 = c + 8 h, quarters (t0 & 1) + {0, 2};
        //   v_permlane16_swap a[c + 8 q], a[c + 4 + 8 q] (c < 4): odd rows of vdst <-> even rows of src.
        typedef unsigned u2_ __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const u2_ sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[c].x), __float_as_uint(a[c + 8].x), false, false);
            const u2_ sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[c].y), __float_as_uint(a[c + 8].y), false, false);
            a[c] = cf{__uint_as_float(sx.x), __uint_as_float(sy.x)};
            a[c + 8] = cf{__uint_as_float(sx.y), __uint_as_float(sy.y)};
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const u2_ sx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[c + 8 * q].x), __float_as_uint(a[c + 4 + 8 * q].x), false, false);
                const u2_ sy = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[c + 8 * q].y), __float_as_uint(a[c + 4 + 8 * q].y), false, false);
                a[c + 8 * q] = cf{__uint_as_float(sx.x), __uint_as_float(sy.x)};
                a[c + 4 + 8 * q] = cf{__uint_as_float(sx.y), __uint_as_float(sy.y)};
            }
        cf w1 = tw.w64[0], w2 = tw.w64[1], w3 = tw.w64[2];
        tw_opaque(w1); tw_opaque(w2); tw_opaque(w3);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            r4<INV>(a[c], a[c + 4], a[c + 8], a[c + 12]); // over the quarters g -> b
            a[c + 4] = twmul_rt<INV>(a[c + 4], w1);
            a[c + 8] = twmul_rt<INV>(a[c + 8], w2);
            a[c + 12] = twmul_rt<INV>(a[c + 12], w3);
        }
        // S4: element (k1 = c + 4 t0, b, t1) goes to lane k1 + 16 b, register t1 (below)
    }
    // S4 / S5: register k of lane t is written to wbase + wdst(k); lane t then reads element j 65 + t into register j
    const int wbase = (P == 32) ? (t & 31) * 65 + 16 * (t >> 5) : (P == 16) ? (t & 15) * 65 + 4 * (t >> 4) : t * 65;
    auto wdst = [](int k) constexpr -> int { return (P == 32) ? (k < 16 ? k : k + 16) : (P == 16) ? (k & 3) + 16 * (k >> 2) : k; };
    if constexpr (!HX) {
        cf* wr = xr + wbase;
#pragma unroll
        for (int k = 0; k < P; k++) wr[wdst(k)] = a[k];
        wave_fence();
#pragma unroll
        for (int j = 0; j < P; j++) a[j] = xr[j * 65 + t];
        wave_fence();
    } else {
        float* xf = reinterpret_cast<float*>(xr);
        float* wr = xf + wbase;
#pragma unroll
        for (int k = 0; k < P; k++) wr[wdst(k)] = a[k].x;
        wave_fence();
#pragma unroll
        for (int j = 0; j < P; j++) a[j].x = xf[j * 65 + t];
        wave_fence();
#pragma unroll
        for (int k = 0; k < P; k++) wr[wdst(k)] = a[k].y;
        wave_fence();
#pragma unroll
        for (int j = 0; j < P; j++) a[j].y = xf[j * 65 + t];
        wave_fence();
    }
    rP<P, INV>(a);
}

template <int N, int XF, bool HX = false> __device__ __forceinline__ void wxform(cf (&a)[N / 64], cf* xr, int t, const TwWave<N / 64>& tw)
{
    if constexpr (XF == XF_FWD) wave_fft<N, false, HX>(a, xr, t, tw);
    if constexpr (XF == XF_INV) wave_fft<N, true, HX>(a, xr, t, tw);
}

// a[l] = row[t + 64 l]; `band`: the columns beyond the 2/3 band limit of a SQUARE grid count as zero and are not
// fetched (compile-time classes per 64-column block, as load_rows of fft_lds.hip)
template <int N> __device__ __forceinline__ void wload_row(cf (&a)[N / 64], const cf* __restrict__ row, const int t, const bool band)
{
    constexpr int P = N / 64, LB = N / 3;
    const char* __restrict__ sb = reinterpret_cast<const char*>(row);
    const unsigned b0 = (unsigned)t * 8u;
    auto at = [&](unsigned off, int imm) -> cf { return row_load<N>(reinterpret_cast<const cf*>(sb + off + imm)); }; // (non-temporal from FDES_NT_MIN points on: fft_dev.inc)
#pragma unroll
    for (int l = 0; l < P; l++) {
        const int lo = 64 * l, hi = 64 * l + 63;
        const int cls = (hi <= LB || lo >= N - LB) ? 0 : ((lo > LB && hi < N - LB) ? 1 : 2);
        const unsigned bj = b0 + (unsigned)((l / 8) * 8 * 512);
        const int imm = (l % 8) * 512;
        if (cls == 0) a[l] = at(bj, imm);
        else if (cls == 1) {
            a[l] = cf{0.f, 0.f};
            if (!band) a[l] = at(bj, imm);
        } else {
            const int c = t + 64 * l;
            const bool dd = band && c > LB && c < N - LB;
            const cf v = at(dd ? b0 : bj + (unsigned)imm, 0);
            a[l] = dd ? cf{0.f, 0.f} : v;
        }
    }
}

// The same request with the accumulation registers (AGPRs) as the landing zone: a wave alone on its SIMD owns 256 of
// them beside its 256 vector registers, global loads may target them directly, and nothing else of a pass can live
// there.  The look-ahead operands of the software pipeline (PIPE) land there while the current row group is being
// transformed.  The loads are inline assembly (the compiler would land them in vector registers and copy them over
// behind an s_waitcnt vmcnt(0), i.e. wait for the look-ahead at once), so the wait is explicit too: wtake_row() is
// only called behind wave_wait_loads().  `band`, classes: as wload_row.
#ifndef FDES_W_ACC_PREFETCH
#define FDES_W_ACC_PREFETCH 1
#endif
__device__ __forceinline__ void acc_load64(cf& dst, unsigned byte_off, const void* base)
{
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=a"(dst) : "v"(byte_off), "s"(base) : "memory");
}
__device__ __forceinline__ void acc_load32(float& dst, unsigned byte_off, const void* base)
{
    asm volatile("global_load_dword %0, %1, %2" : "=a"(dst) : "v"(byte_off), "s"(base) : "memory");
}
__device__ __forceinline__ void wave_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N> __device__ __forceinline__ void wload_row_acc(cf (&an)[N / 64], const cf* __restrict__ row, const int t, const bool band)
{
    constexpr int P = N / 64, LB = N / 3;
    const unsigned b0 = (unsigned)t * 8u;
#pragma unroll
    for (int l = 0; l < P; l++) {
        const int lo = 64 * l, hi = 64 * l + 63;
        const int cls = (hi <= LB || lo >= N - LB) ? 0 : ((lo > LB && hi < N - LB) ? 1 : 2);
        const unsigned bj = b0 + (unsigned)(l * 512);
        if (cls == 0) acc_load64(an[l], bj, row);
        else if (cls == 1) {
            if (!band) acc_load64(an[l], bj, row);
            else an[l] = cf{0.f, 0.f};
        } else {
            const int c = t + 64 * l;
            const bool dd = band && c > LB && c < N - LB;
            acc_load64(an[l], dd ? b0 : bj, row);
        }
    }
}
// a <- the landed look-ahead.  Explicit v_accvgpr_read_b32 (volatile, like the loads: program order among them is kept),
// so that the landing registers are read BEFORE the next request overwrites them - left to the compiler the copies sink
// below the new loads and a second register set plus 2 P moves per iteration appear.  Only called for operands whose
// loads are known to have landed (the wait of wave_wait_loads() / wave_wait_loads_but()).
__device__ __forceinline__ float acc_read(float x)
{
    float r;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(x));
    return r;
}
template <int N> __device__ __forceinline__ void wtake_row(cf (&a)[N / 64], cf (&an)[N / 64], const int t, const bool band)
{
    constexpr int P = N / 64, LB = N / 3;
#pragma unroll
    for (int l = 0; l < P; l++) {
        const int lo = 64 * l, hi = 64 * l + 63;
        const int cls = (hi <= LB || lo >= N - LB) ? 0 : ((lo > LB && hi < N - LB) ? 1 : 2);
        const cf v = cf{acc_read(an[l].x), acc_read(an[l].y)};
        if (cls == 2) {
            const int c = t + 64 * l;
            const bool dd = band && c > LB && c < N - LB;
            a[l] = dd ? cf{0.f, 0.f} : v;
        } else {
            a[l] = v;
        }
    }
}
// waits until at most the n youngest vector-memory operations of this wave are outstanding (n a compile-time count of
// the stores issued after the look-ahead loads: those need not have drained)
template <int NKEEP> __device__ __forceinline__ void wave_wait_loads_but()
{
    constexpr int n = NKEEP > 63 ? 63 : NKEEP;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory");
}

template <int P, class F, class M> __device__ __forceinline__ void wexpiv_all(cf (&a)[P], F f, M mag)
{
    float big = 0.f;
#pragma unroll
    for (int l = 0; l < P; l++) big = fmaxf(big, mag(a[l]));
    if (__builtin_expect(big <= kSincosFast, 1)) {
#pragma unroll
        for (int l = 0; l < P; l++) a[l] = f(a[l], std::false_type{});
    } else {
#pragma unroll
        for (int l = 0; l < P; l++) a[l] = f(a[l], std::true_type{});
    }
}
// t = exp(-imPot v) (cos v, sin v), v = a.x (potential2Transmission, src/multisliceSimulation.cu:41-52)
// Sine / cosine through a 64-entry table of (cos, sin)(2 pi k / 64) in LDS (FDES_W_SINCOS_TAB; an experiment): x = k h + r,
// |r| <= pi / 64, so cos r and sin r are two- and one-term polynomials (truncation 2e-11 / 2.4e-9) and the result is one
// complex product with the table entry: about 15 vector instructions and one LDS read per value against 26 for
// sincos_cw.  h is split into a 12-bit head (k h1 is exact for |k| < 4096, i.e. |x| < 400: the fast range) and a tail.
#ifndef FDES_W_SINCOS_TAB
#define FDES_W_SINCOS_TAB 0
#endif
__device__ __forceinline__ cf sincos_tab(float x, const cf* __restrict__ tab)
{
    const float kf = rintf(x * 10.1859163578813f); // 64 / (2 pi)
    const int k = (int)kf & 63;
    float r = fmaf(-kf, 0.09814453125f, x);          // 2 pi / 64 = 0.09817477042...: head with 12 significant bits
    r = fmaf(-kf, 3.02391747e-05f, r);               // tail
    const float r2 = r * r;
    const float sr = r * fmaf(r2, -1.66666672e-1f, 1.0f);
    const float cr = fmaf(r2, fmaf(r2, 4.16666679e-2f, -0.5f), 1.0f);
    const cf t = tab[k];
    return cf{t.x * cr - t.y * sr, t.y * cr + t.x * sr};
}
template <int P> __device__ __forceinline__ void wtransmission_tab(cf (&a)[P], const float impot, const cf* __restrict__ tab)
{
    float big = 0.f;
#pragma unroll
    for (int l = 0; l < P; l++) big = fmaxf(big, fabsf(a[l].x));
    if (__builtin_expect(big <= 390.f, 1)) {
#pragma unroll
        for (int l = 0; l < P; l++) {
            const float v = a[l].x;
            cf t = sincos_tab(v, tab);
            if (impot != 0.f) t = t * __expf(-(v * impot));
            a[l] = t;
        }
    } else {
#pragma unroll
        for (int l = 0; l < P; l++) {
            const float v = a[l].x;
            float sn, cs;
            sincos_wide(v, sn, cs);
            const float e = (impot != 0.f) ? __expf(-(v * impot)) : 1.f;
            a[l] = cf{e * cs, e * sn};
        }
    }
}
template <int P> __device__ __forceinline__ void wtransmission(cf (&a)[P], const float impot)
{
    if (impot == 0.f) {
        wexpiv_all(a, [&](cf w, auto wide) {
            float sn, cs;
            sincos_sel<decltype(wide)::value>(w.x, sn, cs);
            return cf{cs, sn};
        }, [](cf w) { return fabsf(w.x); });
    } else {
        wexpiv_all(a, [&](cf w, auto wide) {
            const float v = w.x;
            float sn, cs;
            const float e = __expf(-(v * impot));
            sincos_sel<decltype(wide)::value>(v, sn, cs);
            return cf{e * cs, e * sn};
        }, [](cf w) { return fabsf(w.x); });
    }
}

// 64-column blocks that lie wholly beyond the band limit N / 3 of a square grid
template <int N> __host__ __device__ constexpr int dead_blocks()
{
    int n = 0;
    for (int it = 0; it < N / 64; it++) n += ((64 * it > N / 3) && (64 * it + 63 < N - N / 3)) ? 1 : 0;
    return n;
}

// largest live |column| of frequency row i2 under the radial 2/3 limit (zeroHighFreq, src/multisliceSimulation.cu:225-250;
// integer form as MID_MASK of fft_lds.hip); -1: nothing live
__device__ __forceinline__ int live_cols(int i2, int md2)
{
    const int q = md2 - 9 * i2 * i2;
    int Lr = (int)(sqrtf((float)(q > 0 ? q : 0)) * (1.0f / 3.0f));
    Lr += (9 * (Lr + 1) * (Lr + 1) <= q) ? 1 : 0;
    Lr -= (9 * Lr * Lr > q) ? 1 : 0;
    return Lr;
}

// FDES_W_WALK = 1 (experiment, round 4): the kernels of pass_threads = 65 become plain WALKERS - half as many workgroups as
// the chip has slots, each walking its row groups WITHOUT look-ahead at two waves per SIMD - so that the kernels of two
// lanes share every CU for their whole life (one workgroup of each) instead of following each other generation by generation.
#ifndef FDES_W_WALK
#define FDES_W_WALK 0
#endif
#ifndef FDES_W_P5_PREFETCH
#define FDES_W_P5_PREFETCH 1 // the second operand of the product is requested together with the first (no extra registers: while one operand is in its butterflies the other one only sits)
#endif
#ifndef FDES_W_PTAB_EARLY
#define FDES_W_PTAB_EARLY 1 // the propagator's column factors are requested together with the row
#endif

// The body of a pass.  PIPE = false: one row group per workgroup, two workgroups per CU at 2048 points (256 VGPRs).
// PIPE = true: a workgroup is alone on its CU (one wave per SIMD, 512 VGPRs) and walks the row groups
// blockIdx.x, blockIdx.x + gridDim.x, ...; the operands of the NEXT group are requested into a second register set
// before the butterflies of the current one start, so that the CU's loads are in flight while its vector units work and
// its stores drain while the next group is transformed (software pipeline with the register file as the landing zone).
// Half-size LDS regions (wave_fft HX; the staged transposed store also goes re / im): 4 N bytes of LDS per row, so that
// four workgroups = sixteen rows fit on a CU where the registers allow it (<= 128: the one-operand passes at 2048 points).
// FDES_W_HALFX is a mask of the passes built that way: 1 = band limit (P4), 2 = propagator (P6) at 2048 points (measured
// 2-6 % slower there: two workgroups per CU fit anyway); 4 = P4, 8 = P6 at 4096 points, where a full-size
// region (33 KiB per row) admits ONE workgroup per CU, whose load, transform and store phases then run strictly one after
// the other, and half-size regions two (66.7 KiB each, <= 256 registers): P4 65.4 -> 53.2 us, P6 67.5 -> 54.4 us (round 4,
// tools/bench_hx4096.py) - the default for these two passes at 4096 points, whatever kernels the other passes use.
#ifndef FDES_W_HALFX
#define FDES_W_HALFX 76 // 4 + 8: P4, P6 at 4096 points (round 4); 64: P1' at 4096 points (round 5: C5 +1.5 %; 16, the filter pass P2: -2 % again)
#endif
template <int N, int MID, bool PIPE> constexpr bool whalfx()
{
    return ((N == 2048 && !PIPE && (((FDES_W_HALFX & 1) && MID == MID_MASK) || ((FDES_W_HALFX & 2) && MID == MID_PTAB))) ||
            (N == 4096 && !PIPE && (((FDES_W_HALFX & 4) && MID == MID_MASK) || ((FDES_W_HALFX & 8) && MID == MID_PTAB) || ((FDES_W_HALFX & 16) && MID == MID_GTAB) ||
                                     ((FDES_W_HALFX & 32) && MID == MID_EXPIV_PAIR) || ((FDES_W_HALFX & 64) && MID == MID_ATOMS))));
}
template <int N, int MID, bool PIPE> constexpr size_t wlds_bytes()
{
    return whalfx<N, MID, PIPE>() ? sizeof(float) * (size_t)WaveGeo<N>::ROWP * WaveGeo<N>::R + 64 : WaveGeo<N>::LDS_BYTES;
}

template <int N, int PRE, int MID, int POST, bool STORE_T, bool PIPE, int RR = 4>
__device__ __forceinline__ void wpass_body(const PassArgs& A, cf* __restrict__ lds)
{
    using G_ = WaveGeo<N, RR>;
    constexpr int P = G_::P, R = G_::R, THR = G_::THR;
    constexpr bool HX = RR == 4 && whalfx<N, MID, PIPE>();
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), t = tid & 63;
    cf* __restrict__ xr = HX ? reinterpret_cast<cf*>(reinterpret_cast<float*>(lds) + w * G_::ROWP) : lds + w * G_::ROWP;
    const int nvirt = A.nvirt > 0 ? A.nvirt : (int)gridDim.x;
    int vb = (int)blockIdx.x + (PIPE ? 0 : A.vb0);
    if (vb >= nvirt) return;
    const int vstride = PIPE ? (int)gridDim.x : nvirt; // gridDim.x is a multiple of 8 when a workgroup walks (vb % 8 = blockIdx.x % 8)
    // staggered start: wave w of the s-th workgroup of a CU (taken as vb / ncu: workgroups are dealt breadth first; for
    // speed only) starts (w + R s) * stagger * 64 cycles late, so that the rows of a CU are in different phases
    if (A.stagger < 0 && A.ncu > 0 && vb < A.ncu * (int)(N <= 1024 ? 4 : (whalfx<N, MID, PIPE>() ? (N <= 2048 ? 4 : 2) : ((N <= 2048 && !PIPE) ? 2 : 1)))) cu_class_delay(-A.stagger); // fft_dev.inc: the CUs of the chip in different phases
    if (A.stagger > 0) {
        const int slot = A.ncu > 0 ? vb / A.ncu : 0;
        const int n = (w + R * (slot & 1)) * A.stagger;
        for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(1);
    }
    TwWave<P> tw;
    if constexpr (PRE != XF_NONE || POST != XF_NONE) tw_load<N>(tw, reinterpret_cast<const cf*>(A.tw0), t);
    cf* const sctab = lds + (size_t)G_::ROWP * R + 8; // behind the row regions
    if constexpr (MID == MID_EXPIV_PAIR && FDES_W_SINCOS_TAB) {
        if (tid < 64) {
            float sn, cs;
            sincos_cw((float)tid * 0.0981747704246810f, sn, cs);
            sctab[tid] = cf{cs, sn};
        }
        __syncthreads();
    }
    // first row of the row group that virtual workgroup v owns: XCD-aware remap (fft_lds.hip: workgroups of one XCD own
    // consecutive row groups, so that the 32-byte segments of their transposed stores meet in that XCD's L2), then the
    // live groups only when the rows are frequencies
    auto row0_of = [&](int v) -> int {
        const int nwg = nvirt, q = nwg >> 3, rem = nwg & 7, xcd = v & 7, k = v >> 3;
        int bg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
        if (A.live_rows_only) {
            const int L = A.band_L;
            const int g_lo = L / R + 1, g_hi = (A.nrows - L) / R;
            if (bg >= g_lo) bg = g_hi + (bg - g_lo);
        }
        return bg * R;
    };
    const unsigned pin = A.pitch_in ? (unsigned)A.pitch_in : (unsigned)N;
    const unsigned ldt0 = A.pitch_out ? (unsigned)A.pitch_out : (unsigned)A.nrows;
    // batch of grids in one launch (grid.z, PassArgs::nbatch; not with PIPE): this workgroup's grid
    const int bz = (int)blockIdx.z;
    const size_t zoff_in = (A.nbatch > 1) ? (size_t)(A.use_zin ? A.zin[bz] : bz) * A.bstride_in0 : (size_t)0;
    const size_t zoff_out = (A.nbatch > 1) ? (size_t)bz * A.bstride_out : (size_t)0;
    const size_t zoff_in1 = (A.nbatch > 1) ? (size_t)bz * A.bstride_in1 : (size_t)0;
    cf* const out0 = reinterpret_cast<cf*>(A.out) + zoff_out + ((MID == MID_ATOMS) ? (size_t)blockIdx.y * A.species_stride : (size_t)0);

    // operands requested ahead: the row(s) of in0 / in1 and the filter values of this wave's row in the (next) group
    constexpr bool LOADS_ROW = (MID != MID_ATOMS && MID != MID_GTABN);
    // (4096-point rows: two operands of 128 registers each and a third one on its way are what fits beside the temporaries;
    //  the second operand of the current group is then requested at the top of the iteration)
    constexpr bool PRE_B = (MID == MID_MULPSI) && (FDES_W_P5_PREFETCH || PIPE) && !(PIPE && N > 2048);
    constexpr bool PV_HELD = (MID == MID_PTAB) && (FDES_W_PTAB_EARLY || PIPE) && !(PIPE && N > 2048) && !HX;
    constexpr bool WALK = PIPE && FDES_W_WALK;
    constexpr bool ACC = PIPE && FDES_W_ACC_PREFETCH && !WALK; // look-ahead operands land in the accumulation registers
    cf an[LOADS_ROW ? P : 1];
    cf bn[PRE_B ? P : 1];
    constexpr bool G_LATE = (MID == MID_GTAB) && HX && N > 2048; // the filter values are read at the point of use: 64 registers fewer across the first transform
    float gn[(MID == MID_GTAB && !G_LATE) ? P : 1];
    auto request = [&](int row0_) {
        const size_t rb = (size_t)(row0_ + w) * pin; // wave-uniform
        if constexpr (ACC) {
            if constexpr (LOADS_ROW) wload_row_acc<N>(an, reinterpret_cast<const cf*>(A.in0) + rb, t, (A.skip_dead_loads & 1) != 0);
            if constexpr (PRE_B) wload_row_acc<N>(bn, reinterpret_cast<const cf*>(A.in1) + rb, t, (A.skip_dead_loads & 2) != 0);
            if constexpr (MID == MID_GTAB && !G_LATE) {
#pragma unroll
                for (int l = 0; l < P; l++) acc_load32(gn[l], (unsigned)(t + 64 * l) * 4u, A.gtab + rb);
            }
        } else {
            if constexpr (LOADS_ROW) wload_row<N>(an, reinterpret_cast<const cf*>(A.in0) + rb + zoff_in, t, (A.skip_dead_loads & 1) != 0);
            if constexpr (PRE_B) wload_row<N>(bn, reinterpret_cast<const cf*>(A.in1) + rb + zoff_in1, t, (A.skip_dead_loads & 2) != 0);
            if constexpr (MID == MID_GTAB && !G_LATE) {
#pragma unroll
                for (int l = 0; l < P; l++) gn[l] = A.gtab[rb + t + 64 * l];
            }
        }
    };
    // (every path through an iteration must define the look-ahead registers: a request skipped under a condition would
    //  keep their OLD values alive across the whole iteration - 64 to 256 registers)
    auto no_request = [&]() {
        if constexpr (LOADS_ROW) {
#pragma unroll
            for (int l = 0; l < P; l++) an[l] = cf{0.f, 0.f};
        }
        if constexpr (PRE_B) {
#pragma unroll
            for (int l = 0; l < P; l++) bn[l] = cf{0.f, 0.f};
        }
        if constexpr (MID == MID_GTAB && !G_LATE) {
#pragma unroll
            for (int l = 0; l < P; l++) gn[l] = 0.f;
        }
    };
    if constexpr (!WALK) request(row0_of(vb));
    // column factors of the propagator: the same for every row
    cf pv[(MID == MID_PTAB) ? P : 1];
    if constexpr (PV_HELD) {
        const cf* __restrict__ pcol = reinterpret_cast<const cf*>(A.pcol);
#pragma unroll
        for (int l = 0; l < P; l++) pv[l] = pcol[t + 64 * l];
    }

    if constexpr (ACC) wave_wait_loads();
    do { // PIPE: over this workgroup's row groups; otherwise the body runs once
    unsigned ldt = ldt0;
    if constexpr (PIPE) asm volatile("" : "+s"(ldt)); // per iteration: otherwise the 32-64 scalar row pointers of the transposed store are hoisted out of the loop and spilled
    const int row0 = row0_of(vb);
    if constexpr (WALK) request(row0);
    const int grow = row0 + w;
    const size_t rbase = (size_t)grow * pin; // wave-uniform
    const cf* __restrict__ in0 = A.in0 ? reinterpret_cast<const cf*>(A.in0) + rbase + zoff_in : nullptr;
    const cf* __restrict__ in1 = A.in1 ? reinterpret_cast<const cf*>(A.in1) + rbase + zoff_in1 : nullptr;
    const float* __restrict__ gtab = A.gtab ? A.gtab + rbase : nullptr;

    cf a[P];
    cf b[(MID == MID_MULPSI) ? P : 1]; // second operand of the product
    float gvv[(MID == MID_GTAB && !G_LATE) ? P : 1];
    float vim[(MID == MID_EXPIV_PAIR) ? P : 1];
    if constexpr (ACC) {
        // the look-ahead of this group has landed: waited for before the loop resp. at the end of the previous iteration,
        // i.e. before any copy the compiler may place on the loop's back edge
        if constexpr (LOADS_ROW) wtake_row<N>(a, an, t, (A.skip_dead_loads & 1) != 0);
        if constexpr (PRE_B) wtake_row<N>(b, bn, t, (A.skip_dead_loads & 2) != 0);
        if constexpr (MID == MID_GTAB && !G_LATE) {
#pragma unroll
            for (int l = 0; l < P; l++) gvv[l] = acc_read(gn[l]);
        }
    } else {
    if constexpr (LOADS_ROW) {
#pragma unroll
        for (int l = 0; l < P; l++) a[l] = an[l];
    }
    if constexpr (PRE_B) {
#pragma unroll
        for (int l = 0; l < P; l++) b[l] = bn[l];
    }
    if constexpr (MID == MID_GTAB && !G_LATE) {
#pragma unroll
        for (int l = 0; l < P; l++) gvv[l] = gn[l];
    }
    }
    // the next group's operands: requested now, consumed one iteration later (4096-point product pass: three operands
    // of 128 registers do not fit beside the temporaries, so the request waits until the product has freed one)
    constexpr bool LATE_REQ = PIPE && !WALK && N > 2048 && MID == MID_MULPSI;
    if constexpr (PIPE && !LATE_REQ && !WALK) {
        __builtin_amdgcn_sched_barrier(0); // the landed operands are taken before their registers are requested again
        if (vb + vstride < nvirt) request(row0_of(vb + vstride));
        else no_request();
        __builtin_amdgcn_sched_barrier(0); // the scheduler must not sink these loads towards their use
    }
    if constexpr (MID == MID_GTABN) {
        cf acc[P];
#pragma unroll
        for (int l = 0; l < P; l++) acc[l] = cf{0.f, 0.f};
        for (int z = 0; z < A.nspecies; z++) {
            const size_t zo = (size_t)z * A.species_stride;
#pragma unroll
            for (int l = 0; l < P; l++) a[l] = in0[zo + t + 64 * l];
            float gv[P];
#pragma unroll
            for (int l = 0; l < P; l++) gv[l] = gtab[zo + t + 64 * l];
            wxform<N, PRE, HX>(a, xr, t, tw);
#pragma unroll
            for (int l = 0; l < P; l++) {
                acc[l].x += a[l].x * gv[l];
                acc[l].y += a[l].y * gv[l];
            }
        }
#pragma unroll
        for (int l = 0; l < P; l++) a[l] = acc[l];
    } else {
        if constexpr (MID == MID_ATOMS) {
            // squareAtoms_d (src/crystalMaker.cu:73-123) from the (slice, species, row)-sorted records, as MID_ATOMS of
            // fft_lds.hip: the rows are zeroed in LDS (each wave its own region, natural order), ONE wave adds the bilinear
            // weights of all four rows with LDS float atomics in sorted order, every wave then picks up its row
            const AtomRec* __restrict__ recs = reinterpret_cast<const AtomRec*>(A.recs) + ((A.nbatch > 1) ? (size_t)bz * A.bstride_recs : (size_t)0);
            const int rlo = row0 > 0 ? row0 - 1 : 0;
            const int rhi = (row0 + R + 1 < A.nrows) ? row0 + R + 1 : A.nrows;
            int plo[2] = {0, 0}, phi[2] = {0, 0};
#pragma unroll
            for (int comp = 0; comp < 2; comp++) {
                const int qb = (A.nbatch > 1) ? (comp ? A.zq1[bz] : A.zq0[bz]) : (comp ? A.q1 : A.q0);
                const int q = qb < 0 ? -1 : qb + (int)blockIdx.y;
                if (q >= 0) {
                    const int* __restrict__ rs = A.rowstart + ((A.nbatch > 1) ? (size_t)bz * A.bstride_rowstart : (size_t)0) + (size_t)q * (size_t)(A.nrows + 1);
                    plo[comp] = rs[rlo];
                    phi[comp] = rs[rhi];
                }
            }
            if (phi[0] - plo[0] + phi[1] - plo[1] == 0) { // workgroup-uniform
                if constexpr (STORE_T) {
#pragma unroll
                    for (int it = 0; it < P; it++) {
                        const int e = it * THR + tid;
                        (out0 + row0)[(unsigned)(e / R) * ldt + (unsigned)(e & (R - 1))] = cf{0.f, 0.f};
                    }
                }
                continue;
            }
            if constexpr (HX) {
                // half-size regions (4096 points, two workgroups per CU): the tile holds ONE component at a time - slice q0's
                // deposit (real part) first, then slice q1's (imaginary part); same records, same order, same sums
                float* const xf = reinterpret_cast<float*>(xr);
                float* const ldsf = reinterpret_cast<float*>(lds);
#pragma unroll 1
                for (int comp = 0; comp < 2; comp++) {
#pragma unroll
                    for (int l = 0; l < P; l++) xf[t + 64 * l] = 0.f;
                    __syncthreads();
                    if (tid < 64) {
#pragma unroll 1
                        for (int base = plo[comp]; base < phi[comp]; base += 64) {
                            const int i = base + tid;
                            if (i < phi[comp]) {
                                const AtomRec ar = recs[i];
                                const float a1 = fabsf(ar.r1), a2 = fabsf(ar.r2);
                                const int s1 = ar.r1 < 0.f ? -1 : 1, s2 = ar.r2 < 0.f ? -1 : 1;
#pragma unroll
                                for (int px = 0; px < 4; px++) {
                                    const int c = ar.i1 + ((px == 2 || px == 3) ? s1 : 0);
                                    const int rr = ar.i2 + ((px == 1 || px == 2) ? s2 : 0) - row0;
                                    const float wgt = ((px == 2 || px == 3) ? a1 : (1 - a1)) * ((px == 1 || px == 2) ? a2 : (1 - a2)) * ar.occ;
                                    if (rr >= 0 && rr < R && c >= 0 && c < N) atomicAdd(&ldsf[rr * G_::ROWP + c], wgt);
                                }
                            }
                        }
                    }
                    __syncthreads();
                    if (comp == 0) {
#pragma unroll
                        for (int l = 0; l < P; l++) a[l].x = xf[t + 64 * l];
                    } else {
#pragma unroll
                        for (int l = 0; l < P; l++) a[l].y = xf[t + 64 * l];
                    }
                }
                wave_fence();
            } else {
#pragma unroll
            for (int l = 0; l < P; l++) xr[t + 64 * l] = cf{0.f, 0.f};
            __syncthreads();
            if (tid < 64) {
                float* ldsf = reinterpret_cast<float*>(lds);
#pragma unroll 1
                for (int comp = 0; comp < 2; comp++) {
#pragma unroll 1
                    for (int base = plo[comp]; base < phi[comp]; base += 64) {
                        const int i = base + tid;
                        if (i < phi[comp]) {
                            const AtomRec ar = recs[i];
                            const float a1 = fabsf(ar.r1), a2 = fabsf(ar.r2);
                            const int s1 = ar.r1 < 0.f ? -1 : 1, s2 = ar.r2 < 0.f ? -1 : 1;
#pragma unroll
                            for (int px = 0; px < 4; px++) {
                                // pixel order of the reference: (i1,i2), (i1,i2+s2), (i1+s1,i2+s2), (i1+s1,i2)
                                const int c = ar.i1 + ((px == 2 || px == 3) ? s1 : 0);
                                const int rr = ar.i2 + ((px == 1 || px == 2) ? s2 : 0) - row0;
                                const float wgt = ((px == 2 || px == 3) ? a1 : (1 - a1)) * ((px == 1 || px == 2) ? a2 : (1 - a2)) * ar.occ;
                                if (rr >= 0 && rr < R && c >= 0 && c < N) atomicAdd(&ldsf[2 * (rr * G_::ROWP + c) + comp], wgt);
                            }
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int l = 0; l < P; l++) a[l] = xr[t + 64 * l];
            wave_fence();
            }
        }
        if constexpr (MID == MID_MULPSI && !PRE_B && PIPE) wload_row<N>(b, in1, t, (A.skip_dead_loads & 2) != 0);
        if constexpr (MID == MID_PTAB && !PV_HELD && !PIPE && !(HX && N > 2048)) {
            const cf* __restrict__ pcol = reinterpret_cast<const cf*>(A.pcol);
#pragma unroll
            for (int l = 0; l < P; l++) pv[l] = pcol[t + 64 * l];
        }
        wxform<N, PRE, HX>(a, xr, t, tw);
        if constexpr (MID == MID_EXPIV_PAIR) {
            // two slices share one potential grid (V_s = Re, V_(s+1) = Im): first slice now, the second slice's potential
            // waits as one float per pixel
#pragma unroll
            for (int l = 0; l < P; l++) vim[l] = a[l].y;
            if constexpr (FDES_W_SINCOS_TAB) wtransmission_tab(a, A.scale, sctab);
            else wtransmission(a, A.scale);
        } else if constexpr (MID == MID_MASK) {
            const int Lr = live_cols(iwc(grow, A.nrows), A.mindim * A.mindim);
            const int tlo = Lr, thi = N - Lr;
#pragma unroll
            for (int l = 0; l < P; l++) {
                const bool live = (l < P / 2) ? (t <= tlo - 64 * l) : (t >= thi - 64 * l);
                const float f = live ? A.scale : 0.f;
                a[l] = cf{a[l].x * f, a[l].y * f};
            }
        } else if constexpr (MID == MID_SCALE) {
#pragma unroll
            for (int l = 0; l < P; l++) a[l] = cf{a[l].x * A.scale, a[l].y * A.scale};
        } else if constexpr (MID == MID_GTAB && G_LATE) {
#pragma unroll
            for (int l = 0; l < P; l++) { const float gv = gtab[t + 64 * l]; a[l] = cf{a[l].x * gv, a[l].y * gv}; }
        } else if constexpr (MID == MID_GTAB) {
#pragma unroll
            for (int l = 0; l < P; l++) a[l] = cf{a[l].x * gvv[l], a[l].y * gvv[l]};
        } else if constexpr (MID == MID_PTAB) {
            // psi-hat * P, P(k_row, k_col) = prow[row] pcol[col] inside the radial band limit (separable Fresnel propagator,
            // src/multisliceSimulation.cu:225-274, 594-603)
            const cf pr = reinterpret_cast<const cf*>(A.prow)[grow];
            const int Lr = live_cols(iwc(grow, A.nrows), A.mindim * A.mindim);
            const int tlo = Lr, thi = N - Lr;
#pragma unroll
            for (int l = 0; l < P; l++) {
                const bool live = (l < P / 2) ? (t <= tlo - 64 * l) : (t >= thi - 64 * l);
                const cf wv = cmul3(pr, (!PV_HELD && (PIPE || (HX && N > 2048))) ? reinterpret_cast<const cf*>(A.pcol)[t + 64 * l] : pv[l]); // (table value at the point of use: it sits in the caches)
                const cf v = cmul3(a[l], wv);
                a[l] = live ? v : cf{0.f, 0.f};
            }
        } else if constexpr (MID == MID_MULPSI) {
            if constexpr (!PRE_B && !PIPE) wload_row<N>(b, in1, t, (A.skip_dead_loads & 2) != 0);
            wxform<N, PRE, HX>(b, xr, t, tw);
#pragma unroll
            for (int l = 0; l < P; l++) a[l] = cmul3(a[l], b[l]); // f0 = t, f1 = psi
            if constexpr (LATE_REQ) {
                __builtin_amdgcn_sched_barrier(0);
                if (vb + vstride < nvirt) request(row0_of(vb + vstride));
                else no_request();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    auto store_row = [&](cf (&v)[P], cf* outp) {
        if constexpr (!STORE_T) {
            const unsigned pout = A.pitch_out ? (unsigned)A.pitch_out : (unsigned)N;
            cf* __restrict__ on = outp + (size_t)grow * pout;
#pragma unroll
            for (int l = 0; l < P; l++) on[t + 64 * l] = v[l];
            if constexpr (ACC) wave_wait_loads_but<P>(); // the next group's operands have landed; the P stores may still drain
        } else {
            // every wave leaves its row in natural order in its own region; after the barrier all threads write R
            // contiguous elements (one 32-byte segment) per output row
            float re[HX ? P : 1];
            float* const ldsf = reinterpret_cast<float*>(lds);
            const int hx_rd = (tid & (R - 1)) * G_::ROWP + tid / R;
            if constexpr (HX) { // real parts first: staged, picked up by the threads that will store them; then the imaginary parts
                float* xf = reinterpret_cast<float*>(xr);
#pragma unroll
                for (int l = 0; l < P; l++) xf[t + 64 * l] = v[l].x;
                __syncthreads();
#pragma unroll
                for (int it = 0; it < P; it++) re[it] = ldsf[hx_rd + it * 64];
                __syncthreads();
#pragma unroll
                for (int l = 0; l < P; l++) xf[t + 64 * l] = v[l].y;
            } else {
#pragma unroll
                for (int l = 0; l < P; l++) xr[t + 64 * l] = v[l];
            }
            __syncthreads();
            // output element (column c, row row0 + rr) at byte offset (c ldt + rr) 8 from dst: one running 32-bit offset per
            // thread (made opaque per step: unrolled, the compiler otherwise forms P scalar row pointers and spills them)
            char* __restrict__ dst = reinterpret_cast<char*>(outp + row0);
            const int rr = tid & (R - 1), c0 = tid / R; // c0 < 64
            unsigned off = ((unsigned)c0 * ldt + (unsigned)rr) * 8u;
            const unsigned step = 64u * 8u * ldt;
            // two straight-line variants under one uniform branch (a test per step would cut the LDS reads and the stores
            // into P basic blocks, each waiting for its own read): with the band limit of a square grid the dead 64-column
            // blocks are known at compile time (the classes of wload_row)
            auto emit = [&](auto skip_dead) {
#pragma unroll
                for (int it = 0; it < P; it++) {
                    const bool dead = decltype(skip_dead)::value && (64 * it > N / 3) && (64 * it + 63 < N - N / 3);
                    if constexpr (HX) {
                        if (!dead) *reinterpret_cast<cf*>(dst + off) = cf{re[it], ldsf[hx_rd + it * 64]};
                    } else {
                        if (!dead) *reinterpret_cast<cf*>(dst + off) = lds[rr * G_::ROWP + c0 + it * 64];
                    }
                    off += step;
                    asm volatile("" : "+v"(off));
                }
                // the next group's operands have landed; this iteration's stores (a compile-time count) may still drain
                if constexpr (ACC) wave_wait_loads_but<decltype(skip_dead)::value ? P - dead_blocks<N>() : P>();
            };
            if (A.skip_dead_stores) emit(std::true_type{});
            else emit(std::false_type{});
        }
    };
    wxform<N, POST, HX>(a, xr, t, tw);
    store_row(a, out0);
    if constexpr (MID == MID_EXPIV_PAIR) {
#pragma unroll
        for (int l = 0; l < P; l++) a[l] = cf{vim[l], 0.f};
        if constexpr (FDES_W_SINCOS_TAB) wtransmission_tab(a, A.scale, sctab);
        else wtransmission(a, A.scale);
        __syncthreads(); // every wave has read the staged tile before the regions are exchange buffers again
        wxform<N, POST, HX>(a, xr, t, tw);
        store_row(a, reinterpret_cast<cf*>(A.out2) + ((A.nbatch > 1) ? (size_t)bz * A.bstride_out2 : (size_t)0));
    }
    if constexpr (PIPE && (STORE_T || MID == MID_ATOMS)) __syncthreads(); // the staged tile has been read: the regions are free for the next group
    } while (PIPE && (vb += vstride) < nvirt);
}

template <int N, int PRE, int MID, int POST, bool STORE_T, bool PIPE, int RR = 4>
__global__ __launch_bounds__((WaveGeo<N, RR>::THR), (N <= 1024 ? (MID == MID_GTABN ? 2 : 4) : (whalfx<N, MID, PIPE>() ? (N <= 2048 ? 4 : 2) : ((N <= 2048 && (!PIPE || FDES_W_WALK)) ? 2 : 1)))) void k_wpass(PassArgs A)
{
    extern __shared__ cf wlds[];
    wpass_body<N, PRE, MID, POST, STORE_T, PIPE, RR>(A, wlds);
}

#undef float2
#undef make_float2

template <int N, int PRE, int MID, int POST, bool ST, bool PIPE, int RR = 4> hipError_t wlaunch(const PassArgs& a, hipStream_t st)
{
    using G_ = WaveGeo<N, RR>;
    static std::atomic<unsigned long long> attr_set{0}; // dynamic-LDS limit: per function and device (fft_lds.hip, launch)
    auto kern = k_wpass<N, PRE, MID, POST, ST, PIPE, RR>;
    constexpr size_t kLds = (RR == 4) ? wlds_bytes<N, MID, PIPE>() : G_::LDS_BYTES;
    int dev = 0;
    {
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
    }
    if (dev < 0 || dev >= 64 || !((attr_set.load(std::memory_order_acquire) >> dev) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    if (a.nrows % G_::R != 0) return hipErrorInvalidValue;
    int groups = a.nrows / G_::R;
    PassArgs w = a;
    if (a.live_rows_only) {
        if (a.band <= 0) return hipErrorInvalidValue;
        const int L = a.band_L;
        const int g_lo = L / G_::R + 1, g_hi = (a.nrows - L) / G_::R;
        if (g_hi > g_lo) groups = g_lo + (a.nrows / G_::R - g_hi);
        else w.live_rows_only = 0; // everything is live
    }
    const int ny = (MID == MID_ATOMS) ? (a.nspecies > 0 ? a.nspecies : 1) : 1;
    if (a.nbatch > 16 || (PIPE && a.nbatch > 1)) return hipErrorInvalidValue;
    const int nz = a.nbatch > 1 ? a.nbatch : 1; // grid.z = batch
    w.nvirt = 0;
    w.vb0 = 0;
    int grid = groups;
    if (PIPE) { // one workgroup per CU walks the row groups (a multiple of 8 workgroups: the XCD of a walker does not change)
        const int ncu = a.ncu > 8 ? (a.ncu & ~7) : 8;
        if (groups > ncu) { grid = ncu; w.nvirt = groups; }
    }
    if (a.ev_start && a.ev_stop) {
        w.ev_start = w.ev_stop = nullptr;
        hipExtLaunchKernelGGL(kern, dim3(grid, ny, nz), dim3(G_::THR), kLds, st, (hipEvent_t)a.ev_start, (hipEvent_t)a.ev_stop, 0, w);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(grid, ny, nz), dim3(G_::THR), kLds, st, w);
    return hipGetLastError();
}

// Passes whose pipelined form does not fit the register file (its look-ahead operands sit in the accumulation registers,
// which the compiler then lacks as spill space: scratch traffic, and a spilled landing register would be copied before
// its load has landed) run as one row group per workgroup instead.
// (4096-point product pass: its look-ahead could only be requested late in the iteration, and the compiler reuses the
// landing registers as spill space in between, which tools/check_acc_landing.py cannot tell from a premature read)
constexpr bool pipe_fits(int n, int mid) { return !(n > 2048 && (mid == MID_GTAB || mid == MID_GTABN || mid == MID_EXPIV_PAIR || mid == MID_PTAB || mid == MID_MULPSI)); }
template <int N, bool PIPE, int RR = 4> hipError_t wdispatch(int pre, int mid, int post, bool st_t, const PassArgs& a, hipStream_t st)
{
#define CASE(P_, M_, Q_, S_) if (pre == P_ && mid == M_ && post == Q_ && st_t == S_) return wlaunch<N, P_, M_, Q_, S_, PIPE && pipe_fits(N, M_), RR>(a, st);
    CASE(XF_NONE, MID_NONE, XF_NONE, false)
    CASE(XF_NONE, MID_NONE, XF_NONE, true)
    CASE(XF_NONE, MID_SCALE, XF_NONE, true)
    CASE(XF_FWD, MID_NONE, XF_NONE, false)
    CASE(XF_INV, MID_NONE, XF_NONE, false)
    CASE(XF_INV, MID_SCALE, XF_NONE, false)
    CASE(XF_FWD, MID_NONE, XF_NONE, true)
    CASE(XF_INV, MID_NONE, XF_NONE, true)
    CASE(XF_FWD, MID_ATOMS, XF_NONE, true)
    CASE(XF_INV, MID_EXPIV_PAIR, XF_FWD, true)
    CASE(XF_FWD, MID_GTAB, XF_INV, true)
    CASE(XF_FWD, MID_GTABN, XF_INV, true)
    CASE(XF_FWD, MID_MASK, XF_INV, true)
    CASE(XF_INV, MID_MULPSI, XF_FWD, true)
    CASE(XF_NONE, MID_MULPSI, XF_FWD, true)
    CASE(XF_FWD, MID_PTAB, XF_INV, true)
#undef CASE
    return hipErrorInvalidValue;
}

} // namespace

bool wave_pass_supported_len(int n) { return n == 1024 || n == 2048 || n == 4096; }
// passes that run on these kernels whatever workgroup geometry was asked for: the ones with two workgroups per CU at
// 4096 points (half-size LDS regions, above)
bool wave_pass_preferred(int n, int pre, int mid, int post, bool st_t)
{
    if (!(n == 4096 && st_t)) return false;
    if (pre == XF_INV && post == XF_FWD) return mid == MID_EXPIV_PAIR && whalfx<4096, MID_EXPIV_PAIR, false>();
    if (pre == XF_FWD && post == XF_NONE) return mid == MID_ATOMS && whalfx<4096, MID_ATOMS, false>();
    if (!(pre == XF_FWD && post == XF_INV)) return false;
    return (mid == MID_MASK && whalfx<4096, MID_MASK, false>()) || (mid == MID_PTAB && whalfx<4096, MID_PTAB, false>()) ||
           (mid == MID_GTAB && whalfx<4096, MID_GTAB, false>());
}

hipError_t wave_pass(int n, int pre, int mid, int post, bool st_t, const PassArgs& a_in, hipStream_t st)
{
    PassArgs a = a_in;
    if (a.band > 0 && a.band_L != n / 3) a.skip_dead_stores = 0; // the kernels' column classes assume the band of a square grid
    const bool pipe = a.wg == 65 && a.nbatch <= 1;
    switch (n) {
    case 1024: return wdispatch<1024, false>(pre, mid, post, st_t, a, st); // (a pass over 1024 rows is one generation of workgroups: nothing to pipeline)
    case 2048: return pipe ? wdispatch<2048, true>(pre, mid, post, st_t, a, st) : (a.wg == 128 ? wdispatch<2048, false, 8>(pre, mid, post, st_t, a, st) : wdispatch<2048, false>(pre, mid, post, st_t, a, st));
    case 4096: return pipe ? wdispatch<4096, true>(pre, mid, post, st_t, a, st) : wdispatch<4096, false>(pre, mid, post, st_t, a, st);
    default: return hipErrorInvalidValue;
    }
}

} // namespace fdes
