// fft_gen.hip — LDS-resident row passes for grids that are not a power of two (gfx950).
//
// cuFFT serves any size alike (src/paramStructure.cu:676-679) and the reference's own examples use 320-, 800- and
// 1000-point grids (bin/dataFDES.cnf; bin/test.qsc: m = 2 nx, src/rwQsc.cu:943-948; ExampleSpecimens/Si_001_11k_cnf).
// The kernels of fft_lds.hip / fft_wave.hip keep a row in registers and are unrolled per power-of-two length; here the
// same pass structure (coalesced row loads -> [row FFT] -> point-wise operation -> [row FFT] -> natural or transposed
// store, fft_lds.h) runs with the rows in LDS and the length as a run-time value, for any N = 2^a 3^b 5^c 7^d 11^e 13^f in
// [256, 4096]: mixed-radix Stockham stages (radices 13, 11, 10, 8, 7, 5, 4, 3, 2) between two LDS images of the row tile, one
// work item per butterfly, twiddles from a table of the N-th roots of unity (double-precision values rounded once;
// every twiddle is ONE table entry), results in natural order after the last stage.  One workgroup = R rows (8 up to
// 512 points, 4 up to 2048, 2 beyond - two images of 2 x 4096 elements are 128 KiB, the twiddle table then stays in
// global memory, i.e. in the caches: 64-, 32- resp. 16-byte segments in the transposed store; round 4: .qsc inputs give
// m = 2 nx, src/rwQsc.cu:943-948, so nx = 1280 ... 2000 means 2560-, 3000-, 3072-, 3200-, 3600-, 4000-point rows).  The slice loop of the engine then runs
// these sizes with the same 4.5 launches per slice as the power-of-two grids instead of rocFFT + point-wise kernels.
// Run-time compilation (round 5, gen_jit.cpp): the same source compiled by hipRTC at plan creation with -DFDES_GEN_JIT_N=<length>
// gives a length without a compiled-in kernel the compile-time form (k_gpass<NC != 0>: twice the rate of the run-time-length
// kernels); under __HIPCC_RTC__ only the device side is compiled and the pass kernels of that ONE length are exported by name.
#include "fft_lds.h"
#include "geometry.h"

#ifndef __HIPCC_RTC__
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "gen_jit.h"
#else
// hipRTC has no <type_traits>: the two tag types fft_dev.inc selects its sine / cosine range with
namespace std {
struct false_type { static constexpr bool value = false; };
struct true_type { static constexpr bool value = true; };
} // namespace std
#endif

namespace fdes {

namespace {

#include "fft_dev.inc"

constexpr int kGenMaxLen = 8192; // longest row: a tile of two rows (one image) is 128 KiB of LDS
constexpr int kRxMax = 23; // largest radix of a stage: registers of one butterfly (13 in the run-time-length kernels; 17, 19, 23: compile-time kernels only)

struct GenFac {
    int n = 0;       // row length
    int rows = 0;    // rows per workgroup (4 or 8)
    int lrows = 0;   // log2(rows)
    int nf = 0;      // stages
    int radix[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // per stage: butterflies per row n / radix, sub-transform length Ns so far, table step n / (Ns radix) of its twiddle,
    // ceil(2^32 / Ns) (j / Ns = umulhi(j, magic) for j < 2^16: the kernels never divide)
    int nbf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ns[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tws[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned magic[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// Round 5: the compile-time lengths beyond 1024 points run THREE stages with composite radices (12 = 3 x 4, 15 = 3 x 5,
// 16 = 4 x 4, 20 = 4 x 5, 25 = 5 x 5 as two levels of small butterflies inside a thread) instead of four or five of radix
// <= 10: a stage is one trip of the whole tile through LDS (the VGPR -> LDS path, about 85 B/clk/CU, is what these kernels
// are bound by), so a third fewer stages is a third less of that traffic.  First radix odd where the length allows it: the
// first stage writes element i of butterfly j to j RX + i, a stride of 2 RX dwords between lanes, conflict-free for odd RX.
#ifndef FDES_GEN_SPECIALISED
#define FDES_GEN_SPECIALISED 1 // compile-time kernels for the grids the reference ships and a few round ones (gen_pass below)
#endif
#ifndef FDES_GEN_CHAIN
#define FDES_GEN_CHAIN 1
#endif
#ifndef FDES_GEN_NT_ABOVE
#define FDES_GEN_NT_ABOVE 2048 // measured: 3000^2 +7 % with non-temporal row loads; 2000^2 and 1280^2: see profiles/r05_nt_loads.txt
#endif
#ifndef FDES_GEN_REV
#define FDES_GEN_REV 1 // trailing transform of a chained pass in reversed stage order, fed from registers
#endif
#ifndef FDES_GEN_PREFETCH_B
#define FDES_GEN_PREFETCH_B 1 // product pass of the chained kernels: both operands requested before the first transform
#endif
__host__ __device__ constexpr bool gen_three_stages(int n, int (&r)[3])
{
    switch (n) {
    // (stage orders measured in round 5, profiles/r05_mixed_radix.txt: the fastest of four to seven per length)
    case 1280: r[0] = 10; r[1] = 16; r[2] = 8; return true; // (5, 16, 16: -7 %)
    case 1600: r[0] = 10; r[1] = 16; r[2] = 10; return true;
    case 2000: r[0] = 5; r[1] = 20; r[2] = 20; return true;
    case 2560: r[0] = 10; r[1] = 16; r[2] = 16; return true;
    case 3000: r[0] = 15; r[1] = 20; r[2] = 10; return true; // (15, 10, 20: -4 %; 20, 15, 10: -3 %)
    case 3072: r[0] = 12; r[1] = 16; r[2] = 16; return true;
    case 3200: r[0] = 10; r[1] = 16; r[2] = 20; return true;
    case 3600: r[0] = 15; r[1] = 16; r[2] = 15; return true;
    case 4000: r[0] = 10; r[1] = 20; r[2] = 20; return true;
    default: return false;
    }
}

// n = a b c with radices the compile-time kernels have butterflies for, for a length outside the table above: the most balanced
// triple (smallest largest radix: fewest registers per butterfly); first radix odd where one exists (conflict-free first-stage
// writes; not 25), else 10, 12, 8, 16, 20, 4, 2 in this order; then a last radix other than 16, then the larger middle radix - the choices
// the measured table makes for 1280 ... 4000 (3200 excepted: 10 x 16 x 20 there, within a few per cent), checked beyond 4096 points
// (profiles/r05_rows_beyond_4096.txt: 5000, 4800, 8000, 3840)
__host__ __device__ constexpr bool gen_auto_three_stages(int n, int (&r)[3])
{
    const int S[17] = {25, 23, 20, 19, 17, 16, 15, 13, 12, 11, 10, 8, 7, 5, 4, 3, 2};
    const int evenpref[7] = {10, 12, 8, 16, 20, 4, 2};
    long best = -1;
    for (int ia = 0; ia < 17; ia++)
        for (int ib = 0; ib < 17; ib++) {
            const int a = S[ia], b = S[ib];
            if (n % (a * b) != 0) continue;
            const int c = n / (a * b);
            bool ok = false;
            for (int ic = 0; ic < 17; ic++) ok = ok || S[ic] == c;
            if (!ok) continue;
            const int mx = a > b ? (a > c ? a : c) : (b > c ? b : c);
            int apref = 0; // larger is better
            if (a % 2 == 1 && a <= 15) apref = 100 + a; // (25 as the first radix: 5000 = 25 x 25 x 8 ran 15 % behind 10 x 25 x 20)
            else
                for (int e = 0; e < 7; e++)
                    if (evenpref[e] == a) apref = 50 - e;
            // (a last radix of 16 keeps the trailing transform of a chained pass from starting in registers: 4800 = 15 x 16 x 20 ran 9 % ahead of 15 x 20 x 16)
            const long score = (long)(64 - mx) * 1000000 + (long)apref * 1000 + (c != 16 ? 500 : 0) + b;
            if (score > best) { best = score; r[0] = a; r[1] = b; r[2] = c; }
        }
    return best >= 0;
}

// the same with FOUR stages for the lengths three cannot serve (8192 = 8 x 16 x 16 x 4, 6144 = 12 x 16 x 8 x 4, 7168 = 7 x 16 x 16 x 4 ...: rows beyond 4096 points,
// run-time compilation only): smallest largest radix, odd first radix where one exists, then the larger second radix
__host__ __device__ constexpr bool gen_auto_four_stages(int n, int (&r)[4])
{
    const int S[17] = {25, 23, 20, 19, 17, 16, 15, 13, 12, 11, 10, 8, 7, 5, 4, 3, 2};
    if (n % 64 == 0) { // measured (profiles/r05_rows_beyond_4096.txt): a x 16 x c x 4 with a medium first radix - 8192 = 8 x 16 x 16 x 4 (+24 % over 2 x 16 x 16 x 16),
                       // 6144 = 12 x 16 x 8 x 4 (+11 % over 8 x 12 x 8 x 8), 7680 = 12 x 16 x 10 x 4 (+22 %), 7168 = 7 x 16 x 16 x 4; a first radix of 16 loses 5 ... 20 %
        const int pref[17] = {12, 10, 8, 15, 7, 5, 13, 11, 3, 20, 25, 17, 19, 23, 16, 4, 2};
        const int rest = n / 64;
        for (int ia = 0; ia < 17; ia++) {
            const int a = pref[ia];
            if (rest % a != 0) continue;
            const int c = rest / a;
            bool ok = false;
            for (int ic = 0; ic < 17; ic++) ok = ok || S[ic] == c;
            if (ok) { r[0] = a; r[1] = 16; r[2] = c; r[3] = 4; return true; }
        }
    }
    long best = -1;
    for (int ia = 0; ia < 17; ia++)
        for (int ib = 0; ib < 17; ib++)
            for (int ic = 0; ic < 17; ic++) {
                const int a = S[ia], b = S[ib], c = S[ic];
                if (n % (a * b * c) != 0) continue;
                const int d = n / (a * b * c);
                bool ok = false;
                for (int id = 0; id < 17; id++) ok = ok || S[id] == d;
                if (!ok) continue;
                int mx = a > b ? a : b;
                mx = mx > c ? mx : c;
                mx = mx > d ? mx : d;
                const long score = (long)(64 - mx) * 1000000 + (long)((a % 2 == 1) ? 100 + a : (a == 16 ? 0 : 50 - a)) * 1000 + b * 30 + c;
                if (score > best) { best = score; r[0] = a; r[1] = b; r[2] = c; r[3] = d; }
            }
    return best >= 0;
}

// factors and stage tables of an n-point row; false if n has a prime factor above 23 or more than 8 stages
__host__ __device__ constexpr bool factorize(int n, GenFac& f)
{
    f.n = n;
    f.nf = 0;
    int m = n;
#if FDES_GEN_CHAIN && FDES_GEN_SPECIALISED
    {   // three stages for the compile-time lengths beyond 1024 points (composite radices 12, 15, 16, 20, 25: round 5)
        int cr[3] = {0, 0, 0};
        if (gen_three_stages(n, cr)) {
            int Ns = 1;
            for (int q = 0; q < 3; q++) {
                f.radix[q] = cr[q];
                f.nbf[q] = n / cr[q];
                f.ns[q] = Ns;
                f.tws[q] = n / (Ns * cr[q]);
                f.magic[q] = (unsigned)(((1ull << 32) + (unsigned long long)Ns - 1) / (unsigned long long)Ns);
                Ns *= cr[q];
            }
            f.nf = 3;
            return Ns == n;
        }
    }
#endif
    // large radices first: fewer stages (each stage is one trip of the tile through LDS)
    const int cand[12] = {23, 19, 17, 13, 11, 10, 8, 7, 5, 4, 3, 2}; // (round 5, second half: 11 and 13; 17, 19, 23 for the compile-time kernels only: gen_needs_compiled)
    for (int ci = 0; ci < 12; ci++)
        while (m % cand[ci] == 0 && m > 1) {
            if (f.nf == 8) return false;
            f.radix[f.nf++] = cand[ci];
            m /= cand[ci];
        }
    if (m != 1) return false;
#ifdef FDES_GEN_JIT_N
    {   // run-time compilation of ONE length (gen_jit.cpp): three stages with composite radices where the single radices need more,
        // chosen as the hand-measured table above suggests (gen_auto_three_stages)
        int cr[3] = {0, 0, 0}, c4[4] = {0, 0, 0, 0};
#ifdef FDES_GEN_JIT_R0 // tuning knob (FDES_JIT_STAGES=a,b,c[,d] -> gen_jit.cpp): the stage radices of this compilation, given outright
        if (n == FDES_GEN_JIT_N && n > 512 && (long)FDES_GEN_JIT_R0 * FDES_GEN_JIT_R1 * FDES_GEN_JIT_R2 * FDES_GEN_JIT_R3 == n) { // (beyond 512 points only: the two-image stages of shorter rows know the single radices alone)
            c4[0] = FDES_GEN_JIT_R0; c4[1] = FDES_GEN_JIT_R1; c4[2] = FDES_GEN_JIT_R2; c4[3] = FDES_GEN_JIT_R3;
            f.nf = FDES_GEN_JIT_R3 > 1 ? 4 : 3;
            for (int q = 0; q < 8; q++) f.radix[q] = q < f.nf ? c4[q] : 0;
        } else
#endif
        if (n == FDES_GEN_JIT_N && n > 512 && f.nf > 3 && gen_auto_three_stages(n, cr)) {
            f.nf = 3;
            for (int q = 0; q < 8; q++) f.radix[q] = q < 3 ? cr[q] : 0;
        } else if (n == FDES_GEN_JIT_N && n > 4096 && f.nf > 4 && gen_auto_four_stages(n, c4)) {
            f.nf = 4;
            for (int q = 0; q < 8; q++) f.radix[q] = q < 4 ? c4[q] : 0;
        }
    }
#endif
    int Ns = 1;
    for (int q = 0; q < f.nf; q++) {
        f.nbf[q] = n / f.radix[q];
        f.ns[q] = Ns;
        f.tws[q] = n / (Ns * f.radix[q]);
        f.magic[q] = (unsigned)(((1ull << 32) + (unsigned long long)Ns - 1) / (unsigned long long)Ns);
        Ns *= f.radix[q];
    }
    return true;
}
// Round 5 (FDES_GEN_T1024): the compile-time lengths beyond 2048 points as FOUR-row tiles with 1024 threads - one workgroup per
// CU with the sixteen waves that two 512-thread workgroups of two rows have, but 32-byte segments in the transposed store (a
// bare transposing copy at 4096^2 takes 81 us with two rows per workgroup against 50 us with four: profiles/r05_xpose_floor_two_rows.txt)
// and the twiddle table copied once per four rows
#ifndef FDES_GEN_T1024
#define FDES_GEN_T1024 0
#endif
#ifndef FDES_GEN_ROWS4
#define FDES_GEN_ROWS4 0 // experiment: four-row tiles (32-byte segments, one workgroup per CU) for the specialised lengths beyond 2048 points
#endif
#ifndef FDES_GEN_TW_LDS_LIMIT
#define FDES_GEN_TW_LDS_LIMIT 81920 // one-image kernels keep the twiddle table in LDS while tile + table fit this many bytes
#endif
__host__ __device__ constexpr bool gen_specialised(int n)
{
    return n == 320 || n == 800 || n == 1000 || n == 400 || n == 500 || n == 640 || n == 1280 || n == 1600 || n == 2000 || n == 2560 || n == 3000 ||
           n == 3072 || n == 3200 || n == 3600 || n == 4000;
}
__host__ __device__ constexpr int gen_rows(int n)
{
#ifdef FDES_GEN_JIT_ROWS // run-time compilation for a grid whose OTHER dimension the length's own tile rows do not divide (gen_pass_tile_rows)
    if (n == FDES_GEN_JIT_N) return FDES_GEN_JIT_ROWS;
#endif
    return n > 2048 ? (((FDES_GEN_ROWS4 || FDES_GEN_T1024) && gen_specialised(n)) ? 4 : 2) : (n > 512 ? 4 : 8);
}
constexpr int kGenThreads = 512; // threads of a workgroup (gen_threads below: 1024 for the four-row tiles of FDES_GEN_T1024)
#ifndef FDES_GEN_BIG_T1024
#define FDES_GEN_BIG_T1024 1 // rows beyond 4096 points (one two-row tile per CU): 1024 threads, i.e. sixteen waves instead of eight behind the same tile
#endif
__host__ __device__ constexpr bool gen_t1024(int n) { return (FDES_GEN_T1024 && n > 2048 && gen_specialised(n)) || (FDES_GEN_BIG_T1024 && n > 4096 && (n & (n - 1)) != 0); } // (measured, profiles/r05_rows_beyond_4096.txt: 4800^2 +33 %, 8000^2 +31 %, 6144^2 +11 %, 5000^2 +4 %; 8192^2 -16 %: it keeps 512 threads and 256 registers)
__host__ __device__ constexpr int gen_threads(int n)
{
#ifdef FDES_GEN_JIT_THREADS // run-time compilation: 256 threads for small tiles (gen_pass_threads_for)
    if (n == FDES_GEN_JIT_N) return FDES_GEN_JIT_THREADS;
#endif
    return gen_t1024(n) ? 1024 : kGenThreads;
}
__host__ __device__ constexpr int gen_lthreads(int n) { return gen_threads(n) == 1024 ? 10 : (gen_threads(n) == 256 ? 8 : 9); }
__host__ __device__ constexpr int gen_lrows(int rows) { return rows == 2 ? 1 : (rows == 4 ? 2 : 3); }
// rows beyond 2048 points: the tile images fill the LDS, the twiddle table is read from global memory
__host__ __device__ constexpr bool gen_tw_in_lds(int n) { return n <= 2048; }
// the whole description of an n-point pass, as a compile-time constant for the specialised kernels (k_gpass<NC != 0>)
__host__ __device__ constexpr GenFac make_fac(int n)
{
    GenFac f;
    factorize(n, f);
    f.rows = gen_rows(n);
    f.lrows = gen_lrows(f.rows);
    return f;
}
// the same length with the stages in REVERSE order (round 5): the trailing transform of a chained pass starts with the radix the
// leading one ended with, so that a thread's outputs of that last stage - element j + i N / RX of the row - are exactly the inputs
// of its first butterfly of the trailing transform and never leave the registers
__host__ __device__ constexpr GenFac make_fac_rev(int n)
{
    GenFac f = make_fac(n);
    GenFac r = f;
    int Ns = 1;
    for (int q = 0; q < f.nf; q++) {
        const int rx = f.radix[f.nf - 1 - q];
        r.radix[q] = rx;
        r.nbf[q] = n / rx;
        r.ns[q] = Ns;
        r.tws[q] = n / (Ns * rx);
        r.magic[q] = (unsigned)(((1ull << 32) + (unsigned long long)Ns - 1) / (unsigned long long)Ns);
        Ns *= rx;
    }
    return r;
}

// 512 threads = 8 waves per workgroup (two workgroups per CU at 1000 points: 2 waves per SIMD hide the LDS round trips
// of the stages); a row belongs to 512 / rows consecutive threads


// s = +1 forward, -1 inverse: multiply by -i (forward) / +i (inverse)
__device__ __forceinline__ cf mi_s(cf a, float s) { return cf{a.y * s, -a.x * s}; }
// a * (c - i s_ sgn): the constant root of unity exp(-i phi) (forward) / exp(+i phi) (inverse), c = cos phi, sn = sin phi
// (s is a constant at every call site - the transform direction is a template argument of the kernels - so the select folds;
//  the products are the packed forms of fft_dev.inc: one packed multiply + one packed FMA instead of four scalar operations)
__device__ __forceinline__ cf wmul_s(cf a, cf w, float s) { return s > 0.f ? cmul_rt(a, w) : cmulc_rt(a, w); } // a w (forward) / a conj(w) (inverse)
__device__ __forceinline__ cf rot_s(cf a, float c, float sn, float s) { return wmul_s(a, cf{c, -sn}, s); }

__device__ __forceinline__ void dft2(cf (&x)[kRxMax])
{
    const cf t = x[0] - x[1];
    x[0] = x[0] + x[1];
    x[1] = t;
}
__device__ __forceinline__ void dft3(cf (&x)[kRxMax], float s)
{
    const cf t1 = x[1] + x[2];
    const cf t2 = x[0] - t1 * 0.5f;
    const cf t3 = mi_s((x[1] - x[2]) * 0.866025403784438647f, s);
    x[0] = x[0] + t1;
    x[1] = t2 + t3;
    x[2] = t2 - t3;
}
__device__ __forceinline__ void dft4(cf (&x)[kRxMax], float s)
{
    const cf a = x[0] + x[2], b = x[0] - x[2], c = x[1] + x[3], d = mi_s(x[1] - x[3], s);
    x[0] = a + c;
    x[1] = b + d;
    x[2] = a - c;
    x[3] = b - d;
}
__device__ __forceinline__ void dft5_(cf& x0, cf& x1, cf& x2, cf& x3, cf& x4, float s)
{
    constexpr float c1 = 0.309016994374947424f, c2 = -0.809016994374947424f; // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float s1 = 0.951056516295153572f, s2 = 0.587785252292473129f;  // sin(2 pi / 5), sin(4 pi / 5)
    const cf a1 = x1 + x4, a2 = x2 + x3, b1 = x1 - x4, b2 = x2 - x3;
    const cf m1 = x0 + a1 * c1 + a2 * c2, m2 = x0 + a1 * c2 + a2 * c1;
    const cf n1 = mi_s(b1 * s1 + b2 * s2, s), n2 = mi_s(b1 * s2 - b2 * s1, s);
    x0 = x0 + a1 + a2;
    x1 = m1 + n1;
    x4 = m1 - n1;
    x2 = m2 + n2;
    x3 = m2 - n2;
}
__device__ __forceinline__ void dft5(cf (&x)[kRxMax], float s) { dft5_(x[0], x[1], x[2], x[3], x[4], s); }
// radix 7 (round 4: 7-smooth lengths such as 448, 896, 1400, 1792, 2016, 3584): the six non-trivial outputs from the three
// symmetric sums a_j = x_j + x_(7-j) and differences b_j = x_j - x_(7-j), X_k = x_0 + sum_j a_j cos(2 pi j k / 7) -/+ i sum_j b_j sin(2 pi j k / 7)
__device__ __forceinline__ void dft7(cf (&x)[kRxMax], float s)
{
    constexpr float c1 = 0.623489801858733530f, c2 = -0.222520933956314404f, c3 = -0.900968867902419126f; // cos(2 pi j / 7)
    constexpr float s1 = 0.781831482468029809f, s2 = 0.974927912181823607f, s3 = 0.433883739117558120f;  // sin(2 pi j / 7)
    const cf a1 = x[1] + x[6], a2 = x[2] + x[5], a3 = x[3] + x[4];
    const cf b1 = x[1] - x[6], b2 = x[2] - x[5], b3 = x[3] - x[4];
    const cf m1 = x[0] + a1 * c1 + a2 * c2 + a3 * c3; // k = 1, 6: cos(2 pi j k / 7) = c1, c2, c3
    const cf m2 = x[0] + a1 * c2 + a2 * c3 + a3 * c1; // k = 2, 5: c2, c4 = c3, c6 = c1
    const cf m3 = x[0] + a1 * c3 + a2 * c1 + a3 * c2; // k = 3, 4: c3, c6 = c1, c9 = c2
    const cf n1 = mi_s(b1 * s1 + b2 * s2 + b3 * s3, s); // k = 1: sin(2 pi j / 7) = s1, s2, s3
    const cf n2 = mi_s(b1 * s2 - b2 * s3 - b3 * s1, s); // k = 2: sin(4 pi j / 7) = s2, -s3, -s1
    const cf n3 = mi_s(b1 * s3 - b2 * s1 + b3 * s2, s); // k = 3: sin(6 pi j / 7) = s3, -s1, s2
    x[0] = x[0] + a1 + a2 + a3;
    x[1] = m1 + n1; x[6] = m1 - n1;
    x[2] = m2 + n2; x[5] = m2 - n2;
    x[3] = m3 + n3; x[4] = m3 - n3;
}
__device__ __forceinline__ void dft8(cf (&x)[kRxMax], float s)
{
    // two radix-4 over the even / odd inputs, then the radix-2 level with W_8^k
    cf e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    {
        const cf a = e0 + e2, b = e0 - e2, c = e1 + e3, d = mi_s(e1 - e3, s);
        e0 = a + c; e1 = b + d; e2 = a - c; e3 = b - d;
    }
    {
        const cf a = o0 + o2, b = o0 - o2, c = o1 + o3, d = mi_s(o1 - o3, s);
        o0 = a + c; o1 = b + d; o2 = a - c; o3 = b - d;
    }
    constexpr float h = 0.707106781186547524f;
    o1 = rot_s(o1, h, h, s);
    o2 = mi_s(o2, s);
    o3 = rot_s(o3, -h, h, s);
    x[0] = e0 + o0; x[4] = e0 - o0;
    x[1] = e1 + o1; x[5] = e1 - o1;
    x[2] = e2 + o2; x[6] = e2 - o2;
    x[3] = e3 + o3; x[7] = e3 - o3;
}
__device__ __forceinline__ void dft10(cf (&x)[kRxMax], float s)
{
    // X[k] = E[k mod 5] + W_10^k O[k mod 5]: two radix-5 over the even / odd inputs
    cf e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], e4 = x[8], o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7], o4 = x[9];
    dft5_(e0, e1, e2, e3, e4, s);
    dft5_(o0, o1, o2, o3, o4, s);
    o1 = rot_s(o1, 0.809016994374947424f, 0.587785252292473129f, s);  // W_10^1
    o2 = rot_s(o2, 0.309016994374947424f, 0.951056516295153572f, s);  // W_10^2
    o3 = rot_s(o3, -0.309016994374947424f, 0.951056516295153572f, s); // W_10^3
    o4 = rot_s(o4, -0.809016994374947424f, 0.587785252292473129f, s); // W_10^4
    x[0] = e0 + o0; x[5] = e0 - o0;
    x[1] = e1 + o1; x[6] = e1 - o1;
    x[2] = e2 + o2; x[7] = e2 - o2;
    x[3] = e3 + o3; x[8] = e3 - o3;
    x[4] = e4 + o4; x[9] = e4 - o4;
}

// radix 11 and 13 (grids such as 1100 = 2 nx of a .qsc with nx = 550, 1430, 2600, 3300: cuFFT serves any size alike,
// src/paramStructure.cu:676-679): an odd prime P from the (P - 1) / 2 symmetric sums a_j = x_j + x_(P-j) and differences
// b_j = x_j - x_(P-j), X_k = x_0 + sum_j a_j cos(2 pi j k / P) -/+ i sum_j b_j sin(2 pi j k / P), X_(P-k) its mirror image
// (the form of dft7 above; cosines and sines from the double-precision constants below, rounded once)
constexpr double kCos11[5] = {0.84125353283118116886, 0.41541501300188642553, -0.14231483827328514044, -0.65486073394528506406, -0.95949297361449738989};
constexpr double kSin11[5] = {0.54064081745559758211, 0.90963199535451837141, 0.98982144188093273238, 0.75574957435425828377, 0.28173255684142969771};
constexpr double kCos13[6] = {0.88545602565320989590, 0.56806474673115580251, 0.12053668025532305335, -0.35460488704253562597, -0.74851074817110109863, -0.97094181742605202716};
constexpr double kSin13[6] = {0.46472317204376854566, 0.82298386589365639458, 0.99270887409805399280, 0.93501624268541482344, 0.66312265824079520238, 0.23931566428755776715};
// 17, 19, 23 (lengths such as 1088 = 64 x 17, 1216 = 64 x 19, 1472 = 64 x 23): in the compile-time kernels only, i.e. compiled at plan creation
constexpr double kCos17[8] = {0.93247222940435580457, 0.73900891722065911592, 0.4457383557765382674, 0.09226835946330199524, -0.27366299007208286354, -0.60263463637925638918, -0.85021713572961415213, -0.98297309968390177828};
constexpr double kSin17[8] = {0.36124166618715294874, 0.67369564364655721171, 0.89516329135506232207, 0.99573417629503452187, 0.96182564317281907041, 0.79801722728023950333, 0.52643216287735580024, 0.18374951781657033157};
constexpr double kCos19[9] = {0.94581724170063467902, 0.78914050939639359922, 0.54694815812242687471, 0.24548548714079914892, -0.0825793454723323246, -0.40169542465296945752, -0.67728157162574107476, -0.87947375120648907139, -0.9863613034027223736};
constexpr double kSin19[9] = {0.32469946920468348741, 0.61421271268966781744, 0.83716647826252857481, 0.96940026593933041674, 0.99658449300666984982, 0.91577332665505743992, 0.73572391067313162477, 0.47594739303707354443, 0.16459459028073389414};
constexpr double kCos23[11] = {0.96291728734779929502, 0.85441940454648855255, 0.68255314321865408287, 0.46006503773115212604, 0.20345601305263378988, -0.068242413364670975921, -0.33487961217098615196, -0.57668032211486714125, -0.77571129070441980704, -0.91721130150545301784, -0.99068594603633075234};
constexpr double kSin23[11] = {0.26979677115702427125, 0.51958395003543357813, 0.73083596427812410165, 0.88788521840237523498, 0.97908408768232287563, 0.99766876919053919845, 0.94226092211882049562, 0.81696989301044201697, 0.63108794432605278937, 0.398401089846241458, 0.13616664909624659076};
template <int P> __host__ __device__ constexpr double prime_cos(int q) { return P == 11 ? kCos11[q] : (P == 13 ? kCos13[q] : (P == 17 ? kCos17[q] : (P == 19 ? kCos19[q] : kCos23[q]))); }
template <int P> __host__ __device__ constexpr double prime_sin(int q) { return P == 11 ? kSin11[q] : (P == 13 ? kSin13[q] : (P == 17 ? kSin17[q] : (P == 19 ? kSin19[q] : kSin23[q]))); }
template <int P> __device__ __forceinline__ void dft_prime(cf (&x)[kRxMax], float s)
{
    static_assert(P == 11 || P == 13 || P == 17 || P == 19 || P == 23, "radix");
    constexpr int H = (P - 1) / 2;
    cf a[H], b[H];
#pragma unroll
    for (int j = 1; j <= H; j++) {
        a[j - 1] = x[j] + x[P - j];
        b[j - 1] = x[j] - x[P - j];
    }
    cf sum = x[0];
#pragma unroll
    for (int j = 0; j < H; j++) sum = sum + a[j];
    cf m[H], n[H];
#pragma unroll
    for (int k = 1; k <= H; k++) {
        cf mk = x[0], nk = cf{0.f, 0.f};
#pragma unroll
        for (int j = 1; j <= H; j++) {
            const int q = (j * k) % P;            // angle 2 pi q / P; cos(2 pi q / P) = cos(2 pi (P - q) / P), sin changes sign
            const int qq = q <= H ? q : P - q;
            const float c = (float)prime_cos<P>(qq - 1);
            const float sn = (float)prime_sin<P>(qq - 1) * (q <= H ? 1.f : -1.f);
            mk = mk + a[j - 1] * c;
            nk = nk + b[j - 1] * sn;
        }
        m[k - 1] = mk;
        n[k - 1] = mi_s(nk, s);
    }
    x[0] = sum;
#pragma unroll
    for (int k = 1; k <= H; k++) {
        x[k] = m[k - 1] + n[k - 1];
        x[P - k] = m[k - 1] - n[k - 1];
    }
}

// One Stockham stage of radix RX over the R rows of the tile: butterfly j of a row takes src[j + i N/RX], twiddles
// W_N^(i k N / (Ns RX)) with k = j mod Ns, and leaves dst[(j / Ns) Ns RX + k + i Ns].  `row` / `jt` / `tpr`: this thread's
// row, its index among the tpr threads of that row.
template <int RX>
__device__ __forceinline__ void gen_stage(const cf* __restrict__ src, cf* __restrict__ dst, const cf* __restrict__ twl, const int N, const int nb,
                                          const int Ns, const int tws, const unsigned magic, const float s, const int row, const int jt,
                                          const int tpr)
{
    const cf* __restrict__ srow = src + row * N;
    cf* __restrict__ drow = dst + row * N;
    for (int j = jt; j < nb; j += tpr) {
        const int k = (Ns > 1) ? j - (int)__umulhi((unsigned)j, magic) * Ns : 0;
        cf x[kRxMax];
#pragma unroll
        for (int i = 0; i < RX; i++) x[i] = srow[j + i * nb];
        if (Ns > 1) {
            const int dk = k * tws; // i k tws < RX Ns N / (Ns RX) = N: the table index needs no reduction
#pragma unroll
            for (int i = 1; i < RX; i++) x[i] = wmul_s(x[i], twl[i * dk], s); // * w (forward) or conj(w) (inverse)
        }
        if constexpr (RX == 2) dft2(x);
        if constexpr (RX == 3) dft3(x, s);
        if constexpr (RX == 4) dft4(x, s);
        if constexpr (RX == 5) dft5(x, s);
        if constexpr (RX == 7) dft7(x, s);
        if constexpr (RX == 8) dft8(x, s);
        if constexpr (RX == 10) dft10(x, s);
        if constexpr (RX == 11 || RX == 13 || RX == 17 || RX == 19 || RX == 23) dft_prime<RX>(x, s);
        cf* __restrict__ out = drow + (j - k) * RX + k;
#pragma unroll
        for (int i = 0; i < RX; i++) out[i * Ns] = x[i];
    }
}

// twiddle table of a pass: 1 = all N roots in LDS, 2 = the first N / 2 in LDS (one-image kernels whose tile + full table
// would not fit twice on a CU: 3600 and 4000 points; measured 1.48 k -> see profiles/r04_mixed_radix_beyond_2048.txt; even N
// only), 0 = read from global memory
__host__ __device__ constexpr int gen_tw_mode(int n, int rows, bool one_image)
{
    if (!one_image) return n <= 2048 ? 1 : 0;
    const size_t limit = (gen_t1024(n) || n > 4096) ? (size_t)160 * 1024 : (size_t)FDES_GEN_TW_LDS_LIMIT; // (the four-row tiles, and every tile of rows beyond 4096 points, are alone on their CU)
    if (sizeof(float) * 2 * ((size_t)rows * n + n) + 64 <= limit) return 1;
    if (n % 2 == 0 && sizeof(float) * 2 * ((size_t)rows * n + n / 2) + 64 <= limit) return 2;
    return 0;
}

// ---- butterflies of any supported radix on RX registers of a thread (round 5) ------------------------------------------
// Compile-time roots of unity for the composite radices: cos / sin (2 pi m / M) from a Taylor series in double precision
// (|angle| <= pi: 1e-15 absolute), multiples of a quarter turn exact.
constexpr double ct_pi = 3.14159265358979323846264338327950288;
constexpr double ct_angle(int m, int M) { m %= M; if (2 * m > M) m -= M; return 2.0 * ct_pi * (double)m / (double)M; }
constexpr double ct_cos(int m, int M)
{
    m = ((m % M) + M) % M;
    if ((4 * m) % M == 0) { const int q = 4 * m / M; return q == 0 ? 1.0 : (q == 2 ? -1.0 : 0.0); }
    const double x = ct_angle(m, M), x2 = x * x;
    double t = 1.0, r = 1.0;
    for (int n = 1; n <= 16; n++) { t *= -x2 / (double)((2 * n - 1) * (2 * n)); r += t; }
    return r;
}
constexpr double ct_sin(int m, int M)
{
    m = ((m % M) + M) % M;
    if ((4 * m) % M == 0) { const int q = 4 * m / M; return q == 1 ? 1.0 : (q == 3 ? -1.0 : 0.0); }
    const double x = ct_angle(m, M), x2 = x * x;
    double t = x, r = x;
    for (int n = 1; n <= 16; n++) { t *= -x2 / (double)((2 * n) * (2 * n + 1)); r += t; }
    return r;
}
template <int M> struct CtRoots {
    float c[M], s[M];
    constexpr CtRoots() : c(), s()
    {
        for (int m = 0; m < M; m++) { c[m] = (float)ct_cos(m, M); s[m] = (float)ct_sin(m, M); }
    }
};
// small butterflies on named registers
__device__ __forceinline__ void pdft3(cf& x0, cf& x1, cf& x2, float s)
{
    const cf t1 = x1 + x2;
    const cf t2 = x0 - t1 * 0.5f;
    const cf t3 = mi_s((x1 - x2) * 0.866025403784438647f, s);
    x0 = x0 + t1;
    x1 = t2 + t3;
    x2 = t2 - t3;
}
__device__ __forceinline__ void pdft4(cf& x0, cf& x1, cf& x2, cf& x3, float s)
{
    const cf a = x0 + x2, b = x0 - x2, c = x1 + x3, d = mi_s(x1 - x3, s);
    x0 = a + c;
    x1 = b + d;
    x2 = a - c;
    x3 = b - d;
}
// R-point DFT in place on x[0], x[ST], x[2 ST], ... (R = 2 ... 5)
template <int R, int ST> __device__ __forceinline__ void pdft(cf* x, float s)
{
    if constexpr (R == 2) { const cf t = x[0] - x[ST]; x[0] = x[0] + x[ST]; x[ST] = t; }
    if constexpr (R == 3) pdft3(x[0], x[ST], x[2 * ST], s);
    if constexpr (R == 4) pdft4(x[0], x[ST], x[2 * ST], x[3 * ST], s);
    if constexpr (R == 5) dft5_(x[0], x[ST], x[2 * ST], x[3 * ST], x[4 * ST], s);
}
// M = A B points in place: input n = B n1 + n2 in x[n]; A-point butterflies over n1, twiddle W_M^(n2 k1), B-point butterflies
// over n2; output X[k1 + A k2] is left in x[B k1 + k2] (rdx_slot below: no register is moved)
template <int A, int B> __device__ __forceinline__ void cdft(cf* x, float s)
{
    constexpr int M = A * B;
    constexpr CtRoots<M> W{};
#pragma unroll
    for (int n2 = 0; n2 < B; n2++) {
        pdft<A, B>(x + n2, s);
#pragma unroll
        for (int k1 = 1; k1 < A; k1++) {
            const int m = (n2 * k1) % M;
            if (m == 0) continue;
            cf& v = x[n2 + B * k1];
            if ((4 * m) % M == 0) {
                const int q = 4 * m / M; // W^m = (-i)^q forward
                v = q == 1 ? mi_s(v, s) : (q == 2 ? -v : -mi_s(v, s));
            } else {
                v = rot_s(v, W.c[m], W.s[m], s);
            }
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < A; k1++) pdft<B, 1>(x + B * k1, s);
}
constexpr bool rdx_composite(int rx) { return rx == 12 || rx == 15 || rx == 16 || rx == 20 || rx == 25; }
constexpr int rdx_a(int rx) { return rx == 12 ? 3 : (rx == 15 ? 3 : (rx == 16 ? 4 : (rx == 20 ? 4 : 5))); }
// register of a thread's butterfly that holds output i of an RX-point butterfly (rdx_dft)
template <int RX> __host__ __device__ constexpr int rdx_slot(int i)
{
    if constexpr (rdx_composite(RX)) return (RX / rdx_a(RX)) * (i % rdx_a(RX)) + i / rdx_a(RX);
    else return i;
}
template <int RX> __device__ __forceinline__ void rdx_dft(cf (&x)[RX], float s)
{
    if constexpr (RX == 2 || RX == 3 || RX == 4 || RX == 5) pdft<RX, 1>(x, s);
    else if constexpr (RX == 7 || RX == 8 || RX == 10 || RX == 11 || RX == 13 || RX == 17 || RX == 19 || RX == 23) {
        cf t[kRxMax];
#pragma unroll
        for (int i = 0; i < RX; i++) t[i] = x[i];
        if constexpr (RX == 7) dft7(t, s);
        if constexpr (RX == 8) dft8(t, s);
        if constexpr (RX == 10) dft10(t, s);
        if constexpr (RX == 11 || RX == 13 || RX == 17 || RX == 19 || RX == 23) dft_prime<RX>(t, s);
#pragma unroll
        for (int i = 0; i < RX; i++) x[i] = t[i];
    } else {
        static_assert(rdx_composite(RX), "radix");
        cdft<rdx_a(RX), RX / rdx_a(RX)>(x, s);
    }
}

// One stage of the compile-time kernels, IN PLACE (round 4) and in three pieces (round 5): a thread takes the inputs of ALL
// its butterflies into registers (NBT = ceil(nb / tpr) butterflies of RX elements: the thread's share of the row), transforms
// them, and writes them back to the SAME image at the Stockham positions; the row's threads meet between the reads and the
// writes.  One image instead of two halves the LDS of a workgroup: two workgroups per CU where the two-image form admits one.
// The pieces let a transform take its first stage's inputs straight from global memory (element j + i nb of a row: consecutive
// threads read consecutive elements) and leave its last stage's outputs in registers, where the point-wise operation of the
// pass finds them: two trips of the tile through LDS fewer per transform pair.
template <int NC, int Q, bool REV = false> struct GStage {
    static constexpr GenFac F = REV ? make_fac_rev(NC) : make_fac(NC);
    static constexpr int RX = F.radix[Q], nb = F.nbf[Q], Ns = F.ns[Q], tws = F.tws[Q];
    static constexpr unsigned magic = F.magic[Q];
    static constexpr int tpr = gen_threads(NC) >> F.lrows;
    static constexpr int NBT = (nb + tpr - 1) / tpr;
    static constexpr bool half_tw = gen_tw_mode(NC, F.rows, true) == 2;
    static constexpr bool last = Q + 1 == F.nf;
};
template <int NC, int Q, bool REV = false> using XArr = cf[GStage<NC, Q, REV>::NBT][GStage<NC, Q, REV>::RX]; // a thread's registers of stage Q
template <int NC, int Q, bool REV = false> __device__ __forceinline__ int gs_k(int j)
{
    using S = GStage<NC, Q, REV>;
    return (S::Ns > 1) ? j - (int)__umulhi((unsigned)j, S::magic) * S::Ns : 0;
}
// column of output i of butterfly j after stage Q
template <int NC, int Q, bool REV = false> __device__ __forceinline__ int gs_out_col(int j, int i)
{
    using S = GStage<NC, Q, REV>;
    const int k = gs_k<NC, Q, REV>(j);
    return (j - k) * S::RX + k + i * S::Ns;
}
template <int NC, int Q, bool REV = false> __device__ __forceinline__ void gs_load_lds(XArr<NC, Q, REV>& x, const cf* __restrict__ rowp, const int jt)
{
    using S = GStage<NC, Q, REV>;
#pragma unroll
    for (int b = 0; b < S::NBT; b++) {
        const int j = jt + b * S::tpr;
        if (j < S::nb) {
#pragma unroll
            for (int i = 0; i < S::RX; i++) x[b][i] = rowp[j + i * S::nb];
        }
    }
}
// from a row in global memory; `band`: columns beyond the band limit count as zero and are not fetched
template <int NC, int Q> __device__ __forceinline__ void gs_load_global(XArr<NC, Q>& x, const cf* __restrict__ grow, const int jt,
                                                                         const bool band, const int bandv)
{
    using S = GStage<NC, Q>;
#pragma unroll
    for (int b = 0; b < S::NBT; b++) {
        const int j = jt + b * S::tpr;
        if (j < S::nb) {
#pragma unroll
            for (int i = 0; i < S::RX; i++) {
                const int c = j + i * S::nb;
                const bool dead = band && dead_index(iwc(c, NC), bandv);
                x[b][i] = dead ? cf{0.f, 0.f} : row_load<(NC > FDES_GEN_NT_ABOVE ? 1 << 20 : 0)>(grow + c); // (non-temporal for the lengths beyond FDES_GEN_NT_ABOVE: fft_dev.inc)
            }
        }
    }
}
template <int NC, int Q, bool REV = false> __device__ __forceinline__ void gs_compute(XArr<NC, Q, REV>& x, const cf* __restrict__ twl, const float s, const int jt)
{
    using S = GStage<NC, Q, REV>;
#pragma unroll
    for (int b = 0; b < S::NBT; b++) {
        const int j = jt + b * S::tpr;
        if (j < S::nb) {
            if constexpr (S::Ns > 1) {
                const int dk = gs_k<NC, Q, REV>(j) * S::tws; // i k tws < N: the table index needs no reduction
#pragma unroll
                for (int i = 1; i < S::RX; i++) {
                    if constexpr (S::half_tw) { // the table holds W_N^m for m < N / 2 only: W_N^(m + N/2) = -W_N^m
                        const int m = i * dk, h = NC >> 1;
                        const cf w = twl[m >= h ? m - h : m];
                        x[b][i] = wmul_s(x[b][i], m >= h ? -w : w, s);
                    } else {
                        x[b][i] = wmul_s(x[b][i], twl[i * dk], s);
                    }
                }
            }
            rdx_dft<S::RX>(x[b], s);
        }
    }
}
template <int NC, int Q, bool REV = false> __device__ __forceinline__ void gs_store_lds(const XArr<NC, Q, REV>& x, cf* __restrict__ rowp, const int jt)
{
    using S = GStage<NC, Q, REV>;
#pragma unroll
    for (int b = 0; b < S::NBT; b++) {
        const int j = jt + b * S::tpr;
        if (j < S::nb) {
            cf* __restrict__ out = rowp + gs_out_col<NC, Q, REV>(j, 0);
#pragma unroll
            for (int i = 0; i < S::RX; i++) out[i * S::Ns] = x[b][rdx_slot<S::RX>(i)];
        }
    }
}
// all stages of an NC-point row in place, image to image
template <int NC, int Q> __device__ __forceinline__ void gen_inplace_stages(cf* __restrict__ img, const cf* __restrict__ twl, const float s, const int row, const int jt)
{
    constexpr GenFac F = make_fac(NC);
    if constexpr (Q < F.nf) {
        cf x[GStage<NC, Q>::NBT][GStage<NC, Q>::RX];
        gs_load_lds<NC, Q>(x, img + row * NC, jt);
        __syncthreads(); // every input of the stage has been read
        gs_compute<NC, Q>(x, twl, s, jt);
        gs_store_lds<NC, Q>(x, img + row * NC, jt);
        __syncthreads();
        gen_inplace_stages<NC, Q + 1>(img, twl, s, row, jt);
    }
}
// The trailing transform of a chained pass in the REVERSED stage order (make_fac_rev): stage 0 takes `x0` - the caller's registers,
// the leading transform's last-stage outputs after the point-wise operation, renamed into butterfly-input order - stages 1 ... image
// to image; the result is in the image in natural order.
template <int NC, int Q> __device__ __forceinline__ void gen_rev_stages(cf* __restrict__ img, const cf* __restrict__ twl, const float s, const int row, const int jt)
{
    constexpr GenFac F = make_fac_rev(NC);
    if constexpr (Q < F.nf) {
        XArr<NC, Q, true> x;
        gs_load_lds<NC, Q, true>(x, img + row * NC, jt);
        __syncthreads();
        gs_compute<NC, Q, true>(x, twl, s, jt);
        gs_store_lds<NC, Q, true>(x, img + row * NC, jt);
        __syncthreads();
        gen_rev_stages<NC, Q + 1>(img, twl, s, row, jt);
    }
}
template <int NC> __device__ __forceinline__ void gen_rev_from_regs(cf* __restrict__ img, const cf* __restrict__ twl, const float s, const int row, const int jt,
                                                                    XArr<NC, 0, true>& x0)
{
    gs_compute<NC, 0, true>(x0, twl, s, jt);
    gs_store_lds<NC, 0, true>(x0, img + row * NC, jt);
    __syncthreads();
    gen_rev_stages<NC, 1>(img, twl, s, row, jt);
}
// The stages Q ... nf - 2 image to image, then the last stage from the image into the caller's registers `xl` (round 5).
// FROM_GLOBAL (Q = 0 only): the first stage reads its inputs from `grow` instead of the image (nothing of the image is read
// before the first barrier, but other threads may still be reading what the PREVIOUS phase left there: the caller has passed a
// barrier since).
template <int NC, int Q, bool FROM_GLOBAL, bool PRELOADED = false>
__device__ __forceinline__ void gen_chain_to_regs(cf* __restrict__ img, const cf* __restrict__ grow, const bool band, const int bandv, const cf* __restrict__ twl,
                                                  const float s, const int row, const int jt, XArr<NC, make_fac(NC).nf - 1>& xl, XArr<NC, 0>& x0)
{
    constexpr GenFac F = make_fac(NC);
    if constexpr (Q + 1 < F.nf) {
        if constexpr (Q == 0 && FROM_GLOBAL) {
            // (pre: the caller requested this row's first-stage inputs earlier - the second operand of the product, while the
            //  first one was being transformed - and they are used where they landed)
            if constexpr (!PRELOADED) gs_load_global<NC, 0>(x0, grow, jt, band, bandv);
            gs_compute<NC, 0>(x0, twl, s, jt);
            gs_store_lds<NC, 0>(x0, img + row * NC, jt);
        } else {
            XArr<NC, Q> x;
            gs_load_lds<NC, Q>(x, img + row * NC, jt);
            __syncthreads();
            gs_compute<NC, Q>(x, twl, s, jt);
            gs_store_lds<NC, Q>(x, img + row * NC, jt);
        }
        __syncthreads();
        gen_chain_to_regs<NC, Q + 1, false>(img, grow, band, bandv, twl, s, row, jt, xl, x0);
    } else {
        gs_load_lds<NC, Q>(xl, img + row * NC, jt);
        __syncthreads(); // the image is free again
        gs_compute<NC, Q>(xl, twl, s, jt);
    }
}
// one image for the compile-time lengths beyond 512 points (round 4: beyond 1024)
#ifndef FDES_GEN_ONE_ABOVE
#define FDES_GEN_ONE_ABOVE 512 // round 5: 640, 800 and 1000 points too (with the chained transforms: 800^2 +18 %, 1000^2 +23 %; rows up to 512 points
                               // belong to one wave each, whose stages need no workgroup barrier between two images)
#endif
__host__ __device__ constexpr bool gen_one_image(int nc) { return nc > FDES_GEN_ONE_ABOVE; }
// twiddle table in LDS while the workgroup then still fits twice on a CU (80 KiB), else read from global memory

// row FFTs of the whole tile; on return `cur` points at the image that holds the result (natural order)
// (CT: F is a compile-time constant - the stage loop is unrolled, every stage's radix, counts and index arithmetic fold)
template <bool CT, bool ONE = false, int NCI = 320>
__device__ __forceinline__ void gen_fft(cf*& cur, cf*& other, const cf* __restrict__ twl, const GenFac& F, const bool inverse)
{
    const float s = inverse ? -1.f : 1.f;
    const int tpr = (CT ? gen_threads(NCI) : kGenThreads) >> F.lrows, row = (int)threadIdx.x / tpr, jt = (int)threadIdx.x - row * tpr; // tpr is a power of two
    if constexpr (ONE) {
        gen_inplace_stages<NCI, 0>(cur, twl, s, row, jt);
        return;
    }
    auto stage = [&](const int q) {
        const int rx = F.radix[q];
        switch (rx) {
        case 2: gen_stage<2>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        case 3: gen_stage<3>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        case 4: gen_stage<4>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        case 5: gen_stage<5>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        case 7: gen_stage<7>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        case 8: gen_stage<8>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        case 11: gen_stage<11>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        case 13: gen_stage<13>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
#ifdef FDES_GEN_JIT_N // (17, 19, 23: kernels compiled at plan creation only - lengths up to 512 points run their stages between two images, through here)
        case 17: gen_stage<17>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        case 19: gen_stage<19>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        case 23: gen_stage<23>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
#endif
        default: gen_stage<10>(cur, other, twl, F.n, F.nbf[q], F.ns[q], F.tws[q], F.magic[q], s, row, jt, tpr); break;
        }
        // a stage of a row touches that row only: with 64 threads per row (rows up to 512 points) a row belongs to ONE wave,
        // whose LDS accesses complete in program order - between its stages no workgroup barrier is needed, only the wave's
        // own fence (behind the last stage the callers read the tile across rows again: barrier)
        if (tpr == 64 && q + 1 < F.nf) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
        cf* t_ = cur; cur = other; other = t_;
    };
    if constexpr (CT) {
#pragma unroll
        for (int q = 0; q < F.nf; q++) stage(q);
    } else {
#pragma unroll 1
        for (int q = 0; q < F.nf; q++) stage(q);
    }
}

// (float)(i1^2 + i2^2) * 9 / mindim^2 > 1: zeroHighFreq's test (src/multisliceSimulation.cu:241)
__device__ __forceinline__ bool gen_outside(int i1, int i2, float md) { return ((float)(i1 * i1 + i2 * i2) * 9.f / (md * md)) > 1.f; }

// EPT: elements a thread owns (element e = tid + 512 i of the R x N tile, row-major)
// NC: 0 = any supported length, described by the run-time argument; otherwise THE length, the description a constant
// (the reference's shipped grids 320, 800, 1000: the generic kernel spends ten times the vector instructions per point
// of the power-of-two kernels on stage bookkeeping and index arithmetic)
template <int NC, int EPT, int PRE, int MID, int POST, bool STORE_T>
__device__ __forceinline__ void gpass_body(const PassArgs& A, const GenFac& Frt)
{
    constexpr bool CT = NC != 0;
    constexpr GenFac FC = make_fac(CT ? NC : 320);
    const GenFac F = CT ? FC : Frt;
    extern __shared__ cf glds[];
    const int N = F.n, R = F.rows, tid = threadIdx.x;
    constexpr int THR = gen_threads(NC), LTHR = gen_lthreads(NC);
    const int tile = R * N;
    constexpr bool ONE = CT && gen_one_image(NC);
    cf* cur = glds;
    cf* other = ONE ? glds : glds + tile;
    const int nimg = ONE ? 1 : 2;
    const int tw_mode = CT ? gen_tw_mode(FC.n, FC.rows, ONE) : (gen_tw_in_lds(N) ? 1 : 0);
    const bool tw_lds = tw_mode != 0;
    const cf* twl = tw_lds ? glds + nimg * tile : reinterpret_cast<const cf*>(A.tw0);
    const int nvirt = (int)gridDim.x, vb = (int)blockIdx.x;
    int bg;
    {   // XCD-aware remap (fft_lds.hip): workgroups of one XCD own consecutive row groups
        const int q = nvirt >> 3, rem = nvirt & 7, xcd = vb & 7, k = vb >> 3;
        bg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    }
    if (A.live_rows_only) {
        const int L = A.band_L;
        const int g_lo = L / R + 1, g_hi = (A.nrows - L) / R;
        if (bg >= g_lo) bg = g_hi + (bg - g_lo);
    }
    const int row0 = bg * R;
    // The LAST tile of a grid whose row count R does not divide is partial (round 5: 750^2, odd lengths): rows row0 + rvalid ... are not
    // there.  Their threads run along (every barrier is the workgroup's), read the tile's last valid row instead of memory behind the
    // grid (erl / rowl below: a load address only), and nothing of theirs is stored.
    const int rvalid = (A.nrows - row0 < R) ? A.nrows - row0 : R;
    const unsigned pin = A.pitch_in ? (unsigned)A.pitch_in : (unsigned)N;
    const unsigned ldt = A.pitch_out ? (unsigned)A.pitch_out : (unsigned)A.nrows;
    const size_t gbase = (size_t)row0 * pin;
    // batch of grids in one launch (grid.z, PassArgs::nbatch): this workgroup's grid
    const int bz = (int)blockIdx.z;
    const size_t zoff_in = (A.nbatch > 1) ? (size_t)(A.use_zin ? A.zin[bz] : bz) * A.bstride_in0 : (size_t)0;
    const size_t zoff_out = (A.nbatch > 1) ? (size_t)bz * A.bstride_out : (size_t)0;
    const cf* __restrict__ in0 = A.in0 ? reinterpret_cast<const cf*>(A.in0) + gbase + zoff_in : nullptr;
    const cf* __restrict__ in1 = A.in1 ? reinterpret_cast<const cf*>(A.in1) + gbase + ((A.nbatch > 1) ? (size_t)bz * A.bstride_in1 : (size_t)0) : nullptr;
    const float* __restrict__ gtab = A.gtab ? A.gtab + gbase : nullptr;
    cf* const out0 = reinterpret_cast<cf*>(A.out) + zoff_out + ((MID == MID_ATOMS) ? (size_t)blockIdx.y * A.species_stride : (size_t)0);
    // round 5: the slice-loop passes of the one-image kernels chain their transforms through registers (below)
    constexpr bool CHAIN = FDES_GEN_CHAIN && ONE && PRE != XF_NONE && POST != XF_NONE && STORE_T &&
                           (MID == MID_GTAB || MID == MID_EXPIV_PAIR || MID == MID_MASK || MID == MID_MULPSI || MID == MID_PTAB);
    if constexpr ((PRE != XF_NONE || POST != XF_NONE) && !CHAIN) {
        if (tw_lds) {
            const cf* __restrict__ tw = reinterpret_cast<const cf*>(A.tw0);
            cf* twd = glds + nimg * tile;
            for (int i = tid; i < (tw_mode == 2 ? N / 2 : N); i += THR) twd[i] = tw[i];
        }
    }
    // this thread's elements: columns jt + tpr i of ONE row (the 512 / R threads of a row are consecutive: their loads of a
    // row are contiguous, and row and column need no arithmetic per element); EPT >= N / tpr = R N / 512
    int er[EPT], ec[EPT];
    {
        const int tpr = THR >> F.lrows, r = tid >> (LTHR - F.lrows), c = tid & (tpr - 1);
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            er[i] = r;
            ec[i] = c + i * tpr;
        }
    }
    const int erl = (er[0] < rvalid) ? er[0] : rvalid - 1; // this thread's row as a LOAD address (every element of a thread lies in one row)
    auto valid = [&](int i) { return ec[i] < N; };
    auto load_tile = [&](const cf* __restrict__ src, cf* __restrict__ dstl, const bool band) {
#pragma unroll
        for (int i = 0; i < EPT; i++)
            if (valid(i)) {
                const bool dead = band && dead_index(iwc(ec[i], N), A.band);
                dstl[er[i] * N + ec[i]] = dead ? cf{0.f, 0.f} : src[(unsigned)erl * pin + (unsigned)ec[i]];
            }
    };

    // ---- round 5: the slice-loop passes of the one-image kernels with their transforms chained through registers: the first
    // stage of the leading transform reads the rows from global memory, its last stage leaves the spectrum (resp. the real-space
    // row) in registers, the point-wise operation works there, and the result enters the image as that stage's output
    if constexpr (CHAIN) {
        constexpr int NN = CT ? NC : 1280; // (CHAIN implies a compile-time length; the alternative only keeps the templates below well-formed)
        constexpr int QL = make_fac(NN).nf - 1;
        using SL = GStage<NN, QL>;
        const int row = tid >> (LTHR - FC.lrows), jt = tid & (SL::tpr - 1);
        const int grow = row0 + row;
        const int rowl = (row < rvalid) ? row : rvalid - 1, growl = row0 + rowl; // (load addresses of a partial tile's missing rows)
        const float sp = (PRE == XF_INV) ? -1.f : 1.f, sq = (POST == XF_INV) ? -1.f : 1.f;
        cf* __restrict__ rowp = cur + row * NN;
        cf xa[SL::NBT][SL::RX];
        cf xb[(MID == MID_MULPSI) ? SL::NBT : 1][(MID == MID_MULPSI) ? SL::RX : 1];
        float keep[(MID == MID_EXPIV_PAIR) ? SL::NBT : 1][(MID == MID_EXPIV_PAIR) ? SL::RX : 1];
        {
            // the rows are requested first (both operands of the product: the second one lands while the first is being
            // transformed), the twiddle table moves into LDS behind the requests (the first stage needs no twiddle)
            XArr<NN, 0> x0, xpre;
            if constexpr (MID == MID_MULPSI) {
                gs_load_global<NN, 0>(x0, in1 + (unsigned)rowl * pin, jt, (A.skip_dead_loads & 2) != 0, A.band);
                if constexpr (FDES_GEN_PREFETCH_B) gs_load_global<NN, 0>(xpre, in0 + (unsigned)rowl * pin, jt, (A.skip_dead_loads & 1) != 0, A.band);
            } else {
                gs_load_global<NN, 0>(x0, in0 + (unsigned)rowl * pin, jt, (A.skip_dead_loads & 1) != 0, A.band);
            }
            if (tw_lds) {
                const cf* __restrict__ tw = reinterpret_cast<const cf*>(A.tw0);
                cf* twd = glds + nimg * tile;
                for (int i = tid; i < (tw_mode == 2 ? N / 2 : N); i += THR) twd[i] = tw[i];
            }
            if constexpr (MID == MID_MULPSI) {
                gen_chain_to_regs<NN, 0, true, true>(cur, nullptr, false, 0, twl, sp, row, jt, xb, x0);
                if constexpr (FDES_GEN_PREFETCH_B) gen_chain_to_regs<NN, 0, true, true>(cur, nullptr, false, 0, twl, sp, row, jt, xa, xpre);
                else gen_chain_to_regs<NN, 0, true>(cur, in0 + (unsigned)rowl * pin, (A.skip_dead_loads & 1) != 0, A.band, twl, sp, row, jt, xa, x0);
            } else {
                gen_chain_to_regs<NN, 0, true, true>(cur, nullptr, false, 0, twl, sp, row, jt, xa, x0);
            }
        }
        const float md = (float)A.mindim;
#pragma unroll
        for (int b = 0; b < SL::NBT; b++) {
            const int j = jt + b * SL::tpr;
            if (j < SL::nb) {
#pragma unroll
                for (int i = 0; i < SL::RX; i++) {
                    const int p = rdx_slot<SL::RX>(i);
                    const int col = j + i * SL::Ns; // last stage: k = j
                    cf v = xa[b][p];
                    if constexpr (MID == MID_GTAB) v = v * gtab[(unsigned)rowl * pin + (unsigned)col];
                    if constexpr (MID == MID_MASK) v = gen_outside(iwc(col, NN), iwc(grow, A.nrows), md) ? cf{0.f, 0.f} : v * A.scale;
                    if constexpr (MID == MID_PTAB) {
                        const cf pr = reinterpret_cast<const cf*>(A.prow)[growl], pc = reinterpret_cast<const cf*>(A.pcol)[col];
                        v = gen_outside(iwc(col, NN), iwc(grow, A.nrows), md) ? cf{0.f, 0.f} : cmul3(v, cmul3(pr, pc));
                    }
                    if constexpr (MID == MID_MULPSI) v = cmul3(v, xb[b][p]); // f0 = t, f1 = psi
                    if constexpr (MID == MID_EXPIV_PAIR) {
                        keep[b][p] = v.y;
                        float sn, cs;
                        const float e = (A.scale == 0.f) ? 1.f : __expf(-(v.x * A.scale));
                        if (fabsf(v.x) <= kSincosFast) sincos_cw(v.x, sn, cs);
                        else sincos_wide(v.x, sn, cs);
                        v = cf{e * cs, e * sn};
                    }
                    xa[b][p] = v;
                }
            }
        }
        auto store_t = [&](cf* outp) { // transposed grid: consecutive threads write the R consecutive elements of one output row
            cf* __restrict__ dst = outp + row0;
            for (int e = tid; e < tile; e += THR) {
                const int c = e >> F.lrows, rr = e & (R - 1);
                if (rr >= rvalid) continue; // (partial last tile)
                if (A.skip_dead_stores && dead_index(iwc(c, N), A.band)) continue;
                dst[(unsigned)c * ldt + (unsigned)rr] = cur[rr * N + c];
            }
        };
        // the trailing transform runs the stages in reverse order: its first butterfly of thread (b, jt) takes exactly the elements
        // j + i N / RX that the leading transform's last stage left in xa[b] (output i in register rdx_slot(i)): no trip through LDS
        auto trailing = [&]() {
            if constexpr (FDES_GEN_REV && SL::RX != 16) { // (a first stage of radix 16 writes its outputs 32 dwords apart: 16-way bank conflicts - 1280^2 -6 %, 2560^2 -8 % measured)
            static_assert(GStage<NN, 0, true>::RX == SL::RX && GStage<NN, 0, true>::NBT == SL::NBT && GStage<NN, 0, true>::nb == SL::nb, "reversed plan");
            XArr<NN, 0, true> x0;
#pragma unroll
            for (int b = 0; b < SL::NBT; b++)
#pragma unroll
                for (int i = 0; i < SL::RX; i++) x0[b][i] = xa[b][rdx_slot<SL::RX>(i)];
            gen_rev_from_regs<NN>(cur, twl, sq, row, jt, x0);
            } else {
            gs_store_lds<NN, QL>(xa, rowp, jt);
            __syncthreads();
            gen_inplace_stages<NN, 0>(cur, twl, sq, row, jt);
            }
        };
        trailing();
        store_t(out0);
        if constexpr (MID == MID_EXPIV_PAIR) {
#pragma unroll
            for (int b = 0; b < SL::NBT; b++)
#pragma unroll
                for (int p = 0; p < SL::RX; p++) {
                    const float v = keep[b][p];
                    float sn, cs;
                    const float e = (A.scale == 0.f) ? 1.f : __expf(-(v * A.scale));
                    if (fabsf(v) <= kSincosFast) sincos_cw(v, sn, cs);
                    else sincos_wide(v, sn, cs);
                    xa[b][p] = cf{e * cs, e * sn};
                }
            __syncthreads(); // the first slice's tile has been read
            trailing();
            store_t(reinterpret_cast<cf*>(A.out2) + ((A.nbatch > 1) ? (size_t)bz * A.bstride_out2 : (size_t)0));
        }
        return;
    }

    float keep_f[(MID == MID_EXPIV_PAIR) ? EPT : 1]; // second slice's potential of a pair
    if constexpr (MID == MID_GTABN) {
        // sum over species in Fourier space (phaseGrating's species loop), then one inverse transform
        cf acc[EPT];
#pragma unroll
        for (int i = 0; i < EPT; i++) acc[i] = cf{0.f, 0.f};
        for (int z = 0; z < A.nspecies; z++) {
            const size_t zo = (size_t)z * A.species_stride;
            load_tile(in0 + zo, cur, false);
            __syncthreads();
            if constexpr (PRE != XF_NONE) gen_fft<CT, ONE, (CT ? NC : 320)>(cur, other, twl, F, PRE == XF_INV);
#pragma unroll
            for (int i = 0; i < EPT; i++)
                if (valid(i)) {
                    const float gv = gtab[zo + (unsigned)erl * pin + (unsigned)ec[i]];
                    const cf v = cur[er[i] * N + ec[i]];
                    acc[i].x += v.x * gv;
                    acc[i].y += v.y * gv;
                }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < EPT; i++)
            if (valid(i)) cur[er[i] * N + ec[i]] = acc[i];
        __syncthreads();
    } else {
        if constexpr (MID == MID_ATOMS) {
            // squareAtoms_d (src/crystalMaker.cu:73-123) from the (slice, species, row)-sorted records, as MID_ATOMS of
            // fft_lds.hip: tile zeroed in LDS, ONE wave adds the bilinear weights with LDS float atomics in sorted order
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
            if (phi[0] - plo[0] + phi[1] - plo[1] == 0) { // workgroup-uniform: the spectrum of an empty row group is zero
                if constexpr (STORE_T) {
                    for (int e = tid; e < tile; e += THR)
                        if ((e & (R - 1)) < rvalid) (out0 + row0)[(unsigned)(e >> F.lrows) * ldt + (unsigned)(e & (R - 1))] = cf{0.f, 0.f};
                }
                return;
            }
            for (int e = tid; e < tile; e += THR) cur[e] = cf{0.f, 0.f};
            __syncthreads();
            if (tid < 64) {
                float* ldsf = reinterpret_cast<float*>(cur);
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
                                if (rr >= 0 && rr < R && c >= 0 && c < N) atomicAdd(&ldsf[2 * (rr * N + c) + comp], wgt);
                            }
                        }
                    }
                }
            }
        } else if constexpr (MID != MID_MULPSI) {
            load_tile(in0, cur, (A.skip_dead_loads & 1) != 0);
        }
        cf keep_b[(MID == MID_MULPSI) ? EPT : 1]; // second operand of the product, transformed, in registers
        if constexpr (MID == MID_MULPSI) {
            // Two tile images serve both operands (a third one would leave room for ONE workgroup per CU: measured, the pass
            // then does not overlap with the other lanes' at all): the first operand is requested into registers, the
            // second one goes through the images, is transformed and parked in registers, then the first one moves in.
            cf areg[EPT];
            const bool band0 = (A.skip_dead_loads & 1) != 0;
#pragma unroll
            for (int i = 0; i < EPT; i++)
                if (valid(i)) {
                    const bool dead = band0 && dead_index(iwc(ec[i], N), A.band);
                    areg[i] = dead ? cf{0.f, 0.f} : in0[(unsigned)erl * pin + (unsigned)ec[i]];
                }
            load_tile(in1, cur, (A.skip_dead_loads & 2) != 0);
            __syncthreads();
            if constexpr (PRE != XF_NONE) gen_fft<CT, ONE, (CT ? NC : 320)>(cur, other, twl, F, PRE == XF_INV);
#pragma unroll
            for (int i = 0; i < EPT; i++)
                if (valid(i)) keep_b[i] = cur[er[i] * N + ec[i]];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < EPT; i++)
                if (valid(i)) cur[er[i] * N + ec[i]] = areg[i];
        }
        __syncthreads();
        if constexpr (PRE != XF_NONE) gen_fft<CT, ONE, (CT ? NC : 320)>(cur, other, twl, F, PRE == XF_INV);
        // ---- point-wise operation on this thread's elements
        const float md = (float)A.mindim;
#pragma unroll
        for (int i = 0; i < EPT; i++)
            if (valid(i)) {
                cf v = cur[er[i] * N + ec[i]];
                const int grow = row0 + er[i];
                if constexpr (MID == MID_SCALE) v = v * A.scale;
                if constexpr (MID == MID_GTAB) v = v * gtab[(unsigned)erl * pin + (unsigned)ec[i]];
                if constexpr (MID == MID_MASK) v = gen_outside(iwc(ec[i], N), iwc(grow, A.nrows), md) ? cf{0.f, 0.f} : v * A.scale;
                if constexpr (MID == MID_PTAB) {
                    const cf pr = reinterpret_cast<const cf*>(A.prow)[row0 + erl], pc = reinterpret_cast<const cf*>(A.pcol)[ec[i]];
                    v = gen_outside(iwc(ec[i], N), iwc(grow, A.nrows), md) ? cf{0.f, 0.f} : cmul3(v, cmul3(pr, pc));
                }
                if constexpr (MID == MID_MULPSI) v = cmul3(v, keep_b[i]); // f0 = t, f1 = psi
                if constexpr (MID == MID_EXPIV_PAIR) {
                    keep_f[i] = v.y;
                    float sn, cs;
                    const float e = (A.scale == 0.f) ? 1.f : __expf(-(v.x * A.scale));
                    if (fabsf(v.x) <= kSincosFast) sincos_cw(v.x, sn, cs);
                    else sincos_wide(v.x, sn, cs);
                    v = cf{e * cs, e * sn};
                }
                cur[er[i] * N + ec[i]] = v;
            }
        __syncthreads();
    }
    auto store_tile = [&](cf* outp) {
        if constexpr (!STORE_T) {
            const unsigned pout = A.pitch_out ? (unsigned)A.pitch_out : (unsigned)N;
            cf* __restrict__ on = outp + (size_t)row0 * pout;
#pragma unroll
            for (int i = 0; i < EPT; i++)
                if (valid(i) && er[i] < rvalid) on[(unsigned)er[i] * pout + (unsigned)ec[i]] = cur[er[i] * N + ec[i]];
        } else {
            // transposed grid: N rows of length ldt; consecutive threads write the R consecutive elements of one output row
            cf* __restrict__ dst = outp + row0;
            for (int e = tid; e < tile; e += THR) {
                const int c = e >> F.lrows, rr = e & (R - 1);
                if (rr >= rvalid) continue; // (partial last tile)
                if (A.skip_dead_stores && dead_index(iwc(c, N), A.band)) continue;
                dst[(unsigned)c * ldt + (unsigned)rr] = cur[rr * N + c];
            }
        }
    };
    if constexpr (POST != XF_NONE) gen_fft<CT, ONE, (CT ? NC : 320)>(cur, other, twl, F, POST == XF_INV);
    store_tile(out0);
    if constexpr (MID == MID_EXPIV_PAIR) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EPT; i++)
            if (valid(i)) {
                const float v = keep_f[i];
                float sn, cs;
                const float e = (A.scale == 0.f) ? 1.f : __expf(-(v * A.scale));
                if (fabsf(v) <= kSincosFast) sincos_cw(v, sn, cs);
                else sincos_wide(v, sn, cs);
                cur[er[i] * N + ec[i]] = cf{e * cs, e * sn};
            }
        __syncthreads();
        if constexpr (POST != XF_NONE) gen_fft<CT, ONE, (CT ? NC : 320)>(cur, other, twl, F, POST == XF_INV);
        store_tile(reinterpret_cast<cf*>(A.out2) + ((A.nbatch > 1) ? (size_t)bz * A.bstride_out2 : (size_t)0));
    }
}
// waves per SIMD the register allocation is held to: EPT <= 8 (rows up to 1024 points) and the one-image kernels: two workgroups
// per CU (the species loop of MID_GTABN needs more than 128 registers)
#define FDES_GPASS_BOUNDS(NC, EPT, MID) __launch_bounds__(fdes::gen_threads(NC), ((((EPT) <= 8 || ((NC) != 0 && fdes::gen_one_image(NC))) && (MID) != fdes::MID_GTABN && (NC) <= 4096) ? 4 : (fdes::gen_threads(NC) == 1024 ? 4 : 2))) // (rows beyond 4096 points: a tile of two rows is 80 ... 128 KiB, one workgroup per CU may use 256 registers)
template <int NC, int EPT, int PRE, int MID, int POST, bool STORE_T>
__global__ FDES_GPASS_BOUNDS(NC, EPT, MID) void k_gpass(PassArgs A, GenFac Frt)
{
    gpass_body<NC, EPT, PRE, MID, POST, STORE_T>(A, Frt);
}

#undef float2
#undef make_float2

#ifdef __HIPCC_RTC__
} // namespace
} // namespace fdes
// run-time compilation: the passes of ONE length, exported by name (gen_jit.cpp looks them up as fdes_jit_gpass_<pre>_<mid>_<post>_<t>)
#define FDES_JIT_KERNEL(P_, M_, Q_, S_)                                                                                          \
    extern "C" __global__ FDES_GPASS_BOUNDS(FDES_GEN_JIT_N, FDES_GEN_JIT_EPT, M_) void fdes_jit_gpass_##P_##_##M_##_##Q_##_##S_(     \
        fdes::PassArgs A, fdes::GenFac F)                                                                                        \
    {                                                                                                                            \
        fdes::gpass_body<FDES_GEN_JIT_N, FDES_GEN_JIT_EPT, P_, M_, Q_, (S_ != 0)>(A, F);                                         \
    }
FDES_JIT_KERNEL(0, 0, 0, 0)
FDES_JIT_KERNEL(0, 0, 0, 1)
FDES_JIT_KERNEL(0, 7, 0, 1)
FDES_JIT_KERNEL(1, 0, 0, 0)
FDES_JIT_KERNEL(2, 0, 0, 0)
FDES_JIT_KERNEL(2, 7, 0, 0)
FDES_JIT_KERNEL(1, 0, 0, 1)
FDES_JIT_KERNEL(2, 0, 0, 1)
FDES_JIT_KERNEL(1, 9, 0, 1)
FDES_JIT_KERNEL(2, 12, 1, 1)
FDES_JIT_KERNEL(1, 2, 2, 1)
FDES_JIT_KERNEL(1, 8, 2, 1)
FDES_JIT_KERNEL(1, 4, 2, 1)
FDES_JIT_KERNEL(2, 5, 1, 1)
FDES_JIT_KERNEL(0, 5, 1, 1)
FDES_JIT_KERNEL(1, 6, 2, 1)
#else

// launch geometry of a pass over the rows of `a` (shared by the compiled-in and the run-time-compiled kernels)
struct GLaunch {
    PassArgs w;
    dim3 grid;
    size_t lds_bytes = 0;
};
inline hipError_t gen_launch_geometry(const PassArgs& a, const GenFac& f, bool ct, int mid, GLaunch& L)
{
    // one or two images of the tile + the twiddle table
    const bool one = ct && gen_one_image(f.n);
    const int tw_mode = ct ? gen_tw_mode(f.n, f.rows, one) : (gen_tw_in_lds(f.n) ? 1 : 0);
    L.lds_bytes = sizeof(float) * 2 * ((size_t)f.rows * f.n * (one ? 1 : 2) + (tw_mode == 1 ? (size_t)f.n : (tw_mode == 2 ? (size_t)f.n / 2 : (size_t)0))) + 64;
    if (L.lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    const int all_groups = (a.nrows + f.rows - 1) / f.rows; // (the last tile may be partial)
    int groups = all_groups;
    L.w = a;
    if (a.live_rows_only) {
        if (a.band <= 0) return hipErrorInvalidValue;
        const int L_ = a.band_L;
        const int g_lo = L_ / f.rows + 1, g_hi = (a.nrows - L_) / f.rows;
        if (g_hi > g_lo) groups = g_lo + (all_groups - g_hi);
        else L.w.live_rows_only = 0;
    }
    const int ny = (mid == MID_ATOMS) ? (a.nspecies > 0 ? a.nspecies : 1) : 1;
    if (a.nbatch > 16) return hipErrorInvalidValue;
    const int nz = a.nbatch > 1 ? a.nbatch : 1;
    L.grid = dim3(groups, ny, nz);
    L.w.jit = nullptr; // host side only
    return hipSuccess;
}

template <int NC, int EPT, int PRE, int MID, int POST, bool ST> hipError_t glaunch(const PassArgs& a, const GenFac& f, hipStream_t st)
{
    static std::atomic<unsigned long long> attr_set{0};
    auto kern = k_gpass<NC, EPT, PRE, MID, POST, ST>;
    int dev = 0;
    {
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
    }
    if (dev < 0 || dev >= 64 || !((attr_set.load(std::memory_order_acquire) >> dev) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    GLaunch L;
    {
        hipError_t e = gen_launch_geometry(a, f, NC != 0, MID, L);
        if (e != hipSuccess) return e;
    }
    PassArgs& w = L.w;
    if (a.ev_start && a.ev_stop) {
        w.ev_start = w.ev_stop = nullptr;
        hipExtLaunchKernelGGL(kern, L.grid, dim3(gen_threads(NC)), L.lds_bytes, st, (hipEvent_t)a.ev_start, (hipEvent_t)a.ev_stop, 0, w, f);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, L.grid, dim3(gen_threads(NC)), L.lds_bytes, st, w, f);
    return hipGetLastError();
}

// the same pass on the kernels that hipRTC compiled for this length (gen_jit.cpp): k_gpass<n, ...> of the module
inline hipError_t glaunch_module(const GenJitKernels* k, int pre, int mid, int post, bool st_t, const PassArgs& a, const GenFac& f, hipStream_t st)
{
    hipFunction_t fn = reinterpret_cast<hipFunction_t>(gen_jit_function(k, pre, mid, post, st_t));
    if (!fn) return hipErrorInvalidValue;
    GLaunch L;
    {
        hipError_t e = gen_launch_geometry(a, f, true, mid, L);
        if (e != hipSuccess) return e;
    }
    PassArgs w = L.w;
    GenFac ff = f;
    void* params[] = {&w, &ff};
    if (a.ev_start && a.ev_stop) {
        w.ev_start = w.ev_stop = nullptr;
        return hipExtModuleLaunchKernel(fn, L.grid.x * (unsigned)k->threads, L.grid.y, L.grid.z, (unsigned)k->threads, 1, 1, L.lds_bytes, st, params, nullptr,
                                        (hipEvent_t)a.ev_start, (hipEvent_t)a.ev_stop, 0);
    }
    return hipModuleLaunchKernel(fn, L.grid.x, L.grid.y, L.grid.z, (unsigned)k->threads, 1, 1, (unsigned)L.lds_bytes, st, params, nullptr);
}

template <int NC, int EPT> hipError_t gdispatch(int pre, int mid, int post, bool st_t, const PassArgs& a, const GenFac& f, hipStream_t st)
{
#define CASE(P_, M_, Q_, S_) if (pre == P_ && mid == M_ && post == Q_ && st_t == S_) return glaunch<NC, EPT, P_, M_, Q_, S_>(a, f, st);
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

// lengths that exist as compile-time kernels only, i.e. compiled at plan creation (gen_jit.cpp): rows beyond 4096 points (one tile image of
// two rows; the run-time-length form needs two) and lengths with a factor 17, 19 or 23 (the run-time-length kernels stop at radix 13)
bool gen_pass_needs_compiled(int n) { return n > 4096 || n % 17 == 0 || n % 19 == 0 || n % 23 == 0; }

bool gen_pass_supported_len(int n)
{
    if (n > 4096) { // rows of 4098 ... 8192 points (round 5)
        GenFac f8;
        return n <= kGenMaxLen && n % 2 == 0 && gen_jit_available() && factorize(n, f8);
    }
    if (n < 256) return false;
    if ((n & (n - 1)) == 0) return false; // powers of two have kernels of their own
    if (gen_pass_needs_compiled(n) && !gen_jit_available()) return false;
    GenFac f;
    return factorize(n, f);
}
int gen_pass_rows(int n) { return gen_rows(n); }
// Rows per tile for n-point rows of a grid with `nrows` of them: the length's own number (8 up to 512 points, 4 up to 2048, 2
// beyond) or, where that does not divide nrows, the next smaller power of two that does - 500^2 runs four-row tiles, 750^2 (m = 2 nx
// of a .qsc with an odd nx) two-row tiles; where none does (an ODD nrows: 375^2, 1001^2) the length's own number with a PARTIAL last
// tile (k_gpass: rvalid).  Measured (profiles/r05_smaller_tiles.txt): exact smaller tiles run level with or ahead of partial larger
// ones (500^2 73.7 against 65.3 k, 2002^2 9.4 against 8.3 k), so the partial tile is the last resort.  FDES_GEN_PARTIAL_TILES = 1
// (measurement knob): always the length's own rows with a partial last tile; = 0: never a partial tile (odd nrows: rocFFT).
int gen_pass_tile_rows(int n, int nrows)
{
    int r0 = gen_rows(n);
    if (const char* e = std::getenv("FDES_GEN_TILE_ROWS")) { // measurement knob: smaller tiles where the default would do
        const int v = std::atoi(e);
        if ((v == 2 || v == 4) && v < r0) r0 = v;
    }
    const char* pt = std::getenv("FDES_GEN_PARTIAL_TILES");
    if (nrows < 1) return 0;
    if (pt && pt[0] == '1') return r0;
    for (int r = r0; r >= 2; r >>= 1)
        if (nrows % r == 0) return r;
    return (pt && pt[0] == '0') ? 0 : r0;
}
bool gen_pass_compiled_in(int n) { return FDES_GEN_SPECIALISED && gen_specialised(n); }
int gen_pass_threads(int n) { return gen_threads(n); }
// threads of a workgroup of kernels compiled at plan creation for (length, tile rows): two-row tiles of short rows keep 50 ... 250 of the
// 256 threads of a row busy and leave four rows per CU in flight: 256-thread workgroups (four per CU) up to 1280 points - measured
// (profiles/r05_smaller_tiles.txt): 750^2 +13 %, 1250^2 +24 %, 1430^2 +-0, 2002^2 -15 %
int gen_pass_threads_for(int n, int rows)
{
    if (const char* e = std::getenv("FDES_GEN_SMALL_TILE_THREADS")) { if (std::atoi(e) == 512) return gen_threads(n); } // (measurement knob)
    return (rows == 2 && n <= 1280) ? 256 : gen_threads(n);
}

void gen_pass_twiddles(int n, float* tw)
{
    const double w = -2.0 * 3.14159265358979323846 / (double)n;
    for (int k = 0; k < n; k++) {
        tw[2 * k] = (float)std::cos(w * (double)k);
        tw[2 * k + 1] = (float)std::sin(w * (double)k);
    }
}

hipError_t gen_pass(int n, int pre, int mid, int post, bool st_t, const PassArgs& a, hipStream_t st)
{
    GenFac f;
    if (!factorize(n, f) || n < 256 || n > kGenMaxLen) return hipErrorInvalidValue; // (gen_pass_supported_len was asked when the plan was made)
    f.rows = (a.tile_rows == 2 || a.tile_rows == 4 || a.tile_rows == 8) && a.tile_rows <= gen_rows(n) ? a.tile_rows : gen_rows(n);
    f.lrows = gen_lrows(f.rows);
    const int ept = (f.rows * n + kGenThreads - 1) / kGenThreads;
    if (a.jit) { // kernels compiled for this length (and these tile rows) at plan creation (gen_jit.cpp)
        const GenJitKernels* k = static_cast<const GenJitKernels*>(a.jit);
        int dev = -1;
        if (k->n == n && k->rows == f.rows && hipGetDevice(&dev) == hipSuccess && dev == k->device) return glaunch_module(k, pre, mid, post, st_t, a, f, st);
    }
    if (FDES_GEN_SPECIALISED && f.rows == gen_rows(n)) { // (the compiled-in kernels have the length's own tile rows) // the grids the reference ships (bin/dataFDES.cnf, bin/test.qsc, Si_001_11k_cnf) and a few round ones
        if (n == 320) return gdispatch<320, 8>(pre, mid, post, st_t, a, f, st);
        if (n == 800) return gdispatch<800, 8>(pre, mid, post, st_t, a, f, st);
        if (n == 1000) return gdispatch<1000, 8>(pre, mid, post, st_t, a, f, st);
        if (n == 400) return gdispatch<400, 8>(pre, mid, post, st_t, a, f, st);
        if (n == 500) return gdispatch<500, 8>(pre, mid, post, st_t, a, f, st);
        if (n == 640) return gdispatch<640, 8>(pre, mid, post, st_t, a, f, st);
        if (n == 1280) return gdispatch<1280, 16>(pre, mid, post, st_t, a, f, st);
        if (n == 1600) return gdispatch<1600, 16>(pre, mid, post, st_t, a, f, st);
        if (n == 2000) return gdispatch<2000, 16>(pre, mid, post, st_t, a, f, st);
        // m = 2 nx of a .qsc with nx = 1280, 1500, 1536, 1600, 1800, 2000 (src/rwQsc.cu:943-948): two-row tiles
        if (n == 2560) return gdispatch<2560, (FDES_GEN_ROWS4 ? 32 : 16)>(pre, mid, post, st_t, a, f, st);
        if (n == 3000) return gdispatch<3000, (FDES_GEN_ROWS4 ? 32 : 16)>(pre, mid, post, st_t, a, f, st);
        if (n == 3072) return gdispatch<3072, (FDES_GEN_ROWS4 ? 32 : 16)>(pre, mid, post, st_t, a, f, st);
        if (n == 3200) return gdispatch<3200, (FDES_GEN_ROWS4 ? 32 : 16)>(pre, mid, post, st_t, a, f, st);
        if (n == 3600) return gdispatch<3600, (FDES_GEN_ROWS4 ? 32 : 16)>(pre, mid, post, st_t, a, f, st);
        if (n == 4000) return gdispatch<4000, (FDES_GEN_ROWS4 ? 32 : 16)>(pre, mid, post, st_t, a, f, st);
    }
    if (gen_pass_needs_compiled(n)) return hipErrorInvalidValue; // (no run-time-length form: Fft2D::create takes rocFFT when the compilation fails)
    if (ept <= 8) return gdispatch<0, 8>(pre, mid, post, st_t, a, f, st);
    if (ept <= 16) return gdispatch<0, 16>(pre, mid, post, st_t, a, f, st);
    return hipErrorInvalidValue;
}

} // namespace fdes
#endif // __HIPCC_RTC__
