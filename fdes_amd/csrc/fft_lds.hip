// fft_lds.hip — LDS row-FFT pass kernels for gfx950 (see fft_lds.h).
//
// Row FFT of length N = 16 * 16 * R3 (R3 = N/256 in {1,2,4,8,16}), T = N/16 threads per row, each
// thread holding 16 complex values in registers (Stockham autosort, decimation in frequency):
//   stage 0: thread t loads x[t + T*l], radix-16 butterfly, twiddle W_N^(t*k), LDS write at 16t+k
//   stage 1: reads x[t + T*l] again (autosort keeps the read pattern), radix-16, twiddle
//            W_(N/16)^(p*k) (p = t/16), LDS write at q + 256p + 16k (q = t%16)
//   stage 2: reads x[t + T*l], 16/R3 radix-R3 butterflies over registers {i + (16/R3) j}: no
//            twiddle, and the outputs are thread-local and in natural order: register l of thread
//            t is X[t + T*l] - exactly the input pattern of stage 0, so two transforms chain
//            through registers (FFT -> point-wise -> inverse FFT without touching LDS in between).
// LDS rows are padded by one element per 16 (index i -> i + i/16): the stride-16 writes of stage 0
// become stride 17 (conflict-free for ds_write_b64's 16-lane groups), reads stay contiguous.
// Workgroup geometries (template parameter WG, struct WGeo below): 256 threads x 2 rows per thread = 8192/N rows per
// workgroup (4 rows at N = 2048, 68 KiB of LDS: TWO workgroups per CU; the default above 1024 points), 512 threads x 2
// rows per thread for 4096-point rows (4 rows, 136 KiB, one workgroup per CU: 256 threads would cut the transposed-store
// segments to 16 bytes), and N/4 threads x ONE row per thread (four rows per workgroup) for grids up to 1024^2, where a
// pass lasts as long as its slowest workgroup.  The transposed store stages the R x N tile through the
// same LDS (swizzled so both the row-wise write and the column-wise read are conflict-free) and
// writes R contiguous elements per output row; blockIdx is remapped so that workgroups sharing an
// XCD (blockIdx % 8 equal) own consecutive row groups and their partial 128-byte lines merge in
// that XCD's L2.
#include "fft_lds.h"
#include "geometry.h"

#include <atomic>
#include <cmath>
#include <type_traits>

namespace fdes {

namespace {

#ifndef FDES_EXP_NOVALU
#define FDES_EXP_NOVALU 0 // timing experiments only (results are garbage): transforms without arithmetic / without LDS exchanges
#endif
#ifndef FDES_EXP_NOLDS
#define FDES_EXP_NOLDS 0
#endif

#include "fft_dev.inc"

// Workgroup geometries (template parameter WG): even -> WG threads x 2 rows per thread (256: default up to 2048
// points, 512: 4096 points); odd -> (WG - 1) threads x 1 row per thread, instantiated as N / 4 + 1 only, i.e. four
// rows per workgroup (the shortest dependency chain per workgroup and the most workgroups: grids of 1024^2 and below
// are bound by the latency of one workgroup, not by throughput; at 2048^2 it is 10 % slower than 256 x 2).
template <int WG> struct WGeo {
    static constexpr int THR = (WG & 1) ? WG - 1 : WG;
    static constexpr int NRV = (WG & 1) ? 1 : 2;
};

// WG threads per workgroup (512: one workgroup per CU; 256: two per CU, 2 waves per SIMD either way ->
// 256 VGPRs per lane), each thread owning WGeo<WG>::NRV rows x 16 elements.
template <int N, int WG> struct Geo {
    static constexpr int T = N / 16;                 // threads per row
    static constexpr int R = WGeo<WG>::THR * WGeo<WG>::NRV * 16 / N;       // rows per workgroup
    static constexpr int RH = R / WGeo<WG>::NRV;                // rows per "half": thread (r, t) owns rows r and r + RH
    static constexpr int R3 = N / 256;               // radix of the last stage
    static constexpr int G = (R3 >= 1) ? 16 / R3 : 16;
    static constexpr int LDROW = N + N / 16;         // padded row length in float2
    static constexpr int SH = (R >= 16) ? 0 : ((R == 8) ? 1 : ((R == 4) ? 2 : 3)); // transposed-tile swizzle shift
};

// Synchronisation among the waves that share a row (T/64 of them; a row pair never leaves its waves):
// rows of <= 1024 points live in ONE wave, whose LDS instructions execute in order, so a compiler fence
// suffices; longer rows use a counting barrier on an LDS word private to the group.  Unlike
// __syncthreads() this lets the row groups of a workgroup drift apart: one group transforms while
// another still waits for its global loads.
// Measured (2048^2, MI355X): the counting barrier is correct but not faster than s_barrier (the row groups of
// a workgroup contend for the same LDS/VALU) and costs registers; concurrency comes from a second stream.
constexpr bool kGroupSpin = false;
struct GroupSync {
    unsigned* cnt;   // LDS counter of this wave group
    unsigned expect; // arrivals expected so far
};
template <int NW> __device__ __forceinline__ void group_sync(GroupSync& g)
{
    if constexpr (NW <= 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        asm volatile("" ::: "memory");
    } else if constexpr (!kGroupSpin) {
        __syncthreads();
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's LDS writes have landed
        if ((threadIdx.x & 63) == 0) atomicAdd(g.cnt, 1u);
        g.expect += NW;
        while ((int)(*(volatile unsigned*)g.cnt - g.expect) < 0) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    }
}

// WGeo<WG>::NRV row FFTs at once, data in a[h][16] (a[h][l] = x_h[t + T l] on entry, X_h[t + T l] on exit).
// Stage twiddles either sit in registers for the whole pass (TWR, 60 VGPRs: pays where the kernel stays under 256)
// or are fetched from the L1/L2-resident tables at the point of use.
struct Tw {
    const float2* g0;
    const float2* g1;
    float2 r0[16], r1[16];
    float2 b0[6], b1[6]; // TWM_POW: w^1, w^2, w^3, w^4, w^8, w^12 of each stage
};
// Twiddle modes of a pass: fetched from the tables at the point of use (no registers, 30 loads per transform in the
// middle of the arithmetic), all 30 in registers for the whole pass (60 VGPRs: only where the kernel stays under 256
// without spilling), or six powers of each stage in registers (w^1, w^2, w^3, w^4, w^8, w^12: 24 VGPRs) and the other
// nine as ONE complex product each per transform (+36 packed instructions, no load).  One product, not a chain: the
// twiddle errors are the same in every transform of every slice, so they add up coherently over a slice loop and
// dominate the float32 error of an exit wave (2048^2 x 256 slices against the float64 oracle: 2.0e-5 with table
// twiddles, 3.8e-5 with chains of up to three products from w^1, w^2, w^4, w^8).  The products are rebuilt in every
// transform: an empty asm makes the base values opaque, otherwise the compiler computes all powers once and keeps
// them alive across the pass (which is the 60-register mode again).
enum { TWM_FETCH = 0, TWM_REGS = 1, TWM_POW = 2 };
__device__ __forceinline__ void tw_opaque(cf& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void tw_powers(const cf (&b)[6], cf (&w)[16])
{
    cf lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[3], b[4], b[5]}; // w^1..w^3, w^4, w^8, w^12
#pragma unroll
    for (int i = 0; i < 3; i++) { tw_opaque(lo[i]); tw_opaque(hi[i]); }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        w[1 + i] = lo[i];
        w[4 * (i + 1)] = hi[i];
    }
#pragma unroll
    for (int h = 0; h < 3; h++)
#pragma unroll
        for (int l = 0; l < 3; l++) w[4 * (h + 1) + l + 1] = cmul_rt(hi[h], lo[l]);
}
template <int N, int WG, bool INV, bool WAR0, int TWR>
__device__ __forceinline__ void row_fft(float2 (&a)[WGeo<WG>::NRV][16], float2* __restrict__ lds, const int r, const int t, const Tw& tw,
                                        GroupSync& gs)
{
    using G_ = Geo<N, WG>;
    constexpr int T = G_::T;
    constexpr int NW = (T + 63) / 64;
    const int rd0 = t + (t >> 4); // read base: padi(t)
    // ---- stage 0
#if !FDES_EXP_NOVALU
#pragma unroll
    for (int h = 0; h < WGeo<WG>::NRV; h++) r16<INV>(a[h]);
#endif
#if !FDES_EXP_NOVALU
    {
        float2 wp[16];
        if constexpr (TWR == TWM_POW) tw_powers(tw.b0, wp);
#pragma unroll
        for (int k = 1; k < 16; k++) {
            const float2 w = TWR == TWM_POW ? wp[k] : (TWR == TWM_REGS ? tw.r0[k] : tw.g0[k * T + t]);
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++) a[h][k] = twmul_rt<INV>(a[h][k], w);
        }
    }
#endif
#if !FDES_EXP_NOLDS
    if (WAR0) group_sync<NW>(gs);
#pragma unroll
    for (int h = 0; h < WGeo<WG>::NRV; h++) {
        float2* row = lds + (r + h * G_::RH) * G_::LDROW;
#pragma unroll
        for (int k = 0; k < 16; k++) row[17 * t + k] = a[h][k]; // padi(16 t + k)
    }
    group_sync<NW>(gs);
#pragma unroll
    for (int h = 0; h < WGeo<WG>::NRV; h++) {
        const float2* row = lds + (r + h * G_::RH) * G_::LDROW;
#pragma unroll
        for (int l = 0; l < 16; l++) a[h][l] = row[rd0 + (T + T / 16) * l]; // padi(t + T l): T is a multiple of 16
    }
#endif
    // ---- stage 1
#if !FDES_EXP_NOVALU
#pragma unroll
    for (int h = 0; h < WGeo<WG>::NRV; h++) r16<INV>(a[h]);
#endif
    if constexpr (G_::R3 > 1) {
        const int q = t & 15, p = t >> 4;
#if !FDES_EXP_NOVALU
        float2 wp[16];
        if constexpr (TWR == TWM_POW) tw_powers(tw.b1, wp);
#pragma unroll
        for (int k = 1; k < 16; k++) {
            const float2 w = TWR == TWM_POW ? wp[k] : (TWR == TWM_REGS ? tw.r1[k] : tw.g1[k * (T / 16) + p]);
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++) a[h][k] = twmul_rt<INV>(a[h][k], w);
        }
#endif
#if !FDES_EXP_NOLDS
        group_sync<NW>(gs);
#pragma unroll
        for (int h = 0; h < WGeo<WG>::NRV; h++) {
            float2* row = lds + (r + h * G_::RH) * G_::LDROW;
#pragma unroll
            for (int k = 0; k < 16; k++) row[q + 272 * p + 17 * k] = a[h][k]; // padi(q + 256 p + 16 k), q < 16
        }
        group_sync<NW>(gs);
#pragma unroll
        for (int h = 0; h < WGeo<WG>::NRV; h++) {
            const float2* row = lds + (r + h * G_::RH) * G_::LDROW;
#pragma unroll
            for (int l = 0; l < 16; l++) a[h][l] = row[rd0 + (T + T / 16) * l]; // padi(t + T l): T is a multiple of 16
        }
#endif
        // ---- stage 2: G butterflies of radix R3 over registers {i + G j}
#if !FDES_EXP_NOVALU
#pragma unroll
        for (int h = 0; h < WGeo<WG>::NRV; h++) {
            if constexpr (G_::R3 == 2) {
#pragma unroll
                for (int i = 0; i < 8; i++) r2<INV>(a[h][i], a[h][i + 8]);
            } else if constexpr (G_::R3 == 4) {
#pragma unroll
                for (int i = 0; i < 4; i++) r4<INV>(a[h][i], a[h][i + 4], a[h][i + 8], a[h][i + 12]);
            } else if constexpr (G_::R3 == 8) {
#pragma unroll
                for (int i = 0; i < 2; i++)
                    r8<INV>(a[h][i], a[h][i + 2], a[h][i + 4], a[h][i + 6], a[h][i + 8], a[h][i + 10], a[h][i + 12], a[h][i + 14]);
            } else {
                r16<INV>(a[h]);
            }
        }
#endif
    }
}

template <int N, int WG, int XF, bool WAR0, int TWR>
__device__ __forceinline__ void xform(float2 (&a)[WGeo<WG>::NRV][16], float2* lds, int r, int t, const Tw& tw, GroupSync& gs)
{
    if constexpr (XF == XF_FWD) row_fft<N, WG, false, WAR0, TWR>(a, lds, r, t, tw, gs);
    if constexpr (XF == XF_INV) row_fft<N, WG, true, WAR0, TWR>(a, lds, r, t, tw, gs);
}

// Row loads, a[h][l] = src[row h][t + T l].  BAND: the columns beyond the 2/3 band limit of a SQUARE grid
// (|iwc(c)| > N / 3, the largest i with 9 i^2 <= N^2) count as zero and are not fetched.  Columns come in 16
// blocks of T, so after unrolling each (h, l) is one of three compile-time cases: the block is live (plain load),
// dead (no load at all) or straddles the limit (two of the 16: a dead lane re-reads element t, a line that is
// fetched anyway, and is zeroed by a select).  lds_pass() clears the flag when the band limit is not N / 3.
#ifndef FDES_WAVES
#define FDES_WAVES 2 // waves per SIMD the 256/512-thread geometries are compiled for
#endif
// two-operand passes need the registers of two waves per SIMD
constexpr int pass_waves(int mid) { return (mid == MID_MULPSI || mid == MID_GTABN) ? 2 : FDES_WAVES; }
#ifndef FDES_PAIR_TWR
#define FDES_PAIR_TWR 0 // register twiddles in the two-slice transmission pass: 16 spilled registers at 2048; measured 38.3 / 30.7 us (one / two streams) against 37.2 / 28.7 us with fetched twiddles
#endif
#ifndef FDES_TWPOW
#define FDES_TWPOW 3 // 0 never, 1 passes without room for 60 twiddle registers, 2 every two-rows-per-thread pass, 3 every pass; measured at 2048^2 (two lanes): 0 -> 13.1k, 1 -> 13.25k, 2 -> 14.3k slice-propagations/s; 3 vs 2 at 1024^2 (C4, three lanes): 30.6k vs 28.0k
#endif
#ifndef FDES_NO_TWR
#define FDES_NO_TWR 0 // A/B switch: 1 = every pass fetches its stage twiddles at the point of use
#endif
#ifndef FDES_PTAB_TWR
#define FDES_PTAB_TWR 0 // register twiddles in the propagator pass: 13 spilled registers at 2048, measured slower (21.9 vs 19.5 us)
#endif
#ifndef FDES_PSEP_LATE
#define FDES_PSEP_LATE 1
#endif
#ifndef FDES_P5_PREFETCH
#define FDES_P5_PREFETCH 0 // requesting the second operand with the first: measured, no gain (A/B 12.1k vs 12.1k), 14 more VGPRs
#endif
template <int N, int WG>
__device__ __forceinline__ void load_rows(float2 (&a)[WGeo<WG>::NRV][16], const float2* __restrict__ src,
                                          const unsigned (&rbase)[WGeo<WG>::NRV], const int t, const bool band)
{
    constexpr int T = N / 16, LB = N / 3;
    // `band` is uniform.  The straddling blocks are common code for both settings (the flag only feeds the select),
    // the dead blocks are loads under a uniform branch: written as two whole variants under one `if`, the compiler
    // waited for the straddling loads inside the branch, one HBM round trip before the other 28 loads were issued.
    // Addresses: uniform base (scalar registers) + a 32-bit byte offset per thread + an immediate.  Written as element
    // indices the compiler widens every index to 64 bits (one v_add_u32 and one v_lshl_add_u64 per load); as byte offsets
    // one 32-bit add serves the GS loads whose immediates fit the instruction's 12 bits.
    constexpr int GS = 4095 / (T * 8) + 1;
    const char* __restrict__ sb = reinterpret_cast<const char*>(src);
    auto at = [&](unsigned off, int imm) -> float2 { return row_load<N>(reinterpret_cast<const float2*>(sb + off + imm)); }; // (non-temporal from FDES_NT_MIN points on: fft_dev.inc)
#pragma unroll
    for (int h = 0; h < WGeo<WG>::NRV; h++) {
        const unsigned b0 = (rbase[h] + (unsigned)t) * 8u;
#pragma unroll
        for (int l = 0; l < 16; l++) {
            const int lo = T * l, hi = T * l + T - 1;
            const int cls = (hi <= LB || lo >= N - LB) ? 0 : ((lo > LB && hi < N - LB) ? 1 : 2);
            const unsigned bj = b0 + (unsigned)((l / GS) * GS * T * 8);
            const int imm = (l % GS) * T * 8;
            if (cls == 0) a[h][l] = at(bj, imm);
            else if (cls == 1) {
                a[h][l] = make_float2(0.f, 0.f);
                if (!band) a[h][l] = at(bj, imm);
            } else {
                const int c = t + T * l;
                const bool dd = band && c > LB && c < N - LB;
                const float2 v = at(dd ? b0 : bj + (unsigned)imm, 0);
                a[h][l] = dd ? make_float2(0.f, 0.f) : v;
            }
        }
    }
}

// In-kernel stamps (diagnostic build only, -DFDES_STAMPS; no stamp executes in the product): lane 0 of every wave
// writes the shader clock at phase boundaries to A.dbg[(block * waves + wave) * 16 + slot]; WAITV drains the wave's
// vector-memory queue first so that the stamp separates "requested" from "landed".
#ifdef FDES_STAMPS
#define STAMP(slot, WAITV)                                                                                               \
    do {                                                                                                                 \
        if (A.dbg) {                                                                                                     \
            if (WAITV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            const unsigned long long t_ = (slot) == 15 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
            if ((threadIdx.x & 63) == 0) A.dbg[((size_t)blockIdx.x * (WGeo<WG>::THR / 64) + threadIdx.x / 64) * 16 + (slot)] = t_; \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
        }                                                                                                                \
    } while (0)
#else
#define STAMP(slot, WAITV) do { } while (0)
#endif

// The body of a pass for the workgroup blockIdx.x + A.vb0 of A.nvirt.
template <int N, int WG, int PRE, int MID, int POST, bool STORE_T>
__device__ __forceinline__ void pass_body(const PassArgs& A, float2* __restrict__ lds)
{
    using G_ = Geo<N, WG>;
    constexpr int T = G_::T, R = G_::R, RH = G_::RH;
    STAMP(0, false);
    STAMP(15, false); // wall clock (100 MHz) of the same instant
    constexpr bool TWREG = !FDES_NO_TWR && (pass_waves(MID) == 2) && (WGeo<WG>::NRV == 2) && (MID != MID_MULPSI && (MID != MID_PTAB || FDES_PTAB_TWR) && MID != MID_GTABN && (MID != MID_EXPIV_PAIR || (FDES_PAIR_TWR && N <= 2048)));
    // FDES_TWPOW: 0 = never, 1 = in the passes that cannot afford the 60 twiddle registers, 2 = in every two-rows-per-thread pass
    constexpr int TWR = ((FDES_TWPOW == 2 && WGeo<WG>::NRV == 2) || FDES_TWPOW == 3) ? TWM_POW : (TWREG ? TWM_REGS : ((FDES_TWPOW == 1 && WGeo<WG>::NRV == 2) ? TWM_POW : TWM_FETCH));
    Tw tw;
    tw.g0 = reinterpret_cast<const float2*>(A.tw0);
    tw.g1 = reinterpret_cast<const float2*>(A.tw1);
    // per-group barrier words live behind the row buffers
    constexpr int NWG = (T + 63) / 64;
    unsigned* cnts = reinterpret_cast<unsigned*>(lds + (size_t)G_::LDROW * R);
    GroupSync gs;
    gs.cnt = cnts + (NWG > 1 ? threadIdx.x / (64 * NWG) : 0);
    gs.expect = 0;
    if constexpr (NWG > 1) {
        if (threadIdx.x < WGeo<WG>::THR / 64) cnts[threadIdx.x] = 0;
        __syncthreads();
    }
    const int tid = threadIdx.x;
    const int r = tid / T, t = tid % T;
    // A pass may be launched in parts (A.nvirt > 0: this launch covers the row groups vb0 .. vb0 + gridDim.x - 1 of
    // nvirt; vb0 is a multiple of 8, so vb % 8 = blockIdx.x % 8): a part takes only a fraction of the chip's
    // workgroup slots and leaves the rest of every CU to the kernel of another lane.  (An in-kernel loop over row
    // groups was measured first: the loop alone costs the register allocator 40-200 spilled registers.)
    const int nvirt = A.nvirt > 0 ? A.nvirt : (int)gridDim.x;
    const int vb = (int)blockIdx.x + A.vb0;
    if (vb >= nvirt) return;
    // Staggered start (PassArgs::stagger, units of 64 cycles; 0 = off): where two workgroups share a CU and a pass runs
    // for several generations (4096-point rows with 256 threads), workgroups that start together stay in lockstep - both
    // load, then both transform, then both store.  The second workgroup of a CU (taken as (vb / ncu) odd: workgroups are
    // dealt breadth first; for speed only) starts late by about half a workgroup's life, once; later generations inherit
    // the offset because a slot is refilled when its workgroup retires.
    if (A.stagger > 0 && A.ncu > 0 && ((vb / A.ncu) & 1) && vb < 2 * A.ncu) {
        for (int i = 0; i < A.stagger; i++) __builtin_amdgcn_s_sleep(1);
    }
    // stagger < 0 (round 5): the CUs of the CHIP in different phases.  All workgroups of a pass start together, so all CUs
    // request their rows together (the memory system at its limit), then all transform (memory idle), then all store: the
    // first generation on the odd CUs starts -stagger x 64 cycles late, later generations inherit the offset.
    if (A.stagger < 0 && A.ncu > 0 && vb < A.ncu * (WGeo<WG>::THR == 512 ? 1 : 2)) cu_class_delay(-A.stagger);
    // this thread's stage twiddles (they serve every transform of the pass and both rows); issued first so that their
    // latency hides behind the row loads.  Per row group: hoisted out of the walk loop they would stay live across the
    // whole body and cost more registers than the reload does time.
    if constexpr (TWR == TWM_REGS && (PRE != XF_NONE || POST != XF_NONE)) {
#pragma unroll
        for (int k = 1; k < 16; k++) tw.r0[k] = tw.g0[k * T + t];
        if constexpr (G_::R3 > 1) {
#pragma unroll
            for (int k = 1; k < 16; k++) tw.r1[k] = tw.g1[k * (T / 16) + (t >> 4)];
        }
    }
    if constexpr (TWR == TWM_POW && (PRE != XF_NONE || POST != XF_NONE)) {
        constexpr int kBase[6] = {1, 2, 3, 4, 8, 12};
#pragma unroll
        for (int j = 0; j < 6; j++) tw.b0[j] = tw.g0[kBase[j] * T + t];
        if constexpr (G_::R3 > 1) {
#pragma unroll
            for (int j = 0; j < 6; j++) tw.b1[j] = tw.g1[kBase[j] * (T / 16) + (t >> 4)];
        }
    }
    // XCD-aware remap: blocks with equal blockIdx % 8 share an XCD; give them consecutive row groups
    int bg;
    {   // bijective for any grid size: XCD x (= vb % 8) owns a contiguous run of q or q + 1 groups
        const int nwg = nvirt, q = nwg >> 3, rem = nwg & 7, xcd = vb & 7, k = vb >> 3;
        bg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    }
    if (A.live_rows_only) {
        // live rows are [0, L] and [nrows - L, nrows): the launch covers g_lo low groups, then the groups from g_hi on
        const int L = A.band_L;
        const int g_lo = L / R + 1, g_hi = (A.nrows - L) / R;
        if (bg >= g_lo) bg = g_hi + (bg - g_lo);
    }
    const int row0 = bg * R;
    // uniform row-group base (scalar registers) + 32-bit per-thread element offsets (< R * pitch)
    const unsigned pin = A.pitch_in ? (unsigned)A.pitch_in : (unsigned)N;
    const unsigned ldt = A.pitch_out ? (unsigned)A.pitch_out : (unsigned)A.nrows; // leading dimension of a transposed output
    const size_t gbase = (size_t)row0 * pin;
    unsigned rbase[WGeo<WG>::NRV];
    int grow[WGeo<WG>::NRV];
#pragma unroll
    for (int h = 0; h < WGeo<WG>::NRV; h++) {
        grow[h] = row0 + r + h * RH;
        rbase[h] = (unsigned)(r + h * RH) * pin;
    }
    // batch of grids in one launch (grid.z, PassArgs::nbatch): this workgroup's grid
    const int bz = (int)blockIdx.z;
    const size_t zoff_in = (A.nbatch > 1) ? (size_t)(A.use_zin ? A.zin[bz] : bz) * A.bstride_in0 : (size_t)0;
    const size_t zoff_out = (A.nbatch > 1) ? (size_t)bz * A.bstride_out : (size_t)0;
    const float2* __restrict__ in0 = A.in0 ? reinterpret_cast<const float2*>(A.in0) + gbase + zoff_in : nullptr;
    const float2* __restrict__ in1 = A.in1 ? reinterpret_cast<const float2*>(A.in1) + gbase + ((A.nbatch > 1) ? (size_t)bz * A.bstride_in1 : (size_t)0) : nullptr;
    const float* __restrict__ gtab = A.gtab ? A.gtab + gbase : nullptr;
    float2* __restrict__ zsrc = A.zsrc ? reinterpret_cast<float2*>(A.zsrc) + gbase : nullptr;
    float2* const out0 = reinterpret_cast<float2*>(A.out) + zoff_out + ((MID == MID_ATOMS) ? (size_t)blockIdx.y * A.species_stride : (size_t)0); // MID_ATOMS: one launch covers every species

    float2 a[WGeo<WG>::NRV][16];
    float2 b[(MID == MID_MULPSI) ? WGeo<WG>::NRV : 1][16]; // second operand
    float vim[(MID == MID_EXPIV_PAIR) ? WGeo<WG>::NRV : 1][16]; // potential of the second slice of a pair
    if constexpr (MID == MID_GTABN) {
        // sum over species in Fourier space, then one inverse transform (phaseGrating's species loop)
        float2 acc[WGeo<WG>::NRV][16];
#pragma unroll
        for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
            for (int l = 0; l < 16; l++) acc[h][l] = make_float2(0.f, 0.f);
        for (int z = 0; z < A.nspecies; z++) {
            const size_t zo = (size_t)z * A.species_stride;
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
                for (int l = 0; l < 16; l++) a[h][l] = in0[zo + rbase[h] + t + T * l];
            xform<N, WG, PRE, true, TWR>(a, lds, r, t, tw, gs);
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
                for (int l = 0; l < 16; l++) {
                    const float gv = gtab[zo + rbase[h] + t + T * l];
                    acc[h][l].x += a[h][l].x * gv;
                    acc[h][l].y += a[h][l].y * gv;
                }
        }
#pragma unroll
        for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
            for (int l = 0; l < 16; l++) a[h][l] = acc[h][l];
    } else {
        if constexpr (MID == MID_ATOMS) {
            // squareAtoms_d (src/crystalMaker.cu:73-123) without a deposit grid: the few atoms whose bilinear
            // footprint touches this row group are read from the (slice, species, row)-sorted records.
            const AtomRec* __restrict__ recs = reinterpret_cast<const AtomRec*>(A.recs) + ((A.nbatch > 1) ? (size_t)bz * A.bstride_recs : (size_t)0);
            // candidate ranges of both components (4 independent loads), then the records are staged through LDS
            // (free before the first exchange) so that the walk below is not a chain of dependent global loads
            const int rlo = row0 > 0 ? row0 - 1 : 0;
            const int rhi = (row0 + R + 1 < A.nrows) ? row0 + R + 1 : A.nrows;
            int plo[2] = {0, 0}, phi[2] = {0, 0};
#pragma unroll
            for (int comp = 0; comp < 2; comp++) {
                const int qb = (A.nbatch > 1) ? (comp ? A.zq1[bz] : A.zq0[bz]) : (comp ? A.q1 : A.q0);
                const int q = qb < 0 ? -1 : qb + (int)blockIdx.y; // blockIdx.y = species
                if (q >= 0) {
                    const int* __restrict__ rs = A.rowstart + ((A.nbatch > 1) ? (size_t)bz * A.bstride_rowstart : (size_t)0) + (size_t)q * (size_t)(A.nrows + 1);
                    plo[comp] = rs[rlo];
                    phi[comp] = rs[rhi];
                }
            }
            if (phi[0] - plo[0] + phi[1] - plo[1] == 0) {
                // empty row group: its spectrum is zero
                if constexpr (STORE_T) {
#pragma unroll
                    for (int it = 0; it < WGeo<WG>::NRV * 16; it++) {
                        const int e = it * WGeo<WG>::THR + tid;
                        (out0 + row0)[(unsigned)(e / R) * ldt + (unsigned)(e & (R - 1))] = make_float2(0.f, 0.f);
                    }
                }
                return;
            }
            // The row tile is zeroed in LDS, ONE wave adds the bilinear weights with LDS float atomics (lane = atom,
            // 64 at a time, in the sorted order of the records), and every thread then picks up its own 32 pixels.
            // One wave only: instructions of a wave reach the LDS in program order and colliding lanes of one
            // instruction are serialised by the hardware in a fixed order, so the sums do not depend on timing.
            {
                const int rd0 = t + (t >> 4);
#pragma unroll
                for (int h = 0; h < WGeo<WG>::NRV; h++) {
                    float2* row = lds + (r + h * RH) * G_::LDROW;
#pragma unroll
                    for (int l = 0; l < 16; l++) row[rd0 + (T + T / 16) * l] = make_float2(0.f, 0.f);
                }
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
                                    const float w = ((px == 2 || px == 3) ? a1 : (1 - a1)) * ((px == 1 || px == 2) ? a2 : (1 - a2)) * ar.occ;
                                    if (rr >= 0 && rr < R && c >= 0 && c < N) atomicAdd(&ldsf[2 * (rr * G_::LDROW + c + (c >> 4)) + comp], w);
                                }
                            }
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int h = 0; h < WGeo<WG>::NRV; h++) {
                    const float2* row = lds + (r + h * RH) * G_::LDROW;
#pragma unroll
                    for (int l = 0; l < 16; l++) a[h][l] = row[rd0 + (T + T / 16) * l];
                }
            }
            __syncthreads(); // the staging area becomes the exchange buffer
        } else {
        load_rows<N, WG>(a, in0, rbase, t, (A.skip_dead_loads & 1) != 0);
        }
        STAMP(1, false);
        if constexpr (MID == MID_ZSRC) {
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
                for (int l = 0; l < 16; l++) zsrc[rbase[h] + t + T * l] = make_float2(0.f, 0.f);
        }
        // table operands are requested together with the rows (one burst of independent loads) and stay in
        // registers across the first transform; left at their point of use the compiler issues them one by one
        float2 pv[(MID == MID_PTAB) ? 1 : 1][16]; // column factors of the propagator: the same for every row of the thread
        float2 pr[WGeo<WG>::NRV];                  // row factors
        float gvv[(MID == MID_GTAB) ? WGeo<WG>::NRV : 1][16];
        auto load_prop = [&]() {
            const float2* __restrict__ pcol = reinterpret_cast<const float2*>(A.pcol);
            const float2* __restrict__ prow = reinterpret_cast<const float2*>(A.prow);
#pragma unroll
            for (int l = 0; l < 16; l++) pv[0][l] = pcol[t + T * l];
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++) pr[h] = prow[grow[h]];
        };
        // the 1-D propagator tables sit in the caches: requested after the first transform (FDES_PSEP_LATE) they do not
        // hold 36 registers across it, requested with the rows their latency is hidden
        if constexpr (MID == MID_PTAB && !FDES_PSEP_LATE) load_prop();
        if constexpr (MID == MID_GTAB) {
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
                for (int l = 0; l < 16; l++) gvv[h][l] = gtab[rbase[h] + t + T * l];
        }
        // second operand of the product: requested with the first so that its HBM round trip overlaps the first transform
        if constexpr (MID == MID_MULPSI && FDES_P5_PREFETCH) load_rows<N, WG>(b, in1, rbase, t, (A.skip_dead_loads & 2) != 0);
        STAMP(2, true);  // first operand (and tables) landed
        xform<N, WG, PRE, false, TWR>(a, lds, r, t, tw, gs);
        STAMP(3, false); // first transform done
        if constexpr (MID == MID_EXPIV) {
            expiv_all(a, [&](float2 v, auto wide) {
                const float e = __expf(-v.y);
                float sn, cs;
                sincos_sel<decltype(wide)::value>(v.x, sn, cs);
                return make_float2(e * cs, e * sn);
            }, [](float2 v) { return fabsf(v.x); });
        } else if constexpr (MID == MID_EXPIV_RE || MID == MID_EXPIV_IM) {
            // two slices share one potential grid: V_s = Re, V_{s+1} = Im; absorption V.y = imPot * V.x
            expiv_all(a, [&](float2 w, auto wide) {
                const float v = (MID == MID_EXPIV_RE) ? w.x : w.y;
                const float e = __expf(-(v * A.scale));
                float sn, cs;
                sincos_sel<decltype(wide)::value>(v, sn, cs);
                return make_float2(e * cs, e * sn);
            }, [](float2 w) { return (MID == MID_EXPIV_RE) ? fabsf(w.x) : fabsf(w.y); });
        } else if constexpr (MID == MID_EXPIV_PAIR) {
            // transmission function of the first slice now; the second slice's potential (one float per pixel) waits
            // in registers until the first result has been transformed and stored
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
                for (int l = 0; l < 16; l++) vim[h][l] = a[h][l].y;
            pair_transmission(a, A.scale);
        } else if constexpr (MID == MID_MASK) {
            // zeroHighFreq tests (float)(i1^2 + i2^2) * 9 / mindim^2 > 1 (src/multisliceSimulation.cu:241).  On the
            // power-of-two grids this kernel serves, mindim^2 < 2^24, so every float on the deciding side of the
            // threshold is exact and the integer comparison is the same predicate, without 32 divisions per thread.
            // Per row the live columns are |i1| <= Lr, Lr the largest integer with 9 Lr^2 <= mindim^2 - 9 i2^2 (a
            // wave-uniform value, found once per row); the element test is then one compare against t.
            const int md2 = A.mindim * A.mindim;
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++) {
                const int i2 = iwc(grow[h], A.nrows);
                const int q = md2 - 9 * i2 * i2;
                int Lr = (int)(sqrtf((float)(q > 0 ? q : 0)) * (1.0f / 3.0f));
                Lr += (9 * (Lr + 1) * (Lr + 1) <= q) ? 1 : 0;
                Lr -= (9 * Lr * Lr > q) ? 1 : 0; // q < 0: Lr = -1, nothing is live
                const int tlo = Lr, thi = N - Lr; // live: c <= Lr (c < N/2) or c >= N - Lr (c >= N/2; i1 = N/2 squares like -N/2)
#pragma unroll
                for (int l = 0; l < 16; l++) {
                    const bool live = (l < 8) ? (t <= tlo - T * l) : (t >= thi - T * l);
                    const float f = live ? A.scale : 0.f;
                    a[h][l] = make_float2(a[h][l].x * f, a[h][l].y * f);
                }
            }
        } else if constexpr (MID == MID_SCALE) {
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
                for (int l = 0; l < 16; l++) a[h][l] = make_float2(a[h][l].x * A.scale, a[h][l].y * A.scale);
        } else if constexpr (MID == MID_GTAB) {
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
                for (int l = 0; l < 16; l++) {
                    const float gv = gvv[h][l];
                    a[h][l] = make_float2(a[h][l].x * gv, a[h][l].y * gv);
                }
        } else if constexpr (MID == MID_PTAB) {
            if constexpr (FDES_PSEP_LATE) load_prop();
            // psi-hat * P with P(kx, ky) = exp(-i pi lambda d3 kx^2) exp(-i pi lambda d3 ky^2) / (m1 m2) inside the radial 2/3
            // band limit (fresnelPropagatorDevice + zeroHighFreq, src/multisliceSimulation.cu:225-274, 594-603): the phase
            // is a sum of a row term and a column term, so two 1-D tables replace the m1 x m2 grid (a third of this pass's
            // memory traffic); the mask is the integer predicate of MID_MASK.
            const int md2 = A.mindim * A.mindim;
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++) {
                const int i2 = iwc(grow[h], A.nrows);
                const int q = md2 - 9 * i2 * i2;
                int Lr = (int)(sqrtf((float)(q > 0 ? q : 0)) * (1.0f / 3.0f));
                Lr += (9 * (Lr + 1) * (Lr + 1) <= q) ? 1 : 0;
                Lr -= (9 * Lr * Lr > q) ? 1 : 0; // q < 0: Lr = -1, nothing is live
                const int tlo = Lr, thi = N - Lr;
#pragma unroll
                for (int l = 0; l < 16; l++) {
                    const bool live = (l < 8) ? (t <= tlo - T * l) : (t >= thi - T * l);
                    const float2 w = cmul3(pr[h], pv[0][l]);
                    const float2 v = cmul3(a[h][l], w);
                    a[h][l] = live ? v : make_float2(0.f, 0.f);
                }
            }
        } else if constexpr (MID == MID_MULPSI) {
            if constexpr (!FDES_P5_PREFETCH) load_rows<N, WG>(b, in1, rbase, t, (A.skip_dead_loads & 2) != 0);
            STAMP(4, true);  // second operand landed
            xform<N, WG, PRE, true, TWR>(b, lds, r, t, tw, gs);
            STAMP(5, false); // second transform done
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
                for (int l = 0; l < 16; l++) a[h][l] = cmul3(a[h][l], b[h][l]); // f0 = t, f1 = psi
        }
    }
    auto store_rows = [&](float2 (&v)[WGeo<WG>::NRV][16], float2* outp) {
        if constexpr (!STORE_T) {
            const unsigned pout = A.pitch_out ? (unsigned)A.pitch_out : (unsigned)N;
            float2* __restrict__ on = outp + (size_t)row0 * pout;
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
                for (int l = 0; l < 16; l++) on[(unsigned)(r + h * RH) * pout + t + T * l] = v[h][l];
        } else {
            // stage the R x N tile as [c][r] (swizzled) and write R contiguous elements per output row
            __syncthreads();
#pragma unroll
            for (int h = 0; h < WGeo<WG>::NRV; h++) {
                const int rr = r + h * RH;
#pragma unroll
                for (int l = 0; l < 16; l++) {
                    const int c = t + T * l;
                    lds[c * R + ((rr + (c >> G_::SH)) & (R - 1))] = v[h][l];
                }
            }
            __syncthreads();
            STAMP(10, false); // transposed tile staged
            float2* __restrict__ dst = outp + row0; // transposed grid: N rows of length nrows
            const unsigned ld = ldt;
            const int rr = tid & (R - 1), c0 = tid / R; // WG is a multiple of R: rr is the same in every iteration
#pragma unroll
            for (int it = 0; it < WGeo<WG>::NRV * 16; it++) {
                const int c = c0 + it * (WGeo<WG>::THR / R);
                // wave-uniform: skip an iteration only when every column it covers ([it, it + 1) * WG / R) is dead
                if (A.skip_dead_stores && dead_index(iwc(it * (WGeo<WG>::THR / R), N), A.band) &&
                    dead_index(iwc(it * (WGeo<WG>::THR / R) + WGeo<WG>::THR / R - 1, N), A.band) &&
                    (it * (WGeo<WG>::THR / R) > N / 2) == (it * (WGeo<WG>::THR / R) + WGeo<WG>::THR / R - 1 > N / 2))
                    continue;
                (dst + (size_t)(it * (WGeo<WG>::THR / R)) * ld)[(unsigned)c0 * ld + (unsigned)rr] = lds[c * R + ((rr + (c >> G_::SH)) & (R - 1))];
            }
        }
    };
    STAMP(6, false); // point-wise work done
    xform<N, WG, POST, (PRE != XF_NONE), TWR>(a, lds, r, t, tw, gs);
    STAMP(7, false); // last transform done
    store_rows(a, out0);
    STAMP(8, false); // stores issued
    STAMP(9, true);  // stores drained
    if constexpr (MID == MID_EXPIV_PAIR) {
#pragma unroll
        for (int h = 0; h < WGeo<WG>::NRV; h++)
#pragma unroll
            for (int l = 0; l < 16; l++) a[h][l].x = vim[h][l];
        pair_transmission(a, A.scale);
        __syncthreads(); // every wave is done with the transpose tile before the second transform writes the row buffers
        xform<N, WG, POST, true, TWR>(a, lds, r, t, tw, gs);
        store_rows(a, reinterpret_cast<float2*>(A.out2) + ((A.nbatch > 1) ? (size_t)bz * A.bstride_out2 : (size_t)0));
    }
}

template <int N, int WG, int PRE, int MID, int POST, bool STORE_T>
__global__ __launch_bounds__(WGeo<WG>::THR, ((WG & 1) ? 4 : pass_waves(MID))) void k_pass(PassArgs A)
{
    extern __shared__ float2 lds[];
    pass_body<N, WG, PRE, MID, POST, STORE_T>(A, lds);
}

#undef float2
#undef make_float2

template <int N, int WG, int PRE, int MID, int POST, bool ST> hipError_t launch(const PassArgs& a, hipStream_t st)
{
    using G_ = Geo<N, WG>;
    // The dynamic-LDS limit (69-139 KB, above the 64 KB default) is a property of the function ON ONE DEVICE: one bit
    // per device, set by whichever host thread launches this instantiation there first (fetch_or: two threads racing on
    // the same device both make the call, which is idempotent).
    static std::atomic<unsigned long long> attr_set{0};
    constexpr size_t lds_bytes = sizeof(float2) * (size_t)G_::LDROW * G_::R + 64;
    auto kern = k_pass<N, WG, PRE, MID, POST, ST>;
    int dev = 0;
    {
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
    }
    if (dev < 0 || dev >= 64 || !((attr_set.load(std::memory_order_acquire) >> dev) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    if (a.nrows % G_::R != 0) return hipErrorInvalidValue;
    int groups = a.nrows / G_::R;
    if (a.live_rows_only) {
        if (a.band <= 0) return hipErrorInvalidValue;
        const int L = a.band_L;
        const int g_lo = L / G_::R + 1, g_hi = (a.nrows - L) / G_::R;
        if (g_hi > g_lo) groups = g_lo + (a.nrows / G_::R - g_hi); // else: everything is live
        else { PassArgs b = a; b.live_rows_only = 0; b.walk = 1; hipLaunchKernelGGL(kern, dim3(groups, 1, a.nbatch > 1 ? a.nbatch : 1), dim3(WGeo<WG>::THR), lds_bytes, st, b); return hipGetLastError(); }
    }
    PassArgs w = a;
    const int ny = (MID == MID_ATOMS) ? (a.nspecies > 0 ? a.nspecies : 1) : 1; // grid.y = species (q0 / q1 are those of species 0, the grids are species_stride apart)
    if (a.nbatch > 16) return hipErrorInvalidValue;
    const int nz = a.nbatch > 1 ? a.nbatch : 1;                                   // grid.z = batch
    if (a.ev_start && a.ev_stop) { // timed launch: always whole
        w.ev_start = w.ev_stop = nullptr;
        hipExtLaunchKernelGGL(kern, dim3(groups, ny, nz), dim3(WGeo<WG>::THR), lds_bytes, st, (hipEvent_t)a.ev_start, (hipEvent_t)a.ev_stop, 0, w);
        return hipGetLastError();
    }
    int chunk = groups;
    if (a.walk > 1) { // parts of about groups / walk workgroups (a multiple of 8, see the kernel), one launch each
        chunk = ((groups + a.walk - 1) / a.walk + 7) & ~7;
        if (chunk >= groups) chunk = groups;
    }
    if (chunk < groups) w.nvirt = groups;
    for (int v0 = 0; v0 < groups; v0 += chunk) {
        w.vb0 = v0;
        hipLaunchKernelGGL(kern, dim3(chunk < groups - v0 ? chunk : groups - v0, ny, nz), dim3(WGeo<WG>::THR), lds_bytes, st, w);
    }
    return hipGetLastError();
}

template <int N, int WG> hipError_t dispatch(int pre, int mid, int post, bool st_t, const PassArgs& a, hipStream_t st)
{
#define CASE(P, M, Q, S) if (pre == P && mid == M && post == Q && st_t == S) return launch<N, WG, P, M, Q, S>(a, st);
    CASE(XF_NONE, MID_NONE, XF_NONE, false)  // copy (memory floor of the access pattern; micro-benchmark)
    CASE(XF_NONE, MID_NONE, XF_NONE, true)   // transpose-copy
    CASE(XF_NONE, MID_SCALE, XF_NONE, true)  // scaled transpose-copy (empty-slice fast path)
    CASE(XF_FWD, MID_NONE, XF_NONE, false)   // real -> mixed, natural store            (start of a configuration)
    CASE(XF_INV, MID_NONE, XF_NONE, false)   // mixed -> real, natural store            (end of a configuration)
    CASE(XF_INV, MID_SCALE, XF_NONE, false)
    CASE(XF_FWD, MID_NONE, XF_NONE, true)    // generic 2-D FFT passes
    CASE(XF_INV, MID_NONE, XF_NONE, true)
    CASE(XF_FWD, MID_ZSRC, XF_NONE, true)    // P1: deposit grid -> x spectrum
    CASE(XF_FWD, MID_ATOMS, XF_NONE, true)   // P1': atom records -> x spectrum of two slices (re / im)
    CASE(XF_INV, MID_EXPIV_RE, XF_FWD, true) // P3 on the real / imaginary component of a packed potential
    CASE(XF_INV, MID_EXPIV_IM, XF_FWD, true)
    CASE(XF_INV, MID_EXPIV_PAIR, XF_FWD, true) // P3 for both slices of a packed potential
    CASE(XF_FWD, MID_GTAB, XF_INV, true)     // P2: y FFT * f_e/sinc, y IFFT (one species)
    CASE(XF_FWD, MID_GTABN, XF_INV, true)    // P2: y FFT * f_e/sinc, species sum, y IFFT
    CASE(XF_INV, MID_EXPIV, XF_FWD, true)    // P3: x IFFT, exp(iV), x FFT
    CASE(XF_FWD, MID_MASK, XF_INV, true)     // P4: y FFT, band limit, y IFFT
    CASE(XF_INV, MID_MULPSI, XF_FWD, true)   // P5: x IFFT of t and psi, product, x FFT
    CASE(XF_NONE, MID_MULPSI, XF_FWD, true)  // stand-alone propagation unit: product of two real-space grids, x FFT
    CASE(XF_FWD, MID_PTAB, XF_INV, true)     // P6: y FFT, * propagator, y IFFT
#undef CASE
    return hipErrorInvalidValue;
}

template <int N> hipError_t dispatch_wg(int pre, int mid, int post, bool st_t, const PassArgs& a, hipStream_t st)
{
    if (a.wg == 256) return dispatch<N, 256>(pre, mid, post, st_t, a, st);
    if constexpr (N <= 2048) { if (a.wg == 1) return dispatch<N, N / 4 + 1>(pre, mid, post, st_t, a, st); }
    return dispatch<N, 512>(pre, mid, post, st_t, a, st);
}

} // namespace

bool lds_fft_supported_len(int n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096; }
int lds_fft_rows_per_block(int n, int wg)
{
    if (gen_pass_supported_len(n)) return gen_pass_rows(n);
    if (wg == 128) return n == 2048 ? 8 : ((wave_pass_supported_len(n) || n <= 1024) ? 4 : 256 * 2 * 16 / n);
    if (wg == 64 || wg == 65) return (wave_pass_supported_len(n) || n <= 1024) ? 4 : 256 * 2 * 16 / n; // shorter rows: one row per thread, four rows per workgroup
    return wg == 1 ? 4 : (wg == 256 ? 256 * 2 : 512 * 2) * 16 / n;
}

int lds_fft_rows_per_block(int n, int wg, int nrows)
{
    if (gen_pass_supported_len(n)) return gen_pass_tile_rows(n, nrows);
    const int r = lds_fft_rows_per_block(n, wg);
    return (r > 0 && nrows % r == 0) ? r : 0;
}

void lds_fft_twiddles(int n, float* tw0, float* tw1)
{
    const int T = n / 16;
    const double w = -2.0 * 3.14159265358979323846 / (double)n;
    for (int k = 0; k < 16; k++)
        for (int t = 0; t < T; t++) {
            const double a = w * (double)t * (double)k;
            tw0[2 * (k * T + t)] = (float)std::cos(a);
            tw0[2 * (k * T + t) + 1] = (float)std::sin(a);
        }
    const int P = T / 16 > 0 ? T / 16 : 1;
    for (int k = 0; k < 16; k++)
        for (int p = 0; p < P; p++) {
            const double a = w * 16.0 * (double)p * (double)k;
            tw1[2 * (k * P + p)] = (float)std::cos(a);
            tw1[2 * (k * P + p) + 1] = (float)std::sin(a);
        }
}

hipError_t lds_pass(int n, int pre, int mid, int post, bool st_t, const PassArgs& a_in, hipStream_t st)
{
    PassArgs a = a_in;
    if (a.band > 0) {
        a.band_L = live_limit(a.band);
        if (a.band_L != n / 3) a.skip_dead_loads = 0; // the kernel's column classes assume the band of a square grid
    }
    if (gen_pass_supported_len(n)) return gen_pass(n, pre, mid, post, st_t, a, st);
    // (four rows per workgroup there: a row count the plan's own geometry divides but four does not stays with the kernels below)
    if (a.wg != 65 && a.walk <= 1 && a.nrows % 4 == 0 && wave_pass_preferred(n, pre, mid, post, st_t)) { a.wg = 64; return wave_pass(n, pre, mid, post, st_t, a, st); }
    if (a.wg == 64 || a.wg == 65 || a.wg == 128) {
        if (wave_pass_supported_len(n) && a.walk <= 1) return wave_pass(n, pre, mid, post, st_t, a, st);
        a.wg = n <= 1024 ? 1 : 256; // rows without such a kernel (or a pass launched in parts, which only the kernels of this file do): one row per thread up to 1024 points, else two rows per thread
    }
    switch (n) {
    case 256: return dispatch_wg<256>(pre, mid, post, st_t, a, st);
    case 512: return dispatch_wg<512>(pre, mid, post, st_t, a, st);
    case 1024: return dispatch_wg<1024>(pre, mid, post, st_t, a, st);
    case 2048: return dispatch_wg<2048>(pre, mid, post, st_t, a, st);
    case 4096: return dispatch_wg<4096>(pre, mid, post, st_t, a, st);
    default: return hipErrorInvalidValue;
    }
}

} // namespace fdes
