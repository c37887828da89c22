// fft_lds.h — hand-written LDS row-FFT passes for power-of-two grids (gfx950).
//
// A "pass" streams whole rows of a 2-D complex grid through one workgroup: coalesced row loads ->
// [row FFT] -> point-wise operation -> [row FFT] -> store, either in place order (natural) or
// transposed (so that the next pass finds the other axis contiguous).  A 2-D FFT is two passes
// with transposed stores; the slice loop of the multislice engine chains six such passes per slice
// and never runs a stand-alone point-wise kernel ("Fused slice loop").
#ifndef FDES_FFT_LDS_H_
#define FDES_FFT_LDS_H_
#ifndef __HIPCC_RTC__ // (fft_gen.hip is also compiled by hipRTC at run time: the runtime's own declarations are built in there)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#endif

namespace fdes {

enum XfKind { XF_NONE = 0, XF_FWD = 1, XF_INV = 2 };
enum MidKind {
    MID_NONE = 0,
    MID_ZSRC = 1,   // none, but the source rows are overwritten with zeros after loading (deposit grid)
    MID_GTAB = 2,   // spectrum * G[row][col]                            (projected potential filter, 1 species)
    MID_EXPIV = 3,  // v -> exp(-v.y) (cos v.x, sin v.x)               (potential2Transmission)
    MID_MASK = 4,   // radial 2/3 band limit * scale                     (zeroHighFreq + Csscal)
    MID_MULPSI = 5, // (row of in0) (x) (row of in1), both inverse transformed first (multiplyElementwise)
    MID_PTAB = 6,   // spectrum (x) P(row, col) = prow[row] pcol[col] inside the band limit, 0 outside
                    //                                                   (multiplyElementwise with frProp)
    MID_SCALE = 7,  // * scale
    MID_GTABN = 8,  // MID_GTAB with a species loop (nspecies > 1)
    MID_ATOMS = 9,  // source rows are built from the sorted atom records (re: slice q0, im: slice q1), no grid read
    MID_EXPIV_RE = 10, // t = exp(-scale * v) (cos v, sin v) with v = Re(row)  (two slices packed per potential grid)
    MID_EXPIV_IM = 11, // ... v = Im(row)
    MID_EXPIV_PAIR = 12 // both at once: out <- F_x[t(Re)], out2 <- F_x[t(Im)] (one read and one inverse transform for two slices)
};

struct PassArgs {
    const float2* in0 = nullptr;
    const float2* in1 = nullptr;
    float2* out = nullptr;
    float2* out2 = nullptr;      // MID_EXPIV_PAIR: second output grid
    float2* zsrc = nullptr;      // MID_ZSRC: rows to clear (== in0)
    const float* gtab = nullptr; // MID_GTAB: [species][row][col]
    // Fresnel propagator, separable: P(k_row, k_col) = prow[row] * pcol[col] inside the radial band limit (prow carries
    // 1 / (m1 m2)); two 1-D tables instead of a grid (needs mindim)
    const float2* prow = nullptr;
    const float2* pcol = nullptr;
    const float2* tw0 = nullptr; // twiddles of this row length
    const float2* tw1 = nullptr;
    int nrows = 0;               // rows of the input grid (= leading dimension of a transposed output)
    // Row pitches in elements (0: dense).  pitch_in: distance between input rows (in0, in1, ptab, gtab, zsrc);
    // pitch_out: distance between output rows (natural store: default n; transposed store: default nrows).  A pitch
    // that is not a power of two spreads the 2^k-strided segments of a transposed store over the memory channels.
    int pitch_in = 0, pitch_out = 0;
    // walk > 1: the pass is launched in `walk` parts of about 1 / walk of the row groups each (half of the chip's
    // workgroup slots at walk = 2); nvirt, vb0 (set by lds_pass) = number of row groups, first group of this part
    int walk = 1, nvirt = 0, vb0 = 0;
    int nspecies = 1;
    size_t species_stride = 0;   // elements between species grids (in0 and gtab)
    float scale = 1.f;
    int mindim = 0;              // min(m1, m2) for the band limit
    int wg = 512;                // workgroup geometry: 512 or 256 threads x 2 rows per thread, 1 = n/4 threads x 1 row (4 rows per workgroup),
                                 // 64 = one wave per row, 4 rows per workgroup (fft_wave.hip; 2048- and 4096-point rows, else as 256);
                                 // 65 = the same as a software pipeline: one workgroup per CU walks the row groups, the next group's
                                 //      operands are requested into a second register set before the current group is transformed
    // one-wave-per-row passes: wave w of a CU's s-th workgroup starts (w + 4 s) * stagger * 64 cycles late (s taken as
    // (workgroup index) / ncu: workgroups are dealt breadth first over the CUs; speed only)
    int stagger = 0, ncu = 256;
    // Band limit bookkeeping (band = mindim^2 > 0 enables it): frequency index i is dead iff 9 i^2 > band, i.e. outside
    // the radial 2/3 mask whatever the other index is.  Dead rows/columns hold exact zeros that nobody needs to move.
    int band = 0;
    int band_L = 0;              // largest live |i| (filled in by lds_pass)
    int live_rows_only = 0;      // the ROWS of this pass are frequencies: only row groups with a live row are launched
    int skip_dead_loads = 0;     // the COLUMNS of an input are frequencies masked upstream: dead ones are not loaded (= 0);
                                 // bit 0: in0, bit 1: in1 (MID_MULPSI's second operand)
    int skip_dead_stores = 0;    // transposed store: output rows (= our columns) that the next pass never reads are not written
    // MID_ATOMS
    const void* recs = nullptr;  // AtomRec[] sorted by (slice, species, row)
    const int* rowstart = nullptr; // [q][nrows + 1]
    int q0 = -1, q1 = -1;        // (slice * nZ + species) deposited into the real / imaginary component (-1: none)
    // Batch of independent grids in ONE launch (grid.z = nbatch <= 16; fft_lds.hip and fft_gen.hip): grid z reads
    // in0 + (use_zin ? zin[z] : z) * bstride_in0 and writes out + z * bstride_out (out2 + z * bstride_out2); MID_ATOMS:
    // q0 / q1 = zq0[z] / zq1[z].  The potential / transmission passes of several slice pairs of ONE configuration run this
    // way (a single image at 1024^2 and below cannot fill the chip with one slice's rows).
    // A gang of configurations (engine.hip) is the same launch with one CONFIGURATION per grid z: the second
    // operand of MID_MULPSI then also moves by z (bstride_in1) and so do the atom records and their row table of MID_ATOMS
    // (bstride_recs in records, bstride_rowstart in ints; zq0 / zq1 then hold the same slice for every z).
    int nbatch = 1, use_zin = 0;
    size_t bstride_in0 = 0, bstride_out = 0, bstride_out2 = 0;
    size_t bstride_in1 = 0, bstride_recs = 0, bstride_rowstart = 0;
    int zin[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int zq0[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, zq1[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // diagnostic builds (-DFDES_STAMPS): per wave, 16 shader-clock stamps of the pass's phases (tools/stamps.py)
    unsigned long long* dbg = nullptr;
    // host side only: when set, the dispatch is bracketed by these two events through hipExtLaunchKernelGGL, whose
    // timestamps come from the dispatch packet itself (kernel begin / end, what a profiler reports) rather than from
    // marker packets before and after it
    void* ev_start = nullptr;
    void* ev_stop = nullptr;
    // host side only: the kernels hipRTC compiled for this row length at plan creation (GenJitKernels, gen_jit.h), or nullptr
    const void* jit = nullptr;
    // host side only, mixed-radix passes: rows per tile where the length's own number does not divide nrows (gen_pass_tile_rows); 0: default
    int tile_rows = 0;
};

#ifndef __HIPCC_RTC__
// Row lengths the register-resident kernels are instantiated for (powers of two, 256 ... 4096); lds_pass() also takes
// the lengths of gen_pass_supported_len().
bool lds_fft_supported_len(int n);
// rows per workgroup for row length n and wg threads per workgroup (512: one workgroup per CU; 256: two)
int lds_fft_rows_per_block(int n, int wg);
// the same for a grid with `nrows` rows of that length: a mixed-radix length may fall back to smaller tiles (gen_pass_tile_rows); 0: none fits
int lds_fft_rows_per_block(int n, int wg, int nrows);
// fills host arrays (float2 as 2 floats) with the twiddle tables of length n: tw0[16*T], tw1[T]
void lds_fft_twiddles(int n, float* tw0, float* tw1);

// one-wave-per-row passes (fft_wave.hip)
bool wave_pass_supported_len(int n);
bool wave_pass_preferred(int n, int pre, int mid, int post, bool store_transposed);
hipError_t wave_pass(int n, int pre, int mid, int post, bool store_transposed, const PassArgs& a, hipStream_t st);

// LDS-resident mixed-radix passes for row lengths 2^a 3^b 5^c 7^d in [256, 4096] that are not powers of two (fft_gen.hip);
// their twiddle table is the n roots of unity W_n^k (float2 as 2 floats)
bool gen_pass_supported_len(int n);
bool gen_pass_needs_compiled(int n); // ... only as kernels compiled at plan creation (rows beyond 4096 points, factors 17 / 19 / 23)
int gen_pass_rows(int n);
int gen_pass_tile_rows(int n, int nrows);
void gen_pass_twiddles(int n, float* tw);
hipError_t gen_pass(int n, int pre, int mid, int post, bool store_transposed, const PassArgs& a, hipStream_t st);

// Launch one pass over all rows. n = row length, kinds select the template instantiation.
hipError_t lds_pass(int n, int pre, int mid, int post, bool store_transposed, const PassArgs& a, hipStream_t st);
#endif // __HIPCC_RTC__

} // namespace fdes
#endif
