/* fdes_abi_test.h — test, parity-tap and micro-benchmark hooks of libFDES_SHARED_LIB.so.
 *
 * Not part of the drop-in surface (include/fdes_abi.h): nothing here replaces a reference interface a host
 * program would call.  The parity tests (tests/), bench.py's roofline probe and the tools under tools/ use them.
 * Same conventions as fdes_abi.h (plain C, 0 / negative FDES_E* codes).
 */
#ifndef FDES_ABI_TEST_H_
#define FDES_ABI_TEST_H_

#include "fdes_abi.h"

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden and an export list (fdes_amd/csrc/exports_*.txt): only what this header
 * declares is a dynamic symbol. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* How many configurations have been asked which of their slices are empty (option "skip_empty": one D2H and one host
 * wait each); a specimen without empty slices stops being asked after eight configurations in a row (diagnostic). */
int64_t fdes_plan_empty_queries(const fdes_plan* plan);
/* Sum of the HIP-event durations [ms] of the probed launches of the dominant kernel (option
 * "probe_stride" = n > 0 brackets every n-th 2-D FFT with events on the plan's stream) since the
 * last call, and how many were probed.  Synchronises. */
int fdes_plan_probe_ms(fdes_plan* plan, double* total_ms, int64_t* launches);

/* ---- stage taps for parity tests (device results copied to HOST buffers) ---- */
/* Atom coordinates used by configuration (k, j): tilt offset, tilt k, jitter. float[3*nAt]. */
int fdes_plan_tap_coords(fdes_plan* plan, int k, int j, float* xyz);
/* phaseGrating (src/crystalMaker.cu:507-536) of sub-slice s for configuration (k, j):
 * V as float[2*m1*m2] interleaved (.x = sigma*v_z, .y = imPot part). */
int fdes_plan_tap_potential(fdes_plan* plan, int k, int j, int s, float* V);
/* Wave after `nslices` slices of configuration (k, j) (nslices = m3 -> exit wave), before
 * any exit-wave post-processing. float[2*m1*m2]. */
int fdes_plan_tap_wave(fdes_plan* plan, int k, int j, int nslices, float* psi);
/* Band-limited Fresnel propagator as the slice loop applies it
 * (src/multisliceSimulation.cu:594-603). float[2*m1*m2]. */
int fdes_plan_tap_propagator(fdes_plan* plan, float* P);
/* One propagation unit on caller-provided DEVICE buffers (micro-benchmark and parity):
 * psi <- F^-1[ P * F[ t * psi ] ], batch wave functions of m2 x m1 float2 each;
 * t is shared (batch stride 0) or per-wave.  (src/multisliceSimulation.cu:546-548) */
int fdes_plan_propagate_dev(fdes_plan* plan, void* psi_dev, const void* t_dev, int batch, int t_per_wave);

/* Unnormalised 2-D C2C FFT of a HOST grid (float[2*m1*m2], idx = i2*m1 + i1) through the engine's FFT
 * back-end (cufftExecC2C stand-in; test hook).  backend: 0 auto, 1 rocFFT, 2 LDS kernels.
 * Returns the back-end used (1 or 2) or a negative error. */
int fdes_fft2d_host(fdes_ctx* ctx, float* data, int m1, int m2, int inverse, int backend);

/* Micro-benchmark of one LDS row pass on zero-filled n x n scratch grids: mean launch time [us].
 * pre/post: 0 none, 1 forward, 2 inverse row FFT; mid: point-wise op id (fft_lds.h); store_t: transposed store;
 * streams: launches are issued round-robin on this many HIP streams (own grids each), host-timed. */
int fdes_bench_pass(fdes_ctx* ctx, int n, int pre, int mid, int post, int store_t, int iters, int streams, double* us);

/* Engine options for tests and benches only (fdes_set_option):
 *   "probe_stride"  n > 0: bracket every n-th launch of the probed pass class with the start / stop events of the dispatch
 *                   itself (fdes_plan_probe_ms); a library built with TEST_HOOKS=0 refuses these keys (FDES_EINVAL)
 *   "probe_pass"    1 .. 6: the pass class of the fused slice loop that is bracketed (1 = P1' ... 6 = P6, default 5 = P5)
 *   "lanes_active"  n > 0: run_config deals only to the first n lanes from now on (0: all)
 *   "bench_band", "bench_alt", "bench_tall", "bench_pitch", "bench_serial"  shape fdes_bench_pass only */

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FDES_ABI_TEST_H_ */
