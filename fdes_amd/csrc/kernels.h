// kernels.h — host-callable launchers of the point-wise wave-optics kernels (kernels.hip).
#ifndef FDES_KERNELS_H_
#define FDES_KERNELS_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fdes_abi.h"

namespace fdes {

struct GangPar { int n = 0; float f[16] = {}; int k[16] = {}; }; // per-member scalar / index of a gang launch (n = 0: none)

// Scalars the reference's kernels dereference from a device-resident params_t on every launch
// (SURVEY 8a a21); here they travel by value in the kernel argument segment.
struct KP {
    int m1, m2, m3, n1, n2, dn1, dn2, mode;
    float d1, d2, d3, lambda, sigma, imPot;
    float defocspread, illangle, mtfa, mtfb, mtfc, mtfd, ObjAp;
    fdes_aberration ab;
};

struct Kirk { float a0, b0, a1, b1, a2, b2, c0, d0, c1, d1, c2, d2; };
Kirk kirkland_params(int Z);

hipError_t k_fill(float2* f, size_t n, float re, float im, hipStream_t st);
hipError_t k_fill_noise(float* f, size_t n, unsigned seed, hipStream_t st);
hipError_t k_scale(float2* f, size_t n, float alpha, hipStream_t st);
hipError_t k_axpy(float2* y, const float2* x, size_t n, float alpha, hipStream_t st);
// real part of a complex grid <-> float grid (the intensity sums are real: collectives move the float view)
hipError_t k_real_pack(float* dst, const float2* src, size_t n, hipStream_t st);
hipError_t k_real_unpack(float2* dst, const float* src, size_t n, hipStream_t st);
hipError_t k_axpy_real(float2* y, const float* x, size_t n, hipStream_t st); // y.x += x
// Vhat = (first ? 0 : Vhat) + Dhat * g_Z(q) ; Dhat = 0   (projectedPotential_d * divideBySinc * multiplyWith...)
hipError_t k_filter_accum(float2* Vhat, float2* Dhat, const KP& p, const Kirk& kz, int first, hipStream_t st);
hipError_t k_filter_accum_tab(float2* Vhat, float2* Dhat, const float* G, size_t n, int first, hipStream_t st);
hipError_t k_transmit_comp(float2* t, const float2* W, size_t n, int comp, float imPot, hipStream_t st);
hipError_t k_transmit(float2* t, const float2* V, size_t n, hipStream_t st);
hipError_t k_pick_potential(float2* V, const float2* W, size_t n, int comp, float imPot, hipStream_t st);
hipError_t k_mask_scale(float2* f, int m1, int m2, float alpha, hipStream_t st);
hipError_t k_mul(float2* dst, const float2* f0, const float2* f1, size_t n, hipStream_t st); // dst = f0 (x) f1, 3-mult
hipError_t k_build_propagator(float2* P, const KP& p, int transposed, hipStream_t st);
// separable propagator tables of the fused loop: px[m1] (carries 1 / (m1 m2)), py[m2], n-th power
hipError_t k_build_propagator_1d(float2* px, float2* py, const KP& p, int npow, hipStream_t st);
hipError_t k_build_gtab(float* G, const KP& p, const Kirk& kz, int transposed, int pitch, hipStream_t st);
hipError_t k_lens(float2* psi, const KP& p, float defocus_k, hipStream_t st);
// Gang launches (engine.hip): the same point-wise kernels over the members of a gang in ONE launch (grid.y =
// member, the member's grid `stride` elements further); per-member scalars / indices travel in GangPar.
hipError_t k_lens_gang(float2* psi, size_t stride, const KP& p, const GangPar& gp, hipStream_t st);                 // f = defocus
hipError_t k_intensity_gang(float2* I, const float2* psi, size_t n, float pre, const GangPar& gp, hipStream_t st);  // f = weight, k = slot of I
hipError_t k_spatial_incoherence_gang(float2* f, size_t stride, const KP& p, int dp, const GangPar& gp, hipStream_t st); // f = defocus
hipError_t k_mtf_gang(float2* f, size_t stride, int members, const KP& p, float alpha, hipStream_t st);
hipError_t k_noise_gang(float2* f, size_t stride, size_t n, float dose, uint32_t seed, const GangPar& gp, hipStream_t st); // k = measurement
hipError_t k_mask_scale_gang(float2* f, size_t stride, int members, int m1, int m2, float alpha, hipStream_t st);
hipError_t k_tilt_beam_gang(float2* psi, size_t stride, const KP& p, const GangPar& tb0, const GangPar& tb1, int flag, hipStream_t st); // f = tilt
hipError_t k_tukey_gang(float2* psi, size_t stride, int members, const KP& p, hipStream_t st);
hipError_t k_fftshift_gang(float2* out, const float2* in, size_t stride, int members, int m1, int m2, hipStream_t st);
hipError_t k_mask_filter_gang(float2* psi, size_t stride, int members, const KP& p, hipStream_t st);
hipError_t k_crop_gang(float* J, const float2* I, size_t stride, const KP& p, const GangPar& gp, hipStream_t st);   // k = image of the stack
hipError_t k_intensity_axpy(float2* I, const float2* psi, size_t n, float pre_scale, float alpha, hipStream_t st);
hipError_t k_tilt_beam(float2* psi, const KP& p, float tb0, float tb1, int flag, hipStream_t st);
hipError_t k_tukey(float2* psi, const KP& p, hipStream_t st);
hipError_t k_fftshift(float2* out, const float2* in, int m1, int m2, hipStream_t st);
hipError_t k_mask_filter(float2* psi, const KP& p, hipStream_t st);
hipError_t k_spatial_incoherence(float2* f, const KP& p, float defocus_k, int dp, hipStream_t st);
hipError_t k_mtf(float2* f, const KP& p, float alpha, hipStream_t st);
hipError_t k_noise(float2* f, size_t n, float dose, uint32_t seed, int k, hipStream_t st);
hipError_t k_crop(float* J, const float2* I, const KP& p, hipStream_t st);
// ||f||_2 -> *out (device scalar); then f *= target / *out  (cublasScnrm2 + Csscal of incomingWave)
hipError_t k_normalize_to(float2* f, size_t n, float target, float* scratch, hipStream_t st);

} // namespace fdes
#endif
