// kernels.hip — point-wise wave-optics kernels of the FDES engine (gfx950).
//
// Each kernel names the reference kernel(s) it replaces (paths relative to the FDES tree).  The
// reference launches one 1024-thread block per 1024 pixels, reads every scalar through a device
// pointer to params_t and runs most steps as separate passes.  Here:
// 256-thread blocks (4 waves), grid-stride over at most 2048 blocks (8 per CU), scalars by value,
// and adjacent passes fused where that does not change the arithmetic order.
// Data layout: float2 (re, im), idx = i2 * m1 + i1, i1 fastest (include/coordArithmetic.h:36-40).
#include <cfloat>
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "rng.h"

namespace fdes {

static const float kirkland_tab[104][12] = {
#include "kirkland_table.inc"
};

Kirk kirkland_params(int Z)
{
    const float* k = kirkland_tab[(Z >= 1 && Z <= 103) ? Z : 0]; // fallback row: constant 1 (projectedPotential.cu:2985)
    Kirk r = {k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9], k[10], k[11]};
    return r;
}

#define PI_F 3.141592654f

__device__ __forceinline__ int iw(int i, int m) { return (i > m / 2) ? i - m : i; } // iwCoordIp
__device__ __forceinline__ int ow(int i, int m) { return i - m / 2; }               // owCoordIp

static inline dim3 grid_for(size_t n, int bs = 256)
{
    size_t b = (n + bs - 1) / bs;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return dim3((unsigned)b);
}
#define GS_LOOP(i, n) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

// ---- initialValues (src/complexMath.cu:64), cublasCsscal, cublasCaxpy --------------------------
__global__ void kk_fill(float2* __restrict__ f, size_t n, float re, float im)
{
    GS_LOOP(i, n) f[i] = make_float2(re, im);
}
// pseudo-random fill in [-1, 1) for the micro-benchmarks (zero-filled grids run at a higher clock than real data)
__global__ void kk_fill_noise(float* __restrict__ f, size_t n, unsigned seed)
{
    GS_LOOP(i, n) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        f[i] = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;
    }
}
__global__ void kk_scale(float2* __restrict__ f, size_t n, float a)
{
    GS_LOOP(i, n) { float2 v = f[i]; v.x *= a; v.y *= a; f[i] = v; }
}
__global__ void kk_axpy(float2* __restrict__ y, const float2* __restrict__ x, size_t n, float a)
{
    GS_LOOP(i, n) { float2 v = y[i]; const float2 u = x[i]; v.x += a * u.x; v.y += a * u.y; y[i] = v; }
}

__global__ void kk_real_pack(float* __restrict__ d, const float2* __restrict__ s, size_t n) { GS_LOOP(i, n) d[i] = s[i].x; }
__global__ void kk_real_unpack(float2* __restrict__ d, const float* __restrict__ s, size_t n) { GS_LOOP(i, n) d[i] = make_float2(s[i], 0.f); }
// y.x += x: a partial intensity sum arrives as its real view (its imaginary part is identically zero)
__global__ void kk_axpy_real(float2* __restrict__ y, const float* __restrict__ x, size_t n) { GS_LOOP(i, n) { float2 v = y[i]; v.x += x[i]; y[i] = v; } }

// The Fourier-space filter a deposit grid is multiplied with: Kirkland's f_e(q^2) (three Lorentzians + three Gaussians, q in
// 1/Angstrom from the frequency indices and the pixel sizes in Angstrom; projectedPotential_d, src/projectedPotential.cu:30-73)
// x the reference's constant and grid normalisation (`scale`), x the inverse transform of the pixel-wide top-hat along both
// axes, u / sin u with u = pi i / m (divideBySinc, src/crystalMaker.cu:136-158).  Float32 in the reference's order of
// operations: ((f_e scale) (x-factor y-factor)).
__device__ __forceinline__ float potential_filter(int i1, int i2, const KP& p, const Kirk& kz, float d1_angstrom, float d2_angstrom, float scale)
{
    const float q1 = ((float)i1) / (d1_angstrom * ((float)p.m1));
    const float q2 = ((float)i2) / (d2_angstrom * ((float)p.m2));
    const float qsq = q1 * q1 + q2 * q2;
    float fe = kz.a0 / (qsq + kz.b0) + kz.c0 * expf(-kz.d0 * qsq);
    fe += kz.a1 / (qsq + kz.b1) + kz.c1 * expf(-kz.d1 * qsq);
    fe += kz.a2 / (qsq + kz.b2) + kz.c2 * expf(-kz.d2 * qsq);
    const float u1 = ((float)i1) / ((float)p.m1) * PI_F;
    const float u2 = PI_F * (((float)i2) / ((float)p.m2));
    const float unsinc1 = (u1 + FLT_EPSILON) / (sinf(u1) + FLT_EPSILON);
    const float unsinc2 = (u2 + FLT_EPSILON) / (sinf(u2) + FLT_EPSILON);
    return (fe * scale) * (unsinc1 * unsinc2);
}

// ---- projectedPotential_d (src/projectedPotential.cu:30-73) * divideBySinc (src/crystalMaker.cu:
// 136-158) * multiplyWithProjectedPotential_d (:160-172) + cublasCaxpy (:532), in Fourier space.
// The reference materialises f_e(q) per slice and species and sums the species in real space after
// nZ inverse FFTs; the sum commutes with the (linear) inverse FFT, so it is taken here.
__global__ void kk_filter_accum(float2* __restrict__ Vh, float2* __restrict__ Dh, KP p, Kirk kz, int first)
{
    const size_t n = (size_t)p.m1 * p.m2;
    const float d1 = 1e10f * p.d1, d2 = 1e10f * p.d2;
    const float scale = (4.78776452e-9f * p.sigma) / (d1 * d2 * ((float)(p.m1 * p.m2)));
    GS_LOOP(i, n)
    {
        const int j1 = (int)(i % (size_t)p.m1), j2 = (int)(i / (size_t)p.m1);
        const int i1 = iw(j1, p.m1), i2 = iw(j2, p.m2);
        const float g = potential_filter(i1, i2, p, kz, d1, d2, scale);
        const float2 d = Dh[i];
        float2 v = first ? make_float2(0.f, 0.f) : Vh[i];
        v.x += d.x * g;
        v.y += d.y * g;
        Vh[i] = v;
        Dh[i] = make_float2(0.f, 0.f); // ready for the next deposit (initialValues, :522)
    }
}

// Vh (+)= G * Dh with the tabulated filter (kk_gtab, natural layout); Dh is cleared for the next deposit
__global__ void kk_filter_accum_tab(float2* __restrict__ Vh, float2* __restrict__ Dh, const float* __restrict__ G, size_t n, int first)
{
    GS_LOOP(i, n)
    {
        const float g = G[i];
        const float2 d = Dh[i];
        float2 v = first ? make_float2(0.f, 0.f) : Vh[i];
        v.x += d.x * g;
        v.y += d.y * g;
        Vh[i] = v;
        Dh[i] = make_float2(0.f, 0.f);
    }
}

// potential2Transmission on one component of a packed pair potential: v = Re or Im, absorption imPot * v
__global__ void kk_transmit_comp(float2* __restrict__ t, const float2* __restrict__ W, size_t n, int comp, float imPot)
{
    GS_LOOP(i, n)
    {
        const float2 w = W[i];
        const float v = comp ? w.y : w.x;
        const float a = expf(-(v * imPot));
        float s, c;
        sincosf(v, &s, &c);
        t[i] = make_float2(a * c, a * s);
    }
}

// Same filter as kk_filter_accum, tabulated once per plan and species (it does not depend on the
// slice): g_Z(q) = f_e(q^2) * 4.78776452e-9 sigma / (d1 d2 m12) * x/sin x * y/sin y.
__global__ void kk_gtab(float* __restrict__ G, KP p, Kirk kz, int transposed, int pitch)
{
    const size_t n = (size_t)p.m1 * p.m2;
    const float d1 = 1e10f * p.d1, d2 = 1e10f * p.d2;
    const float scale = (4.78776452e-9f * p.sigma) / (d1 * d2 * ((float)(p.m1 * p.m2)));
    GS_LOOP(i, n)
    {
        const int j1 = transposed ? (int)(i / (size_t)p.m2) : (int)(i % (size_t)p.m1);
        const int j2 = transposed ? (int)(i % (size_t)p.m2) : (int)(i / (size_t)p.m1);
        const int i1 = iw(j1, p.m1), i2 = iw(j2, p.m2);
        G[(transposed && pitch > 0) ? (size_t)j1 * (size_t)pitch + (size_t)j2 : i] = potential_filter(i1, i2, p, kz, d1, d2, scale); // pitch: padded rows of the transposed table
    }
}

// ---- potential2Transmission (src/multisliceSimulation.cu:41-52) -------------------------------
__global__ void kk_transmit(float2* __restrict__ t, const float2* __restrict__ V, size_t n)
{
    GS_LOOP(i, n)
    {
        const float2 v = V[i];
        const float a = expf(-v.y);
        float s, c;
        sincosf(v.x, &s, &c);
        t[i] = make_float2(a * c, a * s);
    }
}

// ---- zeroHighFreq (src/multisliceSimulation.cu:225-250) fused with the cublasCsscal that follows
// it in bandwidthLimit (:557-559; the scale commutes with the inverse FFT in between) ----------
__device__ __forceinline__ bool outside_band(int i1, int i2, float mindim)
{
    return ((float)(i1 * i1 + i2 * i2) * 9.f / (mindim * mindim)) > 1.f;
}
__global__ void kk_mask_scale(float2* __restrict__ f, int m1, int m2, float alpha, size_t gstride)
{
    f += (size_t)blockIdx.y * gstride;
    const size_t n = (size_t)m1 * m2;
    const float mindim = (float)(m1 < m2 ? m1 : m2);
    GS_LOOP(i, n)
    {
        const int i1 = iw((int)(i % (size_t)m1), m1), i2 = iw((int)(i / (size_t)m1), m2);
        float2 v = f[i];
        if (outside_band(i1, i2, mindim)) v = make_float2(0.f, 0.f);
        else { v.x *= alpha; v.y *= alpha; }
        f[i] = v;
    }
}

// Packed potential W = V_s + i V_{s+1} -> V of one slice as the reference holds it: (v, imPot * v)
__global__ void kk_pick_potential(float2* __restrict__ V, const float2* __restrict__ W, size_t n, int comp, float imPot)
{
    GS_LOOP(i, n) { const float2 w = W[i]; const float v = comp ? w.y : w.x; V[i] = make_float2(v, v * imPot); }
}

// ---- multiplyElementwise (src/complexMath.cu:44-62): 3-multiply product, f0 = (a, b), f1 = (c, d)
__device__ __forceinline__ float2 cmul3(float2 f0, float2 f1)
{
    const float a = f0.x, b = f0.y, c = f1.x, d = f1.y;
    const float shared = a * (c + d);
    const float sub_re = d * (a + b);
    const float add_im = c * (b - a);
    return make_float2(shared - sub_re, shared + add_im);
}
__global__ void kk_mul(float2* __restrict__ dst, const float2* __restrict__ f0, const float2* __restrict__ f1, size_t n)
{
    GS_LOOP(i, n) dst[i] = cmul3(f0[i], f1[i]);
}

// ---- fresnelPropagatorDevice + zeroHighFreq + Csscal (src/multisliceSimulation.cu:253-274, 594-603)
// The reference rebuilds this table every slice; it only depends on the plan.
// transposed != 0: element (kx, ky) is stored at kx * m2 + ky (layout of the fused LDS passes).
__global__ void kk_propagator(float2* __restrict__ P, KP p, int transposed)
{
    const size_t n = (size_t)p.m1 * p.m2;
    const float mindim = (float)(p.m1 < p.m2 ? p.m1 : p.m2);
    const float alpha = 1.f / ((float)(p.m1 * p.m2));
    GS_LOOP(i, n)
    {
        const int j1 = transposed ? (int)(i / (size_t)p.m2) : (int)(i % (size_t)p.m1);
        const int j2 = transposed ? (int)(i % (size_t)p.m2) : (int)(i / (size_t)p.m1);
        const int i1 = iw(j1, p.m1), i2 = iw(j2, p.m2);
        // k d3 per axis (k = i / (m d)), then the Fresnel phase -pi lambda d3 k^2 as -pi (k d3)^2 (lambda / d3)
        const float kd1 = ((float)(i1) / ((float)p.m1)) * (p.d3 / p.d1);
        const float kd2 = ((float)(i2) / ((float)p.m2)) * (p.d3 / p.d2);
        const float lambda_over_d3 = p.lambda / p.d3;
        const float phase = -PI_F * (kd1 * kd1 + kd2 * kd2) * lambda_over_d3;
        float2 v = make_float2(cosf(phase), sinf(phase));
        if (outside_band(i1, i2, mindim)) v = make_float2(0.f, 0.f);
        v.x *= alpha;
        v.y *= alpha;
        P[i] = v;
    }
}

// Separable form of the same propagator for the fused slice loop: P(i1, i2) = px[i1] * py[i2] inside the band limit,
// px[i1] = alpha exp(i n phi1(i1)), py[i2] = exp(i n phi2(i2)), phi = -pi t^2 lambda / d3 with t as in kk_propagator; the
// phases, their n-fold multiples (n consecutive slices without atoms are n Fresnel steps: one transform pair, so one
// alpha) and the sines / cosines are taken in double from the same float inputs, so the product of the two factors is
// within float32 rounding of the exact value (the reference's float32 phase is a few 1e-8 rad further away).
__global__ void kk_propagator_1d(float2* __restrict__ px, float2* __restrict__ py, KP p, int npow)
{
    const size_t n = (size_t)p.m1 + (size_t)p.m2;
    const double alpha = 1.0 / ((double)p.m1 * (double)p.m2);
    GS_LOOP(i, n)
    {
        const bool isx = i < (size_t)p.m1;
        const int m = isx ? p.m1 : p.m2;
        const int j = iw((int)(isx ? i : i - (size_t)p.m1), m);
        const double t = ((double)j / (double)m) * ((double)p.d3 / (double)(isx ? p.d1 : p.d2));
        const double ph = -3.14159265358979323846 * t * t * ((double)p.lambda / (double)p.d3) * (double)npow;
        double sn, cs;
        sincos(ph, &sn, &cs);
        const double a = isx ? alpha : 1.0;
        (isx ? px[i] : py[i - (size_t)p.m1]) = make_float2((float)(a * cs), (float)(a * sn));
    }
}

// ---- multiplyLensFunction (src/multisliceSimulation.cu:277-343) --------------------------------
// (gang launches, kernels.h: blockIdx.y = member, its grid `gstride` elements further, its scalar in gp)
__global__ void kk_lens(float2* __restrict__ psi, KP p, float defocus_k, size_t gstride, GangPar gp)
{
    psi += (size_t)blockIdx.y * gstride;
    if (gp.n) defocus_k = gp.f[blockIdx.y];
    const size_t n = (size_t)p.m1 * p.m2;
    const fdes_aberration ab = p.ab;
    GS_LOOP(i, n)
    {
        const int i1 = iw((int)(i % (size_t)p.m1), p.m1);
        const int i2 = -iw((int)(i / (size_t)p.m1), p.m2); // row index points up
        // scattering angle (nu, azimuth phi) of this frequency
        const float nu_x = (((float)i1) / ((float)p.m1)) * (p.lambda / p.d1);
        const float nu_y = (((float)i2) / ((float)p.m2)) * (p.lambda / p.d2);
        const float phi = atan2f(nu_y, nu_x);
        const float nu = sqrtf(nu_x * nu_x + nu_y * nu_y);
        float2 out = make_float2(0.f, 0.f);
        if (nu < p.ObjAp) {
            const float W =
                nu * nu *
                (0.5f * (ab.A1_0 * cosf(2.f * (phi - ab.A1_1)) + ab.C1_0 + defocus_k) +
                 nu * (1.f / 3.f * (ab.A2_0 * cosf(3.f * (phi - ab.A2_1)) + ab.B2_0 * cosf(phi - ab.B2_1)) +
                       nu * (0.25f * (ab.A3_0 * cosf(4.f * (phi - ab.A3_1)) + ab.S3_0 * cosf(2.f * (phi - ab.S3_1)) + ab.C3_0) +
                             nu * (0.2f * (ab.A4_0 * cosf(5.f * (phi - ab.A4_1)) + ab.B4_0 * cosf(phi - ab.B4_1) +
                                           ab.D4_0 * cosf(3.f * (phi - ab.D4_1))) +
                                   nu * (1.f / 6.f *
                                         (ab.A5_0 * cosf(6.f * (phi - ab.A5_1)) + ab.R5_0 * cosf(4.f * (phi - ab.R5_1)) +
                                          ab.S5_0 * cosf(2.f * (phi - ab.S5_1)) + ab.C5_0))))));
            // temporal-coherence envelope (image mode only), then the transfer function envelope exp(-2 pi i W / lambda)
            float envelope = 1.f;
            if (p.mode == 0) {
                const float spread = p.defocspread * nu * nu / p.lambda;
                envelope = expf(-2.f * spread * spread);
            }
            const float waves = W / p.lambda;
            const float ctf_re = envelope * cosf(2.f * PI_F * waves);
            const float ctf_im = envelope * sinf(-2.f * PI_F * waves);
            const float2 v = psi[i];
            out.x = ctf_re * v.x - ctf_im * v.y;
            out.y = ctf_re * v.y + ctf_im * v.x;
        }
        psi[i] = out;
    }
}

// ---- cublasCsscal(1/m12) of applyLensFunction (:621) / sqrt(1/m12) of diffractionPattern
// (src/crystalMaker.cu:716) + intensityValues (src/multisliceSimulation.cu:346-359) + cublasCaxpy
// into the running sum (src/crystalMaker.cu:353,359,365) -----------------------------------------
// (roundings pinned - which product of the sum of squares gets fused is otherwise the compiler's choice per kernel, and the
//  gang kernel below must add exactly what this one adds)
__device__ __forceinline__ float2 intensity_add(float2 a, float2 v, float pre, float alpha)
{
    v.x = __fmul_rn(v.x, pre);
    v.y = __fmul_rn(v.y, pre);
    const float q = __fmaf_rn(v.x, v.x, __fmul_rn(v.y, v.y));
    a.x = __fmaf_rn(alpha, q, a.x);
    a.y = __fmaf_rn(alpha, 0.f, a.y);
    return a;
}
__global__ void kk_intensity_axpy(float2* __restrict__ I, const float2* __restrict__ psi, size_t n, float pre, float alpha)
{
    GS_LOOP(i, n) I[i] = intensity_add(I[i], psi[i], pre, alpha);
}

// the same for the members of a gang in ONE launch: member g adds into slot gp.k[g] of I; a thread owns a pixel and goes
// through the members in order, so members that share a slot (the configurations of one measurement) add exactly as
// their separate launches did
__global__ void kk_intensity_gang(float2* __restrict__ I, const float2* __restrict__ psi, size_t n, float pre, GangPar gp)
{
    GS_LOOP(i, n)
    {
        for (int g = 0; g < gp.n; g++) {
            float2* dst = I + (size_t)gp.k[g] * n + i;
            *dst = intensity_add(*dst, psi[(size_t)g * n + i], pre, gp.f[g]);
        }
    }
}

// ---- tiltBeam_d (src/multisliceSimulation.cu:89-120) -------------------------------------------
__global__ void kk_tilt_beam(float2* __restrict__ psi, KP p, float tb0, float tb1, int flag, size_t gstride, GangPar gp, GangPar gp1)
{
    psi += (size_t)blockIdx.y * gstride;
    if (gp.n) { tb0 = gp.f[blockIdx.y]; tb1 = gp1.f[blockIdx.y]; }
    const size_t n = (size_t)p.m1 * p.m2;
    GS_LOOP(i, n)
    {
        const int i1 = ow((int)(i % (size_t)p.m1), p.m1), i2 = ow((int)(i / (size_t)p.m1), p.m2);
        // phase ramp 2 pi (x tilt_x + y tilt_y) / lambda; flag = -1 takes the tilt out again (diffractionPattern)
        const float signed_lambda = p.lambda * ((float)flag);
        const float ramp_x = ((float)i1) * (p.d1 / signed_lambda) * tb1;
        const float ramp_y = ((float)i2) * (p.d2 / signed_lambda) * tb0;
        const float angle = 2.f * PI_F * (ramp_x + ramp_y);
        const float sn = sinf(angle);
        const float cs = cosf(angle);
        const float2 v = psi[i];
        psi[i] = make_float2(cs * v.x - sn * v.y, sn * v.x + cs * v.y);
    }
}

// ---- taperedCosineWindow_d (src/multisliceSimulation.cu:123-156) --------------------------------
__device__ __forceinline__ float tukey1(int i, int dim, int dn)
{
    float w = 1.f;
    const float alpha = 2.f * (((float)dn) / ((float)dim));
    const float x = ((float)i) / ((float)(dim - 1));
    if (x < alpha * 0.5f) w = 0.5f * (1.f + cosf(PI_F * (2.f * x / alpha - 1.f)));
    else if (x > 1.f - 0.5f * alpha) w = 0.5f * (1.f + cosf(PI_F * (2.f * x / alpha + 1.f - 2.f / alpha)));
    return w;
}
__global__ void kk_tukey(float2* __restrict__ psi, KP p, size_t gstride)
{
    psi += (size_t)blockIdx.y * gstride;
    const size_t n = (size_t)p.m1 * p.m2;
    GS_LOOP(i, n)
    {
        float w = tukey1((int)(i % (size_t)p.m1), p.m1, p.dn1);
        w *= tukey1((int)(i / (size_t)p.m1), p.m2, p.dn2);
        float2 v = psi[i];
        v.x *= w;
        v.y *= w;
        psi[i] = v;
    }
}

// ---- cufftShift2D_h (src/complexMath.cu:510-557): 4 strip kernels x 64 strips -> one out-of-place pass
__global__ void kk_fftshift(float2* __restrict__ out, const float2* __restrict__ in, int m1, int m2, size_t gstride)
{
    out += (size_t)blockIdx.y * gstride;
    in += (size_t)blockIdx.y * gstride;
    const size_t n = (size_t)m1 * m2;
    GS_LOOP(i, n)
    {
        const int i1 = (int)(i % (size_t)m1), i2 = (int)(i / (size_t)m1);
        const int j1 = (i1 < m1 - m1 / 2) ? i1 + m1 / 2 : i1 - (m1 - m1 / 2);
        const int j2 = (i2 < m2 - m2 / 2) ? i2 + m2 / 2 : i2 - (m2 - m2 / 2);
        out[(size_t)j2 * m1 + j1] = in[i];
    }
}

// ---- areaMask + areaWeighting (src/multisliceSimulation.cu:468-510, src/crystalMaker.cu:187-224)
__global__ void kk_mask_filter(float2* __restrict__ psi, KP p, size_t gstride)
{
    psi += (size_t)blockIdx.y * gstride;
    const size_t n = (size_t)p.m1 * p.m2;
    GS_LOOP(i, n)
    {
        const int i1 = (int)(i % (size_t)p.m1), i2 = (int)(i / (size_t)p.m1);
        float w = 1.f;
        if (i1 <= p.dn1 - 1) w *= 0.5f * (1 - cosf(3.1415927f * (float)i1 / (float)p.dn1));
        if (i1 >= p.m1 - p.dn1) w *= 0.5f * (1 - cosf(3.1415927f * (float)(p.m1 - i1) / (float)p.dn1));
        if (i2 <= p.dn2 - 1) w *= 0.5f * (1 - cosf(3.1415927f * (float)i2 / (float)p.dn2));
        if (i2 >= p.m2 - p.dn2) w *= 0.5f * (1 - cosf(3.1415927f * (float)(p.m2 - i2) / (float)p.dn2));
        const float2 v = psi[i];
        psi[i] = make_float2(1.f * (1 - w) + v.x * w, 0.f * (1 - w) + v.y * w);
    }
}

// ---- multiplySpatialIncoherence / ...DP (src/multisliceSimulation.cu:391-442) -------------------
__global__ void kk_spatial(float2* __restrict__ f, KP p, float defocus_k, int dp, size_t gstride, GangPar gp)
{
    f += (size_t)blockIdx.y * gstride;
    if (gp.n) defocus_k = gp.f[blockIdx.y];
    const size_t n = (size_t)p.m1 * p.m2;
    GS_LOOP(i, n)
    {
        const int i1 = iw((int)(i % (size_t)p.m1), p.m1), i2 = iw((int)(i / (size_t)p.m1), p.m2);
        float damp;
        if (!dp) { // image: Gaussian in the scattering angle, width set by the illumination angle and this measurement's defocus
            const float nu_x = (((float)i1) / ((float)p.m1)) * (p.lambda / p.d1);
            const float nu_y = (((float)i2) / ((float)p.m2)) * (p.lambda / p.d2);
            const float nusq = nu_x * nu_x + nu_y * nu_y;
            const float width = PI_F * p.illangle * defocus_k;
            damp = expf(-nusq * width * width);
        } else { // diffraction pattern / CBED: Gaussian in the distance from the origin
            const float x = ((float)i1) * p.d1;
            const float y = ((float)i2) * p.d2;
            const float rsq = x * x + y * y;
            const float width = PI_F * p.illangle / p.lambda;
            damp = expf(-width * width * rsq);
        }
        float2 v = f[i];
        v.x *= damp;
        v.y *= damp;
        f[i] = v;
    }
}

// ---- multiplyMtf (src/multisliceSimulation.cu:362-388) + the cublasCsscal after it (crystalMaker.cu:609)
__global__ void kk_mtf(float2* __restrict__ f, KP p, float alpha, size_t gstride)
{
    f += (size_t)blockIdx.y * gstride;
    const size_t n = (size_t)p.m1 * p.m2;
    GS_LOOP(i, n)
    {
        const int i1 = iw((int)(i % (size_t)p.m1), p.m1), i2 = iw((int)(i / (size_t)p.m1), p.m2);
        // detector response a exp(-c nu) + b exp(-d nu^2) at the frequency nu (in units of the sampling frequency), times the
        // transform of the square pixel, sin u / u per axis
        const float nu_x = ((float)i1) / ((float)p.m1);
        const float nu_y = ((float)i2) / ((float)p.m2);
        const float nu = sqrtf(nu_x * nu_x + nu_y * nu_y);
        const float response = (p.mtfa * expf(-p.mtfc * nu) + p.mtfb * expf(-p.mtfd * nu * nu));
        const float u_x = nu_x * PI_F;
        const float u_y = nu_y * PI_F;
        const float mtf = response * (((sinf(u_x) + FLT_EPSILON) / (u_x + FLT_EPSILON)) * ((sinf(u_y) + FLT_EPSILON) / (u_y + FLT_EPSILON)));
        float2 v = f[i];
        v.x = (v.x * mtf) * alpha;
        v.y = (v.y * mtf) * alpha;
        f[i] = v;
    }
}

// ---- ascombeNoise_d (src/crystalMaker.cu:50-70); deviates from Philox stream 1, key (seed, k, pixel)
__global__ void kk_noise(float2* __restrict__ f, size_t n, float dose, uint32_t seed, uint32_t k, size_t gstride, GangPar gp)
{
    f += (size_t)blockIdx.y * gstride;
    if (gp.n) k = (uint32_t)gp.k[blockIdx.y];
    GS_LOOP(i, n)
    {
        float2 v = f[i];
        const float fi = v.x * dose;
        if (fi > 1e-2f) {
            // a deviate in the variance-stabilised (Anscombe) domain around the transform of the expected count, back to counts
            const float deviate = normal(seed, 1u, k, 0u, (uint32_t)i) * sqrtf(1 - expf(-fi / 0.777134f));
            const float anscombe = deviate + (2.f * sqrtf(fi + 0.375f) - 0.25f / sqrtf(fi));
            float counts = roundf(0.25f * anscombe * anscombe - 0.375f);
            if (counts < FLT_MIN) counts = 0.f;
            v.x = counts / dose;
            f[i] = v;
        }
    }
}

// ---- copyMiddleOut (src/optimFunctions.cu:109-121) ---------------------------------------------
__global__ void kk_crop(float* __restrict__ J, const float2* __restrict__ I, KP p, size_t gstride, GangPar gp)
{
    const size_t n = (size_t)p.n1 * p.n2;
    I += (size_t)blockIdx.y * gstride;
    if (gp.n) J += (size_t)gp.k[blockIdx.y] * n; // image gp.k[member] of the stack
    GS_LOOP(j, n)
    {
        const int i1 = (int)(j % (size_t)p.n1), i2 = (int)(j / (size_t)p.n1);
        J[j] = I[(size_t)(i1 + p.dn1) + (size_t)p.m1 * (i2 + p.dn2)].x;
    }
}

// ---- cublasScnrm2 + Csscal of the CBED probe (src/multisliceSimulation.cu:578-580) -------------
// Deterministic: fixed grid, every block leaves its partial sum (grid-stride order, shuffle tree, four wave sums added
// in order) in part[blockIdx.x], one block then adds the partials in a fixed tree.  (Round 1 added the block sums with a
// float atomicAdd, whose order - and with it the last bits of every CBED image - changed from run to run.)
constexpr int kSumBlocks = 1024;
__global__ void kk_sumsq(const float2* __restrict__ f, size_t n, float* __restrict__ part)
{
    float s = 0.f;
    GS_LOOP(i, n) { const float2 v = f[i]; s += v.x * v.x + v.y * v.y; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ float ws[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) ws[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[1 + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ void kk_sum_partials(float* __restrict__ part, int nblocks)
{
    __shared__ float sh[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += part[1 + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[0] = sh[0];
}
__global__ void kk_scale_by(float2* __restrict__ f, size_t n, float target, const float* __restrict__ acc)
{
    const float a = target / sqrtf(*acc);
    GS_LOOP(i, n) { float2 v = f[i]; v.x *= a; v.y *= a; f[i] = v; }
}

#define LAUNCH(kern, n, st, ...)                                                \
    do {                                                                        \
        hipLaunchKernelGGL(kern, grid_for(n), dim3(256), 0, st, __VA_ARGS__);   \
        return hipGetLastError();                                               \
    } while (0)

hipError_t k_fill(float2* f, size_t n, float re, float im, hipStream_t st) { LAUNCH(kk_fill, n, st, f, n, re, im); }
hipError_t k_fill_noise(float* f, size_t n, unsigned seed, hipStream_t st) { LAUNCH(kk_fill_noise, n, st, f, n, seed); }
hipError_t k_scale(float2* f, size_t n, float a, hipStream_t st) { LAUNCH(kk_scale, n, st, f, n, a); }
hipError_t k_axpy(float2* y, const float2* x, size_t n, float a, hipStream_t st) { LAUNCH(kk_axpy, n, st, y, x, n, a); }
hipError_t k_real_pack(float* d, const float2* s, size_t n, hipStream_t st) { LAUNCH(kk_real_pack, n, st, d, s, n); }
hipError_t k_real_unpack(float2* d, const float* s, size_t n, hipStream_t st) { LAUNCH(kk_real_unpack, n, st, d, s, n); }
hipError_t k_axpy_real(float2* y, const float* x, size_t n, hipStream_t st) { LAUNCH(kk_axpy_real, n, st, y, x, n); }
hipError_t k_filter_accum(float2* Vh, float2* Dh, const KP& p, const Kirk& kz, int first, hipStream_t st)
{
    LAUNCH(kk_filter_accum, (size_t)p.m1 * p.m2, st, Vh, Dh, p, kz, first);
}
hipError_t k_transmit(float2* t, const float2* V, size_t n, hipStream_t st) { LAUNCH(kk_transmit, n, st, t, V, n); }
hipError_t k_transmit_comp(float2* t, const float2* W, size_t n, int comp, float imPot, hipStream_t st) { LAUNCH(kk_transmit_comp, n, st, t, W, n, comp, imPot); }
hipError_t k_filter_accum_tab(float2* Vh, float2* Dh, const float* G, size_t n, int first, hipStream_t st) { LAUNCH(kk_filter_accum_tab, n, st, Vh, Dh, G, n, first); }
hipError_t k_pick_potential(float2* V, const float2* W, size_t n, int comp, float imPot, hipStream_t st) { LAUNCH(kk_pick_potential, n, st, V, W, n, comp, imPot); }
hipError_t k_mask_scale(float2* f, int m1, int m2, float alpha, hipStream_t st)
{
    LAUNCH(kk_mask_scale, (size_t)m1 * m2, st, f, m1, m2, alpha, (size_t)0);
}
hipError_t k_mul(float2* dst, const float2* f0, const float2* f1, size_t n, hipStream_t st) { LAUNCH(kk_mul, n, st, dst, f0, f1, n); }
hipError_t k_build_propagator(float2* P, const KP& p, int transposed, hipStream_t st) { LAUNCH(kk_propagator, (size_t)p.m1 * p.m2, st, P, p, transposed); }
hipError_t k_build_propagator_1d(float2* px, float2* py, const KP& p, int npow, hipStream_t st) { LAUNCH(kk_propagator_1d, (size_t)p.m1 + p.m2, st, px, py, p, npow); }
hipError_t k_build_gtab(float* G, const KP& p, const Kirk& kz, int transposed, int pitch, hipStream_t st) { LAUNCH(kk_gtab, (size_t)p.m1 * p.m2, st, G, p, kz, transposed, pitch); }
hipError_t k_lens(float2* psi, const KP& p, float dk, hipStream_t st) { LAUNCH(kk_lens, (size_t)p.m1 * p.m2, st, psi, p, dk, (size_t)0, GangPar{}); }
// gang launches: grid.y = member
#define LAUNCH_G(kern, n, members, st, ...)                                     \
    do {                                                                        \
        dim3 g_ = grid_for(n);                                                  \
        g_.y = (unsigned)(members);                                             \
        hipLaunchKernelGGL(kern, g_, dim3(256), 0, st, __VA_ARGS__);            \
        return hipGetLastError();                                               \
    } while (0)
hipError_t k_lens_gang(float2* psi, size_t stride, const KP& p, const GangPar& gp, hipStream_t st) { LAUNCH_G(kk_lens, (size_t)p.m1 * p.m2, gp.n, st, psi, p, 0.f, stride, gp); }
hipError_t k_intensity_gang(float2* I, const float2* psi, size_t n, float pre, const GangPar& gp, hipStream_t st) { LAUNCH(kk_intensity_gang, n, st, I, psi, n, pre, gp); }
hipError_t k_spatial_incoherence_gang(float2* f, size_t stride, const KP& p, int dp, const GangPar& gp, hipStream_t st) { LAUNCH_G(kk_spatial, (size_t)p.m1 * p.m2, gp.n, st, f, p, 0.f, dp, stride, gp); }
hipError_t k_mtf_gang(float2* f, size_t stride, int members, const KP& p, float alpha, hipStream_t st) { LAUNCH_G(kk_mtf, (size_t)p.m1 * p.m2, members, st, f, p, alpha, stride); }
hipError_t k_noise_gang(float2* f, size_t stride, size_t n, float dose, uint32_t seed, const GangPar& gp, hipStream_t st) { LAUNCH_G(kk_noise, n, gp.n, st, f, n, dose, seed, 0u, stride, gp); }
hipError_t k_mask_scale_gang(float2* f, size_t stride, int members, int m1, int m2, float alpha, hipStream_t st) { LAUNCH_G(kk_mask_scale, (size_t)m1 * m2, members, st, f, m1, m2, alpha, stride); }
hipError_t k_tilt_beam_gang(float2* psi, size_t stride, const KP& p, const GangPar& tb0, const GangPar& tb1, int flag, hipStream_t st) { LAUNCH_G(kk_tilt_beam, (size_t)p.m1 * p.m2, tb0.n, st, psi, p, 0.f, 0.f, flag, stride, tb0, tb1); }
hipError_t k_tukey_gang(float2* psi, size_t stride, int members, const KP& p, hipStream_t st) { LAUNCH_G(kk_tukey, (size_t)p.m1 * p.m2, members, st, psi, p, stride); }
hipError_t k_fftshift_gang(float2* out, const float2* in, size_t stride, int members, int m1, int m2, hipStream_t st) { LAUNCH_G(kk_fftshift, (size_t)m1 * m2, members, st, out, in, m1, m2, stride); }
hipError_t k_mask_filter_gang(float2* psi, size_t stride, int members, const KP& p, hipStream_t st) { LAUNCH_G(kk_mask_filter, (size_t)p.m1 * p.m2, members, st, psi, p, stride); }
hipError_t k_crop_gang(float* J, const float2* I, size_t stride, const KP& p, const GangPar& gp, hipStream_t st) { LAUNCH_G(kk_crop, (size_t)p.n1 * p.n2, gp.n, st, J, I, p, stride, gp); }
hipError_t k_intensity_axpy(float2* I, const float2* psi, size_t n, float pre, float alpha, hipStream_t st)
{
    LAUNCH(kk_intensity_axpy, n, st, I, psi, n, pre, alpha);
}
hipError_t k_tilt_beam(float2* psi, const KP& p, float tb0, float tb1, int flag, hipStream_t st)
{
    LAUNCH(kk_tilt_beam, (size_t)p.m1 * p.m2, st, psi, p, tb0, tb1, flag, (size_t)0, GangPar{}, GangPar{});
}
hipError_t k_tukey(float2* psi, const KP& p, hipStream_t st) { LAUNCH(kk_tukey, (size_t)p.m1 * p.m2, st, psi, p, (size_t)0); }
hipError_t k_fftshift(float2* out, const float2* in, int m1, int m2, hipStream_t st)
{
    LAUNCH(kk_fftshift, (size_t)m1 * m2, st, out, in, m1, m2, (size_t)0);
}
hipError_t k_mask_filter(float2* psi, const KP& p, hipStream_t st) { LAUNCH(kk_mask_filter, (size_t)p.m1 * p.m2, st, psi, p, (size_t)0); }
hipError_t k_spatial_incoherence(float2* f, const KP& p, float dk, int dp, hipStream_t st)
{
    LAUNCH(kk_spatial, (size_t)p.m1 * p.m2, st, f, p, dk, dp, (size_t)0, GangPar{});
}
hipError_t k_mtf(float2* f, const KP& p, float alpha, hipStream_t st) { LAUNCH(kk_mtf, (size_t)p.m1 * p.m2, st, f, p, alpha, (size_t)0); }
hipError_t k_noise(float2* f, size_t n, float dose, uint32_t seed, int k, hipStream_t st)
{
    LAUNCH(kk_noise, n, st, f, n, dose, seed, (uint32_t)k, (size_t)0, GangPar{});
}
hipError_t k_crop(float* J, const float2* I, const KP& p, hipStream_t st) { LAUNCH(kk_crop, (size_t)p.n1 * p.n2, st, J, I, p, (size_t)0, GangPar{}); }
hipError_t k_normalize_to(float2* f, size_t n, float target, float* scratch, hipStream_t st)
{
    hipError_t e;
    hipLaunchKernelGGL(kk_sumsq, dim3(kSumBlocks), dim3(256), 0, st, f, n, scratch); // scratch: 1 + kSumBlocks floats
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(kk_sum_partials, dim3(1), dim3(256), 0, st, scratch, kSumBlocks);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(kk_scale_by, grid_for(n), dim3(256), 0, st, f, n, target, scratch);
    return hipGetLastError();
}

} // namespace fdes
