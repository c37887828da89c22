// rng.h — counter-based RNG of the engine (device + host inline).
//
// The reference draws frozen-phonon displacements and detector noise from cuRAND XORWOW states
// that are initialised once and carried sequentially across all (k, j) configurations
// (src/crystalMaker.cu:28-48, 264-265, 291-295): results depend on the order configurations are
// processed in, which forbids sharding them over GPUs.  Here every deviate is a pure function of
// (seed, stream, k, j, element): Philox4x32-10 (Salmon et al., SC'11) + Box-Muller, the
// construction curand_normal uses.  ln and sin are evaluated with explicit fmaf polynomials so
// that host and device produce identical bits (this TU is built with -ffp-contract=off).
#ifndef FDES_RNG_H_
#define FDES_RNG_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdes {

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ inline float bits2f(uint32_t b)
{
    union { uint32_t u; float f; } v;
    v.u = b;
    return v.f;
}
__host__ __device__ inline uint32_t f2bits(float f)
{
    union { uint32_t u; float f; } v;
    v.f = f;
    return v.u;
}

// ln(x) for normal x in (0, 1]
__host__ __device__ inline float det_logf(float x)
{
    uint32_t b = f2bits(x);
    int e = (int)((b >> 23) & 255u) - 127;
    float m = bits2f((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float q = 0.111111112f;
    q = fmaf(q, z, 0.142857149f);
    q = fmaf(q, z, 0.2f);
    q = fmaf(q, z, 0.333333343f);
    q = fmaf(q, z, 1.0f);
    const float lnm = (2.0f * s) * q;
    return fmaf((float)e, 0.693147182f, lnm);
}

// sin(2 pi u), u in [0, 1)
__host__ __device__ inline float det_sin2pi(float u)
{
    const float t = u * 4.0f;
    const int q = (int)t;
    float f = t - (float)q;
    bool swap = false;
    if (f > 0.5f) { f = 1.0f - f; swap = true; }
    const float y = f * 1.57079637f;
    const float y2 = y * y;
    float sp = 2.75573192e-6f;
    sp = fmaf(sp, y2, -1.98412701e-4f);
    sp = fmaf(sp, y2, 8.33333377e-3f);
    sp = fmaf(sp, y2, -1.66666672e-1f);
    sp = fmaf(sp, y2, 1.0f);
    const float sn = y * sp;
    float cp = -2.75573199e-7f;
    cp = fmaf(cp, y2, 2.48015876e-5f);
    cp = fmaf(cp, y2, -1.38888892e-3f);
    cp = fmaf(cp, y2, 4.16666679e-2f);
    cp = fmaf(cp, y2, -0.5f);
    cp = fmaf(cp, y2, 1.0f);
    const float s_ = swap ? cp : sn;
    const float c_ = swap ? sn : cp;
    switch (q & 3) {
    case 0: return s_;
    case 1: return c_;
    case 2: return -s_;
    default: return -c_;
    }
}

__host__ __device__ inline float det_normal(uint32_t a, uint32_t b)
{
    const float u1 = ((float)(a >> 9) + 0.5f) * 1.1920929e-7f;
    const float u2 = (float)(b >> 8) * 5.96046448e-8f;
    const float r = sqrtf(-2.0f * det_logf(u1));
    return r * det_sin2pi(u2);
}

// stream 0: frozen phonons (element = coordinate index), stream 1: detector noise (element = pixel)
__host__ __device__ inline float normal(uint32_t seed, uint32_t stream, uint32_t k, uint32_t j, uint32_t i)
{
    uint32_t o[4];
    philox4x32_10(i, j, k, stream, seed, 0x46444553u, o);
    return det_normal(o[0], o[1]);
}

} // namespace fdes
#endif
