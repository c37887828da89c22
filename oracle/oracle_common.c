/* oracle_common.c — precision-independent part of the CPU oracle (TEST INFRASTRUCTURE ONLY,
 * see oracle.h for scope and pinning status).  Parameters, atom geometry and the
 * counter-based RNG.  All geometry is float32 exactly as the reference computes it, also in
 * the float64 variant of the oracle, so that both variants bin atoms identically.
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

static int g_threads = 0;
void oracle_set_threads(int n)
{
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}
int oracle_get_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; Random123 v1.09).  Replaces cuRAND's
 * XORWOW (src/crystalMaker.cu:28-35): the reference's generator is sequential state carried
 * across (k, j), which cannot be sharded; a counter-based generator keyed on
 * (seed, k, j, coordinate) gives the same statistics independent of the GPU count.
 * Pinned by the Random123 known-answer vectors in tests/test_oracle_rng.py.
 * ------------------------------------------------------------------------------------------ */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Deterministic float32 ln(x), x in (0,1], built from IEEE +,-,*,/ and fmaf only, so that the
 * CPU oracle and the HIP kernel (fdes_amd/csrc/rng.h) produce identical bits. */
static float det_logf(float x)
{
    uint32_t b;
    memcpy(&b, &x, 4);
    int e = (int)((b >> 23) & 255u) - 127;
    b = (b & 0x007FFFFFu) | 0x3F800000u;
    float m;
    memcpy(&m, &b, 4);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float q = 0.111111112f;
    q = fmaf(q, z, 0.142857149f);
    q = fmaf(q, z, 0.2f);
    q = fmaf(q, z, 0.333333343f);
    q = fmaf(q, z, 1.0f);
    float lnm = (2.0f * s) * q;
    return fmaf((float)e, 0.693147182f, lnm);
}

/* sin(2*pi*u), u in [0,1), same construction. */
static float det_sin2pi(float u)
{
    float t = u * 4.0f;
    int q = (int)t;
    float f = t - (float)q;
    int swap = 0;
    if (f > 0.5f) { f = 1.0f - f; swap = 1; }
    float y = f * 1.57079637f;
    float y2 = y * y;
    float sp = 2.75573192e-6f;
    sp = fmaf(sp, y2, -1.98412701e-4f);
    sp = fmaf(sp, y2, 8.33333377e-3f);
    sp = fmaf(sp, y2, -1.66666672e-1f);
    sp = fmaf(sp, y2, 1.0f);
    float sn = y * sp;
    float cp = -2.75573199e-7f;
    cp = fmaf(cp, y2, 2.48015876e-5f);
    cp = fmaf(cp, y2, -1.38888892e-3f);
    cp = fmaf(cp, y2, 4.16666679e-2f);
    cp = fmaf(cp, y2, -0.5f);
    cp = fmaf(cp, y2, 1.0f);
    float cs = cp;
    float s_ = swap ? cs : sn; /* sin(f0*pi/2) with f0 the unswapped fraction */
    float c_ = swap ? sn : cs;
    switch (q & 3) {
    case 0: return s_;
    case 1: return c_;
    case 2: return -s_;
    default: return -c_;
    }
}

/* Box-Muller on two 32-bit words (the construction of curand_normal, src/crystalMaker.cu:44,
 * with deterministic log / sin). */
float oracle_det_normal(uint32_t a, uint32_t b)
{
    float u1 = ((float)(a >> 9) + 0.5f) * 1.1920929e-7f; /* (2k+1)/2^24, exact */
    float u2 = (float)(b >> 8) * 5.96046448e-8f;         /* k/2^24, exact      */
    float r = sqrtf(-2.0f * det_logf(u1));
    return r * det_sin2pi(u2);
}

float oracle_normal(uint32_t seed, uint32_t stream, uint32_t k, uint32_t j, uint32_t i)
{
    uint32_t ctr[4] = {i, j, k, stream}, key[2] = {seed, 0x46444553u}, out[4];
    oracle_philox4x32_10(ctr, key, out);
    return oracle_det_normal(out[0], out[1]);
}

/* defaultParams, src/paramStructure.cu:501-598 */
void oracle_params_default(fdes_params* p, int n3)
{
    float *ts = p->tiltspec, *tb = p->tiltbeam, *df = p->defoci;
    int cap = p->cap;
    memset(p, 0, sizeof(*p));
    p->tiltspec = ts; p->tiltbeam = tb; p->defoci = df; p->cap = cap;
    p->E0 = 200e3f; p->gamma = 1.3913902f; p->lambda = 2.507934e-012f; p->sigma = 7288400.5f;
    p->ab.C1_0 = -6.1334e-008f;
    p->ab.C3_0 = 1e-3f;
    p->mtfa = 1.f;
    p->ObjAp = 11.1e-3f;
    p->m1 = 4; p->m2 = 4; p->m3 = 1;
    p->d1 = 0.25e-10f; p->d2 = 0.25e-10f; p->d3 = 2e-10f;
    p->subSlTh = p->d3;
    p->dn1 = 1; p->dn2 = 1; p->n1 = 2; p->n2 = 2; p->n3 = n3;
    for (int i = 0; i < n3 && i < cap; i++) {
        ts[2 * i] = ts[2 * i + 1] = 0.f;
        tb[2 * i] = tb[2 * i + 1] = 0.f;
        df[i] = 0.f;
    }
    strcpy(p->sample_name, "Empty sample");
    strcpy(p->material, "Nothing");
    strcpy(p->user_name, "John Smith");
    strcpy(p->institution, "Europe University");
    strcpy(p->department, "Electron Microscopy Facility");
    strcpy(p->email, "john.smith@uni.eu");
    strcpy(p->comments, "This is FDES's default comment");
}

/* consitentParams, src/paramStructure.cu:637-673 (MATLAB_TILT_COMPATIBILITY == 0, src/FDES.cu:46) */
void oracle_consistent_params(fdes_params* p)
{
    const float E0 = p->E0;
    const float m0 = 9.1093822f, c = 2.9979246f, e = 1.6021766f, h = 6.6260696f;
    const float pi = 3.141592654f; /* allocParams, :700 */
    p->gamma = 1.f + E0 * e / m0 / c / c * 1e-4f;
    p->lambda = h / sqrtf(2.f * m0 * e) * 1e-9f / sqrtf(E0 * (1.f + E0 * e / 2.f / m0 / c / c * 1e-4f));
    p->sigma = 2.f * pi * p->gamma * p->lambda * m0 * e / h / h * 1e18f;
    p->m1 = p->n1 + 2 * p->dn1;
    p->m2 = p->n2 + 2 * p->dn2;
    float flag = 0.f;
    for (int j = 0; j < p->n3 * 2; j++) flag += fabsf(p->tiltbeam[j]);
    p->doBeamTilt = !(flag < (FLT_MIN * ((float)p->n3 * 2)));
}

/* subSliceRatio + setSubSlices, src/crystalMaker.cu:720-743 */
int oracle_sub_slices(fdes_params* p)
{
    float ratio = 1.f;
    if ((p->subSlTh > 1e-12f) && (p->subSlTh < p->d3)) ratio = ceilf(p->d3 / p->subSlTh);
    p->m3 = (int)(((float)p->m3) * ratio);
    p->d3 /= ratio;
    return (int)ratio;
}

/* cublasSrot semantics: x' = c x + s y ; y' = c y - s x (separately rounded products). */
static void srot(float* x, float* y, int n, float c, float s)
{
    for (int i = 0; i < n; i++) {
        float xi = x[3 * i], yi = y[3 * i];
        float cx = c * xi, sy = s * yi, cy = c * yi, sx = s * xi;
        x[3 * i] = cx + sy;
        y[3 * i] = cy - sx;
    }
}

/* tiltCoordinates, src/crystalMaker.cu:427-454 */
void oracle_tilt_coordinates(float* xyz, int nAt, float t_0, float t_1, float t_2)
{
    float s, c;
    if (fabsf(t_2) > FLT_EPSILON) { c = cosf(t_2); s = -sinf(t_2); srot(&xyz[0], &xyz[1], nAt, c, s); }
    if (fabsf(t_1) > FLT_EPSILON) { c = cosf(t_1); s = -sinf(t_1); srot(&xyz[0], &xyz[2], nAt, c, s); }
    if (fabsf(t_0) > FLT_EPSILON) { c = cosf(t_0); s = -sinf(t_0); srot(&xyz[1], &xyz[2], nAt, c, s); }
}

/* atomJitter_d, src/crystalMaker.cu:37-48: xyz[i] += N(0,1) * 0.112539540 * sqrt(DWF[i/3]) */
void oracle_atom_jitter(float* xyz, const float* dwf, int nAt, uint32_t seed, int k, int j)
{
    for (int i = 0; i < 3 * nAt; i++) {
        float x = oracle_normal(seed, 0u, (uint32_t)k, (uint32_t)j, (uint32_t)i);
        float d = (x * 0.112539540f) * sqrtf(dwf[i / 3]);
        xyz[i] = xyz[i] + d;
    }
}

/* listOfElements, src/crystalMaker.cu:539-570 (first-seen order) */
int oracle_list_of_elements(int* Zlist, int nAt, const int* Z)
{
    int nZ = 1;
    for (int j = 0; j < 103; j++) Zlist[j] = 0;
    if (nAt <= 0) return 0;
    Zlist[0] = Z[0];
    for (int j = 1; j < nAt; j++) {
        int flag = 0;
        for (int k = 0; k < nZ; k++)
            if (Zlist[k] == Z[j]) flag = 1;
        if (!flag && nZ < 103) { Zlist[nZ] = Z[j]; nZ++; }
    }
    return nZ;
}

/* src/crystalMaker.cu:282-283, 330-337 */
void oracle_config_coords(const fdes_params* p, const fdes_atoms* a, int k, int j, uint32_t seed,
                          float* xyz)
{
    memcpy(xyz, a->xyz, sizeof(float) * 3 * (size_t)a->nAt);
    oracle_tilt_coordinates(xyz, a->nAt, p->tilt_offset_x, p->tilt_offset_y, p->tilt_offset_z);
    if (k >= 0) oracle_tilt_coordinates(xyz, a->nAt, p->tiltspec[2 * k], p->tiltspec[2 * k + 1], 0.f);
    if (p->frPh > 0 && j >= 0) oracle_atom_jitter(xyz, a->dwf, a->nAt, seed, k, j);
}
