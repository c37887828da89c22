/* oracle_core.c — wave-optics part of the CPU oracle (TEST INFRASTRUCTURE ONLY; scope and
 * pinning status in oracle.h).  Compiled twice: -DORACLE_F64 gives the float64 "truth",
 * otherwise float32 in the reference's operation order.  Every function cites the reference
 * lines it restates (paths relative to the FDES tree).  Layout: idx = i2*m1 + i1, interleaved
 * (re, im) pairs (cufftComplex), include/coordArithmetic.h:36-40.
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#ifdef ORACLE_F64
typedef double R;
#define S(name) name##_f64
#define SIN sin
#define COS cos
#define EXP exp
#define SQRT sqrt
#define ATAN2 atan2
#define FABS fabs
#define ROUND round
#define RC(x) x
#else
typedef float R;
#define S(name) name##_f32
#define SIN sinf
#define COS cosf
#define EXP expf
#define SQRT sqrtf
#define ATAN2 atan2f
#define FABS fabsf
#define ROUND roundf
#define RC(x) x##f
#endif

static const float kirkland[104][12] = {
#include "kirkland_table.inc"
};

/* iwCoordIp / owCoordIp, include/coordArithmetic.h:32-34 */
static inline int iw(int i, int m) { return (i > m / 2) ? i - m : i; }
static inline int ow(int i, int m) { return i - m / 2; }

/* ============================ FFT (own; the reference calls cuFFT) ========================
 * Unnormalised 2-D C2C, forward = exp(-2 pi i ...), as cufftExecC2C.  Stockham
 * autosort, radices 4/2/3/5 + generic primes; twiddles from a double-precision table. */
typedef struct { R re, im; } cx;

typedef struct {
    int n;
    int nf;
    int fac[32];
    cx* w; /* w[t] = exp(-2 pi i t / n) */
} fft_plan;

static void plan_init(fft_plan* pl, int n)
{
    pl->n = n;
    pl->nf = 0;
    int r = n;
    while (r % 4 == 0) { pl->fac[pl->nf++] = 4; r /= 4; }
    while (r % 2 == 0) { pl->fac[pl->nf++] = 2; r /= 2; }
    while (r % 3 == 0) { pl->fac[pl->nf++] = 3; r /= 3; }
    while (r % 5 == 0) { pl->fac[pl->nf++] = 5; r /= 5; }
    for (int f = 7; r > 1; f += 2)
        while (r % f == 0) { pl->fac[pl->nf++] = f; r /= f; }
    pl->w = (cx*)malloc(sizeof(cx) * (size_t)n);
    for (int t = 0; t < n; t++) {
        double a = -2.0 * 3.14159265358979323846 * (double)t / (double)n;
        pl->w[t].re = (R)cos(a);
        pl->w[t].im = (R)sin(a);
    }
}
static void plan_free(fft_plan* pl) { free(pl->w); }

static inline cx cmul(cx a, cx b) { cx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }

/* One length-n transform of x (contiguous), result in x. y is scratch of length n. */
static void fft1d(const fft_plan* pl, cx* x, cx* y, int inverse)
{
    const int N = pl->n;
    int n = N, s = 1;
    cx *in = x, *out = y;
    for (int fi = 0; fi < pl->nf; fi++) {
        const int r = pl->fac[fi];
        const int m = n / r;
        for (int p = 0; p < m; p++) {
            for (int q = 0; q < s; q++) {
                cx a[64];
                if (r <= 64) {
                    for (int j = 0; j < r; j++) a[j] = in[q + s * (p + j * m)];
                }
                for (int k = 0; k < r; k++) {
                    cx b;
                    if (r == 2) {
                        if (k == 0) { b.re = a[0].re + a[1].re; b.im = a[0].im + a[1].im; }
                        else { b.re = a[0].re - a[1].re; b.im = a[0].im - a[1].im; }
                    } else if (r == 4) {
                        /* forward: w4 = -i ; inverse: +i */
                        cx t0 = {a[0].re + a[2].re, a[0].im + a[2].im};
                        cx t1 = {a[0].re - a[2].re, a[0].im - a[2].im};
                        cx t2 = {a[1].re + a[3].re, a[1].im + a[3].im};
                        cx t3 = {a[1].re - a[3].re, a[1].im - a[3].im};
                        cx t3r; /* t3 * (-i) forward, t3 * (+i) inverse */
                        if (!inverse) { t3r.re = t3.im; t3r.im = -t3.re; }
                        else { t3r.re = -t3.im; t3r.im = t3.re; }
                        switch (k) {
                        case 0: b.re = t0.re + t2.re; b.im = t0.im + t2.im; break;
                        case 1: b.re = t1.re + t3r.re; b.im = t1.im + t3r.im; break;
                        case 2: b.re = t0.re - t2.re; b.im = t0.im - t2.im; break;
                        default: b.re = t1.re - t3r.re; b.im = t1.im - t3r.im; break;
                        }
                    } else {
                        b.re = 0; b.im = 0;
                        for (int j = 0; j < r; j++) {
                            cx aj = (r <= 64) ? a[j] : in[q + s * (p + j * m)];
                            cx w = pl->w[(int)(((long long)j * k % r) * (N / r))];
                            if (inverse) w.im = -w.im;
                            cx t = cmul(aj, w);
                            b.re += t.re; b.im += t.im;
                        }
                    }
                    if (k > 0 && p > 0) {
                        cx w = pl->w[(int)(((long long)p * k * s) % N)];
                        if (inverse) w.im = -w.im;
                        b = cmul(b, w);
                    }
                    out[q + s * (r * p + k)] = b;
                }
            }
        }
        n = m;
        s *= r;
        cx* t = in; in = out; out = t;
    }
    if (in != x) memcpy(x, in, sizeof(cx) * (size_t)N);
}

/* plan cache (two lengths) */
static fft_plan g_pl[2];
static int g_pl_n[2] = {0, 0};
static const fft_plan* get_plan(int n, int slot)
{
#pragma omp critical(oracle_plan)
    {
        if (g_pl_n[slot] != n) {
            if (g_pl_n[slot]) plan_free(&g_pl[slot]);
            plan_init(&g_pl[slot], n);
            g_pl_n[slot] = n;
        }
    }
    return &g_pl[slot];
}

/* cufftExecC2C on an m2 x m1 grid (cufftPlan2d(m2, m1), src/paramStructure.cu:678) */
void S(oracle_fft2)(R* f, int m1, int m2, int inverse)
{
    cx* g = (cx*)f;
    const fft_plan* p1 = get_plan(m1, 0);
    const fft_plan* p2 = get_plan(m2, 1);
#pragma omp parallel
    {
        cx* y = (cx*)malloc(sizeof(cx) * (size_t)(m1 > m2 ? m1 : m2) * 9);
#pragma omp for schedule(static)
        for (int i2 = 0; i2 < m2; i2++) fft1d(p1, g + (size_t)i2 * m1, y, inverse);
        /* columns, 8 at a time through a contiguous buffer */
        cx* col = y + (m1 > m2 ? m1 : m2);
#pragma omp for schedule(static)
        for (int b = 0; b < (m1 + 7) / 8; b++) {
            int c0 = b * 8, nc = (m1 - c0 < 8) ? m1 - c0 : 8;
            for (int i2 = 0; i2 < m2; i2++)
                for (int c = 0; c < nc; c++) col[(size_t)c * m2 + i2] = g[(size_t)i2 * m1 + c0 + c];
            for (int c = 0; c < nc; c++) fft1d(p2, col + (size_t)c * m2, y, inverse);
            for (int i2 = 0; i2 < m2; i2++)
                for (int c = 0; c < nc; c++) g[(size_t)i2 * m1 + c0 + c] = col[(size_t)c * m2 + i2];
        }
        free(y);
    }
}

/* ============================== BLAS-1 stand-ins ========================================= */
static void csscal(R* f, size_t n, R alpha) /* cublasCsscal */
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < 2 * n; i++) f[i] *= alpha;
}
static void caxpy(R* y, const R* x, size_t n, R alpha) /* cublasCaxpy, alpha real */
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < 2 * n; i++) y[i] += alpha * x[i];
}
static void initial_values(R* f, size_t n, R re, R im) /* src/complexMath.cu:64-76 */
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) { f[2 * i] = re; f[2 * i + 1] = im; }
}

/* multiplyElementwise, src/complexMath.cu:44-62 (3-multiply product) */
static void multiply_elementwise(R* f0, const R* f1, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        R a = f0[2 * i], b = f0[2 * i + 1], c = f1[2 * i], d = f1[2 * i + 1];
        R k = a * (c + d);
        d *= a + b;
        c *= b - a;
        f0[2 * i] = k - d;
        f0[2 * i + 1] = k + c;
    }
}

/* zeroHighFreq, src/multisliceSimulation.cu:225-250 */
static void zero_high_freq(R* f, int dim1, int dim2)
{
    float mindim = (float)dim1;
    if ((float)dim2 < mindim) mindim = (float)dim2;
#pragma omp parallel for schedule(static)
    for (int j2 = 0; j2 < dim2; j2++)
        for (int j1 = 0; j1 < dim1; j1++) {
            int i1 = iw(j1, dim1), i2 = iw(j2, dim2);
            if (((float)(i1 * i1 + i2 * i2) * 9.f / (mindim * mindim)) > 1.f) {
                size_t i = (size_t)j2 * dim1 + j1;
                f[2 * i] = 0;
                f[2 * i + 1] = 0;
            }
        }
}

/* cufftShift2D_h, src/complexMath.cu:510-557 (= fftshift along dim1 then dim2) */
static void fft_shift2d(R* f, int n1, int n2)
{
    cx* g = (cx*)f;
    cx* t = (cx*)malloc(sizeof(cx) * (size_t)n1 * n2);
    for (int i2 = 0; i2 < n2; i2++)
        for (int i1 = 0; i1 < n1; i1++) {
            int j1 = (i1 < n1 - n1 / 2) ? i1 + n1 / 2 : i1 - (n1 - n1 / 2);
            int j2 = (i2 < n2 - n2 / 2) ? i2 + n2 / 2 : i2 - (n2 - n2 / 2);
            t[(size_t)j2 * n1 + j1] = g[(size_t)i2 * n1 + i1];
        }
    memcpy(g, t, sizeof(cx) * (size_t)n1 * n2);
    free(t);
}

/* bandwidthLimit, src/multisliceSimulation.cu:552-560 */
static void bandwidth_limit(R* f, const fdes_params* p)
{
    S(oracle_fft2)(f, p->m1, p->m2, 0);
    zero_high_freq(f, p->m1, p->m2);
    S(oracle_fft2)(f, p->m1, p->m2, 1);
    const size_t m12 = (size_t)p->m1 * p->m2;
    csscal(f, m12, (R)(1.f / ((float)m12)));
}

/* ============================== projected potential ===================================== */

/* squareAtoms_d, src/crystalMaker.cu:73-134.  Geometry in float32 in both variants. */
static void square_atoms(R* V, const fdes_params* p, int nAt, const int* Z, int Z0, const float* xyz,
                         float imPot, const float* occ, int s)
{
    const int m1 = p->m1, m2 = p->m2, m3 = p->m3;
    for (int i = 0; i < nAt; i++) {
        if (Z[i] != Z0) continue;
        float x1 = xyz[i * 3 + 0] / p->d1 + ((float)m1) * 0.5f - 0.5f;
        float x2 = xyz[i * 3 + 1] / p->d2 + ((float)m2) * 0.5f - 0.5f;
        int i3 = (int)(roundf(xyz[i * 3 + 2] / p->d3 + ((float)m3) * 0.5f - 0.5f));
        if (((x1 > 1.f) && (x1 < ((float)(m1 - 2)))) && ((x2 > 1.f) && (x2 < ((float)(m2 - 2)))) &&
            ((i3 > s - 1) && (i3 <= s))) {
            int i1 = (int)roundf(x1);
            int i2 = (int)roundf(x2);
            float r1f = x1 - ((float)i1);
            float r2f = x2 - ((float)i2);
            R r1 = (R)r1f, r2 = (R)r2f, oc = (R)occ[i], ip = (R)imPot;
            int sg1 = (r1f < 0.f) ? -1 : 1, sg2 = (r2f < 0.f) ? -1 : 1;
            size_t j;
            R temp;
            j = (size_t)i2 * m1 + i1;
            temp = (1 - FABS(r1)) * (1 - FABS(r2)) * oc;
            V[2 * j] += temp; V[2 * j + 1] += temp * ip;
            i2 += sg2;
            j = (size_t)i2 * m1 + i1;
            temp = (1 - FABS(r1)) * FABS(r2) * oc;
            V[2 * j] += temp; V[2 * j + 1] += temp * ip;
            i1 += sg1;
            j = (size_t)i2 * m1 + i1;
            temp = FABS(r1) * FABS(r2) * oc;
            V[2 * j] += temp; V[2 * j + 1] += temp * ip;
            i2 -= sg2;
            j = (size_t)i2 * m1 + i1;
            temp = FABS(r1) * (1 - FABS(r2)) * oc;
            V[2 * j] += temp; V[2 * j + 1] += temp * ip;
        }
    }
}

/* projectedPotential_d, src/projectedPotential.cu:30-73, then divideBySinc,
 * src/crystalMaker.cu:136-158, into V2 (.y = 0). */
static void projected_potential(R* V2, int Z, const fdes_params* p)
{
    const int m1 = p->m1, m2 = p->m2;
    const float* kp = kirkland[(Z >= 1 && Z <= 103) ? Z : 0];
    const R a0 = kp[0], b0 = kp[1], a1 = kp[2], b1 = kp[3], a2 = kp[4], b2 = kp[5];
    const R c0 = kp[6], d0 = kp[7], c1 = kp[8], d1_ = kp[9], c2 = kp[10], d2_ = kp[11];
    const R d1 = RC(1e10) * (R)p->d1;
    const R d2 = RC(1e10) * (R)p->d2;
    const R pi = (R)3.141592654f;
    const R eps = (R)FLT_EPSILON;
#pragma omp parallel for schedule(static)
    for (int j2 = 0; j2 < m2; j2++)
        for (int j1 = 0; j1 < m1; j1++) {
            int i1 = iw(j1, m1), i2 = iw(j2, m2);
            R qsq = ((R)i1) / (d1 * ((R)m1));
            R Vz = ((R)i2) / (d2 * ((R)m2));
            qsq = qsq * qsq + Vz * Vz;
            Vz = a0 / (qsq + b0) + c0 * EXP(-d0 * qsq);
            Vz += a1 / (qsq + b1) + c1 * EXP(-d1_ * qsq);
            Vz += a2 / (qsq + b2) + c2 * EXP(-d2_ * qsq);
            R vx = Vz * (RC(4.78776452e-9) * (R)p->sigma) / (d1 * d2 * ((R)(m1 * m2)));
            /* divideBySinc */
            R y = pi;
            R x = ((R)i1) / ((R)m1) * y;
            x = (x + eps) / (SIN(x) + eps);
            y *= ((R)i2) / ((R)m2);
            x *= (y + eps) / (SIN(y) + eps);
            size_t i = (size_t)j2 * m1 + j1;
            V2[2 * i] = vx * x;
            V2[2 * i + 1] = 0 * x;
        }
}

/* phaseGrating, src/crystalMaker.cu:507-536 */
void S(oracle_phase_grating)(const fdes_params* p, const float* xyz, const int* Z, const float* occ,
                             int nAt, const int* Zlist, int nZ, int s, R* V)
{
    const size_t m12 = (size_t)p->m1 * p->m2;
    R* V1 = (R*)malloc(sizeof(R) * 2 * m12);
    R* V2 = (R*)malloc(sizeof(R) * 2 * m12);
    initial_values(V, m12, 0, 0);
    for (int j = 0; j < nZ; j++) {
        initial_values(V1, m12, 0, 0);
        square_atoms(V1, p, nAt, Z, Zlist[j], xyz, p->imPot, occ, s);
        projected_potential(V2, Zlist[j], p);
        S(oracle_fft2)(V1, p->m1, p->m2, 0);
        /* multiplyWithProjectedPotential_d, src/crystalMaker.cu:160-172 */
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < m12; i++) {
            R v = V2[2 * i];
            V1[2 * i] *= v;
            V1[2 * i + 1] *= v;
        }
        S(oracle_fft2)(V1, p->m1, p->m2, 1);
        caxpy(V, V1, m12, 1);
    }
    free(V1);
    free(V2);
}

/* ============================== propagation ============================================= */

/* fresnelPropagator + fresnelPropagatorDevice, src/multisliceSimulation.cu:594-603, 253-274 */
void S(oracle_fresnel_propagator)(const fdes_params* p, R* frProp)
{
    const int dim1 = p->m1, dim2 = p->m2;
    const R pi = (R)3.141592654f;
#pragma omp parallel for schedule(static)
    for (int j2 = 0; j2 < dim2; j2++)
        for (int j1 = 0; j1 < dim1; j1++) {
            int i1 = iw(j1, dim1), i2 = iw(j2, dim2);
            R d3 = (R)p->d3;
            const R t1 = ((R)(i1) / ((R)dim1)) * (d3 / (R)p->d1);
            const R t2 = ((R)(i2) / ((R)dim2)) * (d3 / (R)p->d2);
            d3 = (R)p->lambda / d3;
            d3 = -pi * (t1 * t1 + t2 * t2) * d3;
            size_t i = (size_t)j2 * dim1 + j1;
            frProp[2 * i] = COS(d3);
            frProp[2 * i + 1] = SIN(d3);
        }
    zero_high_freq(frProp, dim1, dim2);
    const size_t m12 = (size_t)dim1 * dim2;
    csscal(frProp, m12, (R)(1.f / ((float)m12)));
}

/* potential2Transmission, src/multisliceSimulation.cu:41-52 */
static void potential2transmission(R* t, const R* V, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        R Vx = V[2 * i], Vy = V[2 * i + 1];
        t[2 * i] = EXP(-Vy) * COS(Vx);
        t[2 * i + 1] = EXP(-Vy) * SIN(Vx);
    }
}

/* convolveWithFrProp, src/multisliceSimulation.cu:606-611 */
static void convolve_with_frprop(R* psi, const R* frProp, const fdes_params* p)
{
    S(oracle_fft2)(psi, p->m1, p->m2, 0);
    multiply_elementwise(psi, frProp, (size_t)p->m1 * p->m2);
    S(oracle_fft2)(psi, p->m1, p->m2, 1);
}

/* forwardPropagation, src/multisliceSimulation.cu:538-549 */
void S(oracle_forward_propagation)(const fdes_params* p, R* psi, const R* V, R* frProp, R* t)
{
    const size_t m12 = (size_t)p->m1 * p->m2;
    S(oracle_fresnel_propagator)(p, frProp);
    potential2transmission(t, V, m12);
    bandwidth_limit(t, p);
    multiply_elementwise(t, psi, m12);
    convolve_with_frprop(t, frProp, p);
    memcpy(psi, t, sizeof(R) * 2 * m12); /* cublasCcopy */
}

/* The 4-step loop body of BASELINE.json's north_star on given t and P (micro-benchmark unit):
 * multiplyElementwise(t, psi) ; convolveWithFrProp ; copy — src/multisliceSimulation.cu:546-548.
 * psi is updated in place; t is not modified. */
void S(oracle_propagate_unit)(const fdes_params* p, R* psi, const R* t, const R* P)
{
    const size_t m12 = (size_t)p->m1 * p->m2;
    R* w = (R*)malloc(sizeof(R) * 2 * m12);
    memcpy(w, t, sizeof(R) * 2 * m12);
    multiply_elementwise(w, psi, m12);
    convolve_with_frprop(w, P, p);
    memcpy(psi, w, sizeof(R) * 2 * m12);
    free(w);
}

/* ============================== probe / lens / detector ================================= */

/* tiltBeam_d, src/multisliceSimulation.cu:89-120 */
static void tilt_beam(R* psi, int k, const fdes_params* p, int flag)
{
    const int dim1 = p->m1, dim2 = p->m2;
    const R pi = (R)3.141592654f;
#pragma omp parallel for schedule(static)
    for (int j2 = 0; j2 < dim2; j2++)
        for (int j1 = 0; j1 < dim1; j1++) {
            int i1 = ow(j1, dim1), i2 = ow(j2, dim2);
            R x2 = (R)p->lambda * ((R)flag);
            R x1 = ((R)i1) * ((R)p->d1 / x2) * (R)p->tiltbeam[2 * k + 1];
            x2 = ((R)i2) * ((R)p->d2 / x2) * (R)p->tiltbeam[2 * k];
            x1 = RC(2.) * pi * (x1 + x2);
            x2 = SIN(x1);
            x1 = COS(x1);
            size_t i = (size_t)j2 * dim1 + j1;
            R temp = psi[2 * i];
            psi[2 * i] = x1 * temp - x2 * psi[2 * i + 1];
            psi[2 * i + 1] = x2 * temp + x1 * psi[2 * i + 1];
        }
}

/* taperedCosineWindow_d, src/multisliceSimulation.cu:123-156 */
static void tapered_cosine_window(R* psi, const fdes_params* p)
{
    const int dim1 = p->m1, dim2 = p->m2;
    const R pi = (R)3.141592654f;
#pragma omp parallel for schedule(static)
    for (int i2 = 0; i2 < dim2; i2++)
        for (int i1 = 0; i1 < dim1; i1++) {
            R w = 1, alpha, x;
            alpha = RC(2.) * (((R)p->dn1) / ((R)dim1));
            x = ((R)i1) / ((R)(dim1 - 1));
            if (x < alpha * RC(0.5)) w = RC(0.5) * (1 + COS(pi * (RC(2.) * x / alpha - 1)));
            else if (x > 1 - RC(0.5) * alpha)
                w = RC(0.5) * (1 + COS(pi * (RC(2.) * x / alpha + 1 - RC(2.) / alpha)));
            alpha = RC(2.) * (((R)p->dn2) / ((R)dim2));
            x = ((R)i2) / ((R)(dim2 - 1));
            if (x < alpha * RC(0.5)) w *= RC(0.5) * (1 + COS(pi * (RC(2.) * x / alpha - 1)));
            else if (x > 1 - RC(0.5) * alpha)
                w *= RC(0.5) * (1 + COS(pi * (RC(2.) * x / alpha + 1 - RC(2.) / alpha)));
            size_t i = (size_t)i2 * dim1 + i1;
            psi[2 * i] *= w;
            psi[2 * i + 1] *= w;
        }
}

/* multiplyLensFunction, src/multisliceSimulation.cu:277-343 */
static void multiply_lens_function(R* psi, int k, const fdes_params* p)
{
    const int dim1 = p->m1, dim2 = p->m2;
    const fdes_aberration* ab = &p->ab;
    const R lambda = (R)p->lambda;
#pragma omp parallel for schedule(static)
    for (int j2 = 0; j2 < dim2; j2++)
        for (int j1 = 0; j1 < dim1; j1++) {
            int i1 = iw(j1, dim1), i2 = iw(j2, dim2);
            i2 = -i2;
            R nu = (((R)i1) / ((R)dim1)) * (lambda / (R)p->d1);
            R nu2 = (((R)i2) / ((R)dim2)) * (lambda / (R)p->d2);
            R phi = ATAN2(nu2, nu);
            nu = SQRT(nu * nu + nu2 * nu2);
            size_t i = (size_t)j2 * dim1 + j1;
            if (nu < (R)p->ObjAp) {
                R W = nu * nu *
                      (RC(0.5) * ((R)ab->A1_0 * COS(RC(2.) * (phi - (R)ab->A1_1)) + (R)ab->C1_0 + (R)p->defoci[k]) +
                       nu * (RC(1.) / RC(3.) * ((R)ab->A2_0 * COS(RC(3.) * (phi - (R)ab->A2_1)) +
                                             (R)ab->B2_0 * COS(phi - (R)ab->B2_1)) +
                             nu * (RC(0.25) * ((R)ab->A3_0 * COS(RC(4.) * (phi - (R)ab->A3_1)) +
                                            (R)ab->S3_0 * COS(RC(2.) * (phi - (R)ab->S3_1)) + (R)ab->C3_0) +
                                   nu * (RC(0.2) * ((R)ab->A4_0 * COS(RC(5.) * (phi - (R)ab->A4_1)) +
                                                 (R)ab->B4_0 * COS(phi - (R)ab->B4_1) +
                                                 (R)ab->D4_0 * COS(RC(3.) * (phi - (R)ab->D4_1))) +
                                         nu * (RC(1.) / RC(6.) *
                                               ((R)ab->A5_0 * COS(RC(6.) * (phi - (R)ab->A5_1)) +
                                                (R)ab->R5_0 * COS(RC(4.) * (phi - (R)ab->R5_1)) +
                                                (R)ab->S5_0 * COS(RC(2.) * (phi - (R)ab->S5_1)) + (R)ab->C5_0))))));
                nu2 = lambda;
                R damp = 1;
                if (p->mode == 0) {
                    damp = (R)p->defocspread * nu * nu / nu2;
                    damp = EXP(RC(-2.) * damp * damp);
                }
                nu = (R)3.141592654f;
                phi = damp * COS(RC(2.) * nu * (W / nu2));
                damp = damp * SIN(RC(-2.) * nu * (W / nu2));
                nu = psi[2 * i];
                nu2 = psi[2 * i + 1];
                psi[2 * i] = phi * nu - damp * nu2;
                psi[2 * i + 1] = phi * nu2 + damp * nu;
            } else {
                psi[2 * i] = 0;
                psi[2 * i + 1] = 0;
            }
        }
}

/* cublasScnrm2 */
static R scnrm2(const R* f, size_t n)
{
    double acc = 0;
    for (size_t i = 0; i < 2 * n; i++) acc += (double)f[i] * (double)f[i];
    return (R)sqrt(acc);
}

/* incomingWave, src/multisliceSimulation.cu:563-591 */
void S(oracle_incoming_wave)(const fdes_params* p, int k, R* psi)
{
    const size_t m12 = (size_t)p->m1 * p->m2;
    initial_values(psi, m12, 1, 0);
    if (p->mode == 2) {
        multiply_lens_function(psi, k, p);
        S(oracle_fft2)(psi, p->m1, p->m2, 1);
        fft_shift2d(psi, p->m1, p->m2);
        bandwidth_limit(psi, p);
        R alpha = scnrm2(psi, m12);
        alpha = (R)sqrtf((float)(p->n1 * p->n2)) / alpha;
        csscal(psi, m12, alpha);
    }
    if (p->doBeamTilt) tilt_beam(psi, k, p, 1);
    if (p->doBeamTilt && ((p->mode == 0) || (p->mode == 1))) {
        tapered_cosine_window(psi, p);
        bandwidth_limit(psi, p);
    }
}

/* applyLensFunction, src/multisliceSimulation.cu:614-622 */
void S(oracle_apply_lens)(const fdes_params* p, int k, R* psi)
{
    const size_t m12 = (size_t)p->m1 * p->m2;
    S(oracle_fft2)(psi, p->m1, p->m2, 0);
    multiply_lens_function(psi, k, p);
    S(oracle_fft2)(psi, p->m1, p->m2, 1);
    csscal(psi, m12, (R)(1.f / ((float)m12)));
}

/* intensityValues, src/multisliceSimulation.cu:346-359 */
static void intensity_values(R* psi, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        R re = psi[2 * i], im = psi[2 * i + 1];
        psi[2 * i] = re * re + im * im;
        psi[2 * i + 1] = 0;
    }
}

/* areaMask + areaWeighting (applyMaskFiltering), src/multisliceSimulation.cu:468-510,
 * src/crystalMaker.cu:187-224: psi <- 1*(1-mask) + psi*mask */
static void apply_mask_filtering(R* psi, const fdes_params* p)
{
    const int dim1 = p->m1, dim2 = p->m2, dn1 = p->dn1, dn2 = p->dn2;
    const R pi7 = (R)3.1415927f;
#pragma omp parallel for schedule(static)
    for (int i2 = 0; i2 < dim2; i2++)
        for (int i1 = 0; i1 < dim1; i1++) {
            R w = 1;
            if (i1 <= dn1 - 1) w *= RC(0.5) * (1 - COS(pi7 * (R)i1 / (R)dn1));
            if (i1 >= dim1 - dn1) w *= RC(0.5) * (1 - COS(pi7 * (R)(dim1 - i1) / (R)dn1));
            if (i2 <= dn2 - 1) w *= RC(0.5) * (1 - COS(pi7 * (R)i2 / (R)dn2));
            if (i2 >= dim2 - dn2) w *= RC(0.5) * (1 - COS(pi7 * (R)(dim2 - i2) / (R)dn2));
            size_t i = (size_t)i2 * dim1 + i1;
            R re = 1 * (1 - w) + psi[2 * i] * w;
            R im = 0 * (1 - w) + psi[2 * i + 1] * w;
            psi[2 * i] = re;
            psi[2 * i + 1] = im;
        }
}

/* diffractionPattern, src/crystalMaker.cu:700-718 */
void S(oracle_diffraction_pattern)(const fdes_params* p, int k, R* psi)
{
    const size_t m12 = (size_t)p->m1 * p->m2;
    R alpha = (R)sqrtf(1.f / ((float)m12));
    if (p->doBeamTilt) tilt_beam(psi, k, p, -1);
    if (p->mode == 1) {
        apply_mask_filtering(psi, p);
        bandwidth_limit(psi, p);
    }
    S(oracle_fft2)(psi, p->m1, p->m2, 0);
    fft_shift2d(psi, p->m1, p->m2);
    csscal(psi, m12, alpha);
    intensity_values(psi, m12);
}

/* multiplySpatialIncoherence / ...DP / multiplyMtf, src/multisliceSimulation.cu:391-442, 362-388 */
static void spatial_incoherence(R* psi, int k, const fdes_params* p, int dp)
{
    const int dim1 = p->m1, dim2 = p->m2;
    const R pi = (R)3.141592654f;
#pragma omp parallel for schedule(static)
    for (int j2 = 0; j2 < dim2; j2++)
        for (int j1 = 0; j1 < dim1; j1++) {
            int i1 = iw(j1, dim1), i2 = iw(j2, dim2);
            size_t i = (size_t)j2 * dim1 + j1;
            R damp;
            if (!dp) {
                damp = (R)p->lambda;
                R nusq = (((R)i1) / ((R)dim1)) * (damp / (R)p->d1);
                damp = (((R)i2) / ((R)dim2)) * (damp / (R)p->d2);
                nusq = nusq * nusq + damp * damp;
                damp = pi * (R)p->illangle * (R)p->defoci[k];
                damp = EXP(-nusq * damp * damp);
            } else {
                R x1 = ((R)i1) * (R)p->d1;
                R x2 = ((R)i2) * (R)p->d2;
                x1 = x1 * x1 + x2 * x2;
                x2 = pi * (R)p->illangle / (R)p->lambda;
                damp = EXP(-x2 * x2 * x1);
            }
            psi[2 * i] *= damp;
            psi[2 * i + 1] *= damp;
        }
}

static void multiply_mtf(R* psi, const fdes_params* p)
{
    const int dim1 = p->m1, dim2 = p->m2;
    const R pi = (R)3.141592654f, eps = (R)FLT_EPSILON;
#pragma omp parallel for schedule(static)
    for (int j2 = 0; j2 < dim2; j2++)
        for (int j1 = 0; j1 < dim1; j1++) {
            int i1 = iw(j1, dim1), i2 = iw(j2, dim2);
            R nu1 = ((R)i1) / ((R)dim1);
            R nu2 = ((R)i2) / ((R)dim2);
            R mtf = SQRT(nu1 * nu1 + nu2 * nu2);
            mtf = ((R)p->mtfa * EXP(-(R)p->mtfc * mtf) + (R)p->mtfb * EXP(-(R)p->mtfd * mtf * mtf));
            nu1 *= pi;
            nu2 *= pi;
            mtf *= ((SIN(nu1) + eps) / (nu1 + eps)) * ((SIN(nu2) + eps) / (nu2 + eps));
            size_t i = (size_t)j2 * dim1 + j1;
            psi[2 * i] *= mtf;
            psi[2 * i + 1] *= mtf;
        }
}

/* ascombeNoise_d, src/crystalMaker.cu:50-70; normal deviates from Philox stream 1 keyed on
 * (seed = 1 + n3 (:295), k, pixel). */
static void ascombe_noise(R* f, float dose, size_t size, int k, int n3)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < size; i++) {
        R fi = f[2 * i] * (R)dose;
        if (fi > (R)1e-2f) {
            R x = (R)oracle_normal((uint32_t)(1 + n3), 1u, (uint32_t)k, 0u, (uint32_t)i);
            x *= SQRT(1 - EXP(-fi / (R)0.777134f));
            x += RC(2.) * SQRT(fi + (R)0.375f) - RC(0.25) / SQRT(fi);
            x = ROUND(RC(0.25) * x * x - (R)0.375f);
            if (x < (R)FLT_MIN) x = 0;
            f[2 * i] = x / (R)dose;
        }
    }
}

/* addNoiseAndMtf, src/crystalMaker.cu:579-613 + copyMiddleOut, src/optimFunctions.cu:109-121.
 * I: m12 complex (modified); J: n1*n2 reals of measurement k. */
void S(oracle_add_noise_and_mtf)(const fdes_params* p, int k, R* I, R* J)
{
    const int m1 = p->m1, m2 = p->m2;
    const size_t m12 = (size_t)m1 * m2;
    const R alpha = (R)(1.f / ((float)(m1 * m2)));
    const float dose = p->pD;
    S(oracle_fft2)(I, m1, m2, 0);
    if (fabsf(p->illangle) > FLT_EPSILON) {
        if (p->mode == 0) spatial_incoherence(I, k, p, 0);
        if ((p->mode == 1) || (p->mode == 2)) spatial_incoherence(I, k, p, 1);
    }
    if (dose > FLT_EPSILON) {
        csscal(I, m12, alpha);
        S(oracle_fft2)(I, m1, m2, 1);
        ascombe_noise(I, dose, m12, k, p->n3);
        S(oracle_fft2)(I, m1, m2, 0);
    }
    multiply_mtf(I, p);
    csscal(I, m12, alpha);
    S(oracle_fft2)(I, m1, m2, 1);
    for (int i2 = 0; i2 < p->n2; i2++)
        for (int i1 = 0; i1 < p->n1; i1++)
            J[(size_t)i2 * p->n1 + i1] = I[2 * ((size_t)(i1 + p->dn1) + (size_t)m1 * (i2 + p->dn2))];
}

/* ============================== driver ================================================== */

/* body of the k loop of buildMeasurements, src/crystalMaker.cu:324-373: the configurations j of measurement k, their
 * average (alpha = 1 / count, :302-304) and the detector chain.  p: already sub-sliced. */
static int one_measurement(const fdes_params* p, const fdes_atoms* a, int k, uint32_t seed, R* psi, R* I, R* ew, R* image_k, R* exitwave_k)
{
    const size_t m12 = (size_t)p->m1 * p->m2;
    const int count = (p->frPh > 0) ? p->frPh : 1;
    const R alpha = (R)(1.f / ((float)count));
    int nprop = 0;
    initial_values(I, m12, 0, 0);
    initial_values(ew, m12, 0, 0);
    for (int j = 0; j < count; j++) {
        S(oracle_wave)(p, a, k, j, seed, p->m3, psi);
        nprop += p->m3;
        if (exitwave_k) caxpy(ew, psi, m12, alpha);
        if (p->mode == 0) {
            S(oracle_apply_lens)(p, k, psi);
            intensity_values(psi, m12);
            caxpy(I, psi, m12, alpha);
        } else {
            S(oracle_diffraction_pattern)(p, k, psi);
            caxpy(I, psi, m12, alpha);
        }
    }
    if (exitwave_k) memcpy(exitwave_k, ew, sizeof(R) * 2 * m12);
    S(oracle_add_noise_and_mtf)(p, k, I, image_k);
    return nprop;
}

/* ONE measurement k of the series (the same loop body; every random stream is keyed on (k, j), so the result is the
 * k-th image of oracle_build_measurements): image_k n1*n2.  NOTE: dose noise (pD > 0) advances a per-run state across k
 * in the full driver (src/crystalMaker.cu:295, 603) - use this entry for pD = 0 only. */
int S(oracle_measurement)(const fdes_params* p0, const fdes_atoms* a, int k, uint32_t seed, R* image_k)
{
    fdes_params ps = *p0;
    (void)oracle_sub_slices(&ps);
    const size_t m12 = (size_t)ps.m1 * ps.m2;
    R* psi = (R*)malloc(sizeof(R) * 2 * m12);
    R* I = (R*)malloc(sizeof(R) * 2 * m12);
    R* ew = (R*)malloc(sizeof(R) * 2 * m12);
    const int n = one_measurement(&ps, a, k, seed, psi, I, ew, image_k, NULL);
    free(psi); free(I); free(ew);
    return n;
}



/* Wave of configuration (k, j) after `nslices` sub-slices: src/crystalMaker.cu:334-344.
 * `p` must already be sub-sliced. */
void S(oracle_wave)(const fdes_params* p, const fdes_atoms* a, int k, int j, uint32_t seed, int nslices,
                    R* psi)
{
    const size_t m12 = (size_t)p->m1 * p->m2;
    int Zlist[103];
    int nZ = oracle_list_of_elements(Zlist, a->nAt, a->Z);
    float* xyz = (float*)malloc(sizeof(float) * 3 * (size_t)a->nAt);
    oracle_config_coords(p, a, k, j, seed, xyz);
    R* V = (R*)malloc(sizeof(R) * 2 * m12);
    R* t = (R*)malloc(sizeof(R) * 2 * m12);
    R* fr = (R*)malloc(sizeof(R) * 2 * m12);
    S(oracle_incoming_wave)(p, k, psi);
    for (int s = 0; s < nslices && s < p->m3; s++) {
        S(oracle_phase_grating)(p, xyz, a->Z, a->occ, a->nAt, Zlist, nZ, s, V);
        S(oracle_forward_propagation)(p, psi, V, fr, t);
    }
    free(V); free(t); free(fr); free(xyz);
}

/* buildMeasurements, src/crystalMaker.cu:227-424.  `p0` is the consistent parameter set
 * BEFORE sub-slicing.  image: n1*n2*n3; potential (optional): 2*m12*m3_original;
 * exitwave (optional): 2*m12*n3. Returns the number of slice propagations performed. */
int S(oracle_build_measurements)(const fdes_params* p0, const fdes_atoms* a, uint32_t seed, R* image,
                                 R* potential, R* exitwave)
{
    fdes_params ps = *p0;
    fdes_params* p = &ps;
    const int m3_orig = p->m3;
    const float d3_orig = p->d3;
    int ratio = oracle_sub_slices(p);
    const size_t m12 = (size_t)p->m1 * p->m2;
    const size_t n12 = (size_t)p->n1 * p->n2;
    R* psi = (R*)malloc(sizeof(R) * 2 * m12);
    R* I = (R*)malloc(sizeof(R) * 2 * m12);
    R* ew = (R*)malloc(sizeof(R) * 2 * m12);
    int nprop = 0;
    for (int k = 0; k < p->n3; k++)
        nprop += one_measurement(p, a, k, seed, psi, I, ew, image + n12 * (size_t)k, exitwave ? exitwave + 2 * m12 * (size_t)k : NULL);
    if (potential) {
        /* src/crystalMaker.cu:381-397: un-jittered, tilt-offset-only potential per ORIGINAL
         * slice.  (The reference leaves the buffer uninitialised when ratio==1, frPh==0 and the
         * last specimen tilt is zero; here it is always computed.) */
        fdes_params po = *p;
        {   /* setSubSlices(params, 1.f / subSlTh), src/crystalMaker.cu:387 */
            const float inv = 1.f / (float)ratio;
            po.m3 = (int)(((float)p->m3) * inv);
            po.d3 = p->d3 / inv;
            (void)m3_orig; (void)d3_orig;
        }
        int Zlist[103];
        int nZ = oracle_list_of_elements(Zlist, a->nAt, a->Z);
        float* xyz = (float*)malloc(sizeof(float) * 3 * (size_t)a->nAt);
        oracle_config_coords(&po, a, -1, -1, seed, xyz);
        for (int s = 0; s < po.m3; s++)
            S(oracle_phase_grating)(&po, xyz, a->Z, a->occ, a->nAt, Zlist, nZ, s, potential + 2 * m12 * (size_t)s);
        free(xyz);
    }
    free(psi); free(I); free(ew);
    return nprop;
}
