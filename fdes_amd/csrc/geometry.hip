// geometry.hip — atom geometry on the GPU: tilts, frozen-phonon jitter, slice binning, deposit.
//
// Built with -ffp-contract=off: every product/sum below rounds separately, exactly like the CPU
// oracle, so atom coordinates, slice indices and pixel indices are bit-identical to it (the
// wave optics downstream is compared within a float32 tolerance instead).
//
// Reference functions restated here (paths relative to the FDES tree):
//   tiltCoordinates      src/crystalMaker.cu:427-454  (cublasSrot with c = cos t, s = -sin t)
//   atomJitter(_d)       src/crystalMaker.cu:37-48, 456-462
//   squareAtoms_d        src/crystalMaker.cu:73-134
// Design difference: the reference re-scans ALL atoms for every (slice, species) launch
// (O(nAt * nZ * m3) threads per configuration).  Here atoms are binned once per configuration:
// key = slice * nZ + species, a stable radix sort (rocPRIM) makes every (slice, species) a
// contiguous segment, and the per-slice deposit touches only its own atoms.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "geometry.h"

#include <atomic>
#include "rng.h"

namespace fdes {

__global__ void k_srot(float* __restrict__ xyz, int nAt, int ax, int ay, float c, float s)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nAt; i += gridDim.x * blockDim.x) {
        const float x = xyz[3 * i + ax], y = xyz[3 * i + ay];
        const float cx = c * x, sy = s * y, cy = c * y, sx = s * x;
        xyz[3 * i + ax] = cx + sy;
        xyz[3 * i + ay] = cy - sx;
    }
}

__global__ void k_jitter(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ dwf, int n3,
                         uint32_t seed, uint32_t k, uint32_t j)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += gridDim.x * blockDim.x) {
        const float x = normal(seed, 0u, k, j, (uint32_t)i);
        const float d = (x * 0.112539540f) * sqrtf(dwf[i / 3]); // 1/(pi sqrt 8)
        out[i] = in[i] + d;
    }
}

// Which slices of configuration (k, j) hold an atom that the deposit would use - WITHOUT the binning, from the plan's constant
// tilt-offset coordinates: the tilt of measurement k (k_srot's arithmetic in tiltCoordinates' axis order: (x, z) by t_1, then
// (y, z) by t_0), the frozen-phonon displacement of (k, j) (k_jitter's) and the slice / border test of k_atom_keys, fused per
// atom, so that flags[slice] is set exactly where the binning of that configuration will find atoms.  The engine runs it on
// a small stream of its own: the question "which slices are empty" (option skip_empty) then never waits for the slice loops
// queued on the lane's stream.
__global__ void k_slice_occupancy(const float* __restrict__ xyz0, const float* __restrict__ dwf, int nAt, BinGeom g, int rot1, float c1, float s1,
                                  int rot0, float c0, float s0, int jitter, uint32_t seed, uint32_t k, uint32_t j, int* __restrict__ flags)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nAt; i += gridDim.x * blockDim.x) {
        float x = xyz0[3 * i + 0], y = xyz0[3 * i + 1], z = xyz0[3 * i + 2];
        if (rot1) {
            const float cx = c1 * x, sz = s1 * z, cz = c1 * z, sx = s1 * x;
            x = cx + sz;
            z = cz - sx;
        }
        if (rot0) {
            const float cy = c0 * y, sz = s0 * z, cz = c0 * z, sy = s0 * y;
            y = cy + sz;
            z = cz - sy;
        }
        if (jitter) {
            const float sd = sqrtf(dwf[i]);
            x = x + (normal(seed, 0u, k, j, (uint32_t)(3 * i + 0)) * 0.112539540f) * sd;
            y = y + (normal(seed, 0u, k, j, (uint32_t)(3 * i + 1)) * 0.112539540f) * sd;
            z = z + (normal(seed, 0u, k, j, (uint32_t)(3 * i + 2)) * 0.112539540f) * sd;
        }
        const float x1 = x / g.d1 + ((float)g.m1) * 0.5f - 0.5f;
        const float x2 = y / g.d2 + ((float)g.m2) * 0.5f - 0.5f;
        const float z3 = roundf(z / g.d3 + ((float)g.m3) * 0.5f - 0.5f);
        const bool inside = (x1 > 1.f) && (x1 < (float)(g.m1 - 2)) && (x2 > 1.f) && (x2 < (float)(g.m2 - 2));
        if (inside && z3 >= 0.f && z3 < (float)g.m3) flags[(int)z3] = 1; // (every writer stores the same value)
    }
}

// key = ((i3 * nZ + species) * m2 + i2) for atoms that squareAtoms_d would deposit in some slice (i2 = nearest
// row), else nq * m2 (sorted to the end).  Also records what the deposit needs.
__global__ void k_atom_keys(const float* __restrict__ xyz, const uint8_t* __restrict__ spec, const float* __restrict__ occ, int nAt,
                            BinGeom g, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, AtomRec* __restrict__ recs)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nAt; i += gridDim.x * blockDim.x) {
        const float x1 = xyz[i * 3 + 0] / g.d1 + ((float)g.m1) * 0.5f - 0.5f;
        const float x2 = xyz[i * 3 + 1] / g.d2 + ((float)g.m2) * 0.5f - 0.5f;
        const float z3 = roundf(xyz[i * 3 + 2] / g.d3 + ((float)g.m3) * 0.5f - 0.5f);
        uint32_t key = (uint32_t)(g.m3 * g.nZ) * (uint32_t)g.m2;
        const bool inside = (x1 > 1.f) && (x1 < (float)(g.m1 - 2)) && (x2 > 1.f) && (x2 < (float)(g.m2 - 2));
        const int i1 = (int)roundf(x1), i2 = (int)roundf(x2);
        if (inside && z3 >= 0.f && z3 < (float)g.m3) key = (uint32_t)((int)z3 * g.nZ + (int)spec[i]) * (uint32_t)g.m2 + (uint32_t)i2;
        keys[i] = key;
        vals[i] = (uint32_t)i;
        AtomRec r;
        r.i1 = i1; r.i2 = i2; r.r1 = x1 - (float)i1; r.r2 = x2 - (float)i2; r.occ = occ[i]; r.pad = 0;
        recs[i] = r;
    }
}

__global__ void k_gather_recs(AtomRec* __restrict__ out, const AtomRec* __restrict__ in, const uint32_t* __restrict__ order, int nAt)
{
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < nAt; p += gridDim.x * blockDim.x) out[p] = in[order[p]];
}

__device__ inline int lower_bound_u32(const uint32_t* __restrict__ sorted, int n, uint32_t key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// seg[q] = first sorted position of (slice, species) q = 0..nq (lower bound of q * m2)
__global__ void k_seg_bounds(const uint32_t* __restrict__ sorted, int nAt, int nq, int m2, int* __restrict__ seg)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nq) return;
    seg[q] = lower_bound_u32(sorted, nAt, (uint32_t)q * (uint32_t)m2);
}

// rowstart[q * (m2 + 1) + row] = first sorted position with key >= q * m2 + row, row = 0..m2
__global__ void k_row_starts(const uint32_t* __restrict__ sorted, int nAt, int nq, int m2, int* __restrict__ rowstart)
{
    const size_t n = (size_t)nq * (size_t)(m2 + 1);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const uint32_t q = (uint32_t)(e / (size_t)(m2 + 1)), row = (uint32_t)(e % (size_t)(m2 + 1));
        rowstart[e] = lower_bound_u32(sorted, nAt, q * (uint32_t)m2 + row);
    }
}

// ---- the same steps for the n members of a gang in one launch each (engine.hip).  Member g's atoms are
// [g nAt, (g + 1) nAt) of every per-atom array; its keys are offset by g KS, KS = (nq + 1) m2 (one more than a member's
// largest key), so ONE stable sort leaves every member's records in its own range, in the order its own sort gives.

struct GangGeo { int n; float c1[16], s1[16], c0[16], s0[16]; uint32_t k[16], j[16]; };

// tiltCoordinates per member (geom_srot about axis (0, 2) with (c1, s1), then (1, 2) with (c0, s0); an angle within
// FLT_EPSILON of zero is skipped as there: flags in bit 0 / 1 of do_rot[g]) from the common coordinates `in`
__global__ void k_tilt_gang(float* __restrict__ out, const float* __restrict__ in, int nAt, GangGeo gg, unsigned rot1_mask, unsigned rot0_mask)
{
    const int g = blockIdx.y;
    float* __restrict__ o = out + (size_t)g * 3 * (size_t)nAt;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nAt; i += gridDim.x * blockDim.x) {
        float x = in[3 * i + 0], y = in[3 * i + 1], z = in[3 * i + 2];
        if ((rot1_mask >> g) & 1u) { // axes (0, 2)
            const float c = gg.c1[g], s = gg.s1[g];
            const float cx = c * x, sy = s * z, cy = c * z, sx = s * x;
            x = cx + sy;
            z = cy - sx;
        }
        if ((rot0_mask >> g) & 1u) { // axes (1, 2)
            const float c = gg.c0[g], s = gg.s0[g];
            const float cx = c * y, sy = s * z, cy = c * z, sx = s * y;
            y = cx + sy;
            z = cy - sx;
        }
        o[3 * i + 0] = x; o[3 * i + 1] = y; o[3 * i + 2] = z;
    }
}

// atomJitter per member: out_g = in_g + displacement(seed, k_g, j_g); in_stride = 0: every member starts from `in`
__global__ void k_jitter_gang(float* __restrict__ out, const float* __restrict__ in, size_t in_stride, const float* __restrict__ dwf, int n3,
                              uint32_t seed, GangGeo gg)
{
    const int g = blockIdx.y;
    float* __restrict__ o = out + (size_t)g * (size_t)n3;
    const float* __restrict__ src = in + (size_t)g * in_stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += gridDim.x * blockDim.x) {
        const float x = normal(seed, 0u, gg.k[g], gg.j[g], (uint32_t)i);
        const float d = (x * 0.112539540f) * sqrtf(dwf[i / 3]); // 1/(pi sqrt 8)
        o[i] = src[i] + d;
    }
}

__global__ void k_atom_keys_gang(const float* __restrict__ xyz, const uint8_t* __restrict__ spec, const float* __restrict__ occ, int nAt,
                                 BinGeom g, uint32_t KS, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, AtomRec* __restrict__ recs)
{
    const int m = blockIdx.y;
    const size_t base = (size_t)m * (size_t)nAt;
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < nAt; a += gridDim.x * blockDim.x) {
        const size_t i = base + (size_t)a;
        const float x1 = xyz[i * 3 + 0] / g.d1 + ((float)g.m1) * 0.5f - 0.5f;
        const float x2 = xyz[i * 3 + 1] / g.d2 + ((float)g.m2) * 0.5f - 0.5f;
        const float z3 = roundf(xyz[i * 3 + 2] / g.d3 + ((float)g.m3) * 0.5f - 0.5f);
        uint32_t key = (uint32_t)(g.m3 * g.nZ) * (uint32_t)g.m2;
        const bool inside = (x1 > 1.f) && (x1 < (float)(g.m1 - 2)) && (x2 > 1.f) && (x2 < (float)(g.m2 - 2));
        const int i1 = (int)roundf(x1), i2 = (int)roundf(x2);
        if (inside && z3 >= 0.f && z3 < (float)g.m3) key = (uint32_t)((int)z3 * g.nZ + (int)spec[a]) * (uint32_t)g.m2 + (uint32_t)i2;
        keys[i] = key + (uint32_t)m * KS;
        vals[i] = (uint32_t)i;
        AtomRec r;
        r.i1 = i1; r.i2 = i2; r.r1 = x1 - (float)i1; r.r2 = x2 - (float)i2; r.occ = occ[a]; r.pad = 0;
        recs[i] = r;
    }
}

__global__ void k_seg_bounds_gang(const uint32_t* __restrict__ sorted, int nAt, int nq, int m2, uint32_t KS, int* __restrict__ seg, size_t seg_stride)
{
    const int m = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nq) return;
    seg[(size_t)m * seg_stride + (size_t)q] = lower_bound_u32(sorted + (size_t)m * (size_t)nAt, nAt, (uint32_t)m * KS + (uint32_t)q * (uint32_t)m2);
}

__global__ void k_row_starts_gang(const uint32_t* __restrict__ sorted, int nAt, int nq, int m2, uint32_t KS, int* __restrict__ rowstart,
                                  size_t rs_stride)
{
    const int m = blockIdx.y;
    const uint32_t* __restrict__ srt = sorted + (size_t)m * (size_t)nAt;
    int* __restrict__ rs = rowstart + (size_t)m * rs_stride;
    const size_t n = (size_t)nq * (size_t)(m2 + 1);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const uint32_t q = (uint32_t)(e / (size_t)(m2 + 1)), row = (uint32_t)(e % (size_t)(m2 + 1));
        rs[e] = lower_bound_u32(srt, nAt, (uint32_t)m * KS + q * (uint32_t)m2 + row);
    }
}

__device__ inline int signum(float x) { return x < 0.f ? -1 : 1; }

// Bilinear "top-hat" deposit of one (slice, species) segment: 4 pixels x (re, im) float atomics per
// atom.  comp selects which float2 component pair receives (w, w*imPot): the engine deposits
// V.x = w*occ and V.y = w*occ*imPot as the reference does.
__global__ void k_deposit(float2* __restrict__ V, const float* __restrict__ xyz, const float* __restrict__ occ,
                          const uint32_t* __restrict__ order, const int* __restrict__ seg, int key, BinGeom g,
                          float imPot)
{
    const int beg = seg[key], end = seg[key + 1];
    for (int p = beg + blockIdx.x * blockDim.x + threadIdx.x; p < end; p += gridDim.x * blockDim.x) {
        const int i = (int)order[p];
        const float x1 = xyz[i * 3 + 0] / g.d1 + ((float)g.m1) * 0.5f - 0.5f;
        const float x2 = xyz[i * 3 + 1] / g.d2 + ((float)g.m2) * 0.5f - 0.5f;
        int i1 = (int)roundf(x1);
        int i2 = (int)roundf(x2);
        const float r1 = x1 - (float)i1;
        const float r2 = x2 - (float)i2;
        const float a1 = fabsf(r1), a2 = fabsf(r2), oc = occ[i];
        const int s1 = signum(r1), s2 = signum(r2);
        float w;
        float* f = reinterpret_cast<float*>(V);
        size_t j = (size_t)i2 * g.m1 + i1;
        w = (1 - a1) * (1 - a2) * oc;
        atomicAdd(f + 2 * j, w);
        atomicAdd(f + 2 * j + 1, w * imPot);
        i2 += s2;
        j = (size_t)i2 * g.m1 + i1;
        w = (1 - a1) * a2 * oc;
        atomicAdd(f + 2 * j, w);
        atomicAdd(f + 2 * j + 1, w * imPot);
        i1 += s1;
        j = (size_t)i2 * g.m1 + i1;
        w = a1 * a2 * oc;
        atomicAdd(f + 2 * j, w);
        atomicAdd(f + 2 * j + 1, w * imPot);
        i2 -= s2;
        j = (size_t)i2 * g.m1 + i1;
        w = a1 * (1 - a2) * oc;
        atomicAdd(f + 2 * j, w);
        atomicAdd(f + 2 * j + 1, w * imPot);
    }
}

// Deposit of a slice PAIR for the packed potential W = V_s0 + i V_s1 (the deposits are real): blockIdx.y = component,
// key0 / key1 = its (slice, species) segment or -1.  No absorptive part here: it is imPot * V at transmission time.
__global__ void k_deposit_pair(float2* __restrict__ V, const float* __restrict__ xyz, const float* __restrict__ occ,
                               const uint32_t* __restrict__ order, const int* __restrict__ seg, int key0, int key1, BinGeom g)
{
    const int comp = (int)blockIdx.y, key = comp ? key1 : key0;
    if (key < 0) return;
    const int beg = seg[key], end = seg[key + 1];
    float* f = reinterpret_cast<float*>(V) + comp;
    for (int p = beg + blockIdx.x * blockDim.x + threadIdx.x; p < end; p += gridDim.x * blockDim.x) {
        const int i = (int)order[p];
        const float x1 = xyz[i * 3 + 0] / g.d1 + ((float)g.m1) * 0.5f - 0.5f;
        const float x2 = xyz[i * 3 + 1] / g.d2 + ((float)g.m2) * 0.5f - 0.5f;
        int i1 = (int)roundf(x1);
        int i2 = (int)roundf(x2);
        const float r1 = x1 - (float)i1;
        const float r2 = x2 - (float)i2;
        const float a1 = fabsf(r1), a2 = fabsf(r2), oc = occ[i];
        const int s1 = signum(r1), s2 = signum(r2);
        atomicAdd(f + 2 * ((size_t)i2 * g.m1 + i1), (1 - a1) * (1 - a2) * oc);
        i2 += s2;
        atomicAdd(f + 2 * ((size_t)i2 * g.m1 + i1), (1 - a1) * a2 * oc);
        i1 += s1;
        atomicAdd(f + 2 * ((size_t)i2 * g.m1 + i1), a1 * a2 * oc);
        i2 -= s2;
        atomicAdd(f + 2 * ((size_t)i2 * g.m1 + i1), a1 * (1 - a2) * oc);
    }
}

// Deterministic deposit ("sorted scatter option"): one workgroup owns R rows of the grid.  The rows are zeroed
// in LDS, ONE wave adds the bilinear weights of the atoms whose footprint touches them - 64 at a time, in the sorted
// (slice, species, row) order of the records - with LDS float atomics, then the tile overwrites the grid rows with
// plain stores.  Instructions of one wave reach the LDS in program order and colliding lanes of one instruction are
// serialised in a fixed order, so the sums do not depend on timing (k_deposit / k_deposit_pair: global float atomics
// as the reference's squareAtoms_d, src/crystalMaker.cu:100-119, whose last bits depend on arrival order).
// q0 -> component x, q1 -> component y (-1: none); with_impot: y = imPot * (the x deposit) instead.
__global__ void k_deposit_tile(float2* __restrict__ V, const AtomRec* __restrict__ recs, const int* __restrict__ rowstart, int q0, int q1,
                               int with_impot, float imPot, int m1, int m2, int R)
{
    extern __shared__ float2 dtile[];
    const int tid = threadIdx.x, row0 = (int)blockIdx.x * R;
    const int nr = (row0 + R <= m2) ? R : m2 - row0;
    for (int e = tid; e < nr * m1; e += blockDim.x) dtile[e] = make_float2(0.f, 0.f);
    __syncthreads();
    if (tid < 64) {
        float* tf = reinterpret_cast<float*>(dtile);
        const int rlo = row0 > 0 ? row0 - 1 : 0;
        const int rhi = (row0 + nr + 1 < m2) ? row0 + nr + 1 : m2;
        for (int comp = 0; comp < (with_impot ? 1 : 2); comp++) {
            const int q = comp ? q1 : q0;
            if (q < 0) continue;
            const int* __restrict__ rs = rowstart + (size_t)q * (size_t)(m2 + 1);
            const int plo = rs[rlo], phi = rs[rhi];
            for (int base = plo; base < phi; base += 64) {
                const int i = base + tid;
                if (i < phi) {
                    const AtomRec ar = recs[i];
                    const float a1 = fabsf(ar.r1), a2 = fabsf(ar.r2);
                    const int s1 = ar.r1 < 0.f ? -1 : 1, s2 = ar.r2 < 0.f ? -1 : 1;
#pragma unroll
                    for (int px = 0; px < 4; px++) {
                        // pixel order of the reference: (i1,i2), (i1,i2+s2), (i1+s1,i2+s2), (i1+s1,i2)
                        const int c = ar.i1 + ((px == 2 || px == 3) ? s1 : 0);
                        const int rr = ar.i2 + ((px == 1 || px == 2) ? s2 : 0) - row0;
                        const float w = ((px == 2 || px == 3) ? a1 : (1 - a1)) * ((px == 1 || px == 2) ? a2 : (1 - a2)) * ar.occ;
                        if (rr >= 0 && rr < nr && c >= 0 && c < m1) {
                            atomicAdd(&tf[2 * (rr * m1 + c) + comp], w);
                            if (with_impot) atomicAdd(&tf[2 * (rr * m1 + c) + 1], w * imPot);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    float2* __restrict__ out = V + (size_t)row0 * m1;
    for (int e = tid; e < nr * m1; e += blockDim.x) out[e] = dtile[e];
}

static inline int blocks_for(int n, int bs, int cap) { int b = (n + bs - 1) / bs; if (b < 1) b = 1; return b > cap ? cap : b; }

hipError_t geom_srot(float* xyz, int nAt, int ax, int ay, float c, float s, hipStream_t st)
{
    if (nAt <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_srot, dim3(blocks_for(nAt, 256, 2048)), dim3(256), 0, st, xyz, nAt, ax, ay, c, s);
    return hipGetLastError();
}

hipError_t geom_jitter(float* out, const float* in, const float* dwf, int nAt, uint32_t seed, int k, int j, hipStream_t st)
{
    if (nAt <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_jitter, dim3(blocks_for(3 * nAt, 256, 2048)), dim3(256), 0, st, out, in, dwf, 3 * nAt, seed,
                       (uint32_t)k, (uint32_t)j);
    return hipGetLastError();
}

hipError_t geom_slice_occupancy(int* flags, const float* xyz0, const float* dwf, int nAt, const BinGeom& g, float t_0, float t_1, bool jitter,
                                uint32_t seed, int k, int j, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(flags, 0, sizeof(int) * (size_t)g.m3, st);
    if (e != hipSuccess || nAt <= 0) return e;
    // tiltCoordinates (src/crystalMaker.cu:427-454): a rotation is skipped when |t| <= FLT_EPSILON; c = cos t, s = -sin t on the host
    const int rot1 = fabsf(t_1) > FLT_EPSILON ? 1 : 0, rot0 = fabsf(t_0) > FLT_EPSILON ? 1 : 0;
    hipLaunchKernelGGL(k_slice_occupancy, dim3(blocks_for(nAt, 256, 512)), dim3(256), 0, st, xyz0, dwf, nAt, g, rot1, cosf(t_1), -sinf(t_1), rot0,
                       cosf(t_0), -sinf(t_0), jitter ? 1 : 0, seed, (uint32_t)k, (uint32_t)j, flags);
    return hipGetLastError();
}

size_t geom_sort_temp_bytes(int nAt)
{
    size_t bytes = 0;
    uint32_t* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)(nAt > 0 ? nAt : 1), 0u, 32u, (hipStream_t)0);
    return bytes;
}

hipError_t geom_bin_atoms(const float* xyz, const uint8_t* spec, const float* occ, int nAt, const BinGeom& g, AtomBins& b,
                          bool with_rows, hipStream_t st)
{
    const int nq = g.m3 * g.nZ;
    if (nAt <= 0) {
        hipError_t e0 = hipMemsetAsync(b.seg, 0, sizeof(int) * (size_t)(nq + 2), st);
        if (e0 == hipSuccess && with_rows && b.rowstart) e0 = hipMemsetAsync(b.rowstart, 0, sizeof(int) * (size_t)nq * (size_t)(g.m2 + 1), st);
        return e0;
    }
    hipLaunchKernelGGL(k_atom_keys, dim3(blocks_for(nAt, 256, 2048)), dim3(256), 0, st, xyz, spec, occ, nAt, g, b.keys, b.vals, b.recs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint64_t nkeys = (uint64_t)nq * (uint64_t)g.m2;
    unsigned bits = 1;
    while ((1ull << bits) <= nkeys && bits < 32) bits++;
    size_t tb = b.tmp_bytes;
    e = rocprim::radix_sort_pairs(b.tmp, tb, b.keys, b.keys_sorted, b.vals, b.order, (size_t)nAt, 0u, bits, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_seg_bounds, dim3((nq + 1 + 255) / 256), dim3(256), 0, st, b.keys_sorted, nAt, nq, g.m2, b.seg);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (with_rows && b.rowstart) {
        hipLaunchKernelGGL(k_gather_recs, dim3(blocks_for(nAt, 256, 2048)), dim3(256), 0, st, b.recs_sorted, b.recs, b.order, nAt);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        const size_t n = (size_t)nq * (size_t)(g.m2 + 1);
        hipLaunchKernelGGL(k_row_starts, dim3(blocks_for((int)(n > 0x7fffffff ? 0x7fffffff : n), 256, 4096)), dim3(256), 0, st, b.keys_sorted,
                           nAt, nq, g.m2, b.rowstart);
        e = hipGetLastError();
    }
    return e;
}

hipError_t geom_tilt_gang(float* out, const float* in, int nAt, int n, const float* t0, const float* t1, hipStream_t st)
{
    if (nAt <= 0 || n <= 0) return hipSuccess;
    GangGeo gg{};
    gg.n = n;
    unsigned m1 = 0, m0 = 0;
    for (int g = 0; g < n; g++) { // cos / sin on the host, c = cos t, s = -sin t, as tilt_coordinates does
        if (fabsf(t1[g]) > FLT_EPSILON) { m1 |= 1u << g; gg.c1[g] = cosf(t1[g]); gg.s1[g] = -sinf(t1[g]); }
        if (fabsf(t0[g]) > FLT_EPSILON) { m0 |= 1u << g; gg.c0[g] = cosf(t0[g]); gg.s0[g] = -sinf(t0[g]); }
    }
    hipLaunchKernelGGL(k_tilt_gang, dim3(blocks_for(nAt, 256, 2048), n), dim3(256), 0, st, out, in, nAt, gg, m1, m0);
    return hipGetLastError();
}

hipError_t geom_jitter_gang(float* out, const float* in, size_t in_stride, const float* dwf, int nAt, int n, uint32_t seed, const int* k,
                            const int* j, hipStream_t st)
{
    if (nAt <= 0 || n <= 0) return hipSuccess;
    GangGeo gg{};
    gg.n = n;
    for (int g = 0; g < n; g++) { gg.k[g] = (uint32_t)k[g]; gg.j[g] = (uint32_t)j[g]; }
    hipLaunchKernelGGL(k_jitter_gang, dim3(blocks_for(3 * nAt, 256, 2048), n), dim3(256), 0, st, out, in, in_stride, dwf, 3 * nAt, seed, gg);
    return hipGetLastError();
}

// Binning of the n members' atoms (coordinates back to back in xyz) with ONE sort: `b` is member 0's view of arrays that
// hold n members back to back (nAt keys / values / records per member, seg_stride / rs_stride ints per member);
// tmp must hold geom_sort_temp_bytes(n * nAt).  Leaves exactly what n calls of geom_bin_atoms(with_rows) leave.
hipError_t geom_bin_atoms_gang(const float* xyz, const uint8_t* spec, const float* occ, int nAt, int n, const BinGeom& g, AtomBins& b,
                               size_t seg_stride, size_t rs_stride, hipStream_t st)
{
    const int nq = g.m3 * g.nZ;
    if (nAt <= 0 || n <= 0) return hipErrorInvalidValue;
    const uint64_t KS = ((uint64_t)nq + 1) * (uint64_t)g.m2;
    if (KS * (uint64_t)n >= (1ull << 32)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_atom_keys_gang, dim3(blocks_for(nAt, 256, 2048), n), dim3(256), 0, st, xyz, spec, occ, nAt, g, (uint32_t)KS, b.keys, b.vals,
                       b.recs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    unsigned bits = 1;
    while ((1ull << bits) <= KS * (uint64_t)n && bits < 32) bits++;
    size_t tb = b.tmp_bytes;
    e = rocprim::radix_sort_pairs(b.tmp, tb, b.keys, b.keys_sorted, b.vals, b.order, (size_t)nAt * (size_t)n, 0u, bits, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_seg_bounds_gang, dim3((nq + 1 + 255) / 256, n), dim3(256), 0, st, b.keys_sorted, nAt, nq, g.m2, (uint32_t)KS, b.seg, seg_stride);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const size_t tot = (size_t)nAt * (size_t)n;
    hipLaunchKernelGGL(k_gather_recs, dim3(blocks_for((int)(tot > 0x7fffffff ? 0x7fffffff : tot), 256, 4096)), dim3(256), 0, st, b.recs_sorted, b.recs,
                       b.order, (int)tot);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const size_t nrs = (size_t)nq * (size_t)(g.m2 + 1);
    hipLaunchKernelGGL(k_row_starts_gang, dim3(blocks_for((int)(nrs > 0x7fffffff ? 0x7fffffff : nrs), 256, 4096), n), dim3(256), 0, st, b.keys_sorted,
                       nAt, nq, g.m2, (uint32_t)KS, b.rowstart, rs_stride);
    return hipGetLastError();
}

hipError_t geom_deposit(float2* V, const float* xyz, const float* occ, const AtomBins& b, int key, const BinGeom& g,
                        float imPot, int blocks, hipStream_t st)
{
    hipLaunchKernelGGL(k_deposit, dim3(blocks), dim3(256), 0, st, V, xyz, occ, b.order, b.seg, key, g, imPot);
    return hipGetLastError();
}

hipError_t geom_deposit_tile(float2* V, const AtomBins& b, int key0, int key1, bool with_impot, float imPot, const BinGeom& g, hipStream_t st)
{
    if (!b.rowstart || !b.recs_sorted) return hipErrorInvalidValue;
    int R = 8192 / (g.m1 > 0 ? g.m1 : 1); // tile of at most 64 KiB (one float2 per pixel), 1 ... 8 rows
    R = R < 1 ? 1 : (R > 8 ? 8 : R);
    const size_t lds = sizeof(float2) * (size_t)R * (size_t)g.m1;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> attr_set{0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64 || !((attr_set.load(std::memory_order_acquire) >> dev) & 1ull)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_deposit_tile), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) attr_set.fetch_or(1ull << dev, std::memory_order_release);
        }
        if (lds > 160 * 1024) return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k_deposit_tile, dim3((g.m2 + R - 1) / R), dim3(256), lds, st, V, b.recs_sorted, b.rowstart, key0, key1, with_impot ? 1 : 0,
                       imPot, g.m1, g.m2, R);
    return hipGetLastError();
}

hipError_t geom_deposit_pair(float2* V, const float* xyz, const float* occ, const AtomBins& b, int key0, int key1, const BinGeom& g,
                             int blocks, hipStream_t st)
{
    hipLaunchKernelGGL(k_deposit_pair, dim3(blocks, 2), dim3(256), 0, st, V, xyz, occ, b.order, b.seg, key0, key1, g);
    return hipGetLastError();
}

} // namespace fdes
