// params.cpp — host-side parameter handling of the FDES engine (no GPU code).
//
// Mirrors the reference's params_t life cycle:
//   allocParams/defaultParams   src/paramStructure.cu:686-705, 501-598
//   consitentParams             src/paramStructure.cu:637-673
//   subSliceRatio/setSubSlices  src/crystalMaker.cu:720-743
//   readAtomsFromArray          src/paramStructure.cu:304-345
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fdes_internal.h"

extern "C" {

int fdes_abi_version(void) { return FDES_ABI_VERSION; }

int fdes_params_init(fdes_params* p, int n3_capacity)
{
    if (!p || n3_capacity < 1) return FDES_EINVAL;
    std::memset(p, 0, sizeof(*p));
    p->cap = n3_capacity;
    p->tiltspec = (float*)std::calloc(2 * (size_t)n3_capacity, sizeof(float));
    p->tiltbeam = (float*)std::calloc(2 * (size_t)n3_capacity, sizeof(float));
    p->defoci = (float*)std::calloc((size_t)n3_capacity, sizeof(float));
    if (!p->tiltspec || !p->tiltbeam || !p->defoci) {
        fdes_params_release(p);
        return FDES_ENOMEM;
    }
    // microscope: 200 kV with the constants the reference hard-codes for it
    p->E0 = 200e3f;
    p->gamma = 1.3913902f;
    p->lambda = 2.507934e-12f;
    p->sigma = 7288400.5f;
    p->ab.C1_0 = -6.1334e-8f;
    p->ab.C3_0 = 1e-3f;
    p->mtfa = 1.f;
    p->ObjAp = 11.1e-3f;
    // imaging
    p->m1 = p->m2 = 4;
    p->m3 = 1;
    p->d1 = p->d2 = 0.25e-10f;
    p->d3 = 2e-10f;
    p->subSlTh = p->d3; // NB: the default d3, not the parsed one (paramStructure.cu:561)
    p->dn1 = p->dn2 = 1;
    p->n1 = p->n2 = 2;
    p->n3 = n3_capacity;
    std::snprintf(p->sample_name, FDES_STR, "Empty sample");
    std::snprintf(p->material, FDES_STR, "Nothing");
    std::snprintf(p->user_name, FDES_STR, "John Smith");
    std::snprintf(p->institution, FDES_STR, "Europe University");
    std::snprintf(p->department, FDES_STR, "Electron Microscopy Facility");
    std::snprintf(p->email, FDES_STR, "john.smith@uni.eu");
    std::snprintf(p->comments, FDES_STR, "This is FDES's default comment");
    return FDES_OK;
}

void fdes_params_release(fdes_params* p)
{
    if (!p) return;
    std::free(p->tiltspec);
    std::free(p->tiltbeam);
    std::free(p->defoci);
    p->tiltspec = p->tiltbeam = p->defoci = nullptr;
    p->cap = 0;
}

int fdes_params_consistent(fdes_params* p)
{
    if (!p || p->n3 < 1 || !p->tiltbeam || p->cap < p->n3) return FDES_EINVAL;
    // Mantissas only; the powers of ten are folded into 1e-4 / 1e-9 / 1e18 (float32 throughout).
    const float E0 = p->E0;
    const float m0 = 9.1093822f, c = 2.9979246f, e = 1.6021766f, h = 6.6260696f;
    const float pi = 3.141592654f;
    p->gamma = 1.f + E0 * e / m0 / c / c * 1e-4f;
    p->lambda = h / sqrtf(2.f * m0 * e) * 1e-9f / sqrtf(E0 * (1.f + E0 * e / 2.f / m0 / c / c * 1e-4f));
    p->sigma = 2.f * pi * p->gamma * p->lambda * m0 * e / h / h * 1e18f;
    p->m1 = p->n1 + 2 * p->dn1;
    p->m2 = p->n2 + 2 * p->dn2;
    float sum = 0.f;
    for (int j = 0; j < 2 * p->n3; j++) sum += fabsf(p->tiltbeam[j]);
    p->doBeamTilt = (sum < FLT_MIN * ((float)p->n3 * 2)) ? 0 : 1;
    return FDES_OK;
}

int fdes_params_sub_slices(fdes_params* p)
{
    if (!p) return FDES_EINVAL;
    float ratio = 1.f;
    if (p->subSlTh > 1e-12f && p->subSlTh < p->d3) ratio = ceilf(p->d3 / p->subSlTh);
    p->m3 = (int)(((float)p->m3) * ratio);
    p->d3 /= ratio;
    return (int)ratio;
}

void fdes_atoms_release(fdes_atoms* a)
{
    if (!a) return;
    std::free(a->Z);
    std::free(a->xyz);
    std::free(a->dwf);
    std::free(a->occ);
    a->Z = nullptr;
    a->xyz = a->dwf = a->occ = nullptr;
    a->nAt = 0;
}

int fdes_atoms_alloc(fdes_atoms* a, int n)
{
    a->nAt = n;
    size_t m = (size_t)(n > 0 ? n : 1);
    a->Z = (int32_t*)std::calloc(m, sizeof(int32_t));
    a->xyz = (float*)std::calloc(3 * m, sizeof(float));
    a->dwf = (float*)std::calloc(m, sizeof(float));
    a->occ = (float*)std::calloc(m, sizeof(float));
    if (!a->Z || !a->xyz || !a->dwf || !a->occ) {
        fdes_atoms_release(a);
        return FDES_ENOMEM;
    }
    return FDES_OK;
}

int fdes_atoms_from_array(fdes_atoms* a, const float* arr, int numAtoms, int truncate_occ)
{
    if (!a || !arr || numAtoms <= 0) return FDES_EINVAL;
    int rc = fdes_atoms_alloc(a, numAtoms);
    if (rc) return rc;
    for (int i = 0; i < numAtoms; i++) {
        a->Z[i] = (int32_t)arr[6 * i + 0];
        a->xyz[3 * i + 0] = arr[6 * i + 1];
        a->xyz[3 * i + 1] = arr[6 * i + 2];
        a->xyz[3 * i + 2] = arr[6 * i + 3];
        a->dwf[i] = arr[6 * i + 4];
        a->occ[i] = truncate_occ ? (float)(int)arr[6 * i + 5] : arr[6 * i + 5];
    }
    return FDES_OK;
}

int fdes_write_binary(const char* file, const float* data, size_t n)
{
    if (!file || !data) return FDES_EINVAL;
    FILE* f = std::fopen(file, "wb");
    if (!f) return FDES_EIO;
    size_t w = std::fwrite(data, sizeof(float), n, f);
    std::fclose(f);
    return w == n ? FDES_OK : FDES_EIO;
}

} // extern "C"

// Deep copy used by the engine (the reference deep-copies to the device, paramStructure.cu:707-731).
int fdes_params_clone(fdes_params* dst, const fdes_params* src)
{
    int cap = src->n3 > 0 ? src->n3 : 1;
    float *ts = (float*)std::calloc(2 * (size_t)cap, sizeof(float));
    float *tb = (float*)std::calloc(2 * (size_t)cap, sizeof(float));
    float *df = (float*)std::calloc((size_t)cap, sizeof(float));
    if (!ts || !tb || !df) { std::free(ts); std::free(tb); std::free(df); return FDES_ENOMEM; }
    *dst = *src;
    int n = src->n3 < src->cap ? src->n3 : src->cap;
    if (src->tiltspec) std::memcpy(ts, src->tiltspec, sizeof(float) * 2 * (size_t)n);
    if (src->tiltbeam) std::memcpy(tb, src->tiltbeam, sizeof(float) * 2 * (size_t)n);
    if (src->defoci) std::memcpy(df, src->defoci, sizeof(float) * (size_t)n);
    dst->tiltspec = ts; dst->tiltbeam = tb; dst->defoci = df; dst->cap = cap;
    return FDES_OK;
}
