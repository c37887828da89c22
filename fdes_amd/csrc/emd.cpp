// emd.cpp — EMD/HDF5 results writer and configuration reader (host only).
//
// Schema of the reference's writer (src/rwHdf5.cu:27-1083; SURVEY Appendix B, cross-checked against
// `h5dump` of ExampleSpecimens/Au_cubeoctahedron_emd/Auparticle.emd) and of its reader
// (src/rwHdf5.cu:1946-2570).  libhdf5 (>= 1.10) is resolved at run time with dlopen, so the engine
// has no build- or link-time dependency on it; without it both entry points return
// FDES_EUNSUPPORTED and Measurements.bin remains the output.
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "fdes_internal.h"

namespace {

typedef int64_t hid_t;
typedef int herr_t;
typedef int htri_t;
typedef unsigned long long hsize_t;

struct H5 {
    void* so = nullptr;
    bool ok = false;
    herr_t (*open)() = nullptr;
    herr_t (*get_libversion)(unsigned*, unsigned*, unsigned*) = nullptr;
    hid_t (*Fcreate)(const char*, unsigned, hid_t, hid_t) = nullptr;
    hid_t (*Fopen)(const char*, unsigned, hid_t) = nullptr;
    herr_t (*Fclose)(hid_t) = nullptr;
    hid_t (*Gcreate2)(hid_t, const char*, hid_t, hid_t, hid_t) = nullptr;
    hid_t (*Gopen2)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*Gclose)(hid_t) = nullptr;
    hid_t (*Screate)(int) = nullptr;
    hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*) = nullptr;
    herr_t (*Sclose)(hid_t) = nullptr;
    int (*Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*) = nullptr;
    int (*Sget_simple_extent_ndims)(hid_t) = nullptr;
    hid_t (*Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
    hid_t (*Dopen2)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*) = nullptr;
    herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*) = nullptr;
    hid_t (*Dget_space)(hid_t) = nullptr;
    herr_t (*Dclose)(hid_t) = nullptr;
    hid_t (*Acreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t) = nullptr;
    hid_t (*Aopen)(hid_t, const char*, hid_t) = nullptr;
    htri_t (*Aexists)(hid_t, const char*) = nullptr;
    herr_t (*Awrite)(hid_t, hid_t, const void*) = nullptr;
    herr_t (*Aread)(hid_t, hid_t, void*) = nullptr;
    hid_t (*Aget_type)(hid_t) = nullptr;
    herr_t (*Aclose)(hid_t) = nullptr;
    hid_t (*Tcopy)(hid_t) = nullptr;
    herr_t (*Tset_size)(hid_t, size_t) = nullptr;
    herr_t (*Tset_strpad)(hid_t, int) = nullptr;
    size_t (*Tget_size)(hid_t) = nullptr;
    herr_t (*Tclose)(hid_t) = nullptr;
    htri_t (*Lexists)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*Eset_auto2)(hid_t, void*, void*) = nullptr;
    hid_t T_FLOAT = -1, T_INT = -1, T_UCHAR = -1, T_C_S1 = -1;
};

void h5_load(H5& h);
// the library is looked for once per process; concurrent first callers (one host thread per GPU) wait for the loader
H5& h5()
{
    static H5 h;
    static std::once_flag once;
    std::call_once(once, [] { h5_load(h); });
    return h;
}
void h5_load(H5& h)
{
    std::vector<std::string> names;
    if (const char* e = std::getenv("FDES_HDF5_LIB")) names.push_back(e);
    for (const char* n : {"libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5_serial.so", "libhdf5_serial.so.103",
                          "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"})
        names.push_back(n);
    for (const auto& n : names) {
        h.so = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h.so) break;
    }
    if (!h.so) return;
    bool all = true;
#define SYM(field, name) *(void**)(&h.field) = dlsym(h.so, name); if (!h.field) all = false;
    SYM(open, "H5open") SYM(get_libversion, "H5get_libversion") SYM(Fcreate, "H5Fcreate") SYM(Fopen, "H5Fopen") SYM(Fclose, "H5Fclose")
    SYM(Gcreate2, "H5Gcreate2") SYM(Gopen2, "H5Gopen2") SYM(Gclose, "H5Gclose") SYM(Screate, "H5Screate")
    SYM(Screate_simple, "H5Screate_simple") SYM(Sclose, "H5Sclose") SYM(Sget_simple_extent_dims, "H5Sget_simple_extent_dims")
    SYM(Sget_simple_extent_ndims, "H5Sget_simple_extent_ndims") SYM(Dcreate2, "H5Dcreate2") SYM(Dopen2, "H5Dopen2") SYM(Dwrite, "H5Dwrite")
    SYM(Dread, "H5Dread") SYM(Dget_space, "H5Dget_space") SYM(Dclose, "H5Dclose") SYM(Acreate2, "H5Acreate2") SYM(Aopen, "H5Aopen")
    SYM(Aexists, "H5Aexists") SYM(Awrite, "H5Awrite") SYM(Aread, "H5Aread") SYM(Aget_type, "H5Aget_type") SYM(Aclose, "H5Aclose")
    SYM(Tcopy, "H5Tcopy") SYM(Tset_size, "H5Tset_size") SYM(Tset_strpad, "H5Tset_strpad") SYM(Tget_size, "H5Tget_size") SYM(Tclose, "H5Tclose")
    SYM(Lexists, "H5Lexists") SYM(Eset_auto2, "H5Eset_auto2")
#undef SYM
    if (!all) return;
    unsigned maj = 0, min = 0, rel = 0;
    if (h.get_libversion(&maj, &min, &rel) < 0 || maj != 1 || min < 10) return; // hid_t is 64-bit from 1.10 on
    if (h.open() < 0) return;
    auto glob = [&](const char* n) -> hid_t { hid_t* p = (hid_t*)dlsym(h.so, n); return p ? *p : -1; };
    h.T_FLOAT = glob("H5T_NATIVE_FLOAT_g");
    h.T_INT = glob("H5T_NATIVE_INT_g");
    h.T_UCHAR = glob("H5T_NATIVE_UCHAR_g");
    h.T_C_S1 = glob("H5T_C_S1_g");
    h.ok = h.T_FLOAT >= 0 && h.T_INT >= 0 && h.T_UCHAR >= 0 && h.T_C_S1 >= 0;
    if (h.ok) h.Eset_auto2(0, nullptr, nullptr); // no HDF5 error stack printing: errors are return codes here
    return;
}

// ---- attribute helpers: scalars are rank-1 [1] arrays (rwHdf5.cu:2591-2604), strings are fixed-size = strlen,
// NULLTERM padded, scalar dataspace (:2641-2655)
template <class T> void attr1(H5& h, hid_t loc, const char* name, hid_t type, T v)
{
    hsize_t one = 1;
    hid_t sp = h.Screate_simple(1, &one, nullptr);
    hid_t a = h.Acreate2(loc, name, type, sp, 0, 0);
    if (a >= 0) { h.Awrite(a, type, &v); h.Aclose(a); }
    h.Sclose(sp);
}
void attr_str(H5& h, hid_t loc, const char* name, const char* v)
{
    size_t len = std::strlen(v);
    if (len == 0) len = 1;
    hid_t t = h.Tcopy(h.T_C_S1);
    h.Tset_size(t, len);
    h.Tset_strpad(t, 0 /* H5T_STR_NULLTERM */);
    hid_t sp = h.Screate(0 /* H5S_SCALAR */);
    hid_t a = h.Acreate2(loc, name, t, sp, 0, 0);
    if (a >= 0) { h.Awrite(a, t, v); h.Aclose(a); }
    h.Sclose(sp);
    h.Tclose(t);
}
hid_t dset(H5& h, hid_t loc, const char* name, hid_t type, int rank, const hsize_t* dims, const void* data)
{
    hid_t sp = h.Screate_simple(rank, dims, nullptr);
    hid_t d = h.Dcreate2(loc, name, type, sp, 0, 0, 0);
    if (d >= 0) h.Dwrite(d, type, 0, 0, 0, data);
    h.Sclose(sp);
    return d;
}
// centred axes: index - (n-1)/2 (x, y everywhere: rwHdf5.cu:108-113, 134-139, 434-437; z of the potential slices: :160-165);
// frame axes of /data/images and /data/exit_wave are the plain index (float) i (:330-335, 493-498, config writer :1393-1398)
void axis(H5& h, hid_t grp, const char* dname, int n, const char* name, const char* units, bool centred = true)
{
    std::vector<float> v((size_t)n);
    for (int i = 0; i < n; i++) v[i] = centred ? (float)(i - (n - 1) / 2.0) : (float)i;
    hsize_t d = (hsize_t)n;
    hid_t ds = dset(h, grp, dname, h.T_FLOAT, 1, &d, v.data());
    if (ds >= 0) { attr_str(h, ds, "name", name); attr_str(h, ds, "units", units); h.Dclose(ds); }
}
void complex_axis(H5& h, hid_t grp)
{
    hid_t t = h.Tcopy(h.T_C_S1);
    h.Tset_size(t, 4);
    h.Tset_strpad(t, 0);
    const char names[8] = {'r', 'e', 'a', 'l', 'i', 'm', 'a', 'g'};
    hsize_t d = 2;
    hid_t ds = dset(h, grp, "dim4", t, 1, &d, names);
    if (ds >= 0) { attr_str(h, ds, "name", "complex"); attr_str(h, ds, "units", "[]"); h.Dclose(ds); }
    h.Tclose(t);
}
void vec_dataset(H5& h, hid_t grp, const char* name, hid_t type, size_t n, const void* data, const char* units)
{
    hsize_t d = (hsize_t)n;
    hid_t ds = dset(h, grp, name, type, 1, &d, data);
    if (ds >= 0) { if (units) attr_str(h, ds, "units", units); h.Dclose(ds); }
}

// [slice][y][x][c] -> [x][y][slice][c]   (rwHdf5.cu:80-88, 248-256)
void complex_stack(H5& h, hid_t grp, const float* src, int m1, int m2, int n, bool centred_z)
{
    std::vector<float> t(2 * (size_t)m1 * m2 * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m2; j++)
            for (int k = 0; k < m1; k++) {
                const size_t o = 2 * ((size_t)k * n * m2 + (size_t)j * n + i), s = 2 * ((size_t)i * m1 * m2 + (size_t)j * m1 + k);
                t[o] = src[s];
                t[o + 1] = src[s + 1];
            }
    hsize_t d[4] = {(hsize_t)m1, (hsize_t)m2, (hsize_t)n, 2};
    hid_t ds = dset(h, grp, "data", h.T_FLOAT, 4, d, t.data());
    if (ds >= 0) h.Dclose(ds);
    axis(h, grp, "dim1", m1, "x", "[m]");
    axis(h, grp, "dim2", m2, "y", "[m]");
    axis(h, grp, "dim3", n, "z", "[m]", centred_z);
    complex_axis(h, grp);
}

bool rd_attr(H5& h, hid_t loc, const char* name, hid_t type, void* out)
{
    if (h.Aexists(loc, name) <= 0) return false;
    hid_t a = h.Aopen(loc, name, 0);
    if (a < 0) return false;
    bool ok = h.Aread(a, type, out) >= 0;
    h.Aclose(a);
    return ok;
}
bool rd_str(H5& h, hid_t loc, const char* name, char* out)
{
    if (h.Aexists(loc, name) <= 0) return false;
    hid_t a = h.Aopen(loc, name, 0);
    if (a < 0) return false;
    hid_t t = h.Aget_type(a);
    size_t n = h.Tget_size(t);
    std::vector<char> buf(n + 1, 0);
    bool ok = h.Aread(a, t, buf.data()) >= 0;
    if (ok) { std::strncpy(out, buf.data(), FDES_STR - 1); out[FDES_STR - 1] = 0; }
    h.Tclose(t);
    h.Aclose(a);
    return ok;
}
long ds_len(H5& h, hid_t loc, const char* name)
{
    if (h.Lexists(loc, name, 0) <= 0) return -1;
    hid_t d = h.Dopen2(loc, name, 0);
    if (d < 0) return -1;
    hid_t sp = h.Dget_space(d);
    hsize_t dims[8] = {0};
    int nd = h.Sget_simple_extent_ndims(sp);
    if (nd >= 1 && nd <= 8) h.Sget_simple_extent_dims(sp, dims, nullptr);
    h.Sclose(sp);
    h.Dclose(d);
    return nd >= 1 ? (long)dims[0] : -1;
}
bool rd_ds(H5& h, hid_t loc, const char* name, hid_t type, void* out)
{
    hid_t d = h.Dopen2(loc, name, 0);
    if (d < 0) return false;
    bool ok = h.Dread(d, type, 0, 0, 0, out) >= 0;
    h.Dclose(d);
    return ok;
}

} // namespace

extern "C" int fdes_emd_available(void) { return h5().ok ? 1 : 0; }

extern "C" int fdes_write_emd(const char* file, const fdes_params* p, const fdes_atoms* atoms, const float* image, const float* potential,
                              const float* exitwave, int print_level)
{
    if (!file || !p) return FDES_EINVAL;
    H5& h = h5();
    if (!h.ok) return FDES_EUNSUPPORTED;
    hid_t f = h.Fcreate(file, 2u /* H5F_ACC_TRUNC */, 0, 0);
    if (f < 0) return FDES_EIO;
    attr1<float>(h, f, "version", h.T_FLOAT, 0.1f);
    // ---- /data
    hid_t gdata = h.Gcreate2(f, "/data", 0, 0, 0);
    const long long m123 = (long long)p->m1 * p->m2 * p->m3;
    if (print_level > 0 && potential && m123 < 268435456LL) { // rwHdf5.cu:64
        hid_t g = h.Gcreate2(gdata, "potential_slices", 0, 0, 0);
        attr1<unsigned char>(h, g, "emd_group_type", h.T_UCHAR, 1);
        complex_stack(h, g, potential, p->m1, p->m2, p->m3, true);
        h.Gclose(g);
    }
    if (print_level > 1 && exitwave) {
        hid_t g = h.Gcreate2(gdata, "exit_wave", 0, 0, 0);
        attr1<unsigned char>(h, g, "emd_group_type", h.T_UCHAR, 1);
        complex_stack(h, g, exitwave, p->m1, p->m2, p->n3, false);
        h.Gclose(g);
    }
    {
        hid_t g = h.Gcreate2(gdata, "images", 0, 0, 0);
        attr1<unsigned char>(h, g, "emd_group_type", h.T_UCHAR, 1);
        const int n1 = p->n1, n2 = p->n2, n3 = p->n3;
        std::vector<float> t((size_t)n1 * n2 * n3, 0.f); // a configuration echo (image == NULL) carries zeros (:1085)
        if (image)
            for (int i = 0; i < n3; i++)
                for (int j = 0; j < n2; j++)
                    for (int k = 0; k < n1; k++) t[(size_t)k * n3 * n2 + (size_t)j * n3 + i] = image[(size_t)i * n1 * n2 + (size_t)j * n1 + k];
        hsize_t d[3] = {(hsize_t)n1, (hsize_t)n2, (hsize_t)n3};
        hid_t ds = dset(h, g, "data", h.T_FLOAT, 3, d, t.data());
        if (ds >= 0) h.Dclose(ds);
        axis(h, g, "dim1", n1, "x", "[m]");
        axis(h, g, "dim2", n2, "y", "[m]");
        axis(h, g, "dim3", n3, "z", "[m]", false);
        h.Gclose(g);
    }
    h.Gclose(gdata);
    // ---- /microscope (+ /aberrations)
    {
        hid_t g = h.Gcreate2(f, "/microscope", 0, 0, 0);
        attr1<float>(h, g, "voltage", h.T_FLOAT, p->E0); attr_str(h, g, "voltage_units", "[v]");
        attr1<float>(h, g, "gamma", h.T_FLOAT, p->gamma);
        attr1<float>(h, g, "wavelength", h.T_FLOAT, p->lambda); attr_str(h, g, "wavelength_units", "[m]");
        attr1<float>(h, g, "interaction_constant", h.T_FLOAT, p->sigma); attr_str(h, g, "interaction_constant_units", "[V^-1][m^-1]");
        attr1<float>(h, g, "focus_spread", h.T_FLOAT, p->defocspread); attr_str(h, g, "focus_spread_units", "[m]");
        attr1<float>(h, g, "illumination_angle", h.T_FLOAT, p->illangle); attr_str(h, g, "illumination_angle_units", "[rad]");
        attr1<float>(h, g, "objective_aperture", h.T_FLOAT, p->ObjAp); attr_str(h, g, "objective_aperture_units", "[rad]");
        attr1<float>(h, g, "mtf_a", h.T_FLOAT, p->mtfa); attr1<float>(h, g, "mtf_b", h.T_FLOAT, p->mtfb);
        attr1<float>(h, g, "mtf_c", h.T_FLOAT, p->mtfc); attr1<float>(h, g, "mtf_d", h.T_FLOAT, p->mtfd);
        hid_t ga = h.Gcreate2(g, "aberrations", 0, 0, 0);
        const fdes_aberration& b = p->ab;
        struct { const char* n; float a0, a1; bool angle; } ab[] = {
            {"C1", b.C1_0, b.C1_1, false}, {"A1", b.A1_0, b.A1_1, true}, {"A2", b.A2_0, b.A2_1, true}, {"B2", b.B2_0, b.B2_1, true},
            {"C3", b.C3_0, b.C3_1, false}, {"A3", b.A3_0, b.A3_1, true}, {"S3", b.S3_0, b.S3_1, true}, {"A4", b.A4_0, b.A4_1, true},
            {"B4", b.B4_0, b.B4_1, true}, {"D4", b.D4_0, b.D4_1, true}, {"C5", b.C5_0, b.C5_1, false}, {"A5", b.A5_0, b.A5_1, true},
            {"R5", b.R5_0, b.R5_1, true}, {"S5", b.S5_0, b.S5_1, true}};
        // round aberrations (C1, C3, C5) carry no angle in the files the reference wrote (Auparticle.emd)
        for (auto& e : ab) {
            attr1<float>(h, ga, (std::string(e.n) + "_amplitude").c_str(), h.T_FLOAT, e.a0);
            if (e.angle) attr1<float>(h, ga, (std::string(e.n) + "_angle").c_str(), h.T_FLOAT, e.a1);
        }
        attr_str(h, ga, "amplitude_units", "[m]");
        attr_str(h, ga, "angle_units", "[rad]");
        h.Gclose(ga);
        h.Gclose(g);
    }
    // ---- /sample
    {
        hid_t g = h.Gcreate2(f, "/sample", 0, 0, 0);
        attr_str(h, g, "name", p->sample_name);
        attr_str(h, g, "material", p->material);
        attr1<float>(h, g, "absorptive_potential_factor", h.T_FLOAT, p->imPot);
        if (atoms && atoms->nAt > 0) {
            const size_t n = (size_t)atoms->nAt;
            std::vector<float> x(n), y(n), z(n);
            for (size_t i = 0; i < n; i++) { x[i] = atoms->xyz[3 * i]; y[i] = atoms->xyz[3 * i + 1]; z[i] = atoms->xyz[3 * i + 2]; }
            vec_dataset(h, g, "atomic_numbers", h.T_INT, n, atoms->Z, nullptr);
            vec_dataset(h, g, "x_coordinates", h.T_FLOAT, n, x.data(), "[m]");
            vec_dataset(h, g, "y_coordinates", h.T_FLOAT, n, y.data(), "[m]");
            vec_dataset(h, g, "z_coordinates", h.T_FLOAT, n, z.data(), "[m]");
            vec_dataset(h, g, "debeye_waller_factors", h.T_FLOAT, n, atoms->dwf, "[m^2]");
            vec_dataset(h, g, "occupancy", h.T_FLOAT, n, atoms->occ, nullptr);
        }
        h.Gclose(g);
    }
    // ---- /imaging
    {
        hid_t g = h.Gcreate2(f, "/imaging", 0, 0, 0);
        attr1<int>(h, g, "mode", h.T_INT, p->mode);
        attr1<int>(h, g, "sample_size_x", h.T_INT, p->m1); attr1<int>(h, g, "sample_size_y", h.T_INT, p->m2);
        attr1<int>(h, g, "sample_size_z", h.T_INT, p->m3); attr_str(h, g, "sample_size_units", "[pix]");
        attr1<float>(h, g, "pixel_size_x", h.T_FLOAT, p->d1); attr1<float>(h, g, "pixel_size_y", h.T_FLOAT, p->d2);
        attr1<float>(h, g, "pixel_size_z", h.T_FLOAT, p->d3); attr_str(h, g, "pixel_size_units", "[m]");
        attr1<int>(h, g, "image_size_x", h.T_INT, p->n1); attr1<int>(h, g, "image_size_y", h.T_INT, p->n2);
        attr1<int>(h, g, "image_size_z", h.T_INT, p->n3); attr_str(h, g, "image_size_units", "[pix]");
        attr1<int>(h, g, "border_size_x", h.T_INT, p->dn1); attr1<int>(h, g, "border_size_y", h.T_INT, p->dn2);
        attr_str(h, g, "border_size_units", "[pix]");
        attr1<float>(h, g, "specimen_tilt_offset_x", h.T_FLOAT, p->tilt_offset_x);
        attr1<float>(h, g, "specimen_tilt_offset_y", h.T_FLOAT, p->tilt_offset_y);
        attr1<float>(h, g, "specimen_tilt_offset_z", h.T_FLOAT, p->tilt_offset_z);
        attr_str(h, g, "specimen_tilt_offset_units", "[rad]");
        attr1<int>(h, g, "frozen_phonons", h.T_INT, p->frPh);
        attr1<float>(h, g, "pixel_dose", h.T_FLOAT, p->pD);
        attr1<float>(h, g, "subpixel_size_z", h.T_FLOAT, p->subSlTh); attr_str(h, g, "subpixel_size_units", "[m]");
        const size_t n3 = (size_t)p->n3;
        std::vector<float> sx(n3), sy(n3), bx(n3), by(n3), df(n3);
        for (size_t i = 0; i < n3; i++) {
            sx[i] = p->tiltspec[2 * i]; sy[i] = p->tiltspec[2 * i + 1];
            bx[i] = p->tiltbeam[2 * i]; by[i] = p->tiltbeam[2 * i + 1];
            df[i] = p->defoci[i];
        }
        vec_dataset(h, g, "specimen_tilt_x", h.T_FLOAT, n3, sx.data(), "[rad]");
        vec_dataset(h, g, "specimen_tilt_y", h.T_FLOAT, n3, sy.data(), "[rad]");
        vec_dataset(h, g, "beam_tilt_x", h.T_FLOAT, n3, bx.data(), "[rad]");
        vec_dataset(h, g, "beam_tilt_y", h.T_FLOAT, n3, by.data(), "[rad]");
        vec_dataset(h, g, "defoci", h.T_FLOAT, n3, df.data(), "[rad]"); // sic: the reference labels metres as [rad] (:1015)
        h.Gclose(g);
    }
    // ---- /user (attributes omitted when empty, :1026-1054), /comments
    {
        hid_t g = h.Gcreate2(f, "/user", 0, 0, 0);
        if (p->user_name[0]) attr_str(h, g, "name", p->user_name);
        if (p->institution[0]) attr_str(h, g, "institution", p->institution);
        if (p->department[0]) attr_str(h, g, "department", p->department);
        if (p->email[0]) attr_str(h, g, "email", p->email);
        h.Gclose(g);
        g = h.Gcreate2(f, "/comments", 0, 0, 0);
        attr_str(h, g, "comment", p->comments);
        h.Gclose(g);
    }
    return h.Fclose(f) < 0 ? FDES_EIO : FDES_OK;
}

// readHdf5, src/rwHdf5.cu:1946-2570: required attributes image_size_{x,y,z}, mode, sample_size_{x,y,z},
// pixel_size_{x,y,z}, border_size_{x,y}; everything else optional.  `p` must come from fdes_params_init with
// capacity >= image_size_z.  Derived constants are the caller's job (fdes_params_consistent).
extern "C" int fdes_read_emd(const char* file, fdes_params* p, fdes_atoms* atoms, int flags)
{
    if (!file || !p || !p->tiltspec || !p->tiltbeam || !p->defoci) return FDES_EINVAL;
    H5& h = h5();
    if (!h.ok) return FDES_EUNSUPPORTED;
    hid_t f = h.Fopen(file, 0u /* H5F_ACC_RDONLY */, 0);
    if (f < 0) return FDES_EIO;
    int rc = FDES_OK;
    hid_t g = h.Lexists(f, "/imaging", 0) > 0 ? h.Gopen2(f, "/imaging", 0) : -1;
    if (g < 0) { h.Fclose(f); return FDES_EINVAL; }
    bool req = rd_attr(h, g, "image_size_x", h.T_INT, &p->n1) && rd_attr(h, g, "image_size_y", h.T_INT, &p->n2) &&
               rd_attr(h, g, "image_size_z", h.T_INT, &p->n3) && rd_attr(h, g, "mode", h.T_INT, &p->mode) &&
               rd_attr(h, g, "sample_size_x", h.T_INT, &p->m1) && rd_attr(h, g, "sample_size_y", h.T_INT, &p->m2) &&
               rd_attr(h, g, "sample_size_z", h.T_INT, &p->m3) && rd_attr(h, g, "pixel_size_x", h.T_FLOAT, &p->d1) &&
               rd_attr(h, g, "pixel_size_y", h.T_FLOAT, &p->d2) && rd_attr(h, g, "pixel_size_z", h.T_FLOAT, &p->d3) &&
               rd_attr(h, g, "border_size_x", h.T_INT, &p->dn1) && rd_attr(h, g, "border_size_y", h.T_INT, &p->dn2);
    if (!req || p->n3 < 1 || p->n3 > p->cap) rc = FDES_EINVAL;
    if (rc == FDES_OK) {
        p->subSlTh = p->d3;
        rd_attr(h, g, "specimen_tilt_offset_x", h.T_FLOAT, &p->tilt_offset_x);
        rd_attr(h, g, "specimen_tilt_offset_y", h.T_FLOAT, &p->tilt_offset_y);
        rd_attr(h, g, "specimen_tilt_offset_z", h.T_FLOAT, &p->tilt_offset_z);
        rd_attr(h, g, "frozen_phonons", h.T_INT, &p->frPh);
        rd_attr(h, g, "pixel_dose", h.T_FLOAT, &p->pD);
        rd_attr(h, g, "subpixel_size_z", h.T_FLOAT, &p->subSlTh);
        const size_t n3 = (size_t)p->n3;
        std::vector<float> a(n3), b(n3);
        auto pair = [&](const char* nx, const char* ny, float* dst) {
            if (ds_len(h, g, nx) == (long)n3 && ds_len(h, g, ny) == (long)n3 && rd_ds(h, g, nx, h.T_FLOAT, a.data()) &&
                rd_ds(h, g, ny, h.T_FLOAT, b.data()))
                for (size_t i = 0; i < n3; i++) { dst[2 * i] = a[i]; dst[2 * i + 1] = b[i]; }
        };
        pair("specimen_tilt_x", "specimen_tilt_y", p->tiltspec);
        pair("beam_tilt_x", "beam_tilt_y", p->tiltbeam);
        if (ds_len(h, g, "defoci") == (long)n3) rd_ds(h, g, "defoci", h.T_FLOAT, p->defoci);
    }
    h.Gclose(g);
    if (rc == FDES_OK && h.Lexists(f, "/microscope", 0) > 0) {
        g = h.Gopen2(f, "/microscope", 0);
        rd_attr(h, g, "voltage", h.T_FLOAT, &p->E0);
        rd_attr(h, g, "focus_spread", h.T_FLOAT, &p->defocspread);
        rd_attr(h, g, "illumination_angle", h.T_FLOAT, &p->illangle);
        rd_attr(h, g, "objective_aperture", h.T_FLOAT, &p->ObjAp);
        rd_attr(h, g, "mtf_a", h.T_FLOAT, &p->mtfa); rd_attr(h, g, "mtf_b", h.T_FLOAT, &p->mtfb);
        rd_attr(h, g, "mtf_c", h.T_FLOAT, &p->mtfc); rd_attr(h, g, "mtf_d", h.T_FLOAT, &p->mtfd);
        if (h.Lexists(g, "aberrations", 0) > 0) {
            hid_t ga = h.Gopen2(g, "aberrations", 0);
            fdes_aberration& b = p->ab;
            struct { const char* n; float *a0, *a1; } ab[] = {
                {"C1", &b.C1_0, &b.C1_1}, {"A1", &b.A1_0, &b.A1_1}, {"A2", &b.A2_0, &b.A2_1}, {"B2", &b.B2_0, &b.B2_1},
                {"C3", &b.C3_0, &b.C3_1}, {"A3", &b.A3_0, &b.A3_1}, {"S3", &b.S3_0, &b.S3_1}, {"A4", &b.A4_0, &b.A4_1},
                {"B4", &b.B4_0, &b.B4_1}, {"D4", &b.D4_0, &b.D4_1}, {"C5", &b.C5_0, &b.C5_1}, {"A5", &b.A5_0, &b.A5_1},
                {"R5", &b.R5_0, &b.R5_1}, {"S5", &b.S5_0, &b.S5_1}};
            for (auto& e : ab) {
                rd_attr(h, ga, (std::string(e.n) + "_amplitude").c_str(), h.T_FLOAT, e.a0);
                rd_attr(h, ga, (std::string(e.n) + "_angle").c_str(), h.T_FLOAT, e.a1);
            }
            h.Gclose(ga);
        }
        h.Gclose(g);
    }
    if (rc == FDES_OK && h.Lexists(f, "/user", 0) > 0) {
        g = h.Gopen2(f, "/user", 0);
        rd_str(h, g, "name", p->user_name); rd_str(h, g, "institution", p->institution);
        rd_str(h, g, "department", p->department); rd_str(h, g, "email", p->email);
        h.Gclose(g);
    }
    if (rc == FDES_OK && h.Lexists(f, "/comments", 0) > 0) {
        g = h.Gopen2(f, "/comments", 0);
        rd_str(h, g, "comment", p->comments);
        h.Gclose(g);
    }
    if (rc == FDES_OK && h.Lexists(f, "/sample", 0) > 0) {
        g = h.Gopen2(f, "/sample", 0);
        rd_str(h, g, "name", p->sample_name);
        rd_str(h, g, "material", p->material);
        rd_attr(h, g, "absorptive_potential_factor", h.T_FLOAT, &p->imPot);
        if (atoms && !(flags & FDES_CNF_SKIP_ATOMS)) {
            long n = ds_len(h, g, "atomic_numbers");
            if (n > 0 && ds_len(h, g, "x_coordinates") == n && ds_len(h, g, "y_coordinates") == n && ds_len(h, g, "z_coordinates") == n) {
                rc = fdes_atoms_alloc(atoms, (int)n);
                if (rc == FDES_OK) {
                    std::vector<float> x((size_t)n), y((size_t)n), z((size_t)n);
                    bool ok = rd_ds(h, g, "atomic_numbers", h.T_INT, atoms->Z) && rd_ds(h, g, "x_coordinates", h.T_FLOAT, x.data()) &&
                              rd_ds(h, g, "y_coordinates", h.T_FLOAT, y.data()) && rd_ds(h, g, "z_coordinates", h.T_FLOAT, z.data());
                    for (long i = 0; i < n; i++) { atoms->xyz[3 * i] = x[i]; atoms->xyz[3 * i + 1] = y[i]; atoms->xyz[3 * i + 2] = z[i]; atoms->occ[i] = 1.f; }
                    if (ds_len(h, g, "debeye_waller_factors") == n) rd_ds(h, g, "debeye_waller_factors", h.T_FLOAT, atoms->dwf);
                    if (ds_len(h, g, "occupancy") == n) rd_ds(h, g, "occupancy", h.T_FLOAT, atoms->occ);
                    if (!ok) rc = FDES_EIO;
                    p->nAt = (int)n;
                }
            } else {
                rc = fdes_atoms_alloc(atoms, 0);
                p->nAt = 0;
            }
        }
        h.Gclose(g);
    }
    h.Fclose(f);
    return rc;
}
