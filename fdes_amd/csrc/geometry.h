// geometry.h — host-callable launchers of geometry.hip.
#ifndef FDES_GEOMETRY_H_
#define FDES_GEOMETRY_H_
#ifndef __HIPCC_RTC__ // (run-time compilation of fft_gen.hip only needs AtomRec)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace fdes {

struct BinGeom { // what squareAtoms_d reads from params_t (src/crystalMaker.cu:81-87)
    int m1, m2, m3, nZ;
    float d1, d2, d3;
};

// What one deposited atom needs (squareAtoms_d, src/crystalMaker.cu:85-99): nearest pixel, signed offsets, occupancy.
struct AtomRec {
    int i1, i2;
    float r1, r2, occ;
    int pad;
};

#ifndef __HIPCC_RTC__
struct AtomBins { // device buffers of the per-configuration binning, key = (slice * nZ + species) * m2 + row
    uint32_t *keys = nullptr, *keys_sorted = nullptr, *vals = nullptr, *order = nullptr;
    int* seg = nullptr;      // [m3*nZ + 2]  first sorted position of every (slice, species)
    int* rowstart = nullptr; // [m3*nZ][m2 + 1] first sorted position of every (slice, species, row); nullptr = not built
    AtomRec *recs = nullptr, *recs_sorted = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
};

hipError_t geom_srot(float* xyz, int nAt, int ax, int ay, float c, float s, hipStream_t st);
hipError_t geom_jitter(float* out, const float* in, const float* dwf, int nAt, uint32_t seed, int k, int j, hipStream_t st);
size_t geom_sort_temp_bytes(int nAt);
hipError_t geom_bin_atoms(const float* xyz, const uint8_t* spec, const float* occ, int nAt, const BinGeom& g, AtomBins& b, bool with_rows,
                          hipStream_t st);
// the same for the n <= 16 members of a gang in one launch each (member g: atoms [g nAt, (g + 1) nAt) of every array)
// flags[s] = 1 for every slice s of configuration (k, j) that holds an atom the deposit will use, 0 elsewhere: tilt (t_0, t_1 of
// measurement k), jitter and the binning's slice / border test recomputed from the constant tilt-offset coordinates (m3 ints)
hipError_t geom_slice_occupancy(int* flags, const float* xyz0, const float* dwf, int nAt, const BinGeom& g, float t_0, float t_1, bool jitter,
                                uint32_t seed, int k, int j, hipStream_t st);
hipError_t geom_tilt_gang(float* out, const float* in, int nAt, int n, const float* t0, const float* t1, hipStream_t st);
hipError_t geom_jitter_gang(float* out, const float* in, size_t in_stride, const float* dwf, int nAt, int n, uint32_t seed, const int* k,
                            const int* j, hipStream_t st);
hipError_t geom_bin_atoms_gang(const float* xyz, const uint8_t* spec, const float* occ, int nAt, int n, const BinGeom& g, AtomBins& b,
                               size_t seg_stride, size_t rs_stride, hipStream_t st);
hipError_t geom_deposit(float2* V, const float* xyz, const float* occ, const AtomBins& b, int key, const BinGeom& g,
                        float imPot, int blocks, hipStream_t st);

// deterministic deposit from the sorted records (needs geom_bin_atoms(..., with_rows = true)): component x <- segment
// key0, component y <- segment key1 (-1: none), or with_impot: y = imPot * x; overwrites V (no clearing needed)
// one whole row of the grid (float2) must fit the LDS tile: rows beyond 20 480 points take the atomic deposit instead
inline bool geom_deposit_tile_fits(int m1) { return m1 > 0 && sizeof(float2) * (size_t)m1 <= (size_t)160 * 1024; }
hipError_t geom_deposit_tile(float2* V, const AtomBins& b, int key0, int key1, bool with_impot, float imPot, const BinGeom& g, hipStream_t st);
hipError_t geom_deposit_pair(float2* V, const float* xyz, const float* occ, const AtomBins& b, int key0, int key1, const BinGeom& g,
                             int blocks, hipStream_t st);
#endif // __HIPCC_RTC__

} // namespace fdes
#endif
