/* oracle.h — CPU restatement of the FDES forward path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is the parity oracle: a plain-C restatement of the reference's CUDA
 * algorithm (operation order of src/crystalMaker.cu:324-373, 507-536 and
 * src/multisliceSimulation.cu:538-611), in float32 (`*_f32`, the reference's arithmetic)
 * and in float64 (`*_f64`, the "truth" the tolerance of SURVEY.md 8c is stated against).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (fdes_amd/csrc) never links, imports or calls anything in here.
 *
 * PINNING STATUS.  The reference cannot be built or run in the authoring container
 * (CUDA-only: nvcc, cuFFT, cuBLAS, cuRAND absent) and ships no tests and no golden
 * wave functions or images.  What IS pinned by reference-produced values:
 *   - consitentParams (gamma, lambda, sigma) at 200 kV (src/paramStructure.cu:509-512) and
 *     at 50 kV (attributes of ExampleSpecimens/Au_cubeoctahedron_emd/Auparticle.emd);
 *   - the Kirkland table (regex-extracted numbers, tools/extract_kirkland.py);
 *   - the .cnf <-> .emd parameter round trip of the shipped Au-309 example.
 * Everything else on the wave-optics path is **parity unpinned**: it is a line-by-line
 * restatement checked by independent means (numpy FFT, Kirkland's real-space closed
 * form, norm conservation), not by reference outputs.
 */
#ifndef FDES_ORACLE_H_
#define FDES_ORACLE_H_

#include <stdint.h>
#include "../include/fdes_abi.h" /* fdes_params / fdes_atoms PODs only */

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- precision independent (oracle_common.c) ---------- */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float oracle_det_normal(uint32_t a, uint32_t b);
float oracle_normal(uint32_t seed, uint32_t stream, uint32_t k, uint32_t j, uint32_t i);
void oracle_params_default(fdes_params* p, int n3); /* arrays must already be allocated */
void oracle_consistent_params(fdes_params* p);
int oracle_sub_slices(fdes_params* p);
void oracle_tilt_coordinates(float* xyz, int nAt, float t0, float t1, float t2);
void oracle_atom_jitter(float* xyz, const float* dwf, int nAt, uint32_t seed, int k, int j);
int oracle_list_of_elements(int* Zlist, int nAt, const int* Z);
/* coordinates of configuration (k, j): tilt offset -> tilt k -> jitter (frPh > 0) */
void oracle_config_coords(const fdes_params* p, const fdes_atoms* a, int k, int j, uint32_t seed,
                          float* xyz_out);
void oracle_set_threads(int n);
int oracle_get_threads(void);
/* Au cuboctahedron / SrTiO3 generators of SURVEY.md 8(d') are in python (tests/specimens.py) */

/* ---------- per precision (oracle_core.c compiled twice) ---------- */
#define ORACLE_DECL(R, S)                                                                        \
    void oracle_fft2_##S(R* f, int m1, int m2, int inverse);                                     \
    void oracle_phase_grating_##S(const fdes_params* p, const float* xyz, const int* Z,          \
                                  const float* occ, int nAt, const int* Zlist, int nZ, int s,    \
                                  R* V);                                                         \
    void oracle_fresnel_propagator_##S(const fdes_params* p, R* P);                              \
    void oracle_forward_propagation_##S(const fdes_params* p, R* psi, const R* V, R* frProp,     \
                                        R* t);                                                   \
    void oracle_propagate_unit_##S(const fdes_params* p, R* psi, const R* t, const R* P);        \
    void oracle_incoming_wave_##S(const fdes_params* p, int k, R* psi);                          \
    void oracle_apply_lens_##S(const fdes_params* p, int k, R* psi);                             \
    void oracle_diffraction_pattern_##S(const fdes_params* p, int k, R* psi);                    \
    void oracle_add_noise_and_mtf_##S(const fdes_params* p, int k, R* I, R* J);                  \
    void oracle_wave_##S(const fdes_params* p, const fdes_atoms* a, int k, int j, uint32_t seed, \
                         int nslices, R* psi);                                                   \
    int oracle_build_measurements_##S(const fdes_params* p, const fdes_atoms* a, uint32_t seed,  \
                                      R* image, R* potential, R* exitwave);                      \
    int oracle_measurement_##S(const fdes_params* p, const fdes_atoms* a, int k, uint32_t seed,  \
                               R* image_k);

ORACLE_DECL(float, f32)
ORACLE_DECL(double, f64)

#ifdef __cplusplus
}
#endif
#endif
