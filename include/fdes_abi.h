/* fdes_abi.h — C-ABI of the MI355X-native FDES forward-multislice engine.
 *
 * Plain C: pointers, sizes, PODs.  No torch / HIP types appear in any signature.
 * Every entry point names the reference interface it replaces (paths relative to
 * the FDES reference tree).  Unless stated otherwise functions return 0 on
 * success and a negative FDES_E* code on failure; they never exit()/abort() the
 * process (the reference does: src/FDESExport.cu:98-127, include/cuda_assert.hpp).
 *
 * Library: fdes_amd/csrc/libFDES_SHARED_LIB.so (name kept from
 * Python/CMakeLists.txt:84 so that Python/pyFDES.py:36-style loaders keep working).
 */
#ifndef FDES_ABI_H_
#define FDES_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden and an export list (fdes_amd/csrc/exports_*.txt): only what this header
 * declares is a dynamic symbol. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define FDES_ABI_VERSION 1
#define FDES_STR 1024 /* BUZZ_SIZE, include/paramStructure.h:39 */

enum {
    FDES_OK = 0,
    FDES_EINVAL = -1,   /* bad argument / inconsistent parameters          */
    FDES_EIO = -2,      /* file could not be read / written                */
    FDES_EGPU = -3,     /* HIP / rocFFT runtime error (see fdes_last_error) */
    FDES_ENOMEM = -4,
    FDES_EUNSUPPORTED = -5
};

/* aberration_t, include/paramStructure.h:56-85 (same member order) */
typedef struct fdes_aberration {
    float C1_0, C1_1, A1_0, A1_1, A2_0, A2_1, B2_0, B2_1, C3_0, C3_1, A3_0, A3_1, S3_0, S3_1;
    float A4_0, A4_1, B4_0, B4_1, D4_0, D4_1, C5_0, C5_1, A5_0, A5_1, R5_0, R5_1, S5_0, S5_1;
} fdes_aberration;

/* params_t without the CUDA handles (include/paramStructure.h:48-162).
 * tiltspec/tiltbeam/defoci have `cap` entries (2*cap, 2*cap, cap floats) and are
 * owned by whoever filled the pointers (fdes_params_init allocates them,
 * fdes_params_release frees them). */
typedef struct fdes_params {
    /* EM_t */
    float E0, gamma, lambda, sigma;
    fdes_aberration ab;
    float defocspread, illangle, mtfa, mtfb, mtfc, mtfd, ObjAp;
    /* IM_t */
    int32_t mode, m1, m2, m3;
    float d1, d2, d3;
    int32_t dn1, dn2, n1, n2, n3, frPh;
    float pD, subSlTh;
    float tilt_offset_x, tilt_offset_y, tilt_offset_z;
    int32_t doBeamTilt;
    int32_t cap;       /* capacity (in measurements) of the three arrays below */
    float* tiltspec;   /* [2*k] -> t_0 (rotates y,z), [2*k+1] -> t_1 (rotates x,z) */
    float* tiltbeam;
    float* defoci;
    /* SAMPLE_t */
    float imPot;
    int32_t nAt;
    /* free text (USER_t, COMMENT_t, SAMPLE_t strings) */
    char user_name[FDES_STR], institution[FDES_STR], department[FDES_STR], email[FDES_STR];
    char comments[FDES_STR], sample_name[FDES_STR], material[FDES_STR];
} fdes_params;

/* Host-side atom list (what the reference keeps as four device arrays
 * Z_d, xyzCoord_d, DWF_d, occ_d; src/paramStructure.cu:274-291). */
typedef struct fdes_atoms {
    int32_t nAt;
    int32_t* Z;  /* [nAt]                      */
    float* xyz;  /* [3*nAt] AoS x,y,z  [m]     */
    float* dwf;  /* [nAt]  Debye-Waller [m^2]  */
    float* occ;  /* [nAt]                      */
} fdes_atoms;

typedef struct fdes_ctx fdes_ctx;   /* one per GPU (replaces the reference's globals) */
typedef struct fdes_plan fdes_plan; /* device-resident state of one simulation        */

/* ---------------- parameters: host only, no GPU needed ---------------- */

/* allocParams + defaultParams, src/paramStructure.cu:686-705, 501-598 */
int fdes_params_init(fdes_params* p, int n3_capacity);
void fdes_params_release(fdes_params* p);
/* consitentParams, src/paramStructure.cu:637-673 (gamma, lambda, sigma, m=n+2dn, doBeamTilt) */
int fdes_params_consistent(fdes_params* p);
/* subSliceRatio + setSubSlices, src/crystalMaker.cu:720-743; returns the ratio (>=1) or <0 */
int fdes_params_sub_slices(fdes_params* p);
/* getParams/readConfig/numberOfAtoms/readCoordinates, src/paramStructure.cu:600-635, 42-302,
 * 1019-1077.  flags: FDES_CNF_BUG_COMPATIBLE reproduces the reference's reader quirks
 * (duplicated last atom, 100-byte fgets splitting, index-advancing blank lines);
 * FDES_CNF_SKIP_ATOMS mirrors `atomsFromExternal` (src/paramStructure.cu:268). */
#define FDES_CNF_BUG_COMPATIBLE 1
#define FDES_CNF_SKIP_ATOMS 2
int fdes_read_cnf(const char* file, fdes_params* p, fdes_atoms* atoms, int flags);
/* writeConfig, src/paramStructure.cu:362-491 */
int fdes_write_cnf(const char* file, const fdes_params* p, const fdes_atoms* atoms);
void fdes_atoms_release(fdes_atoms* a);
/* readAtomsFromArray, src/paramStructure.cu:304-345: flat float[6*n] = {Z,x,y,z,DWF,occ}.
 * truncate_occ != 0 reproduces the reference's (int) cast of the occupancy (:323). */
int fdes_atoms_from_array(fdes_atoms* a, const float* atomsArray, int numAtoms, int truncate_occ);
/* writeBinary, src/rwBinary.cpp:44-51 */
int fdes_write_binary(const char* file, const float* data, size_t n);
/* writeHdf5 (results), src/rwHdf5.cu:27-1083.  libhdf5 is dlopen'ed; FDES_EUNSUPPORTED if absent. */
int fdes_write_emd(const char* file, const fdes_params* p, const fdes_atoms* atoms,
                   const float* image, const float* potential, const float* exitwave, int print_level);

/* readHdf5, src/rwHdf5.cu:1946-2570: parameters + atoms from an EMD configuration/result file.  `p` from
 * fdes_params_init (capacity >= image_size_z); call fdes_params_consistent afterwards.  flags: FDES_CNF_SKIP_ATOMS. */
int fdes_read_emd(const char* file, fdes_params* p, fdes_atoms* atoms, int flags);
/* readQsc, src/rwQsc.cu:8-1101 (+ qstem-libs readparam / readUnitCell / replicateUnitCell): a QSTEM
 * `.qsc` file and the `.cfg` (or `.cssr` / `.dat`) unit cell it names -> parameters (n3 = 1) + the NCELL
 * super cell.  `p` from fdes_params_init; call fdes_params_consistent afterwards.  flags: FDES_CNF_SKIP_ATOMS.
 * Sites with partial or shared occupancy draw their vacancies (species 0) with QSTEM's ran1 from its fixed seed;
 * `Cube:` boxes the (tilted) crystal; `tds: yes` applies QSTEM's Einstein displacements at read time (fixed seed; the
 * reference seeds them from the clock).  FDES_EUNSUPPORTED for non-TEM modes and `.pdb`/`.xyz` cells. */
int fdes_read_qsc(const char* file, fdes_params* p, fdes_atoms* atoms, int flags);
/* Non-zero when libhdf5 (>= 1.10) could be loaded at run time (FDES_HDF5_LIB overrides the search). */
int fdes_emd_available(void);

/* ---------------- engine ---------------- */

/* cudaSetDevice(gpu_index), src/FDES.cu:165 */
int fdes_create(fdes_ctx** ctx, int gpu_index);
int fdes_destroy(fdes_ctx* ctx);
const char* fdes_last_error(const fdes_ctx* ctx);
/* Non-zero when HIP kernels of this library can run (a GPU is visible). */
int fdes_gpu_available(void);

/* buildMeasurements, src/crystalMaker.cu:227-424 / include/crystalMaker.h:77, with HOST
 * pointers and without the file writing.  `p` is the consistent parameter set BEFORE
 * sub-slicing (the call applies fdes_params_sub_slices to a private copy, as the reference
 * does at :246-247).  image: float[n1*n2*n3], image[k*n1*n2 + i2*n1 + i1].
 * potential (may be NULL): float[2*m1*m2*m3] original slices (print_level>0, :381-397).
 * exitwave  (may be NULL): float[2*m1*m2*n3] (print_level>1, :347,370). */
int fdes_build_measurements(fdes_ctx* ctx, const fdes_params* p, const fdes_atoms* atoms,
                            float* image, float* potential, float* exitwave);

/* The same over `ngpu` GPUs of one node from one host process (one host thread per device in `devices`): the (k, j)
 * configurations of src/crystalMaker.cu:324-367 are block-partitioned over the GPUs; the partial sums of a measurement
 * that spans GPUs (intensity, and the coherent exit-wave sum of print_level > 1, :347-365) are added ON THE OWNER'S GPU in
 * ascending order of the list (fdes_plan_accumulate_from: peer copy over xGMI + one axpy per peer), the owner applies
 * addNoiseAndMtf.  potential (may be NULL, print_level > 0): its slices are dealt over the GPUs as well.
 * exitwave (may be NULL, print_level > 1). */
int fdes_build_measurements_multi(int ngpu, const int* devices, const fdes_params* p, const fdes_atoms* atoms, float* image,
                                  float* potential, float* exitwave);

/* ---- resident interface: the same loops, split so that inputs stay in HBM and the
 * (k, j) configurations can be sharded over GPUs (src/crystalMaker.cu:324-373) ---- */

/* Allocations + uploads of :245-295 (deep param copy, scratch, tilt offset, species list,
 * RNG keys).  `p` BEFORE sub-slicing, as above. */
int fdes_plan_create(fdes_ctx* ctx, const fdes_params* p, const fdes_atoms* atoms, fdes_plan** plan);
int fdes_plan_destroy(fdes_plan* plan);
/* Loop-1 head (:326-331): I = 0, exit-wave accumulator = 0, specimen tilt k. */
int fdes_plan_begin_measurement(fdes_plan* plan, int k);
/* Loop-2 body (:334-366) for configuration j of measurement k: incoming wave, frozen-phonon
 * jitter (Philox keyed on (k, j, coordinate): independent of sharding), the slice loop
 * (phaseGrating + forwardPropagation, :339-344), exit-wave post-processing by mode and
 * I += result * weight.  The reference uses weight = 1/count (:302-304). Asynchronous. */
int fdes_plan_run_config(fdes_plan* plan, int k, int j, float weight);
/* addNoiseAndMtf (:372, :579-613) -> J[k]. */
int fdes_plan_end_measurement(fdes_plan* plan, int k);
/* Complete measurements ks[0 .. n): all their configurations with the weight 1 / count and the detector chain; the images
 * are in the plan's stack afterwards (fdes_plan_get_images).  What fdes_build_measurements runs; a series with ONE
 * configuration per measurement goes through it in gangs of measurements (option "gang"), which the per-k calls above
 * cannot form.  Not while an exit-wave output is wanted in a gang plan: then it runs one k after the other. */
int fdes_plan_run_measurements(fdes_plan* plan, const int* ks, int n);
/* Device pointer of the running intensity sum I (float2[m1*m2], .y = 0) so that the host
 * can reduce it across ranks (RCCL) between run_config and end_measurement.  The call first issues what run_config has
 * only queued (gangs) and folds the lanes' partial sums into I, stream-ordered: call fdes_plan_sync before another
 * stream or the host touches the memory. */
int fdes_plan_intensity_ptr(fdes_plan* plan, void** dev_ptr, size_t* bytes);
/* D2D copy of I into (to_plan = 0) or from (to_plan = 1) a caller-owned DEVICE buffer of the size
 * fdes_plan_intensity_ptr reports; stream-ordered with the plan's work, synchronises before return.
 * Lets a host runtime (torch.distributed / RCCL) reduce I without aliasing library memory. */
int fdes_plan_copy_intensity(fdes_plan* plan, void* dev_buf, int to_plan);
/* The same with the real part only: dev_buf is float[m1*m2] (I.y is identically zero, so a collective over this
 * view moves half the bytes: 16 MiB instead of 32 MiB at 2048^2).  to_plan = 1 writes I = (buf, 0). */
int fdes_plan_copy_intensity_real(fdes_plan* plan, void* dev_buf, int to_plan);
/* dst.I += src.I (and dst's exit-wave sum += src's when both plans want it), device to device: the reduction of
 * src/crystalMaker.cu:347-365 for a measurement whose configurations ran on two plans.  Plans on different GPUs of
 * the process: the intensity sum crosses as its real view (float[m1*m2]: its imaginary part is identically zero; 16 MiB
 * at 2048^2), one peer copy into a landing buffer on dst's GPU + one add kernel.  Synchronises dst. */
int fdes_plan_accumulate_from(fdes_plan* dst, fdes_plan* src);

/* The same reduction as ONE collective over RCCL (ncclReduce(sum, float[m1*m2]) to the owner of the measurement),
 * for hosts that run one context per GPU - threads of one process (fdes_build_measurements_multi with FDES_REDUCE=rccl) or
 * one process per GPU.  librccl.so is loaded at run time on the first call (FDES_EUNSUPPORTED without it).
 *   fdes_comm_unique_id   one rank makes the 128-byte id (ncclGetUniqueId) and hands it to the others by its own means;
 *   fdes_comm_create      every rank, concurrently: ncclCommInitRank on the context's device; blocks until all have joined.
 *                         One rank per GPU (RCCL refuses two ranks on one device).  Call it before the ranks create
 *                         plans: it allocates device memory, which must not coincide with another thread's graph capture;
 *   fdes_plan_reduce_intensity  EVERY rank of the communicator, for the SAME measurement, in the same order: the ranks'
 *                         running intensity sums (lanes folded, gangs issued) are added onto `root`'s plan, whose sum then
 *                         holds the total; the other ranks' sums are left as they were.  Plans that accumulate the coherent
 *                         exit-wave sum (fdes_plan_want_exitwave, every rank alike) have it reduced by a second ncclReduce.  Stream-ordered behind the plan's
 *                         work; synchronises.  The order of the additions is RCCL's (fixed for a given node and job shape,
 *                         not the ascending-GPU order of fdes_plan_accumulate_from);
 *   fdes_comm_destroy     after the context's plans are done with it, before fdes_destroy. */
typedef struct fdes_comm fdes_comm;
typedef struct { char bytes[128]; } fdes_comm_id; /* ncclUniqueId */
int fdes_comm_unique_id(fdes_comm_id* id);
int fdes_comm_create(fdes_ctx* ctx, int nranks, int rank, const fdes_comm_id* id, fdes_comm** comm);
int fdes_comm_destroy(fdes_comm* comm);
int fdes_plan_reduce_intensity(fdes_plan* plan, fdes_comm* comm, int root);
/* The same sum for a measurement whose configurations sit on the ranks lo .. hi (lo <= root <= hi) of the communicator only:
 * ONLY those ranks call it, in the same order for the same measurement; every rank of the span sends the float view of its
 * sum (and its exit-wave sum) to `root` in one group of point-to-point transfers (ncclSend / ncclRecv: every peer has an
 * xGMI link of its own to the root), the root adds them in rank order.  A span of the whole communicator is the collective above. */
int fdes_plan_reduce_intensity_span(fdes_plan* plan, fdes_comm* comm, int root, int lo, int hi);
/* Coherent exit-wave average (print_level > 1, src/crystalMaker.cu:347,370): switch the accumulation on before
 * fdes_plan_begin_measurement; fdes_plan_get_exitwave copies the sum of the current measurement, float[2*m1*m2]. */
int fdes_plan_want_exitwave(fdes_plan* plan, int on);
int fdes_plan_get_exitwave(fdes_plan* plan, float* exitwave);
/* Potential output of print_level > 0 (src/crystalMaker.cu:381-397) for the ORIGINAL slices [s_lo, s_hi):
 * float[2*m1*m2*(s_hi - s_lo)].  fdes_plan_original_slices = m3 before sub-slicing. */
int fdes_plan_potential(fdes_plan* plan, int s_lo, int s_hi, float* potential);
int fdes_plan_original_slices(const fdes_plan* plan);
/* Device pointer to J (float[n1*n2*n3]). */
int fdes_plan_images_ptr(fdes_plan* plan, void** dev_ptr, size_t* bytes);
/* D2H of J (:375). Synchronises. */
int fdes_plan_get_images(fdes_plan* plan, float* image);
int fdes_plan_sync(fdes_plan* plan);
/* 1: generic slice loop on rocFFT + point-wise kernels; 2: fused LDS-pass slice loop. */
int fdes_plan_fft_backend(const fdes_plan* plan);
/* Grid axes (0, 1 or 2) of a fused plan whose row passes run kernels compiled for that length at plan creation (hipRTC; option
   "jit", FDES_JIT=0 turns it off): lengths 2^a 3^b 5^c 7^d 11^e 13^f without compiled-in kernels (and every row of 4098 ... 8192 points or with a factor 17, 19, 23) - cufftPlan2d serves any size
   alike (src/paramStructure.cu:676-679), here the compile-time form of the row passes is about twice as fast as the form that
   takes the length at run time.  0 for the power-of-two grids and the lengths whose kernels are part of the library. */
int fdes_plan_jit_kernels(const fdes_plan* plan);
/* The same question for a grid of m1 x m2 points before any plan exists (host only; fft_option as engine option "fft"). */
int fdes_grid_backend(int m1, int m2, int fft_option);
/* Configurations the plan keeps in flight at once (lanes: own HIP stream and buffers each; option "lanes"). */
int fdes_plan_lanes(const fdes_plan* plan);
/* Configurations of one measurement that a lane runs in lockstep, every pass one launch (option "gang"; 1: off). */
int fdes_plan_gang(const fdes_plan* plan);
/* Number of (sub-)slices m3 after sub-slicing; slice-propagations done so far. */
int fdes_plan_num_slices(const fdes_plan* plan);
int64_t fdes_plan_slices_done(const fdes_plan* plan);
/* Mean device time [ms] of the slice loops between the HIP events recorded by
 * run_config since the last call (measurement). Synchronises. */
int fdes_plan_slice_loop_ms(fdes_plan* plan, double* total_ms, int64_t* slices);

/* Engine options (before fdes_plan_create).  Unknown keys -> FDES_EINVAL.  (Test / bench-only keys: fdes_abi_test.h.)
 *   "fft"        0 = auto, 1 = rocFFT, 2 = hand-written LDS FFT kernels (grid lengths 256 ... 4096 that are powers of two,
 *                or 2^a 3^b 5^c 7^d up to 4096: the 320-, 800- and 1000-point grids of the reference's examples, and the 2560 ... 4000-point
 *                grids a .qsc with nx = 1280 ... 2000 gives)
 *   "graph"      1 = replay the slice loop from a hipGraph
 *   "seed"       frozen-phonon seed (reference: 1, src/crystalMaker.cu:292)
 *   "band_skip"  1 (default): rows / columns that the radial 2/3 band limit zeroes whatever the other index is are
 *                neither transformed nor moved in the fused loop (exact: they hold zeros); 0: move everything
 *   "skip_empty" 1 (default): a slice that holds no atom has t = 1 exactly, so only its Fresnel step is run
 *                (2 passes instead of 5-6); 0: every slice goes through the full sequence like the reference
 *   "lanes"      1..8 configurations (or gangs of them) in flight at once in the fused slice loop (default 0: three up to
 *                1024 x 1024 pixels without gangs, two above and whenever the lanes run gangs, one for a small gang job;
 *                never more than the job has configurations resp. gangs): run_config calls are dealt round-robin to
 *                the lanes, partial intensity sums are folded in end_measurement (and by every call that hands out
 *                or moves the sums: fdes_plan_intensity_ptr, fdes_plan_copy_intensity*, fdes_plan_accumulate_from)
 *   "gang"       configurations of one measurement (frozen-phonon configurations of one tilt / defocus) whose slice
 *                loops run in lockstep on a lane, every pass ONE launch with the configurations as grid z: fills the
 *                chip where one grid's rows cannot (up to 1024 x 1024).  -1 auto, 0 / 1 off, 2..16.  run_config then
 *                only queues; the work is issued when the gang is full or its results are asked for.
 *   "pass_threads"  0 auto; 256 or 512 threads x two rows per thread; 1: one row per thread, four rows per workgroup;
 *                64: one wave per row (1024-, 2048- and 4096-point rows), four rows per workgroup; 128: the same with
 *                eight rows per workgroup (2048-point rows); 65: 64 as a software pipeline.  (At 4096 points the band-limit
 *                and propagator passes always run one wave per row with half-size LDS regions: two workgroups per CU.)
 *   "split"      -1 (default): a plan with one lane (single-image jobs; a plan never has more lanes than the job has
 *                configurations) and at least 2^20 pixels runs the potential / transmission passes of its slice loop on a second stream, one
 *                slice pair ahead of the wave's passes; 0 never, 1 always
 *   "batch"      -1 (default): a one-lane plan of at most 2^20 pixels runs the potential / transmission passes of 4 (1024^2)
 *                or 8 (512^2 and below) slice pairs as one launch each, a batch ahead of the wave's passes;
 *                0 / 1 off, 2 ... 8 pairs per launch
 *   "pitch_pad"  -1 (default: 32 for 2048-point rows, 64 from 4096 on) elements of padding per row of the slice loop's grids
 *   "walk"       1 (default) .. 8: launch every pass in that many parts (an experiment of round 2; the multi-wave row kernels
 *                implement it, so walk > 1 selects those instead of the one-wave-per-row kernels; the mixed-radix passes
 *                ignore it)
 *   "deterministic"  1 (default): the deposit of the rocFFT slice loop and of the potential output adds the atoms in sorted
 *                order through LDS (bit-reproducible, like the fused loop); 0: global float atomics as the reference's
 *                squareAtoms_d (src/crystalMaker.cu:100-119)
 *   "stagger"    0 (default) .. 1024: one-wave-per-row passes start the waves of a CU that many x 64 cycles apart;
 *                -1 .. -2048 (experiment, round 5): the first generation of workgroups on the odd CUs starts |value| x 64
 *                cycles late (the CUs of the chip in different phases; measured without effect, profiles/r05_cu_class_stagger_4096.txt)
 *   "peer_copy"  1 (default): fdes_plan_accumulate_from moves a partial sum between GPUs by a peer copy and falls back
 *                to host staging when the runtime refuses it; 0: always stage through host memory
 *   "jit"        -1 (default): unless FDES_JIT=0, a plan on the fused loop whose grid length has no compiled-in mixed-radix
 *                kernels (for its tile rows) gets them compiled by hipRTC when it is created (seconds once per length and
 *                machine: directory cache FDES_JIT_CACHE, else ~/.cache/fdes_amd); 1: always, 0: never - the kernels that take
 *                the length at run time serve then, and whenever libhiprtc is missing (fdes_plan_jit_kernels tells)  */
int fdes_set_option(fdes_ctx* ctx, const char* key, int64_t value);

/* Progress report.  The reference prints a percentage to stderr from inside its slice loop (progressCounter,
 * src/optimFunctions.cu:257, called at src/crystalMaker.cu:341 with j = slice-propagations started, jTot = n3 * count * m3).
 * Here fdes_build_measurements calls `fn(user, done, total)` on the calling host thread, between configurations (never
 * from a captured graph), at most once per `min_interval_ms` and once at the end; `done` counts slice-propagations whose
 * configuration has FINISHED on the GPU.  With a callback installed at most 2 x lanes configurations are kept in
 * flight.  fn = NULL removes it. */
typedef void (*fdes_progress_fn)(void* user, int64_t done, int64_t total);
int fdes_set_progress(fdes_ctx* ctx, fdes_progress_fn fn, void* user, int min_interval_ms);

/* ---------------- legacy symbol ---------------- */
/* src/FDESExport.cu:59-60.  Same arguments; returns normally on error after printing to
 * stderr (dstImage is then left untouched).  atomsArray: float[6*numAtoms]. */
void FDES(int gpu_Index, int print_Level, char* input_name, char* image_name, char* emd_save_name,
          float* atomsArray, int numAtoms, float* dstImage);
/* int-returning twin of the above. */
int fdes_run_file(int gpu_index, int print_level, const char* input_name, const char* image_name,
                  const char* emd_name, const float* atomsArray, int numAtoms, float* dstImage);

int fdes_abi_version(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FDES_ABI_H_ */
