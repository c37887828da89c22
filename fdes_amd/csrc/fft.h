// fft.h — unnormalised 2-D complex-to-complex FFT of an m2 x m1 float2 grid, in place
// (replaces cufftPlan2d(m2, m1, C2C) + cufftExecC2C, src/paramStructure.cu:676-679 and the exec
// sites of SURVEY 2b).  forward = exp(-2 pi i ...), inverse unnormalised.
// Back-ends: hand-written LDS row passes (fft_lds.hip) for power-of-two grids, rocFFT otherwise
// (the shipped examples use 320, 800 and 1000 point grids).
#ifndef FDES_FFT_H_
#define FDES_FFT_H_
#include <hip/hip_runtime.h>

#include <string>

struct rocfft_plan_t;
struct rocfft_execution_info_t;

namespace fdes {

struct Fft2D {
    int m1 = 0, m2 = 0;
    int backend = 0; // 1 rocFFT, 2 hand-written LDS kernels
    // rocFFT
    rocfft_plan_t *fwd = nullptr, *inv = nullptr;
    rocfft_execution_info_t* info = nullptr;
    void* work = nullptr;
    size_t work_bytes = 0;
    // LDS kernels: twiddle tables of the two row lengths and a transposition scratch grid
    float2 *tw0x = nullptr, *tw1x = nullptr, *tw0y = nullptr, *tw1y = nullptr;
    float2* scratch = nullptr;
    int wg = 512; // workgroup geometry of the LDS passes (PassArgs::wg)
    // kernels compiled at run time for a mixed-radix row length without compiled-in ones (gen_jit.h; PassArgs::jit), per axis
    const void *jit_x = nullptr, *jit_y = nullptr;
    int rows_x = 0, rows_y = 0; // mixed-radix axes: rows per tile (PassArgs::tile_rows; gen_pass_tile_rows), 0 for the other kernels
    std::string jit_note; // why a length that could have them runs the run-time-length kernels instead

    static bool lds_supported(int m1, int m2);
    static int pick_wg(int m1, int m2);
    int create(int m1, int m2, int opt, hipStream_t st, std::string* err, bool jit = false);
    hipError_t exec(float2* data, bool inverse, hipStream_t st);
    void destroy();
};

} // namespace fdes
#endif
