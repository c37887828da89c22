// gen_jit.h — run-time compilation of the mixed-radix row passes for one grid length (gen_jit.cpp).
#ifndef FDES_GEN_JIT_H_
#define FDES_GEN_JIT_H_
#include <hip/hip_runtime.h>

#include <string>

namespace fdes {

// the compile-time form of fft_gen.hip's passes for ONE row length, compiled by hipRTC and loaded on one device
struct GenJitKernels {
    static constexpr int kCount = 16; // the pass kinds of gen_pass()
    int n = 0, rows = 0, device = 0, threads = 512; // row length, rows per tile it was compiled for
    void* module = nullptr;           // hipModule_t
    void* fn[kCount] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

// FDES_JIT != "0"
bool gen_jit_default_on();
// libhiprtc can be loaded: grid lengths beyond 4096 points exist on the fused loop only then (they have no other kernels), and only for
// plans that ask for the compilation (option jit / FDES_JIT)
bool gen_jit_available();
// The kernels of the n-point passes on the CURRENT device: from this process, from the directory cache, or compiled now (seconds).
// nullptr: n has compiled-in kernels (for these tile rows) or is no mixed-radix length (note stays empty), or hipRTC is missing / the compilation
// failed (note says why; the run-time-length kernels serve the length).  Call at plan creation, never inside a stream capture.
const GenJitKernels* gen_jit_prepare(int n, int rows, std::string* note); // rows: gen_pass_tile_rows(n, rows of the grid)
// hipFunction_t of one pass kind, nullptr if there is none
void* gen_jit_function(const GenJitKernels* k, int pre, int mid, int post, bool store_transposed);

// fft_gen.hip: lengths whose compile-time kernels are part of the library
bool gen_pass_compiled_in(int n);
// threads of a workgroup of the compile-time kernels of this length (512; 1024 beyond 4096 points)
int gen_pass_threads(int n);
int gen_pass_threads_for(int n, int rows); // ... of the kernels compiled at plan creation for these tile rows

} // namespace fdes
#endif
