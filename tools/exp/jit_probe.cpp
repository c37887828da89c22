// jit_probe.cpp - round 5: can a hipRTC-compiled kernel with > 64 KiB of dynamic LDS and by-value struct arguments be launched through
// hipModuleLaunchKernel / hipExtModuleLaunchKernel (dispatch events), and captured into a hipGraph?  (feasibility of compiling
// k_gpass<N> for a grid length at plan creation.)  Build: hipcc -O2 -o jit_probe jit_probe.cpp -lhiprtc
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define RK(x) do { hiprtcResult e_ = (x); if (e_ != HIPRTC_SUCCESS) { printf("%s: %s (line %d)\n", #x, hiprtcGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Args { float* out; int n; float scale; int pad[20]; };
struct Fac { int radix[8]; unsigned magic[8]; };
static const char* src = R"(
struct Args { float* out; int n; float scale; int pad[20]; };
struct Fac { int radix[8]; unsigned magic[8]; };
template <int N> __device__ void body(const Args& a, const Fac& f) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < N; i += blockDim.x) lds[i] = (float)i * a.scale + (float)f.radix[3];
    __syncthreads();
    for (int i = threadIdx.x; i < a.n; i += blockDim.x) a.out[blockIdx.x * a.n + i] = lds[N - 1 - i];
}
extern "C" __global__ __launch_bounds__(256) void probe(Args a, Fac f) { body<PROBE_N>(a, f); }
)";
int main()
{
    hiprtcProgram prog;
    RK(hiprtcCreateProgram(&prog, src, "probe.hip", 0, nullptr, nullptr));
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-DPROBE_N=30000"};
    auto t0 = std::chrono::steady_clock::now();
    hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) { std::string log(ls, 0); hiprtcGetProgramLog(prog, &log[0]); printf("log: %s\n", log.c_str()); }
    RK(r);
    size_t cs = 0;
    RK(hiprtcGetCodeSize(prog, &cs));
    std::vector<char> code(cs);
    RK(hiprtcGetCode(prog, code.data()));
    printf("compiled %zu bytes in %.1f ms\n", cs, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    hipModule_t mod;
    hipFunction_t fn;
    CK(hipModuleLoadData(&mod, code.data()));
    CK(hipModuleGetFunction(&fn, mod, "probe"));
    const int n = 1000, nb = 64;
    float* out;
    CK(hipMalloc(&out, sizeof(float) * n * nb));
    Args a{out, n, 2.f, {0}};
    Fac f{{0, 0, 0, 7, 0, 0, 0, 0}, {0}};
    void* params[] = {&a, &f};
    const unsigned ldsb = 30000 * 4; // 117 KiB
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    printf("hipFuncSetAttribute on a module function: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipModuleLaunchKernel(fn, nb, 1, 1, 256, 1, 1, ldsb, st, params, nullptr));
    CK(hipStreamSynchronize(st));
    std::vector<float> h((size_t)n * nb);
    CK(hipMemcpy(h.data(), out, sizeof(float) * n * nb, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int b = 0; b < nb; b++) for (int i = 0; i < n; i++) bad += h[(size_t)b * n + i] != (float)(29999 - i) * 2.f + 7.f;
    printf("module launch with %u bytes of dynamic LDS: %zu wrong\n", ldsb, bad);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(out, 0, sizeof(float) * n * nb));
    CK(hipExtModuleLaunchKernel(fn, nb * 256, 1, 1, 256, 1, 1, ldsb, st, params, nullptr, e0, e1, 0));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), out, sizeof(float) * n * nb, hipMemcpyDeviceToHost));
    bad = 0;
    for (int b = 0; b < nb; b++) for (int i = 0; i < n; i++) bad += h[(size_t)b * n + i] != (float)(29999 - i) * 2.f + 7.f;
    printf("ext module launch with events: %.3f ms, %zu wrong\n", ms, bad);
    // graph capture
    CK(hipMemset(out, 0, sizeof(float) * n * nb));
    CK(hipDeviceSynchronize());
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CK(hipModuleLaunchKernel(fn, nb, 1, 1, 256, 1, 1, ldsb, st, params, nullptr));
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    a.scale = 3.f; // the captured node holds a copy of the arguments
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), out, sizeof(float) * n * nb, hipMemcpyDeviceToHost));
    bad = 0;
    for (int b = 0; b < nb; b++) for (int i = 0; i < n; i++) bad += h[(size_t)b * n + i] != (float)(29999 - i) * 2.f + 7.f;
    printf("captured module launch replayed from a hipGraph: %zu wrong\n", bad);
    return 0;
}
