// gen_jit.cpp — run-time compilation of the mixed-radix row passes for ONE grid length (round 5).
//
// cuFFT serves any grid size alike (src/paramStructure.cu:676-679); a user's m1, m2 (or m = 2 nx of a .qsc,
// src/rwQsc.cu:943-948) is whatever the specimen asks for.  fft_gen.hip has two forms of every pass: k_gpass<NC != 0>, in which
// the length and its stage tables are compile-time constants (one tile image, transforms chained through registers, every
// index folded), built into the library for the lengths the reference ships and a few round ones, and k_gpass<0>, which takes
// the length at run time and runs at about half that rate (1100^2: 13.7 k against 25.0 k slice-propagations/s, 2288^2: 3.0 k
// against 6.4 k, profiles/r05_radix_11_13.txt).  Here the compile-time form of ANY supported length is built when a plan for
// it is created: the library carries the text of fft_gen.hip and its three includes (gen_jit_src.inc, written by the
// Makefile), hipRTC compiles it with -DFDES_GEN_JIT_N=<n> (about five seconds), the code object is kept in a directory cache
// ($FDES_JIT_CACHE, else $XDG_CACHE_HOME/fdes_amd, else ~/.cache/fdes_amd; keyed on the source text, the options and the hipRTC version; every file carries the length and a checksum of its code object and is ignored when they do not match), loaded as a module on the plan's
// device, and gen_pass() launches its kernels through hipModuleLaunchKernel.  libhiprtc is resolved with dlopen: without it, or
// when the compilation fails, the plan runs the run-time-length kernels as before (the reason is kept for the caller).
// FDES_JIT=0 (or engine option jit = 0) turns it off.
#include "gen_jit.h"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "fft_lds.h"

namespace fdes {

namespace {

#include "gen_jit_src.inc" // (written by the Makefile into the build directory) kSrc_fft_gen, kSrc_fft_lds_h, kSrc_geometry_h, kSrc_fft_dev_inc

struct Rtc {
    void* so = nullptr;
    hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char* const*) = nullptr;
    hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetCode)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*DestroyProgram)(hiprtcProgram*) = nullptr;
    hiprtcResult (*Version)(int*, int*) = nullptr;
    bool ok = false;
};
Rtc& rtc()
{
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* n : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6", "/opt/rocm/lib/libhiprtc.so"}) {
            r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
        if (!r.so) return;
#define SYM(f_, name_) r.f_ = reinterpret_cast<decltype(r.f_)>(dlsym(r.so, name_))
        SYM(CreateProgram, "hiprtcCreateProgram");
        SYM(CompileProgram, "hiprtcCompileProgram");
        SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
        SYM(GetProgramLog, "hiprtcGetProgramLog");
        SYM(GetCodeSize, "hiprtcGetCodeSize");
        SYM(GetCode, "hiprtcGetCode");
        SYM(DestroyProgram, "hiprtcDestroyProgram");
        SYM(Version, "hiprtcVersion");
#undef SYM
        r.ok = r.CreateProgram && r.CompileProgram && r.GetProgramLogSize && r.GetProgramLog && r.GetCodeSize && r.GetCode && r.DestroyProgram;
    });
    return r;
}

// the pass kinds gen_pass() serves, in the order of fft_gen.hip's FDES_JIT_KERNEL list
struct Kind { int pre, mid, post, t; };
constexpr Kind kKinds[GenJitKernels::kCount] = {{0, 0, 0, 0}, {0, 0, 0, 1}, {0, 7, 0, 1}, {1, 0, 0, 0}, {2, 0, 0, 0}, {2, 7, 0, 0}, {1, 0, 0, 1}, {2, 0, 0, 1},
                                                {1, 9, 0, 1}, {2, 12, 1, 1}, {1, 2, 2, 1}, {1, 8, 2, 1}, {1, 4, 2, 1}, {2, 5, 1, 1}, {0, 5, 1, 1}, {1, 6, 2, 1}};

unsigned long long fnv(unsigned long long h, const char* s, size_t n)
{
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)s[i]; h *= 1099511628211ull; }
    return h;
}

std::string cache_dir()
{
    if (const char* e = std::getenv("FDES_JIT_CACHE")) return e[0] ? std::string(e) : std::string();
    std::string base;
    if (const char* x = std::getenv("XDG_CACHE_HOME")) base = x;
    if (base.empty()) {
        const char* h = std::getenv("HOME");
        if (!h || !h[0]) return std::string();
        base = std::string(h) + "/.cache";
    }
    return base + "/fdes_amd";
}
void make_dirs(const std::string& d)
{
    for (size_t i = 1; i <= d.size(); i++)
        if (i == d.size() || d[i] == '/') (void)mkdir(d.substr(0, i).c_str(), 0755);
}

std::mutex g_mu;
std::map<std::pair<int, int>, std::vector<char>> g_code;        // (length, tile rows) -> code object (compiled or read from the cache once per process)
std::map<std::pair<int, int>, std::string> g_from_file;         // (length, tile rows) -> cache file the code object was READ from (not compiled in this process)
std::map<std::pair<int, int>, std::string> g_failed;            // (length, tile rows) -> why it has no code object (not tried again)
std::map<std::pair<std::pair<int, int>, int>, GenJitKernels*> g_mod; // ((length, tile rows), device) -> loaded module

int ept_of(int n, int rows) { const int t = gen_pass_threads_for(n, rows), e = (rows * n + t - 1) / t; return e <= 8 ? 8 : (e <= 16 ? 16 : 32); }

// code object of the n-point passes: from this process, from the directory cache, or compiled now
const std::vector<char>* code_for(int n, int rows, std::string* note)
{
    const std::pair<int, int> key{n, rows};
    auto it = g_code.find(key);
    if (it != g_code.end()) return &it->second;
    auto fl = g_failed.find(key);
    if (fl != g_failed.end()) { if (note) *note = fl->second; return nullptr; }
    Rtc& R = rtc();
    auto fail = [&](const std::string& why) -> const std::vector<char>* {
        g_failed[key] = why;
        if (note) *note = why;
        return nullptr;
    };
    if (!R.ok) return fail("libhiprtc not found");
    const std::string o_n = "-DFDES_GEN_JIT_N=" + std::to_string(n), o_e = "-DFDES_GEN_JIT_EPT=" + std::to_string(ept_of(n, rows)),
                      o_r = "-DFDES_GEN_JIT_ROWS=" + std::to_string(rows);
    // tuning knob: FDES_JIT_STAGES=a,b,c[,d] gives the stage radices of every compilation of this process outright (product = the length,
    // radices out of 2 ... 5, 7, 8, 10 ... 13, 15, 16, 17, 19, 20, 23, 25; anything else is ignored); part of the cache key like every option
    std::string o_s[4];
    int nstage_opts = 0;
    if (const char* e = std::getenv("FDES_JIT_STAGES")) {
        int r[4] = {0, 0, 0, 1};
        const int got = std::sscanf(e, "%d,%d,%d,%d", &r[0], &r[1], &r[2], &r[3]);
        auto okr = [](int v) { for (int c : {2, 3, 4, 5, 7, 8, 10, 11, 12, 13, 15, 16, 17, 19, 20, 23, 25}) if (v == c) return true; return false; };
        if (got >= 3 && okr(r[0]) && okr(r[1]) && okr(r[2]) && (got == 3 || okr(r[3])) && (long)r[0] * r[1] * r[2] * (got == 3 ? 1 : r[3]) == n) {
            if (got == 3) r[3] = 1;
            for (int q = 0; q < 4; q++) o_s[q] = "-DFDES_GEN_JIT_R" + std::to_string(q) + "=" + std::to_string(r[q]);
            nstage_opts = 4;
        }
    }
    const bool own_rows = rows == gen_pass_rows(n); // (the length's own tile rows: no option, and the cache entries of earlier builds stay valid)
    const char* opts[20] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-DFDES_TEST_HOOKS=0", o_n.c_str(), o_e.c_str()};
    int nopts = 7;
    if (!own_rows) opts[nopts++] = o_r.c_str();
    const std::string o_t = "-DFDES_GEN_JIT_THREADS=" + std::to_string(gen_pass_threads_for(n, rows));
    if (gen_pass_threads_for(n, rows) != gen_pass_threads(n)) opts[nopts++] = o_t.c_str();
    for (int q = 0; q < nstage_opts; q++) opts[nopts++] = o_s[q].c_str();
    // tuning knob: FDES_JIT_FLAGS = up to four more compiler options, space separated (e.g. "-mllvm -amdgpu-sched-strategy=max-ilp")
    std::string o_x[4];
    if (const char* e = std::getenv("FDES_JIT_FLAGS")) {
        std::string all(e);
        size_t pos = 0;
        int nx = 0;
        while (nx < 4 && pos < all.size()) {
            const size_t sp = all.find(' ', pos);
            const std::string tok = all.substr(pos, sp == std::string::npos ? std::string::npos : sp - pos);
            if (!tok.empty()) o_x[nx++] = tok;
            if (sp == std::string::npos) break;
            pos = sp + 1;
        }
        for (int q = 0; q < nx; q++) opts[nopts++] = o_x[q].c_str();
    }
    int vmaj = 0, vmin = 0;
    if (R.Version) (void)R.Version(&vmaj, &vmin);
    unsigned long long h = 1469598103934665603ull;
    h = fnv(h, kSrc_fft_gen, sizeof(kSrc_fft_gen));
    h = fnv(h, kSrc_fft_lds_h, sizeof(kSrc_fft_lds_h));
    h = fnv(h, kSrc_geometry_h, sizeof(kSrc_geometry_h));
    h = fnv(h, kSrc_fft_dev_inc, sizeof(kSrc_fft_dev_inc));
    for (int i = 0; i < nopts; i++) h = fnv(h, opts[i], std::strlen(opts[i]) + 1);
    h = fnv(h, reinterpret_cast<const char*>(&vmaj), sizeof(vmaj));
    h = fnv(h, reinterpret_cast<const char*>(&vmin), sizeof(vmin));
    char name[96];
    std::snprintf(name, sizeof(name), "gpass_%d_%016llx.bin", n, h);
    if (rows != gen_pass_rows(n)) std::snprintf(name, sizeof(name), "gpass_%dr%d_%016llx.bin", n, rows, h);
    const std::string dir = cache_dir(), path = dir.empty() ? std::string() : dir + "/" + name;
    // cache file = "FDESJIT1", length of the code object, its FNV-1a hash, the code object: a file cut short (a full disk, a
    // killed process of an older build without the rename) must never reach hipModuleLoadData, which does not survive one
    if (!path.empty()) {
        if (FILE* f = std::fopen(path.c_str(), "rb")) {
            std::vector<char> buf;
            char tmp[65536];
            size_t got;
            while ((got = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
            std::fclose(f);
            unsigned long long len = 0, sum = 0;
            if (buf.size() > 24 + 64 && !std::memcmp(buf.data(), "FDESJIT1", 8)) {
                std::memcpy(&len, buf.data() + 8, 8);
                std::memcpy(&sum, buf.data() + 16, 8);
            }
            if (len != 0 && len == buf.size() - 24 && sum == fnv(1469598103934665603ull, buf.data() + 24, (size_t)len) && !std::memcmp(buf.data() + 24, "\177ELF", 4)) {
                buf.erase(buf.begin(), buf.begin() + 24);
                g_from_file[key] = path;
                return &(g_code[key] = std::move(buf));
            }
            (void)std::remove(path.c_str()); // not ours or damaged: compiled again below
        }
    }
    hiprtcProgram prog = nullptr;
    const char* hdr_src[] = {kSrc_fft_lds_h, kSrc_geometry_h, kSrc_fft_dev_inc};
    const char* hdr_name[] = {"fft_lds.h", "geometry.h", "fft_dev.inc"};
    if (R.CreateProgram(&prog, kSrc_fft_gen, "fft_gen.hip", 3, hdr_src, hdr_name) != HIPRTC_SUCCESS) return fail("hiprtcCreateProgram failed");
    const hiprtcResult rc = R.CompileProgram(prog, nopts, opts);
    if (rc != HIPRTC_SUCCESS) {
        std::string log;
        size_t ls = 0;
        if (R.GetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) { log.resize(ls); (void)R.GetProgramLog(prog, &log[0]); }
        (void)R.DestroyProgram(&prog);
        if (log.size() > 600) log.resize(600);
        return fail("hiprtcCompileProgram failed for " + std::to_string(n) + " points: " + log);
    }
    size_t cs = 0;
    std::vector<char> buf;
    if (R.GetCodeSize(prog, &cs) != HIPRTC_SUCCESS || cs == 0) { (void)R.DestroyProgram(&prog); return fail("hiprtcGetCodeSize failed"); }
    buf.resize(cs);
    if (R.GetCode(prog, buf.data()) != HIPRTC_SUCCESS) { (void)R.DestroyProgram(&prog); return fail("hiprtcGetCode failed"); }
    (void)R.DestroyProgram(&prog);
    if (!path.empty()) { // written under a private name and renamed: another process never reads half a file
        make_dirs(dir);
        const std::string tmpn = path + "." + std::to_string((long)getpid()) + ".tmp";
        if (FILE* f = std::fopen(tmpn.c_str(), "wb")) {
            const unsigned long long len = buf.size(), sum = fnv(1469598103934665603ull, buf.data(), buf.size());
            bool okw = std::fwrite("FDESJIT1", 1, 8, f) == 8 && std::fwrite(&len, 1, 8, f) == 8 && std::fwrite(&sum, 1, 8, f) == 8;
            okw = okw && std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
            const bool okc = std::fclose(f) == 0;
            if (!(okw && okc && std::rename(tmpn.c_str(), path.c_str()) == 0)) (void)std::remove(tmpn.c_str());
        }
    }
    return &(g_code[key] = std::move(buf));
}

} // namespace

bool gen_jit_default_on()
{
    const char* e = std::getenv("FDES_JIT");
    return !(e && e[0] == '0');
}

bool gen_jit_available() { return rtc().ok; }

const GenJitKernels* gen_jit_prepare(int n, int rows, std::string* note)
{
    if (!gen_pass_supported_len(n) || !(rows == 2 || rows == 4 || rows == 8) || rows > gen_pass_rows(n)) return nullptr;
    if (gen_pass_compiled_in(n) && rows == gen_pass_rows(n)) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { if (note) *note = "hipGetDevice failed"; return nullptr; }
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_mod.find({{n, rows}, dev});
    if (it != g_mod.end()) return it->second;
    const std::vector<char>* code = code_for(n, rows, note);
    if (!code) return nullptr;
    GenJitKernels* k = new GenJitKernels();
    k->n = n;
    k->rows = rows;
    k->device = dev;
    k->threads = gen_pass_threads_for(n, rows);
    hipModule_t mod = nullptr;
    hipError_t le = hipModuleLoadData(&mod, code->data());
    if (le != hipSuccess) {
        (void)hipGetLastError();
        auto ff = g_from_file.find({n, rows});
        if (ff != g_from_file.end()) { // a cache file that does not load (truncated by a full disk, written by another runtime): drop it and compile once
            (void)std::remove(ff->second.c_str());
            g_from_file.erase(ff);
            g_code.erase({n, rows});
            code = code_for(n, rows, note);
            mod = nullptr;
            le = code ? hipModuleLoadData(&mod, code->data()) : hipErrorUnknown;
            if (le != hipSuccess) (void)hipGetLastError();
        }
    }
    if (le != hipSuccess) {
        delete k;
        if (note) *note = "hipModuleLoadData failed for the " + std::to_string(n) + "-point passes";
        return nullptr;
    }
    k->module = mod;
    for (int i = 0; i < GenJitKernels::kCount; i++) {
        char name[64];
        std::snprintf(name, sizeof(name), "fdes_jit_gpass_%d_%d_%d_%d", kKinds[i].pre, kKinds[i].mid, kKinds[i].post, kKinds[i].t);
        hipFunction_t fn = nullptr;
        if (hipModuleGetFunction(&fn, mod, name) != hipSuccess || !fn) {
            (void)hipGetLastError();
            (void)hipModuleUnload(mod);
            delete k;
            if (note) *note = std::string("kernel ") + name + " missing from the compiled module";
            return nullptr;
        }
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        k->fn[i] = fn;
    }
    g_mod[{{n, rows}, dev}] = k; // kept until the process ends: plans of any context on this device share it
    return k;
}

void* gen_jit_function(const GenJitKernels* k, int pre, int mid, int post, bool store_transposed)
{
    if (!k) return nullptr;
    for (int i = 0; i < GenJitKernels::kCount; i++)
        if (kKinds[i].pre == pre && kKinds[i].mid == mid && kKinds[i].post == post && kKinds[i].t == (store_transposed ? 1 : 0)) return k->fn[i];
    return nullptr;
}

} // namespace fdes
