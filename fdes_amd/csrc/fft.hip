// fft.hip — FFT back-ends of the engine (see fft.h).
#include "fft.h"

#include <rocfft/rocfft.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_lds.h"
#include "gen_jit.h"

namespace fdes {

// workgroup geometry whose rows per workgroup divide both grid dimensions (a pass over the rows of one axis has the other
// axis' length as its row count): 512 or 256 threads x two rows per thread, or one row per thread; 0: none
int Fft2D::pick_wg(int m1, int m2)
{
    if (!(lds_fft_supported_len(m1) || gen_pass_supported_len(m1)) || !(lds_fft_supported_len(m2) || gen_pass_supported_len(m2))) return 0;
    for (int wg : {512, 256, 1}) {
        if (wg == 1 && (m1 > 2048 || m2 > 2048)) continue;
        if (lds_fft_rows_per_block(m1, wg, m2) > 0 && lds_fft_rows_per_block(m2, wg, m1) > 0) return wg;
    }
    return 0;
}
bool Fft2D::lds_supported(int m1, int m2) { return pick_wg(m1, m2) != 0; }

// (asynchronous copies on the plan's stream + a wait for that stream: a blocking hipMemcpy goes through the device's legacy
//  stream and is refused - and invalidates the capture - while another host thread captures a slice loop on this device)
static int upload_twiddles(int n, float2** tw0, float2** tw1, hipStream_t st, std::string* err)
{
    const bool gen = gen_pass_supported_len(n); // mixed-radix passes: tw0 = the n roots of unity, tw1 unused
    const int T = n / 16, P = T / 16 > 0 ? T / 16 : 1;
    std::vector<float> h0(gen ? 2 * (size_t)n : 2 * 16 * (size_t)T), h1(2 * 16 * (size_t)P);
    if (gen) gen_pass_twiddles(n, h0.data());
    else lds_fft_twiddles(n, h0.data(), h1.data());
    if (hipMalloc((void**)tw0, h0.size() * sizeof(float)) != hipSuccess || hipMalloc((void**)tw1, h1.size() * sizeof(float)) != hipSuccess ||
        hipMemcpyAsync(*tw0, h0.data(), h0.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(*tw1, h1.data(), h1.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        if (err) *err = "twiddle upload failed";
        return -1;
    }
    return 0;
}

int Fft2D::create(int m1_, int m2_, int opt, hipStream_t st, std::string* err, bool jit)
{
    m1 = m1_;
    m2 = m2_;
    const bool lds_ok = lds_supported(m1, m2);
    if (opt == 2 && !lds_ok) { if (err) *err = "hand-written FFT needs grid lengths 256 ... 4096 that are powers of two, or even 2^a 3^b 5^c 7^d 11^e 13^f lengths up to 8192 (beyond 4096 points with run-time compilation only)"; return -1; }
    backend = (opt == 1 || !lds_ok) ? 1 : 2;
    if (backend == 2) {
        wg = pick_wg(m1, m2);
        // mixed-radix axes: tiles smaller than the length's own where those do not divide the other dimension
        rows_x = gen_pass_supported_len(m1) ? gen_pass_tile_rows(m1, m2) : 0;
        rows_y = gen_pass_supported_len(m2) ? gen_pass_tile_rows(m2, m1) : 0;
        if (jit) { // a mixed-radix length without compiled-in kernels (for these tile rows): compile them now (or take them from the cache); failure is not an error
            jit_x = gen_jit_prepare(m1, rows_x, &jit_note);
            jit_y = (m2 == m1) ? jit_x : gen_jit_prepare(m2, rows_y, &jit_note);
            if (!jit_note.empty() && std::getenv("FDES_JIT_VERBOSE")) std::fprintf(stderr, "  FDES: run-time-length kernels (%s)\n", jit_note.c_str());
        }
        // rows beyond 4096 points (and lengths with a factor 17, 19, 23) exist as compiled kernels only: without them the grid takes rocFFT like any unsupported size
        if ((gen_pass_needs_compiled(m1) && rows_x > 0 && !jit_x) || (gen_pass_needs_compiled(m2) && rows_y > 0 && !jit_y)) {
            if (opt == 2) { if (err) *err = "rows beyond 4096 points and lengths with a factor 17, 19 or 23 need their kernels compiled at plan creation (hipRTC; option jit, FDES_JIT): " + jit_note; return -1; }
            backend = 1;
            jit_x = jit_y = nullptr;
            rows_x = rows_y = 0;
        }
    }
    if (backend == 2) {
        if (upload_twiddles(m1, &tw0x, &tw1x, st, err)) return -1;
        if (upload_twiddles(m2, &tw0y, &tw1y, st, err)) return -1;
        if (hipMalloc((void**)&scratch, sizeof(float2) * (size_t)m1 * m2) != hipSuccess) { if (err) *err = "scratch allocation failed"; return -1; }
        return 0;
    }
    const size_t lengths[2] = {(size_t)m1, (size_t)m2}; // rocFFT: lengths[0] is the fastest dimension
    rocfft_status s;
    s = rocfft_plan_create(&fwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_single, 2, lengths, 1, nullptr);
    if (s != rocfft_status_success) { if (err) *err = "rocfft_plan_create(forward) failed: " + std::to_string((int)s); return -1; }
    s = rocfft_plan_create(&inv, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_single, 2, lengths, 1, nullptr);
    if (s != rocfft_status_success) { if (err) *err = "rocfft_plan_create(inverse) failed: " + std::to_string((int)s); return -1; }
    size_t w1 = 0, w2 = 0;
    rocfft_plan_get_work_buffer_size(fwd, &w1);
    rocfft_plan_get_work_buffer_size(inv, &w2);
    work_bytes = w1 > w2 ? w1 : w2;
    s = rocfft_execution_info_create(&info);
    if (s != rocfft_status_success) { if (err) *err = "rocfft_execution_info_create failed"; return -1; }
    if (work_bytes) {
        if (hipMalloc(&work, work_bytes) != hipSuccess) { if (err) *err = "work buffer allocation failed"; return -1; }
        rocfft_execution_info_set_work_buffer(info, work, work_bytes);
    }
    rocfft_execution_info_set_stream(info, st);
    return 0;
}

hipError_t Fft2D::exec(float2* data, bool inverse, hipStream_t st)
{
    if (backend == 2) {
        // pass 1: rows along x (length m1, m2 rows) -> scratch[kx][y]; pass 2: rows along y -> data[ky][kx]
        const int xf = inverse ? XF_INV : XF_FWD;
        PassArgs a;
        a.in0 = data; a.out = scratch; a.tw0 = tw0x; a.tw1 = tw1x; a.nrows = m2; a.wg = wg; a.jit = jit_x; a.tile_rows = rows_x;
        hipError_t e = lds_pass(m1, xf, MID_NONE, XF_NONE, true, a, st);
        if (e != hipSuccess) return e;
        PassArgs b;
        b.in0 = scratch; b.out = data; b.tw0 = tw0y; b.tw1 = tw1y; b.nrows = m1; b.wg = wg; b.jit = jit_y; b.tile_rows = rows_y;
        return lds_pass(m2, xf, MID_NONE, XF_NONE, true, b, st);
    }
    void* in[1] = {data};
    rocfft_status s = rocfft_execute(inverse ? inv : fwd, in, nullptr, info);
    return s == rocfft_status_success ? hipSuccess : hipErrorUnknown;
}

void Fft2D::destroy()
{
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (info) rocfft_execution_info_destroy(info);
    void* ptrs[] = {work, tw0x, tw1x, tw0y, tw1y, scratch};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    fwd = inv = nullptr;
    info = nullptr;
    work = nullptr;
    tw0x = tw1x = tw0y = tw1y = scratch = nullptr;
}

} // namespace fdes
