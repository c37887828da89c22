// stream_overlap.hip — which HIP streams of one process run kernels side by side?  Creates streams in the order given on
// the command line (n = normal, h = high, l = low priority; upper case: created but left idle), then for every pair launches a spin kernel of ~T us on each
// (one workgroup: they fit beside each other) and prints elapsed / T: ~1 = concurrent, ~2 = one after the other.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/stream_overlap tools/exp/stream_overlap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (cycles < 0) *sink = 1;
}

int main(int argc, char** argv)
{
    const char* spec = argc > 1 ? argv[1] : "nnnn";
    const int n = (int)std::strlen(spec);
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    std::vector<hipStream_t> st(n);
    for (int i = 0; i < n; i++) {
        const char lc = (char)(spec[i] | 0x20);
        if (lc == 'n') hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        else hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, lc == 'h' ? greatest : least);
    }
    int* sink = nullptr;
    hipMalloc(&sink, 4);
    const long long cyc = 100 * 200; // wall_clock64 runs at 100 MHz: 200 us
    for (int i = 0; i < n; i++) { spin<<<1, 64, 0, st[i]>>>(100, sink); }
    hipDeviceSynchronize();
    auto run = [&](int a, int b) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        spin<<<1, 64, 0, st[a]>>>(cyc, sink);
        if (b >= 0) spin<<<1, 64, 0, st[b]>>>(cyc, sink);
        hipStreamSynchronize(st[a]);
        if (b >= 0) hipStreamSynchronize(st[b]);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    const bool quick = argc > 2;
    int first = 0;
    while (first < n && !(spec[first] & 0x20)) first++;
    const double one = run(first, -1);
    if (quick) {
        double r[3];
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) if (spec[i] & 0x20) spin<<<1, 64, 0, st[i]>>>(cyc, sink);
            hipDeviceSynchronize();
            r[rep] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / one;
        }
        std::printf("%-12s all active at once: %.2f %.2f %.2f\n", spec, r[0], r[1], r[2]);
        return 0;
    }
    std::printf("streams %s, one kernel %.0f us; pair time / single:\n     ", spec, one);
    for (int b = 0; b < n; b++) std::printf("  %c%d ", spec[b], b);
    std::printf("\n");
    for (int a = 0; a < n; a++) {
        std::printf("%c%d  ", spec[a], a);
        for (int b = 0; b < n; b++) {
            if (b <= a) { std::printf("   .  "); continue; }
            std::printf(" %4.2f ", run(a, b) / one);
        }
        std::printf("\n");
    }
    // all at once
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) spin<<<1, 64, 0, st[i]>>>(cyc, sink);
    hipDeviceSynchronize();
    std::printf("all %d at once: %.2f\n", n, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / one);
    return 0;
}
