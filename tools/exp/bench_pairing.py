#!/usr/bin/env python3
"""Two streams: does it matter WHICH kernels run side by side?  Time for 100 launches of X on stream 0 and 100 of Y on
stream 1 (concurrently), for same-kernel and mixed pairs; 2048^2, band flags as in the slice loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", 256)
K = {"P3": (2, 3, 1), "P4": (1, 4, 2), "P5": (2, 5, 1), "P6": (1, 6, 2)}
eng.set_option("bench_band", 0)
single, same = {}, {}
for k, v in K.items():
    eng.set_option("bench_alt", -1)
    single[k] = eng.bench_pass(2048, *v, 1, 100, 1)
    same[k] = eng.bench_pass(2048, *v, 1, 100, 2) * 2   # time for one launch on each of the two streams
print("alone      :", {k: round(v, 1) for k, v in single.items()})
print("same pair  :", {k: round(v, 1) for k, v in same.items()}, "(us for one launch on each stream)")
names = list(K)
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        a, b = names[i], names[j]
        eng.set_option("bench_alt", K[b][0] * 10000 + K[b][1] * 100 + K[b][2])
        t = eng.bench_pass(2048, *K[a], 1, 100, 2) * 2
        print(f"{a} || {b}: {t:6.1f} us   (mean of the same-kernel pairs {0.5 * (same[a] + same[b]):6.1f}, sum alone {single[a] + single[b]:6.1f})")
