#!/usr/bin/env python3
"""Micro-benchmark of the LDS row passes (run on the GPU box): launch time and effective GB/s per pass type."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdes_amd
eng = fdes_amd.Engine(0)
NAMES = {(0, 0, 0, 0): "copy natural", (0, 0, 0, 1): "copy transposed", (1, 0, 0, 0): "FFT natural", (1, 0, 0, 1): "FFT transposed",
         (1, 1, 0, 1): "P1 FFT+clear T", (1, 2, 2, 1): "P2 FFT*G IFFT T", (2, 3, 1, 1): "P3 IFFT exp FFT T", (1, 4, 2, 1): "P4 FFT mask IFFT T",
         (2, 5, 1, 1): "P5 2xIFFT mul FFT T", (1, 6, 2, 1): "P6 FFT*P IFFT T"}
BYTES = {(0, 0, 0, 0): 16, (0, 0, 0, 1): 16, (1, 0, 0, 0): 16, (1, 0, 0, 1): 16, (1, 1, 0, 1): 24, (1, 2, 2, 1): 20, (2, 3, 1, 1): 16,
         (1, 4, 2, 1): 16, (2, 5, 1, 1): 24, (1, 6, 2, 1): 24}
for n in [int(x) for x in (sys.argv[1:] or ["2048"])]:
    for key, name in NAMES.items():
        row = f"n={n:5d} {name:22s}"
        for wg, lp in ((256, 1), (512, 1)):
            for ns in (1, 2):
                eng.set_option("pass_threads", wg)
                try:
                    us = eng.bench_pass(n, key[0], key[1], key[2], key[3], 200, ns)
                    row += f" | wg{wg}l{lp}x{ns}: {us:7.2f}"
                except Exception as e:
                    row += f" | wg{wg}l{lp}x{ns}: n/a"
        print(row)
