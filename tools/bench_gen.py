#!/usr/bin/env python3
"""Mixed-radix LDS-resident passes (fft_gen.hip) pass by pass: mean launch time [us] on one / two / three streams, beside
the power-of-two kernels at the neighbouring size.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdes_amd

PASSES = {"copy N": (0, 0, 0, 0), "copy T": (0, 0, 0, 1), "FFT T": (1, 0, 0, 1), "P2 gtab": (1, 2, 2, 1), "P3 pair": (2, 12, 1, 1), "P4 mask": (1, 4, 2, 1),
          "P5 mulpsi": (2, 5, 1, 1), "P6 ptab": (1, 6, 2, 1)}
BAND = {4: 1, 6: 1, 5: 6, 12: 4}
for n in [int(x) for x in (sys.argv[1:] or ["1000", "1024", "800", "320"])]:
    for name, key in PASSES.items():
        eng = fdes_amd.Engine(0, bench_band=BAND.get(key[1], 0))
        row = f"n={n:5d} {name:10s}"
        for ns in (1, 2, 3):
            row += f" | x{ns}: {eng.bench_pass(n, key[0], key[1], key[2], key[3], 200, ns):6.2f}"
        eng.close()
        print(row, flush=True)
