#!/usr/bin/env python3
"""Half-size LDS regions (FDES_W_HALFX builds of fft_wave.hip; FDES_LIB selects the build): P4 / P6 at 2048 points,
mean launch time [us] on one and two streams.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
eng = fdes_amd.Engine(0, pass_threads=64, bench_band=1, bench_pitch=32)
for name, key in (("P4 mask", (1, 4, 2, 1)), ("P6 ptab", (1, 6, 2, 1))):
    out = []
    for rep in range(3):
        out.append("/".join(f"{eng.bench_pass(2048, *key, 200, ns):6.2f}" for ns in (1, 2)))
    print(os.path.basename(os.environ.get("FDES_LIB", "tree")), name, "x1/x2 us:", "  ".join(out), flush=True)
