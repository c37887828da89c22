#!/usr/bin/env python3
"""Round 4, 4096-point rows: band-limit (P4) and propagator (P6) passes with TWO workgroups per CU (one wave per row,
half-size LDS regions: FDES_W_HALFX=12 build of fft_wave.hip, FDES_LIB selects it) against the defaults.
Mean launch time [us] alone / on two streams.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
for wg in (512, 64):
    eng = fdes_amd.Engine(0, pass_threads=wg, bench_band=1, bench_pitch=64)  # (P2 runs over all rows: bench_band is reset for it below)
    for name, key in (("P4 mask", (1, 4, 2, 1)), ("P6 ptab", (1, 6, 2, 1)), ("P2 gtab", (1, 2, 2, 1))):
        eng.set_option("bench_band", 0 if key[1] == 2 else 1)
        out = []
        for rep in range(2):
            out.append("/".join(f"{eng.bench_pass(4096, *key, 100, ns):7.2f}" for ns in (1, 2)))
        print(os.path.basename(os.environ.get("FDES_LIB", "tree")), f"wg={wg}", name, "x1/x2 us:", "  ".join(out), flush=True)
    eng.close()
