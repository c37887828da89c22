#!/usr/bin/env python3
"""4096-point passes: 512 threads (one workgroup per CU) against 256 threads (two per CU) with the second workgroup of a CU
started late (stagger, units of 64 cycles), alone / on two streams.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
PASSES = {"copy T": (0, 0, 0, 1), "FFT T": (1, 0, 0, 1), "P4 mask": (1, 4, 2, 1), "P5 mulpsi": (2, 5, 1, 1), "P6 ptab": (1, 6, 2, 1), "P3 pair": (2, 12, 1, 1)}
BAND = {4: 1, 6: 1, 5: 6, 12: 4}
for name, key in PASSES.items():
    row = f"n= 4096 {name:10s}"
    for wg, stg in ((512, 0), (256, 0), (256, 64), (256, 128), (256, 192), (256, 256), (256, 384)):
        eng = fdes_amd.Engine(0, pass_threads=wg, stagger=stg, bench_band=BAND.get(key[1], 0), bench_pitch=64)
        res = "/".join(f"{eng.bench_pass(4096, key[0], key[1], key[2], key[3], 100, ns):6.1f}" for ns in (1, 2))
        eng.close()
        row += f" | wg{wg} s{stg}: {res}"
    print(row, flush=True)
