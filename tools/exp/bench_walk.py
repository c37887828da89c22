#!/usr/bin/env python3
"""Round 4: walking workgroups (FDES_W_WALK=1 build, pass_threads = 65: 256 workgroups, each walks its row groups, no
look-ahead, two waves per SIMD) against one row group per workgroup (pass_threads = 64), 2048-point rows, pass by pass:
us per launch alone / on two streams.  FDES_LIB selects the build.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
PASSES = {"copy T": (0, 0, 0, 1), "P2 gtab": (1, 2, 2, 1), "P3 pair": (2, 12, 1, 1), "P4 mask": (1, 4, 2, 1), "P5 mulpsi": (2, 5, 1, 1), "P6 ptab": (1, 6, 2, 1)}
BAND = {4: 1, 6: 1, 5: 6, 12: 4}
n = 2048
for name, key in PASSES.items():
    row = f"n={n:5d} {name:10s}"
    for wg in (64, 65):
        eng = fdes_amd.Engine(0, pass_threads=wg, bench_band=BAND.get(key[1], 0), bench_pitch=32)
        res = [f"{eng.bench_pass(n, key[0], key[1], key[2], key[3], 300, ns):6.2f}" for ns in (1, 2)]
        eng.close()
        row += f" | wg{wg}: " + "/".join(res)
    print(row, flush=True)
