#!/usr/bin/env python3
"""Round 4: adaptive start offset of the one-wave-per-row passes (engine option stagger < 0: a workgroup that starts on a CU
less than -stagger x 0.1 us after the previous one waits for the rest of that time, once): 2048-point passes, us per
launch alone / on two streams, against no offset (0) and the fixed per-wave delays of round 3 (> 0).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdes_amd
PASSES = {"P4 mask": (1, 4, 2, 1), "P5 mulpsi": (2, 5, 1, 1), "P6 ptab": (1, 6, 2, 1), "P3 pair": (2, 12, 1, 1)}
BAND = {4: 1, 6: 1, 5: 6, 12: 4}
for name, key in PASSES.items():
    row = f"{name:10s}"
    for stg in (0, -20, -40, -60, -80, 32):
        eng = fdes_amd.Engine(0, pass_threads=64, stagger=stg, bench_band=BAND.get(key[1], 0), bench_pitch=32)
        res = [f"{eng.bench_pass(2048, key[0], key[1], key[2], key[3], 300, ns):6.2f}" for ns in (1, 2)]
        eng.close()
        row += f" | s{stg}: " + "/".join(res)
    print(row, flush=True)
