#!/usr/bin/env python3
"""How much of a pass's rate is the Infinity Cache (256 MiB memory-side cache; rocprofv3 exposes no hit counter for it on
this box, profiles/r03_counters_available.txt)?  The same launch sequence on ONE stream over 1 buffer set (2048^2: about
150 MB, resident) and round-robin over 8 sets (about 1.2 GB: every launch finds its operands in HBM only); 4096^2: one
set is 600 MB already.  Mean launch time [us]; run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdes_amd

PASSES = {"copy T": (0, 0, 0, 1), "P4 mask": (1, 4, 2, 1), "P5 mulpsi": (2, 5, 1, 1), "P6 ptab": (1, 6, 2, 1)}
BAND = {4: 1, 6: 1, 5: 6, 12: 4}
for n in [int(x) for x in (sys.argv[1:] or ["2048", "4096"])]:
    for name, key in PASSES.items():
        row = f"n={n:5d} {name:10s}"
        for sets in (1, 2, 4, 8):
            eng = fdes_amd.Engine(0, pass_threads=0 if n != 2048 else 64, bench_band=BAND.get(key[1], 0), bench_pitch=32 if n == 2048 else 64,
                                  bench_serial=1)
            if n == 4096:
                eng.set_option("pass_threads", 512)
            row += f" | {sets} set(s): {eng.bench_pass(n, key[0], key[1], key[2], key[3], 100, sets):7.2f}"
            eng.close()
        print(row, flush=True)
