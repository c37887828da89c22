#!/usr/bin/env python3
"""Does a padded row pitch (not a power of two) help the transposed stores?  us per launch, 1 and 2 streams."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
eng = fdes_amd.Engine(0)
PASSES = {"copy nat": (0, 0, 0, 0), "copy T": (0, 0, 0, 1), "FFT T": (1, 0, 0, 1), "P4": (1, 4, 2, 1), "P5": (2, 5, 1, 1), "P6": (1, 6, 2, 1)}
for n in [int(x) for x in (sys.argv[1:] or ["2048", "4096"])]:
    for wg in ((256, 512) if n <= 2048 else (512,)):
        eng.set_option("pass_threads", wg)
        for name, key in PASSES.items():
            row = f"n={n} wg={wg} {name:9s}"
            for pitch in (0, 4, 8, 16, 32, 64, 136):
                eng.set_option("bench_pitch", pitch)
                cells = []
                for ns in (1, 2, 3):
                    cells.append(f"{eng.bench_pass(n, key[0], key[1], key[2], key[3], 100, ns):6.1f}")
                row += f" | p{pitch}: " + "/".join(cells)
            print(row, flush=True)
