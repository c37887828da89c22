#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", 256)
for name, key in (("P4", (1, 4, 2, 1)), ("P6", (1, 6, 2, 1)), ("P5", (2, 5, 1, 1)), ("P3", (2, 3, 1, 1)), ("copyT", (0, 0, 0, 1))):
    row = name
    for bb in (0, 1, 2, 4, 6):
        eng.set_option("bench_band", bb)
        row += f" | band{bb}: {eng.bench_pass(2048, *key, 200, 1):6.2f} x2 {eng.bench_pass(2048, *key, 200, 2):6.2f}"
    print(row)
