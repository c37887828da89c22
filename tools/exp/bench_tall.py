#!/usr/bin/env python3
"""Does one launch over twice the rows (a batch of two configurations) beat two launches / two streams?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", 256)
for name, key in (("P4", (1, 4, 2, 1)), ("P6", (1, 6, 2, 1)), ("P5", (2, 5, 1, 1)), ("P3", (2, 3, 1, 1)), ("copyT", (0, 0, 0, 1))):
    row = name
    for tall in (1, 2, 4):
        eng.set_option("bench_tall", tall)
        a = eng.bench_pass(2048, *key, 100, 1) / tall
        b = eng.bench_pass(2048, *key, 100, 2) / tall
        row += f" | rows x{tall}: {a:6.2f} us per 2048 rows (2 streams {b:6.2f})"
    print(row)
