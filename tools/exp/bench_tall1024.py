#!/usr/bin/env python3
"""1024-point passes (one wave per row) over the rows of 1, 2, 4 configurations in one launch, on 1 - 3 streams:
us per configuration.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", 64)
for name, key in (("P4", (1, 4, 2, 1)), ("P6", (1, 6, 2, 1)), ("P5", (2, 5, 1, 1)), ("P3", (2, 12, 1, 1))):
    row = name
    for tall in (1, 2, 4):
        eng.set_option("bench_tall", tall)
        row += f" | x{tall}: " + "/".join(f"{eng.bench_pass(n, *key, 200, ns) / tall:5.2f}" for ns in (1, 2, 3))
    print(row, "us per configuration on 1/2/3 streams", flush=True)
