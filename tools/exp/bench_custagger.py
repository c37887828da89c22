#!/usr/bin/env python3
"""Round 5: start offset by CU class (engine option stagger < 0: the first generation of workgroups on the odd CUs starts
-stagger x 64 cycles late, so that half of the chip is in its transform phase while the other half loads or stores).
Mean launch time [us] of a pass on one / two streams over a sweep of the offset.
   python3 tools/exp/bench_custagger.py n [wg]        (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wg = int(sys.argv[2]) if len(sys.argv) > 2 else (512 if n > 2048 else 64)
PASSES = [("P5 product", 2, 5, 1, 6), ("P6 propagator", 1, 6, 2, 1), ("P4 band limit", 1, 4, 2, 1), ("P3 pair", 2, 12, 1, 4), ("P2 filter", 1, 2, 2, 0)]
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", wg)
for name, pre, mid, post, band in PASSES:
    eng.set_option("bench_band", band)
    for st in (0, -75, -150, -225, -300, -450, 0):
        eng.set_option("stagger", st)
        r = "  ".join("/".join(f"{eng.bench_pass(n, pre, mid, post, 1, 100, ns):7.2f}" for ns in (1, 2)) for _ in range(2))
        print(f"n={n} wg={wg} {name:14s} stagger {st:5d} ({-st * 64 / 2400.0:5.1f} us)  x1/x2 us: {r}", flush=True)
