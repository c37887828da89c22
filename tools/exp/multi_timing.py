#!/usr/bin/env python3
"""Round 5: plan creation of the one-process multi-GPU driver's workers (fdes_build_measurements_multi, FDES_TIMING=1): all
workers on device 0 of this box (one GPU), headline grid; the lines show whether the creations ran side by side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FDES_TIMING"] = "1"
import fdes_amd
from tests import specimens as S
hp, at = S.case_c3(m3=16, frPh=8)
fdes_amd.consistent(hp)
for n in (1, 2, 4):
    t0 = time.perf_counter()
    fdes_amd.build_measurements_multi([0] * n, hp, at)
    print(f"{n} workers: {time.perf_counter() - t0:.3f} s for the whole call", flush=True)
