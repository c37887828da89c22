#!/usr/bin/env python3
"""The reference's bin/test.qsc (SrTiO3 9x9x20 cells, 800^2 wave, 40 slices -> 400 sub-slices, CBED, one configuration)
through the boundary call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hp, at = fdes_amd.read_qsc(os.path.join(ROOT, "tests", "golden", "qsc", "test.qsc"))
fdes_amd.consistent(hp)
q, _ = fdes_amd.sub_sliced(hp)
eng = fdes_amd.Engine(0)
eng.build_measurements(hp, at)
for rep in range(3):
    t0 = time.perf_counter()
    eng.build_measurements(hp, at)
    dt = time.perf_counter() - t0
    print(f"bin/test.qsc: {q.c.m1}^2 x {q.c.m3} sub-slices, 1 configuration: {dt * 1e3:.1f} ms = {q.c.m3 / dt / 1e3:.1f} k slice-propagations/s", flush=True)
