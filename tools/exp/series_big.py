#!/usr/bin/env python3
"""Larger series (one configuration per measurement) through the boundary call: lanes x gang combinations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
from tests import specimens as S
n, n3 = int(sys.argv[1]), int(sys.argv[2])
hp, at = S.case_c4(n3=n3, frPh=0, n=n, dn=n // 2)
fdes_amd.consistent(hp)
for label, opts in (("auto", dict()), ("1 lane x 8", dict(lanes=1, gang=8)), ("2 lanes x 4", dict(lanes=2, gang=4)), ("2 lanes x 8", dict(lanes=2, gang=8)),
                    ("1 lane x 16", dict(lanes=1, gang=16)), ("3 lanes, no gang", dict(gang=0))):
    eng = fdes_amd.Engine(0, **opts)
    eng.build_measurements(hp, at)
    t0 = time.perf_counter()
    eng.build_measurements(hp, at)
    dt = time.perf_counter() - t0
    eng.close()
    print(f"{2 * n}^2 x {n3} tilts x {hp.c.m3} slices, {label:18s}: {n3 * hp.c.m3 / dt / 1e3:7.1f} k slice-propagations/s ({dt * 1e3:7.1f} ms)", flush=True)
