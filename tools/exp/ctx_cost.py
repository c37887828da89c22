#!/usr/bin/env python3
"""What a lane costs to set up: context (stream) creation / destruction, plan creation / destruction with 1, 2, 3 lanes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
from tests import specimens as S
e0 = fdes_amd.Engine(0)
for rep in range(3):
    t0 = time.perf_counter(); e = fdes_amd.Engine(0); t1 = time.perf_counter(); e.close(); t2 = time.perf_counter()
    print(f"context create {1e3 * (t1 - t0):.2f} ms, destroy {1e3 * (t2 - t1):.2f} ms")
hp, at = S.case_c4(n3=32, frPh=0, n=256, dn=128)
fdes_amd.consistent(hp)
for lanes in (1, 2, 3, 1, 2, 3):
    eng = fdes_amd.Engine(0, lanes=lanes, gang=0)
    t0 = time.perf_counter(); pl = eng.plan(hp, at); pl.sync(); t1 = time.perf_counter(); n = pl.lanes(); pl.close(); t2 = time.perf_counter()
    print(f"lanes {n}: plan create {1e3 * (t1 - t0):.2f} ms, destroy {1e3 * (t2 - t1):.2f} ms")
    eng.close()
