#!/usr/bin/env python3
"""C4 as a job (begin / run x 8 / end per tilt) with a given number of lanes and tilts: slice-propagations/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
from tests import specimens as S
nk = int(os.environ.get("NK", "16"))
hp, at = S.case_c4(n3=int(os.environ.get("N3", "64")))
fdes_amd.consistent(hp)
count = max(hp.c.frPh, 1)
DUMMY = int(os.environ.get("DUMMY", "0"))  # 1: another context exists while this one is created; 2: and stays
for lanes in [int(a) for a in sys.argv[1:]] or [3, 4]:
    dummy = [fdes_amd.Engine(0) for _ in range(int(os.environ.get("NDUMMY", "1")))] if DUMMY else []
    eng = fdes_amd.Engine(0, lanes=lanes, gang=int(os.environ.get("GANG", "-1")))
    if DUMMY == 1:
        for d_ in dummy: d_.close()
    pl = eng.plan(hp, at)
    def job():
        for k in range(nk):
            pl.begin_measurement(k)
            for j in range(count):
                pl.run_config(k, j, 1.0 / count)
            pl.end_measurement(k)
        pl.sync()
    warm = int(os.environ.get("WARM", "-1"))
    if warm < 0:
        job()
    else:
        pl.begin_measurement(0)
        for j in range(warm):
            pl.run_config(0, 100 + j, 0.0)
        pl.sync()
    t0 = time.perf_counter()
    job()
    dt = time.perf_counter() - t0
    print(f"lanes {lanes} ({pl.lanes()}) gang {pl.gang()}, {nk} tilts x {count}: {nk * count * pl.m3 / dt:.0f} slice-propagations/s", flush=True)
    pl.close(); eng.close()
