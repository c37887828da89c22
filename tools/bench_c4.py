#!/usr/bin/env python3
"""Throughput of the C4-style workload (SrTiO3, 3 species, 1024^2, beam-tilt series) through plan.run_config."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdes_amd
from tests import specimens as S
n3, frph = 4, int(os.environ.get("FRPH", "8"))
BATCH = int(os.environ.get("BATCH", "-1"))
GANG = int(os.environ.get("GANG", "-1"))
N = int(os.environ.get("N", "512"))   # image size; the wave is 2 N points across
hp, at = S.case_c4(n3=n3, frPh=frph, n=N, dn=N // 2)
fdes_amd.consistent(hp)
for lanes in ([int(a) for a in sys.argv[1:]] or [2, 3]):
    eng = fdes_amd.Engine(0, lanes=lanes, skip_empty=int(os.environ.get('SKIP', '1')), batch=BATCH, gang=GANG)
    pl = eng.plan(hp, at)
    pl.begin_measurement(0)
    for j in range(3):
        pl.run_config(0, 100 + j, 0.0)
    pl.sync()
    t0 = time.perf_counter()
    for k in range(n3):
        pl.begin_measurement(k)
        for j in range(frph):
            pl.run_config(k, j, 1.0 / frph)
        pl.end_measurement(k)
    pl.sync()
    dt = time.perf_counter() - t0
    print(f"C4 SrTiO3 {2 * N}^2 x {hp.c.m3} slices, lanes {lanes} ({pl.lanes()}) batch {BATCH} gang {pl.gang()}: {n3 * frph * hp.c.m3 / dt:.0f} slice-propagations/s ({dt / (n3 * frph) * 1e3:.2f} ms per configuration)")
    pl.close()
