#!/usr/bin/env python3
"""Throughput of the five BASELINE.json configurations through the boundary call with resident plans (slice-propagations/s,
whole job incl. incoming wave, detector chain; engine defaults unless noted).  C1 is the reference-sized CPU case."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fdes_amd
from tests import specimens as S


def run(name, hp, at, configs, skip_empty, reps=1):
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, skip_empty=skip_empty)
    pl = eng.plan(hp, at)
    n3, count = hp.c.n3, max(hp.c.frPh, 1)
    todo = [(i // count, i % count) for i in range(min(configs, n3 * count))]
    def job():
        cur = -1
        for (k, j) in todo:
            if k != cur:
                if cur >= 0:
                    pl.end_measurement(cur)
                pl.begin_measurement(k)
                cur = k
            pl.run_config(k, j, 1.0 / count)
        pl.end_measurement(cur)
        pl.sync()
    job()
    t0 = time.perf_counter()
    for _ in range(reps):
        job()
    dt = (time.perf_counter() - t0) / reps
    img = pl.get_images()
    rate = len(todo) * pl.m3 / dt
    print(f"{name:58s} lanes {pl.lanes()}  {len(todo):4d} configurations x {pl.m3:3d} slices: {dt * 1e3:9.2f} ms  {rate:9.0f} slice-propagations/s  finite {bool(np.isfinite(img).all())}", flush=True)
    pl.close()
    eng.close()


for skip in (0, 1):
    print(f"--- skip_empty = {skip} ({'engine default' if skip else 'every slice the full sequence, as the reference'})")
    hp, at = S.case_c1();              run("C1 SrTiO3 256^2 x 8 slices, 1 configuration", hp, at, 1, skip, reps=20)
    hp, at = S.case_c2();              run("C2 Si[001] 1024^2 x 64 slices, 1 configuration", hp, at, 1, skip, reps=20)
    hp, at = S.case_c3();              run("C3 Au 94 611 atoms 2048^2 x 256 slices, 8 of 32 configurations", hp, at, 8, skip)
    hp, at = S.case_c4();              run("C4 SrTiO3 tilt series 1024^2 x 40 slices, 64 x 8 configurations", hp, at, 512, skip)
    hp, at = S.case_c5();              run("C5 Au 738 221 atoms 4096^2 x 512 slices, 4 of 16 configurations", hp, at, 4, skip)
