#!/usr/bin/env python3
"""Single-image jobs (one configuration: no lanes): slice-propagations/s of C1 / C2-like specimens for the settings of the
one-lane slice loop - single stream + hipGraph (split = 0), two-stream split loop (batch = 0), batched potential chain
(batch = 2 / 4 / 8) - with and without the empty-slice short cut.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fdes_amd
from tests import specimens as S


def rate(hp, at, reps, **opts):
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, **opts)
    pl = eng.plan(hp, at)
    def job():
        pl.begin_measurement(0)
        pl.run_config(0, 0, 1.0)
        pl.end_measurement(0)
        pl.sync()
    job(); job()
    t0 = time.perf_counter()
    for _ in range(reps):
        job()
    dt = (time.perf_counter() - t0) / reps
    r = pl.m3 / dt
    pl.close(); eng.close()
    return r


cases = {"C1 SrTiO3 256^2 x 8": S.case_c1(), "C2 Si[001] 1024^2 x 64": S.case_c2(),
         "Si[001] 512^2 x 64": S.case_c2(n=256, dn=128), "Si[001] 1000^2 x 64": S.case_c2(n=500, dn=250)}
settings = [("split=0 (graph)", dict(split=0)), ("split=1 batch=0", dict(split=1, batch=0)), ("batch=2", dict(batch=2)),
            ("batch=4", dict(batch=4)), ("batch=8", dict(batch=8)), ("default", dict())]
for name, (hp, at) in cases.items():
    for skip in (0, 1):
        row = f"{name:24s} skip_empty={skip}"
        for label, o in settings:
            try:
                row += f" | {label}: {rate(hp, at, 10, skip_empty=skip, **o):8.0f}"
            except Exception as e:
                row += f" | {label}: n/a ({str(e)[:30]})"
        print(row, flush=True)
