"""Single-configuration jobs (one lane, two-stream slice loop): p4_pair on / off.  usage: single_config.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fdes_amd
from tests import specimens as S

def run(name, hp, at, opts, reps):
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, skip_empty=0, **opts)
    pl = eng.plan(hp, at)
    def job():
        pl.begin_measurement(0)
        pl.run_config(0, 0, 1.0)
        pl.end_measurement(0)
        pl.sync()
    job()
    t0 = time.perf_counter()
    for _ in range(reps):
        job()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:44s} {opts}: {dt * 1e3:8.3f} ms  {pl.m3 / dt:8.0f} slice-propagations/s", flush=True)
    pl.close(); eng.close()

for rep in range(2):
    for opts in (dict(p4_pair=0), dict(p4_pair=1), dict(split=0)):
        hp, at = S.case_c1(); run("C1 256^2 x 8", hp, at, opts, 50)
        hp, at = S.case_c2(); run("C2 Si[001] 1024^2 x 64", hp, at, opts, 20)
        hp, at = S.case_c3(); hp.set(frPh=0); run("C3 specimen 2048^2 x 256, one configuration", hp, at, opts, 3)
