"""Single-configuration jobs: two-stream slice loop (split) against the one-stream hipGraph, by grid size and slice count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
from tests import specimens as S

def run(hp, at, opts, reps):
    eng = fdes_amd.Engine(0, skip_empty=0, **opts)
    pl = eng.plan(hp, at)
    def job():
        pl.begin_measurement(0); pl.run_config(0, 0, 1.0); pl.end_measurement(0); pl.sync()
    job(); job()
    t0 = time.perf_counter()
    for _ in range(reps):
        job()
    dt = (time.perf_counter() - t0) / reps
    pl.close(); eng.close()
    return pl.m3 / dt, dt

for m in (256, 512, 1024):
    for m3 in (8, 32, 128):
        hp, at = S.case_tiny(m=m, m3=m3, nz=2, nat=max(200, m3 * 40), seed=4)
        fdes_amd.consistent(hp)
        r = {}
        for name, opts in (("split", dict(split=1)), ("graph", dict(split=0)), ("auto", dict())):
            r[name] = run(hp, at, opts, 30 if m < 1024 else 10)
        print(f"{m}^2 x {m3:3d} slices: split {r['split'][0]:8.0f}/s ({r['split'][1]*1e3:7.3f} ms)  one stream + graph {r['graph'][0]:8.0f}/s ({r['graph'][1]*1e3:7.3f} ms)  auto {r['auto'][0]:8.0f}/s", flush=True)
