"""Single-configuration jobs: workgroup geometry (one row per thread / two rows per thread) by grid size."""
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
    return pl.m3 / dt

for m in (256, 512, 1024, 2048):
    hp, at = S.case_tiny(m=m, m3=64, nz=2, nat=2000, seed=4)
    fdes_amd.consistent(hp)
    r = {k: run(hp, at, dict(pass_threads=k), 20 if m < 2048 else 5) for k in (1, 256, 512)}
    print(f"{m}^2 x 64 slices, one configuration: one row per thread {r[1]:8.0f}/s   256 x 2 rows {r[256]:8.0f}/s   512 x 2 rows {r[512]:8.0f}/s", flush=True)
