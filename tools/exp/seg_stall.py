import sys, os, time
sys.path.insert(0, "/root/repo")
import fdes_amd
from tests import specimens as S
hp, at = S.case_c3()
fdes_amd.consistent(hp)
eng = fdes_amd.Engine(0)
pl = eng.plan(hp, at)
pl.begin_measurement(0)
for rep in range(3):
    pl.sync()
    t0 = time.perf_counter()
    for j in range(8):
        pl.run_config(0, j, 1 / 32)
    pl.sync()
    dt = time.perf_counter() - t0
    print("rep", rep, "rate", 8 * 256 / dt, flush=True)
