import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
from tests import specimens as S
n3, frph = 4, 8
hp, at = S.case_c4(n3=n3, frPh=frph)
fdes_amd.consistent(hp)
for rep in range(2):
    for pad in (0, 16, 32, 64):
        eng = fdes_amd.Engine(0, pitch_pad=pad)
        pl = eng.plan(hp, at)
        pl.begin_measurement(0)
        for j in range(3): pl.run_config(0, 100 + j, 0.0)
        pl.sync()
        t0 = time.perf_counter()
        for k in range(n3):
            pl.begin_measurement(k)
            for j in range(frph): pl.run_config(k, j, 1.0 / frph)
            pl.end_measurement(k)
        pl.sync()
        dt = time.perf_counter() - t0
        print(f"C4 1024^2 pad {pad}: {n3 * frph * hp.c.m3 / dt:.0f}", flush=True)
        pl.close(); eng.close()
