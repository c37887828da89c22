import sys, os
sys.path.insert(0, os.getcwd())
import torch, fdes_amd
from tests import specimens as S
"""Device-memory leak check: plans of every slice-loop flavour (lanes + graph, single image on the batched chain, mixed-radix
grid, one-wave-per-row kernels) created, run and destroyed 25 times; free memory must come back.  Run on the GPU box."""
import itertools
CASES = [dict(m=256, m3=6, nz=2, frPh=2, n3=2, tilt=True, zfrac=0.3), dict(m=1024, m3=5, nz=2, nat=100), dict(m=320, m3=4, nz=1, frPh=3),
         dict(m=2048, m3=3, nz=1, frPh=2, nat=100), dict(m=72, m3=3, nz=2), dict(m=256, m3=4, nz=2, n3=9, tilt=True),
         dict(m=512, m3=4, nz=3, frPh=8, nat=200)]   # (the last two: gangs across measurements / of configurations)
free0 = None
for it, kw in zip(range(int(os.environ.get("ITERS", "25"))), itertools.cycle(CASES)):
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    out = eng.build_measurements(hp, at)
    pl = eng.plan(hp, at)
    pl.begin_measurement(0)
    for j in range(4):
        pl.run_config(0, j, 0.25)
    pl.end_measurement(0)
    pl.sync()
    pl.close()
    eng.close() if hasattr(eng, "close") else None
    del eng
    f, t = torch.cuda.mem_get_info()
    if it == int(os.environ.get("FROM", "2")): free0 = f
    if it % 6 == 0: print(it, "free MiB", f >> 20)
print("leak MiB over 22 iterations:", (free0 - f) >> 20)
