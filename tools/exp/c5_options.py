#!/usr/bin/env python3
"""Round 5: BASELINE config 5 (Au k = 60, 4096^2 x 512 slices) under a list of engine option sets, interleaved over several
rounds in ONE process (box-to-box spread does not enter): slice-propagations/s per set and round, then min / median / max.
   python3 tools/exp/c5_options.py ROUNDS NCONFIG "lanes=2" "batch=2" "pass_threads=256 batch=2" ..."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fdes_amd
from tests import specimens as S
rounds, n = int(sys.argv[1]), int(sys.argv[2])
sets = [dict(kv.split("=") for kv in a.split()) for a in sys.argv[3:]] or [{}]
size = int(os.environ.get("C5_SIZE", "0"))
hp, at = S.case_c5()
if size:   # the same specimen on a smaller / other grid (mixed-radix sizes): pixel size scaled so that the particle fits
    hp.set(n1=size // 2, n2=size // 2, dn1=size // 4, dn2=size // 4, d1=0.25e-10 * 4096 / size, d2=0.25e-10 * 4096 / size)
if os.environ.get("C5_SLICES"):
    hp.set(m3=int(os.environ["C5_SLICES"]))
fdes_amd.consistent(hp)
print("atoms", at.n, "grid", hp.c.m1, "slices", hp.c.m3, flush=True)
res = [[] for _ in sets]
ref = None
for r in range(rounds):
    for i, o in enumerate(sets):
        eng = fdes_amd.Engine(0, skip_empty=0, **{k: int(v) for k, v in o.items()})
        pl = eng.plan(hp, at)
        pl.begin_measurement(0)
        pl.run_config(0, 100, 0.0)
        pl.run_config(0, 101, 0.0)
        pl.sync()
        t0 = time.perf_counter()
        for j in range(n):
            pl.run_config(0, j, 1.0 / n)
        pl.sync()
        dt = time.perf_counter() - t0
        pl.end_measurement(0)
        img = pl.get_images()
        if ref is None:
            ref = img
        dev = float(np.abs(img - ref).max())
        res[i].append(n * hp.c.m3 / dt)
        print(f"round {r} {o}: {res[i][-1]:.0f} slice-propagations/s, image mean {img.mean():.6f}, max |diff to first set| {dev:.2e}", flush=True)
        pl.close()
        eng.close()
print("# min / median / max")
for o, v in zip(sets, res):
    print(f"{str(o):60s} {min(v):7.0f} {statistics.median(v):7.0f} {max(v):7.0f}")
