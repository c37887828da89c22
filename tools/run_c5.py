#!/usr/bin/env python3
"""BASELINE config 5 at full size: Au cuboctahedron k = 60 (738 221 atoms), 4096^2 wave, 512 slices; a few of the 16
frozen-phonon configurations, throughput and a sanity check of the image (finite, mean ~ 1 with imPot = 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fdes_amd
from tests import specimens as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opts = dict(kv.split("=") for kv in sys.argv[2:])   # e.g. pitch_pad=0 lanes=1
hp, at = S.case_c5()
if "slices" in opts:   # shortened run for counter collection: the first slices of the same specimen
    hp.set(m3=int(opts.pop("slices")))
fdes_amd.consistent(hp)
print("atoms", at.n, "grid", hp.c.m1, "slices", hp.c.m3, "options", opts, flush=True)
eng = fdes_amd.Engine(0, skip_empty=0, **{k: int(v) for k, v in opts.items()})
t0 = time.perf_counter()
pl = eng.plan(hp, at)
pl.begin_measurement(0)
pl.run_config(0, 100, 0.0)
pl.run_config(0, 101, 0.0)
pl.sync()
print(f"plan + 2 warm-up configurations: {time.perf_counter() - t0:.2f} s", flush=True)
t0 = time.perf_counter()
for j in range(n):
    pl.run_config(0, j, 1.0 / n)
pl.sync()
dt = time.perf_counter() - t0
pl.end_measurement(0)
img = pl.get_images()
print(f"C5: {n * hp.c.m3 / dt:.0f} slice-propagations/s ({dt / n * 1e3:.1f} ms per configuration), image mean {img.mean():.6f} min {img.min():.4f} max {img.max():.4f} finite {bool(np.isfinite(img).all())}")
pl.close()
