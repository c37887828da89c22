#!/usr/bin/env python3
"""Round 5 edition (three-stage register-chained mixed-radix kernels beyond 1024 points incl. the reversed trailing transform,
non-temporal row loads at 4096 points and beyond 2048 mixed-radix; more weight on those grid lengths).  Round 4 edition (grid lengths with radix 7, one-image mixed-radix kernels beyond 1024 points, two-row tiles beyond 2048,
4096-point rows with two workgroups per CU, the empty-slice question on its own stream).  Original text:
One-off stress of the round-3 launch paths (gangs of configurations / measurements, lanes, all kernel families) against
the oracle: random draws over grid sizes 256 ... 2048 incl. mixed-radix ones, modes, species, tilts, frozen phonons,
measurements, gang / lane options.  python tools/exp/sweep_stress.py [first_seed] [count]   (test infrastructure: uses the
oracle as the checker)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fdes_amd
from tests import specimens as S, oracle_py
oracle_py.lib()
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
worst = 0.0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    m = int(rng.choice([256, 896, 1024, 1280, 1400, 1600, 2000, 2048, 2560, 3000, 3072, 3200, 3600, 4000, 4096, 1280, 1600, 2000, 2560, 3000, 3600, 4000, 640, 1792, 2880]))
    kw = dict(m=m, m3=int(rng.integers(1, 7)), nz=int(rng.integers(1, 4)), frPh=int(rng.choice([0, 0, 2, 3, 5, 6])),
              mode=int(rng.choice([0, 0, 1, 2])), n3=int(rng.integers(1, 7)), seed=int(rng.integers(0, 1000)),
              tilt=bool(rng.integers(0, 2)), beam_tilt=bool(rng.integers(0, 2)), imPot=float(rng.choice([0.0, 0.05, 0.2])),
              nat=int(rng.integers(1, 200)), sub=int(rng.integers(1, 3)), zfrac=float(rng.choice([0.5, 0.3, 0.15])),
              pD=float(rng.choice([0.0, 0.0, 40.0])))
    if m >= 2000:
        kw["n3"] = min(kw["n3"], 2); kw["frPh"] = min(kw["frPh"], 2); kw["m3"] = min(kw["m3"], 4)
    if m >= 3000:
        kw["n3"] = 1; kw["sub"] = 1
    opts = dict(gang=int(rng.choice([-1, -1, 0, 2, 3, 4, 8, 16])), lanes=int(rng.choice([0, 0, 1, 2, 3])), skip_empty=int(rng.integers(0, 2)))
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, **opts)
    out = eng.build_measurements(hp, at)["image"]
    eng.close()
    if kw["pD"] > 0:   # dose noise: compare against the engine itself without gangs / lanes (same Philox streams), bit for bit when nothing is skipped
        eng = fdes_amd.Engine(0, gang=0, lanes=1, skip_empty=opts["skip_empty"])
        ref = eng.build_measurements(hp, at)["image"]
        eng.close()
        e = float(np.abs(out.astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))
        tol = 0.0 if opts["skip_empty"] == 0 and opts["lanes"] in (0, 1) and kw["frPh"] < 2 else 0.5
    else:
        ref = oracle_py.build_measurements(hp, at, prec="f64" if kw["frPh"] == 0 else "f32")["image"]
        e = float(np.sqrt(((out - ref) ** 2).sum() / (ref ** 2).sum()))
        tol = 2e-5
    worst = max(worst, e if kw["pD"] == 0 else 0.0)
    flag = "" if e <= tol else "   <-- FAIL"
    print(f"seed {seed}: m {m} n3 {kw['n3']} frPh {kw['frPh']} mode {kw['mode']} nz {kw['nz']} pD {kw['pD']} {opts}: E = {e:.2e}{flag}", flush=True)
    if flag:
        print(kw)
        sys.exit(1)
print("worst relative error vs oracle:", worst)
