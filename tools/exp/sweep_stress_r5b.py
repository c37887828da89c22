#!/usr/bin/env python3
"""Round 5, second half: randomised stress of what it added - grid lengths with factors 11 and 13, kernels compiled at plan
creation (random lengths out of ALL supported ones, jit on / off at random), smaller tiles where the tile rows do not divide the
other dimension (m = 2 (mod 4); 8 -> 4 rows), rectangular grids of two such lengths - through the whole driver (lanes, gangs,
graphs, modes, frozen phonons) against the oracle.  python tools/exp/sweep_stress_r5b.py [first_seed] [count]
(test infrastructure: uses the oracle as the checker)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fdes_amd
from tests import specimens as S, oracle_py
oracle_py.lib()
lib = fdes_amd.load_library()
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)


def smooth13(n):
    for p in (2, 3, 5, 7, 11, 13):
        while n % p == 0:
            n //= p
    return n == 1


LENGTHS = [n for n in range(256, 4097, 2) if smooth13(n) and (n & (n - 1)) != 0]
WITH_11_13 = [n for n in LENGTHS if n % 11 == 0 or n % 13 == 0]
MOD4_2 = [n for n in LENGTHS if n % 4 == 2 and n <= 2048]
def smooth23(n):
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23):
        while n % p == 0:
            n //= p
    return n == 1


P23 = [n for n in range(256, 4097) if smooth23(n) and not smooth13(n)]   # a factor 17, 19 or 23 (seeds from 3000 on): compile-time kernels only
ODD = [n for n in range(257, 4096, 2) if smooth13(n)]   # odd lengths (seeds from 2000 on): a partial last tile on both axes
BIG = [n for n in range(4098, 8193, 2) if smooth13(n)]   # rows beyond 4096 points (kernels compiled at plan creation only): beside a short other axis, so that the oracle stays cheap
worst = 0.0
for seed in range(first, first + count):
    rng = np.random.default_rng(77000 + seed)
    pool = [LENGTHS, WITH_11_13, MOD4_2][int(rng.integers(0, 3))]
    m1 = int(rng.choice(pool))
    m2 = m1
    big_case = 1000 <= seed < 2000   # seeds 1000 ... 1999: one axis beyond 4096 points
    if seed >= 3000:
        m1 = int(rng.choice(P23))
        m2 = m1
        if rng.integers(0, 2):
            for _ in range(100):
                c = int(rng.choice(P23 + LENGTHS))
                if lib.fdes_grid_backend(m1, c, 0) == 2 and max(m1, c) <= 3 * min(m1, c):
                    m2 = c
                    break
    elif seed >= 2000:
        m1 = int(rng.choice(ODD))
        m2 = m1 if rng.integers(0, 2) else int(rng.choice([n for n in ODD if max(n, m1) <= 3 * min(n, m1)]))
    elif big_case:
        m1 = int(rng.choice(BIG))
        for _ in range(200):
            c = int(rng.choice([256, 320, 400, 500, 512, 572, 640, 750, 800, 1000, 1024]))
            if lib.fdes_grid_backend(m1, c, 0) == 2:
                m2 = c
                break
        if rng.integers(0, 2):
            m1, m2 = m2, m1
    elif rng.integers(0, 3) == 0:   # a rectangular grid of two lengths that the fused loop takes together
        for _ in range(50):
            c = int(rng.choice(LENGTHS + [256, 512, 1024, 2048]))
            if lib.fdes_grid_backend(m1, c, 0) == 2 and max(m1, c) <= 4 * min(m1, c):
                m2 = c
                break
    big = max(m1, m2)
    kw = dict(m=m1, m2=m2, m3=int(rng.integers(1, 7)), nz=int(rng.integers(1, 4)), frPh=int(rng.choice([0, 0, 2, 3, 5])),
              mode=int(rng.choice([0, 0, 1, 2])), n3=int(rng.integers(1, 5)), seed=int(rng.integers(0, 1000)),
              tilt=bool(rng.integers(0, 2)), beam_tilt=bool(rng.integers(0, 2)), imPot=float(rng.choice([0.0, 0.05, 0.2])),
              nat=int(rng.integers(1, 200)), sub=int(rng.integers(1, 3)), zfrac=float(rng.choice([0.5, 0.3, 0.15])))
    if big >= 1500:
        kw["n3"] = min(kw["n3"], 2); kw["frPh"] = min(kw["frPh"], 2); kw["m3"] = min(kw["m3"], 4)
    if big >= 2600:
        kw["n3"] = 1; kw["sub"] = 1
    opts = dict(gang=int(rng.choice([-1, -1, 0, 2, 4, 8])), lanes=int(rng.choice([0, 0, 1, 2, 3])), skip_empty=int(rng.integers(0, 2)), jit=1 if (big_case or seed >= 3000) else int(rng.integers(0, 2)))
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, **opts)
    pl = eng.plan(hp, at)
    backend, axes = pl.fft_backend(), pl.jit_kernels()
    pl.close()
    out = eng.build_measurements(hp, at)["image"]
    eng.close()
    ref = oracle_py.build_measurements(hp, at, prec="f64" if kw["frPh"] == 0 else "f32")["image"]
    e = float(np.sqrt(((out - ref) ** 2).sum() / (ref ** 2).sum()))
    worst = max(worst, e)
    flag = "" if (e <= 2e-5 and backend == 2) else "   <-- FAIL"
    print(f"seed {seed}: {m1} x {m2} (backend {backend}, compiled axes {axes}) n3 {kw['n3']} frPh {kw['frPh']} mode {kw['mode']} nz {kw['nz']} {opts}: E = {e:.2e}{flag}", flush=True)
    if flag:
        print(kw)
        sys.exit(1)
print("worst relative error vs oracle:", worst)
