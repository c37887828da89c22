#!/usr/bin/env python3
"""A tilt series with ONE configuration per measurement through the boundary call (fdes_build_measurements, host buffers in
and out): the reference's own example (bin/dataFDES.cnf: Au-309, 320^2 wave, 25 tilts, 12 slices -> 132 sub-slices) and
SrTiO3 series at 512^2 and 1024^2; gangs across measurements off / automatic.  slice-propagations/s incl. plan creation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fdes_amd
from tests import specimens as S
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = []
hp, at = fdes_amd.read_cnf(os.path.join(ROOT, "tests", "golden", "dataFDES_bin.cnf"))
cases.append(("bin/dataFDES.cnf Au-309 320^2 x 25 tilts", hp, at))
for n in (256, 512):
    hp, at = S.case_c4(n3=32, frPh=0, n=n, dn=n // 2)
    fdes_amd.consistent(hp)
    cases.append((f"SrTiO3 {2 * n}^2 x 32 beam tilts", hp, at))
for name, hp, at in cases:
    q, ratio = fdes_amd.sub_sliced(hp)
    row = f"{name:44s} {q.c.m3:4d} slices:"
    ref = None
    for label, opts in (("gang off", dict(gang=0)), ("auto", dict()), ("8 x 2 lanes", dict(gang=8, lanes=2)), ("16 x 1", dict(gang=16, lanes=1))):
        eng = fdes_amd.Engine(0, **opts)
        eng.build_measurements(hp, at)
        t0 = time.perf_counter()
        img = eng.build_measurements(hp, at)["image"]
        dt = time.perf_counter() - t0
        eng.close()
        if ref is None:
            ref = img
        same = bool(np.array_equal(img, ref))
        row += f"  {label} {hp.c.n3 * q.c.m3 / dt / 1e3:7.1f} k ({dt * 1e3:6.1f} ms{'' if same else ', DIFFERS'})"
    print(row, flush=True)
