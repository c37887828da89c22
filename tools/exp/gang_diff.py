#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fdes_amd
from tests import specimens as S
for kw in (dict(m=256, m3=4, nz=1, nat=100, n3=4, tilt=True), dict(m=256, m3=4, nz=1, nat=100, frPh=4), dict(m=320, m3=5, nz=2, nat=150, n3=5, tilt=True, pD=50.0)):
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    out = {}
    for g in (0, 4):
        eng = fdes_amd.Engine(0, skip_empty=0, gang=g, lanes=1)
        out[g] = eng.build_measurements(hp, at)["image"]
        eng.close()
    d = np.abs(out[0].astype(np.float64) - out[4])
    print(kw, "max abs diff", d.max(), "rel", d.max() / np.abs(out[0]).max(), "differing px", int((d > 0).sum()), "of", d.size, "per image", [(int((np.abs(out[0][k].astype(np.float64) - out[4][k]) > 0).sum())) for k in range(out[0].shape[0])])
