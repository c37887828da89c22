#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

 * tiny_cases.npz   float64 / float32 oracle images of specimens.GOLDEN_CASES (our own CPU
                    restatement: the reference ships no images and cannot run here).
 * au309_atoms.npy  the 309 atom records [Z,x,y,z,DWF,occ] of the shipped
                    ExampleSpecimens/Au_cubeoctahedron_cnf/dataFDES_Auparticle.cnf (data fixture), parsed
                    here with plain python.
Run in the authoring container (needs /root/reference for the second file)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_py as O, specimens as S  # noqa: E402

out = {}
for name, kw in S.GOLDEN_CASES.items():
    hp, at = S.case_tiny(**kw)
    O.consistent(hp)
    out[name + "_f64"] = O.build_measurements(hp, at, prec="f64")["image"]
    out[name + "_f32"] = O.build_measurements(hp, at, prec="f32")["image"].astype(np.float32)
np.savez_compressed(os.path.join(ROOT, "tests/golden/tiny_cases.npz"), **out)

ref = "/root/reference/ExampleSpecimens/Au_cubeoctahedron_cnf/dataFDES_Auparticle.cnf"
if os.path.exists(ref):
    rows = []
    for line in open(ref, errors="replace"):
        t = line.split()
        if t and t[0] == "atom:":
            rows.append([float(x) for x in t[1:7]])
    np.save(os.path.join(ROOT, "tests/golden/au309_atoms.npy"), np.array(rows, np.float32))
print("golden written:", sorted(out))
