#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

 * tiny_cases.npz   float64 / float32 oracle images of specimens.GOLDEN_CASES (our own CPU
                    restatement: the reference ships no images and cannot run here).
 * stage_cases.npz  SURVEY 8c's per-stage goldens at 64 x 64 from the same oracle (case img_2sp): the projected potential of
                    sub-slice 2 (deposit (x) f_e / sinc, float64 + float32), the Fresnel propagator with its band-limit
                    mask, the incoming wave, and ONE full slice step psi' = F^-1[P F[BL(e^{iV}) psi]] from the plane wave.
 * au309_k12.npz    exit-wave intensity |psi|^2 (float64 truth and float32 oracle) of measurement k = 12 of the shipped
                    Au-309 example (tests/golden/dataFDES_Auparticle.cnf: 320 x 320 wave, 132 sub-slices, tilt 12 of 25),
                    and the image of that measurement without dose noise.
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

# per-stage goldens (SURVEY 8c "per-kernel goldens at 64 x 64")
hp, at = S.case_tiny(**S.GOLDEN_CASES["img_2sp"])
O.consistent(hp)
q, _ = O.sub_sliced(hp)
xyz = O.config_coords(q, at, 0, -1)
st = {}
for prec in ("f64", "f32"):
    V = O.phase_grating(q, at, xyz, 2, prec)
    P = O.fresnel_propagator(q, prec)
    psi0 = O.incoming_wave(q, 0, prec)
    st["potential_s2_" + prec] = V
    st["propagator_" + prec] = P
    st["incoming_" + prec] = psi0
    st["one_step_" + prec] = O.forward_propagation(q, psi0, V, prec)
st["band_mask"] = (np.abs(st["propagator_f64"]) > 0).astype(np.uint8)
np.savez_compressed(os.path.join(ROOT, "tests/golden/stage_cases.npz"), **st)

# Au-309, measurement k = 12: needs the product's .cnf reader only for parsing the shipped input
import fdes_amd  # noqa: E402
hp, at = fdes_amd.read_cnf(os.path.join(ROOT, "tests/golden/dataFDES_Auparticle.cnf"), bug_compatible=False)
hp.set(pD=0.0)
O.consistent(hp)
q, _ = O.sub_sliced(hp)
au = {}
for prec in ("f64", "f32"):
    psi = O.wave(q, at, 12, 0, prec=prec)
    au["exit_intensity_" + prec] = (np.abs(psi) ** 2).astype(np.float64 if prec == "f64" else np.float32)
au["image_f64"] = O.measurement(hp, at, 12, prec="f64")
np.savez_compressed(os.path.join(ROOT, "tests/golden/au309_k12.npz"), **au)

ref = "/root/reference/ExampleSpecimens/Au_cubeoctahedron_cnf/dataFDES_Auparticle.cnf"
if os.path.exists(ref):
    rows = []
    for line in open(ref, errors="replace"):
        t = line.split()
        if t and t[0] == "atom:":
            rows.append([float(x) for x in t[1:7]])
    np.save(os.path.join(ROOT, "tests/golden/au309_atoms.npy"), np.array(rows, np.float32))
print("golden written:", sorted(out), sorted(st), sorted(au))
