import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import fdes_amd
from tests import specimens as S, oracle_py as O
hp, at = S.case_c3(frPh=0)
fdes_amd.consistent(hp)
eng = fdes_amd.Engine(0, skip_empty=0)
t0 = time.perf_counter(); out = eng.build_measurements(hp, at)["image"]; print("gpu s", time.perf_counter() - t0, flush=True)
eng2 = fdes_amd.Engine(0, skip_empty=1)
out2 = eng2.build_measurements(hp, at)["image"]
t0 = time.perf_counter(); r32 = O.build_measurements(hp, at, prec="f32")["image"]; print("f32 oracle s", time.perf_counter() - t0, flush=True)
t0 = time.perf_counter(); r64 = O.build_measurements(hp, at, prec="f64")["image"]; print("f64 oracle s", time.perf_counter() - t0, flush=True)
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
print("E(gpu) vs f64", rel(out, r64), "E(gpu, skip_empty) vs f64", rel(out2, r64), "E(cpu_f32) vs f64", rel(r32, r64), "contrast", float(r64.std() / r64.mean()))
