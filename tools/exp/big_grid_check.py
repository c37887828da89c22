"""Round 5: one-off checks of SQUARE grids beyond 4096 points (kernels compiled at plan creation): 2-D FFT at 5000^2 and 8192^2 against
numpy, a whole-driver image at 5000^2 against the float64 oracle.  (test infrastructure: uses the oracle as the checker)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, fdes_amd
from tests import specimens as S, oracle_py
from tests.test_gpu_parity import relerr
oracle_py.lib()
for m in (5000, 8192):
    rng = np.random.default_rng(1)
    f = (rng.standard_normal((m, m), dtype=np.float32) + 1j * rng.standard_normal((m, m), dtype=np.float32)).astype(np.complex64)
    eng = fdes_amd.Engine(0, jit=1)
    o, used = eng.fft2(f, False, backend=0)
    ref = np.fft.fft2(f.astype(np.complex128))
    print(m, "fft backend", used, "rel L2", relerr(o, ref), flush=True)
    del ref, o, f
    eng.close()
hp, at = S.case_tiny(m=5000, m3=2, nz=1, nat=100, tilt=True, seed=3)
fdes_amd.consistent(hp)
t0 = time.time()
ref = oracle_py.build_measurements(hp, at, prec="f64")["image"]
print("oracle s", round(time.time() - t0, 1), flush=True)
eng = fdes_amd.Engine(0, jit=1)
pl = eng.plan(hp, at); print("backend", pl.fft_backend(), "axes", pl.jit_kernels()); pl.close()
img = eng.build_measurements(hp, at)["image"]
eng.close()
print("5000^2 driver image vs f64 oracle:", relerr(img, ref))
