"""GPU tests added in round 5."""
import numpy as np
import pytest

import fdes_amd
from tests import specimens as S
from tests.test_gpu_parity import check, relerr

pytestmark = pytest.mark.gpu


def test_slice_loop_on_a_450_x_4096_grid(oracle):
    """4096-point rows with a row count that two divides but four does not (m1 = 450, 750, 1250 ... x m2 = 4096): the plan's
    geometry check picks two-row workgroups for the y passes; the one-wave-per-row kernels for P4 / P6 (four rows per
    workgroup) must not be preferred there (round 4 returned hipErrorInvalidValue in the first slice)."""
    hp, at = S.case_tiny(m=450, m2=4096, m3=3, nz=2, nat=200, tilt=True, seed=17)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    out = eng.build_measurements(hp, at)["image"]
    eng.close()
    check(out, oracle.build_measurements(hp, at, prec="f64")["image"], None, 1e-5, f"slice loop on a {hp.c.m1} x {hp.c.m2} grid")


_CAPTURE_STRESS = r'''
import sys, threading, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import fdes_amd
from tests import specimens as S

# thread A: configurations with frozen phonons and empty slices at both ends (zfrac 0.2): every new empty-slice pattern
# is a new stream capture of the slice loop; thread B: creates and destroys plans all the while (hipMalloc / hipFree /
# table kernels).  With captures in thread-local mode B's calls must not invalidate A's captures.
hp, at = S.case_tiny(m=256, m3=14, nz=2, nat=50, frPh=6, n3=2, tilt=True, zfrac=0.2, sub=2)
fdes_amd.consistent(hp)
quiet = fdes_amd.Engine(0, skip_empty=1, lanes=2, gang=0)
ref = quiet.build_measurements(hp, at)["image"]
quiet.close()
stop = threading.Event()
made = [0]
err = []
def churn():
    try:
        eng = fdes_amd.Engine(0)
        h2, a2 = S.case_tiny(m=512, m3=3, nz=1, nat=20)
        fdes_amd.consistent(h2)
        while not stop.is_set():
            pl = eng.plan(h2, a2)
            pl.close()
            made[0] += 1
        eng.close()
    except Exception as e:   # noqa
        err.append(repr(e))
t = threading.Thread(target=churn)
t.start()
try:
    for rep in range(6):
        eng = fdes_amd.Engine(0, skip_empty=1, lanes=2, gang=0)
        out = eng.build_measurements(hp, at)["image"]
        eng.close()
        assert np.array_equal(out, ref), f"repetition {rep}: images differ from the quiet run"
finally:
    stop.set()
    t.join()
assert not err, err
assert made[0] >= 3, made
print("ok", made[0])
'''


@pytest.mark.parametrize("lock", ["1", "0"])
def test_slice_loop_capture_beside_plan_creation_in_another_thread(tmp_path, lock):
    """Round 5: the slice loop is captured in hipStreamCaptureModeThreadLocal and plan creation no longer takes a process-wide
    lock (engine.hip, DeviceLocks): one host thread replays / captures slice loops (a new capture per empty-slice pattern) while
    another creates and destroys plans on the same device.  lock = 0 (FDES_CAPTURE_LOCK=0) also drops the per-device lock around
    the capture: the capture mode alone must keep the other thread's hipMalloc / hipFree out."""
    import os
    import subprocess
    import sys
    script = tmp_path / "stress.py"
    script.write_text(_CAPTURE_STRESS)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDES_CAPTURE_LOCK=lock)
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-300:], r.stderr[-1500:])
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]
