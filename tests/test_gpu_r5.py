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
