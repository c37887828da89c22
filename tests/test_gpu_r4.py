"""GPU tests added in round 4: the RCCL reduction entry points (single-rank communicator: one GPU per box here), the float
view of fdes_plan_accumulate_from, the 4096-point band-limit / propagator passes with two workgroups per CU, the device-side
empty-slice decision and the mixed-radix passes beyond 2048 points."""
import numpy as np
import pytest

import fdes_amd
from tests import specimens as S
from tests.test_gpu_parity import check, relerr

pytestmark = pytest.mark.gpu


def test_rccl_reduce_with_one_rank():
    """fdes_comm_* / fdes_plan_reduce_intensity (ncclReduce(sum, float[m1 m2]) to the owner): librccl.so is
    loaded at run time, a communicator of ONE rank is created on this box's GPU (RCCL refuses two ranks on one device, so
    more cannot be rehearsed here), and the reduction of a measurement's running sum - lanes folded, queued gangs issued,
    the real view packed, reduced and written back - must leave the image exactly as it is without the call."""
    hp, at = S.case_tiny(m=256, m3=4, nz=2, frPh=4, nat=100, tilt=True)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    ref = eng.build_measurements(hp, at)["image"]
    eng.close()
    for opts in (dict(lanes=2, gang=0), dict(lanes=1, gang=2), dict()):
        eng = fdes_amd.Engine(0, **opts)
        uid = fdes_amd.Engine.comm_unique_id()
        assert len(uid) == 128
        comm = eng.comm_create(1, 0, uid)
        pl = eng.plan(hp, at)
        pl.begin_measurement(0)
        for j in range(4):
            pl.run_config(0, j, 0.25)
        pl.reduce_intensity(comm, 0)
        pl.end_measurement(0)
        img = pl.get_images()
        e = relerr(img, ref)
        print(f"[parity] one-rank ncclReduce {opts}: image vs plain run {e:.3e}")
        assert e < 2e-6   # (the association order of the lane sums only)
        pl.close()
        eng.comm_destroy(comm)
        eng.close()


def test_accumulate_from_moves_the_real_view():
    """fdes_plan_accumulate_from between plans that do not share a device path (forced through the staged branch with
    peer_copy = 0): the intensity crosses as float[m1 m2] and is added to I.x; bit-identical to the complex path of two
    plans on one device (adding zero imaginary parts changes nothing)."""
    hp, at = S.case_tiny(m=512, m3=3, nz=1, frPh=4, nat=150)
    fdes_amd.consistent(hp)
    imgs = {}
    for peer in (1, 0):
        ea, eb = fdes_amd.Engine(0, peer_copy=peer, lanes=1, gang=0), fdes_amd.Engine(0, lanes=1, gang=0)
        pa, pb = ea.plan(hp, at), eb.plan(hp, at)
        pa.begin_measurement(0)
        pb.begin_measurement(0)
        for j in (0, 1):
            pa.run_config(0, j, 0.25)
        for j in (2, 3):
            pb.run_config(0, j, 0.25)
        pb.sync()
        pa.accumulate_from(pb)
        pa.end_measurement(0)
        imgs[peer] = pa.get_images()
        for q in (pa, pb):
            q.close()
        ea.close()
        eb.close()
    assert np.array_equal(imgs[0], imgs[1])
