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
    r = eng.build_measurements(hp, at, want_exitwave=True)
    ref, ref_ew = r["image"], r["exitwave"][0, ..., 0] + 1j * r["exitwave"][0, ..., 1]
    eng.close()
    for opts in (dict(lanes=2, gang=0), dict(lanes=1, gang=2), dict()):
        eng = fdes_amd.Engine(0, **opts)
        uid = fdes_amd.Engine.comm_unique_id()
        assert len(uid) == 128
        comm = eng.comm_create(1, 0, uid)
        pl = eng.plan(hp, at)
        pl.want_exitwave(True)
        pl.begin_measurement(0)
        for j in range(4):
            pl.run_config(0, j, 0.25)
        pl.reduce_intensity(comm, 0)   # (+ the coherent exit-wave sum: a second reduce)
        with pytest.raises(fdes_amd.FdesError):
            pl.reduce_intensity_span(comm, 0, 0, 1)   # (round 5) a span beyond the communicator is refused, not sent
        ew = pl.get_exitwave()
        pl.end_measurement(0)
        img = pl.get_images()
        e = relerr(img, ref)
        e2 = relerr(ew, ref_ew)
        print(f"[parity] one-rank ncclReduce {opts}: image vs plain run {e:.3e}, exit-wave sum {e2:.3e}")
        assert e < 2e-6 and e2 < 2e-6   # (the association order of the lane sums only)
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


@pytest.mark.parametrize("kw,opts", [(dict(m=256, m3=12, nz=2, nat=60, frPh=3, n3=3, tilt=True, zfrac=0.2, sub=2), dict(lanes=2, gang=0)),
                                     (dict(m=256, m3=12, nz=2, nat=60, frPh=4, n3=2, tilt=True, zfrac=0.2), dict(lanes=1, gang=4)),
                                     (dict(m=512, m3=9, nz=3, nat=40, n3=5, tilt=True, zfrac=0.15), dict()),
                                     (dict(m=1024, m3=10, nz=1, nat=25, frPh=2, tilt=True, zfrac=0.1), dict(lanes=2, gang=0))])
def test_empty_slice_question_on_its_own_stream(oracle, kw, opts):
    """Option skip_empty: which slices of a configuration hold atoms is now answered on a query stream from the constant
    coordinates (tilt of k, jitter of (k, j) and the binning's slice test recomputed: geom_slice_occupancy), not from the
    binning's segment table behind the lane's queued slice loops.  Few atoms, tilts and jitter put atoms next to slice
    boundaries: a slice wrongly taken for empty would lose its atoms.  Images with the short cut against the float32
    oracle (same Philox streams) and against the full sequence on every slice; the question was really asked."""
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    ref = oracle.build_measurements(hp, at, prec="f32")["image"]
    out = {}
    for skip in (1, 0):
        eng = fdes_amd.Engine(0, skip_empty=skip, **opts)
        out[skip] = eng.build_measurements(hp, at)["image"]
        eng.close()
    check(out[1], ref, None, 2e-5, f"empty-slice question {kw} {opts}")
    assert relerr(out[1], out[0].astype(np.float64)) < 2e-5
    eng = fdes_amd.Engine(0, skip_empty=1, **opts)
    pl = eng.plan(hp, at)
    pl.begin_measurement(0)
    pl.run_config(0, 0, 1.0)
    pl.sync()
    assert pl.empty_queries() >= 1
    pl.close()
    eng.close()


def test_qsc_sized_grids_beyond_2048_run_the_fused_loop(oracle):
    """m = 2 nx for .qsc inputs (src/rwQsc.cu:943-948): nx = 1280 ... 2000 gives 2560-, 3000-, 3072-, 3200-, 3600-, 4000-point
    grids, all 2^a 3^b 5^c; cuFFT serves them like any size (src/paramStructure.cu:676-679).  They now run the fused LDS
    passes (mixed-radix rows, two-row tiles) instead of rocFFT + point-wise kernels: backend query, and a rectangular
    2560 x 1280 image (two-row and four-row tiles in one plan) with frozen phonons through the whole driver against the float32 oracle."""
    lib = fdes_amd.load_library()
    for m in (2560, 3000, 3072, 3200, 3600, 4000):
        assert lib.fdes_grid_backend(m, m, 0) == 2, m
    assert lib.fdes_grid_backend(3000, 3000, 1) == 1 and lib.fdes_grid_backend(2006, 2006, 0) == 1   # 2006 = 2 * 17 * 59: a prime factor above 13 leaves the hand-written loop
    assert lib.fdes_grid_backend(2288, 2288, 0) == 2 and lib.fdes_grid_backend(1100, 1100, 0) == 2     # round 5: 2288 = 16 * 11 * 13, 1100 = 2 nx of a .qsc with nx = 550
    hp, at = S.case_tiny(m=2560, m3=3, nz=2, nat=200, frPh=2, tilt=True, rect=True)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 2
    pl.close()
    out = eng.build_measurements(hp, at)["image"]
    eng.close()
    check(out, oracle.build_measurements(hp, at, prec="f32")["image"], None, 2e-5, f"fused loop on a {hp.c.m1} x {hp.c.m2} grid")


def test_stage_goldens_and_au309_measurement_12_on_the_gpu():
    """The committed per-stage goldens (tests/golden/stage_cases.npz: potential of one sub-slice, Fresnel propagator with its
    band-limit mask) and the exit-wave intensity of measurement k = 12 of the shipped Au-309 example (au309_k12.npz; 320 x 320
    wave on the mixed-radix passes, 132 sub-slices, specimen tilt 12 of 25) against the engine's taps - fixtures instead of a
    live oracle run (SURVEY 8c's list)."""
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(G, "stage_cases.npz"))
    hp, at = S.case_tiny(**S.GOLDEN_CASES["img_2sp"])
    fdes_amd.consistent(hp)
    for fft in (1, 0):   # 64 x 64 runs the rocFFT loop either way; the taps are the same code for every path
        eng = fdes_amd.Engine(0, fft=fft, skip_empty=0)
        pl = eng.plan(hp, at)
        check(pl.tap_potential(0, 0, 2), g["potential_s2_f64"], g["potential_s2_f32"], 1e-5, "golden potential of sub-slice 2")
        check(pl.tap_propagator(), g["propagator_f64"], g["propagator_f32"], 1e-5, "golden Fresnel propagator")
        assert np.array_equal(np.abs(pl.tap_propagator()) > 0, g["band_mask"].astype(bool))
        pl.close()
        eng.close()
    a = np.load(os.path.join(G, "au309_k12.npz"))
    hp, at = fdes_amd.read_cnf(os.path.join(G, "dataFDES_Auparticle.cnf"), bug_compatible=False)
    hp.set(pD=0.0)
    fdes_amd.consistent(hp)
    for skip in (0, 1):
        eng = fdes_amd.Engine(0, skip_empty=skip)
        pl = eng.plan(hp, at)
        assert pl.fft_backend() == 2
        psi = pl.tap_wave(12, 0)
        check(np.abs(psi) ** 2, a["exit_intensity_f64"], a["exit_intensity_f32"], 1e-4, f"Au-309 k = 12 exit-wave intensity, skip_empty = {skip}")
        pl.close()
        eng.close()
    eng = fdes_amd.Engine(0)
    img = eng.build_measurements(hp, at)["image"]
    eng.close()
    check(img[12], a["image_f64"], None, 1e-4, "Au-309 image of measurement 12 within the whole series")
