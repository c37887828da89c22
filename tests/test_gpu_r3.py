"""GPU tests added in round 3: the incoming wave of a tilted CBED probe that leaves the 2/3 band (the first product of a
configuration must see all of it, src/multisliceSimulation.cu:546, 583-590) and the exact launch path bench.py times
(two lanes, hipGraph replay, frozen-phonon jitter) against the oracle at the headline size."""
import os

import numpy as np
import pytest

import fdes_amd
from tests import specimens as S
from tests.test_gpu_parity import check, relerr

from tests.conftest import full_only

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", [256, 1024])
@pytest.mark.parametrize("skip_empty", [0, 1])
@pytest.mark.parametrize("tilt", [(0.0, 0.062), (0.055, 0.055)])
def test_tilted_cbed_probe_outside_the_band(oracle, m, skip_empty, tilt):
    """mode 2 with a beam tilt: incomingWave applies the phase ramp AFTER the band limit and does not limit again
    (src/multisliceSimulation.cu:583-590), so the probe disc (ObjAp) shifted by the tilt leaves |k| <= 1/(3 d) once
    (ObjAp + tilt) d / lambda > 1/3.  Here (ObjAp + tilt) d / lambda = 0.44 / 0.41 (> 0.4): the first transmission
    product must use the whole incoming wave, as the reference does (:546); exit wave against the float64 oracle."""
    hp, at = S.case_tiny(m=m, m3=5, nz=2, mode=2, nat=200, seed=31, zfrac=0.5 if not skip_empty else 0.3)
    hp.set(tiltbeam=np.array(tilt, np.float32))
    fdes_amd.consistent(hp)
    assert hp.c.doBeamTilt
    reach = (hp.c.ObjAp + max(abs(t) for t in tilt)) * hp.c.d1 / hp.c.lambda_
    assert reach > 0.4 and reach < 0.5
    q, _ = oracle.sub_sliced(hp)
    eng = fdes_amd.Engine(0, skip_empty=skip_empty)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 2
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    for ns in (1, None):  # after the first slice and at the exit
        psi = pl.tap_wave(0, 0) if ns is None else pl.tap_wave(0, 0, ns)
        r = ref if ns is None else oracle.wave(q, at, 0, 0, nslices=ns, prec="f64")
        check(psi, r, None, 1e-5, f"tilted CBED probe {m}^2 tilt={tilt} skip_empty={skip_empty} slices={ns}")
    # the part of the incoming wave outside the band is not negligible (otherwise this test would not see the bug)
    psi0 = oracle.incoming_wave(q, 0, prec="f64")
    f = np.fft.fft2(psi0)
    kx = np.fft.fftfreq(f.shape[1]) * f.shape[1]
    out = (9 * kx * kx > f.shape[1] ** 2)
    frac = float((np.abs(f[:, out]) ** 2).sum() / (np.abs(f) ** 2).sum())
    print(f"[parity] tilted CBED probe {m}^2 tilt={tilt}: fraction of the incoming intensity at dead kx = {frac:.3f}")
    if abs(tilt[1]) > 0.06 or abs(tilt[0]) > 0.06:
        assert frac > 0.05
    pl.close()
    eng.close()
    # and through the whole driver (lanes / graph as the engine chooses): diffraction pattern image
    eng = fdes_amd.Engine(0, skip_empty=skip_empty)
    img = eng.build_measurements(hp, at)["image"]
    eng.close()
    r64 = oracle.build_measurements(hp, at, prec="f64")["image"]
    assert relerr(img, r64) <= 1e-5


def test_bench_launch_path_against_oracle_at_headline_size(oracle):
    """What bench.py times: C3 at 2048^2 x 256 slices with frozen phonons on, i.e. two lanes, hipGraph replay of the
    slice loop, jitter, band bookkeeping, every slice the full sequence (skip_empty = 0).  Two configurations, image
    against the float32 oracle with identical Philox streams (RNG configurations compare with the CPU
    backend), E <= 5e-5."""
    hp, at = S.case_c3(frPh=2)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, skip_empty=0)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 2 and pl.lanes() == 2
    pl.close()
    img = eng.build_measurements(hp, at)["image"]
    eng.close()
    r32 = oracle.build_measurements(hp, at, prec="f32")["image"]
    e = relerr(img, r32)
    print(f"[parity] bench path (C3 2048^2 x 256, frPh=2, two lanes + graph): E(gpu vs cpu_f32) = {e:.3e}")
    assert e <= 5e-5
    assert np.abs(img - r32).max() <= 1e-3 * np.abs(r32).max()


# ------------------------------------------------------------------------------------------------------------------
# one-wave-per-row passes (fft_wave.hip): engine option pass_threads = 64, 2048- and 4096-point rows
# ------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("threads", [64, 65])
@pytest.mark.parametrize("shape", [(2048, 2048), (4096, 4096), (2048, 4096), (1024, 2048), (1024, 1024), (512, 1024)])
def test_wave_fft_against_numpy(shape, threads):
    """The row FFT with one wave per row (radix-32 / radix-64 butterflies over the registers, one LDS exchange, the
    radix-2 across lane halves by v_permlane32_swap at 2048 points, the radix-4 across lane quarters by v_permlane32_swap +
    v_permlane16_swap at 1024 points) as a 2-D transform against numpy; 512-point rows fall back to the two-rows-per-thread
    kernels."""
    eng = fdes_amd.Engine(0, pass_threads=threads)
    rng = np.random.default_rng(11)
    f = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    for inv in (False, True):
        out, used = eng.fft2(f, inv, backend=2)
        assert used == 2
        ref = np.fft.ifft2(f.astype(np.complex128)) * f.size if inv else np.fft.fft2(f.astype(np.complex128))
        e = relerr(out, ref)
        print(f"[parity] wave fft {shape} inv={inv}: rel L2 {e:.3e}")
        assert e < 5e-7
    eng.close()


@pytest.mark.parametrize("threads", [64, pytest.param(65, marks=full_only), 128])   # (65: the software-pipelined variant, not a default anywhere)
@pytest.mark.parametrize("m,nz,stagger", [(2048, 1, 0), (2048, 2, 16), (4096, 1, 0), (4096, 2, 8), (1024, 1, 0), (1024, 3, 0)])
def test_wave_passes_slice_loop(oracle, m, nz, stagger, threads):
    """Every pass of the slice loop on the one-wave-per-row kernels (P1' atoms, P2 with one and two species, the two-slice
    P3, P4, P5, P6, enter / leave), with and without the staggered start: exit wave after an odd number of slices and
    the potential of both members of a pair against the float64 oracle.  phaseGrating src/crystalMaker.cu:507-536,
    forwardPropagation src/multisliceSimulation.cu:538-549."""
    if threads == 128 and m != 2048:
        pytest.skip("eight-row workgroups exist for 2048-point rows only")
    hp, at = S.case_tiny(m=m, m3=5 if m <= 2048 else 3, nz=nz, nat=300, tilt=True, seed=41 + nz)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    eng = fdes_amd.Engine(0, pass_threads=threads, skip_empty=0, stagger=stagger)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 2
    psi = pl.tap_wave(0, 0)
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    check(psi, ref, oracle.wave(q, at, 0, 0, prec="f32"), 1e-5, f"wave passes: exit wave {m}^2 nz={nz} stagger={stagger}")
    xyz = oracle.config_coords(q, at, 0, -1)
    for s in (1, 2):
        V = pl.tap_potential(0, 0, s)
        check(V, oracle.phase_grating(q, at, xyz, s, "f64"), None, 1e-5, f"wave passes: potential s={s} {m}^2 nz={nz}")
    pl.close()
    eng.close()


@pytest.mark.parametrize("threads", [64, pytest.param(65, marks=full_only), pytest.param(128, marks=full_only)])
def test_wave_passes_equal_lanes_graph_and_empty_slices(oracle, threads):
    """One-wave-per-row passes through the whole driver at 2048^2: two lanes, graph replay, frozen phonons, runs of
    empty slices (P^n steps) - image against the float32 oracle."""
    hp, at = S.case_tiny(m=2048, m3=8, nz=2, nat=400, frPh=3, tilt=True, seed=5, zfrac=0.25)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, pass_threads=threads, stagger=12 if threads == 64 else 0)
    img = eng.build_measurements(hp, at)["image"]
    eng.close()
    r64 = oracle.build_measurements(hp, at, prec="f64")["image"]
    e = relerr(img, r64)
    print(f"[parity] wave passes, driver 2048^2 frPh=3 with empty slices: E = {e:.3e}")
    assert e <= 1e-5


# ------------------------------------------------------------------------------------------------------------------
# mixed-radix LDS-resident passes (fft_gen.hip): the grid sizes of the reference's own examples (320, 800, 1000)
# ------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape", [(320, 320), (800, 800), (1000, 1000), (1000, 512), (640, 960), (1280, 1500), (2000, 1920),
                                   (3000, 3000), (2560, 3072), (3600, 4000), (3200, 4096), (2048, 2560), (2304, 2700), (1600, 1280),
                                   (448, 896), (1400, 1792), (2016, 3584), (1372, 1024),   # 7-smooth lengths (radix 7); 1372 = 4 * 7^3   # (round 4: lengths beyond 2048; 2304 x 2700: run-time-length kernels there)
                                   (1100, 572), pytest.param((1144, 2600), marks=full_only), pytest.param((3300, 2860), marks=full_only), (2288, 1716)])   # round 5: radix 11 and 13 (1100 = 2 nx of a .qsc with nx = 550; 572 = 4 * 11 * 13, 1716 = 4 * 3 * 11 * 13, 2288 = 16 * 11 * 13; two-row tiles beyond 2048)
def test_mixed_radix_fft_against_numpy(engine, shape):
    """Row FFTs of length 2^a 3^b 5^c 7^d 11^e 13^f (Stockham stages of radix 13, 11, 10, 8, 7, 5, 4, 3, 2 in LDS) as a 2-D transform against
    numpy, also paired with a power-of-two length (cufftPlan2d serves any size, src/paramStructure.cu:676-679)."""
    rng = np.random.default_rng(13)
    f = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    for inv in (False, True):
        out, used = engine.fft2(f, inv, backend=2)
        assert used == 2
        ref = np.fft.ifft2(f.astype(np.complex128)) * f.size if inv else np.fft.fft2(f.astype(np.complex128))
        e = relerr(out, ref)
        print(f"[parity] mixed-radix fft {shape} inv={inv}: rel L2 {e:.3e}")
        assert e < 6e-7


@pytest.mark.parametrize("m,nz", [(320, 1), (320, 2), (800, 1), (1000, 2), (3000, 1), (2560, 2), (896, 2), (1400, 1),
                                  (2880, 1), (572, 2), (1100, 1)])   # (round 4: two-row tiles beyond 2048 points; 7-smooth lengths; round 5: 2880 = a length without compile-time specialisation beyond 2048; 572 = 4 * 11 * 13 and 1100 = 4 * 25 * 11: radix 11 / 13)
def test_mixed_radix_slice_loop(oracle, m, nz):
    """The fused slice loop on 320 / 800 / 1000-point grids (before round 3: rocFFT + point-wise kernels, 24 launches per
    slice): exit wave after an odd number of slices and the potential of both members of a pair against the float64
    oracle, one and two species, with and without the empty-slice short cut."""
    hp, at = S.case_tiny(m=m, m3=5, nz=nz, nat=300, tilt=True, seed=51 + nz)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    r32 = oracle.wave(q, at, 0, 0, prec="f32")
    for skip in (0, 1):
        eng = fdes_amd.Engine(0, skip_empty=skip)
        pl = eng.plan(hp, at)
        assert pl.fft_backend() == 2
        psi = pl.tap_wave(0, 0)
        check(psi, ref, r32, 1e-5, f"mixed-radix slice loop {m}^2 nz={nz} skip_empty={skip}")
        if skip == 0:
            xyz = oracle.config_coords(q, at, 0, -1)
            for s in (2, 3):
                V = pl.tap_potential(0, 0, s)
                check(V, oracle.phase_grating(q, at, xyz, s, "f64"), None, 1e-5, f"mixed-radix potential s={s} {m}^2 nz={nz}")
        pl.close()
        eng.close()


@pytest.mark.parametrize("kw", [dict(m=320, m3=6, nz=2, frPh=3, nat=200, tilt=True), dict(m=800, m3=4, nz=2, mode=2, nat=150),
                                dict(m=1000, m3=4, nz=1, mode=1, beam_tilt=True, nat=150), dict(m=640, rect=True, m3=3, nz=2, nat=100)])
def test_mixed_radix_driver_images(oracle, kw):
    """Whole driver (lanes, graph replay, detector chain) on non-power-of-two grids: images against the float64 oracle
    resp. the float32 oracle for frozen-phonon runs; the same through rocFFT for comparison."""
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    img = eng.build_measurements(hp, at)["image"]
    eng.close()
    eng = fdes_amd.Engine(0, fft=1)
    img_r = eng.build_measurements(hp, at)["image"]
    eng.close()
    ref = oracle.build_measurements(hp, at, prec="f64" if kw.get("frPh", 0) == 0 else "f32")["image"]
    e, er = relerr(img, ref), relerr(img_r, ref)
    print(f"[parity] mixed-radix driver {kw}: E(fused) = {e:.3e}, E(rocFFT path) = {er:.3e}")
    assert e <= 1e-5


def test_multi_gpu_driver_host_staged_fallback():
    """fdes_plan_accumulate_from (the sum of src/crystalMaker.cu:347-365 for a measurement whose configurations ran on two
    plans): the peer-copy path and the host-staged fallback it takes when the runtime refuses a peer copy (forced here by
    option peer_copy = 0, which also takes it between two plans of one device) give the single-plan image."""
    hp, at = S.case_tiny(m=256, m3=4, nz=2, frPh=4, nat=100, tilt=True)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    ref = eng.build_measurements(hp, at, want_exitwave=True)
    eng.close()
    for peer in (1, 0):
        ea, eb = fdes_amd.Engine(0, peer_copy=peer), fdes_amd.Engine(0)
        pa, pb = ea.plan(hp, at), eb.plan(hp, at)
        pa.want_exitwave(True)
        pb.want_exitwave(True)
        pa.begin_measurement(0)
        pb.begin_measurement(0)
        for j in (0, 1):
            pa.run_config(0, j, 0.25)
        for j in (2, 3):
            pb.run_config(0, j, 0.25)
        pb.sync()
        pa.accumulate_from(pb)
        ew = pa.get_exitwave()
        pa.end_measurement(0)
        img = pa.get_images()
        e, e2 = relerr(img, ref["image"]), relerr(ew, ref["exitwave"][0, ..., 0] + 1j * ref["exitwave"][0, ..., 1])
        print(f"[parity] accumulate_from peer_copy={peer}: image {e:.3e}, exit-wave sum {e2:.3e}")
        assert e < 2e-6 and e2 < 2e-6
        for q in (pa, pb):
            q.close()
        ea.close()
        eb.close()


# ------------------------------------------------------------------------------------------------------------------
# batched potential chain of one-lane plans (engine.hip, batched_loop)
# ------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("kw", [dict(m=256, m3=11, nz=2, nat=200, tilt=True), dict(m=512, m3=7, nz=1, nat=300, zfrac=0.2),
                                dict(m=1024, m3=9, nz=2, nat=300, mode=2), dict(m=800, m3=6, nz=2, nat=200, zfrac=0.3),
                                dict(m=1280, m3=5, nz=1, nat=200, zfrac=0.3)])   # (1280: the register-chained mixed-radix kernels of round 5 with grid.z batches)
def test_batched_potential_chain_does_not_change_a_bit(oracle, kw):
    """A single image (one configuration, one lane): the potential / transmission passes of several slice pairs as one
    launch each (grid.z), the wave's passes one batch behind.  Same kernels, same operands: the exit wave is bit-identical
    to the unbatched two-stream loop and to the single-stream loop for every batch size, odd slice counts, empty slices at
    both ends (skip_empty 0 and 1), one and two species, a power-of-two and a mixed-radix grid; and equals the oracle."""
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    for skip in (0, 1):
        outs = {}
        for label, opts in (("single stream", dict(split=0)), ("split", dict(split=1, batch=0)), ("batch 2", dict(batch=2)),
                            ("batch 3", dict(batch=3)), ("batch 8", dict(batch=8)), ("default", dict()),
                            ("one wave per row, batch 4", dict(pass_threads=64, batch=4))):
            eng = fdes_amd.Engine(0, skip_empty=skip, **opts)
            pl = eng.plan(hp, at)
            assert pl.lanes() == 1 and pl.fft_backend() == 2
            outs[label] = pl.tap_wave(0, 0)
            pl.close()
            eng.close()
        check(outs["default"], ref, None, 1e-5, f"batched potential chain {kw} skip_empty={skip}")
        for label, o in outs.items():
            if label.startswith("one wave") and kw["m"] == 1024:   # other kernels: equal within rounding, not bit by bit
                assert relerr(o, outs["single stream"]) < 2e-6, (label, kw, skip)
            else:
                assert np.array_equal(o.view(np.uint64), outs["single stream"].view(np.uint64)), (label, kw, skip)


def test_stale_handles_are_refused_everywhere():
    """Every plan entry point checks its handle against the registry of live objects, and a destroyed object's address is
    not reused at once (graveyard of 1024 shells): a stale handle - a late finaliser, a host bug - gets FDES_EINVAL (-1)
    instead of reaching a newer plan that happens to live at the same address."""
    import ctypes as C
    hp, at = S.case_tiny(m=64, m3=2, nz=1, nat=10)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    lib = eng.lib
    pl = eng.plan(hp, at)
    stale = C.c_void_p(pl.h.value)
    pl.close()
    newer = [eng.plan(hp, at) for _ in range(4)]          # would reuse the address without the graveyard
    assert all(q.h.value != stale.value for q in newer)
    img = np.zeros((1, hp.c.n2, hp.c.n1), np.float32)
    assert lib.fdes_plan_run_config(stale, 0, 0, 1.0) == -1
    assert lib.fdes_plan_begin_measurement(stale, 0) == -1
    assert lib.fdes_plan_get_images(stale, img.ctypes.data_as(C.POINTER(C.c_float))) == -1
    assert lib.fdes_plan_sync(stale) == -1 and lib.fdes_plan_destroy(stale) == -1
    assert lib.fdes_plan_fft_backend(stale) == -1 and lib.fdes_plan_lanes(stale) == -1
    for q in newer:
        q.close()
    ctx_stale = C.c_void_p(eng.h.value)
    eng.close()
    assert lib.fdes_set_option(ctx_stale, b"seed", 3) == -1 and lib.fdes_destroy(ctx_stale) == -1


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomised_parameter_sweep_round3_kernels(oracle, seed):
    """Random draws over the parameter surface (mode incl. CBED and diffraction patterns, species, sub-slicing, odd and even
    slice counts, empty slices, specimen and beam tilts, several measurements, frozen phonons, absorption, rectangular
    grids) on the grids the round-3 kernels serve: 1024 and 2048 points (one wave per row; with lanes + graph or as a
    single image on the batched chain), 320 / 640 / 800 / 1000 points (mixed radix) and their mixtures with power-of-two
    lengths: images against the float64 oracle resp. the float32 oracle with the same Philox streams."""
    rng = np.random.default_rng(3000 + seed)
    m = int(rng.choice([1024, 2048, 320, 640, 800, 1000, 1024, 320]))
    rect = bool(rng.integers(0, 3) == 0) and m in (1024, 2048, 640)
    kw = dict(m=m, m3=int(rng.integers(1, 7)), nz=int(rng.integers(1, 4)), frPh=int(rng.choice([0, 0, 2, 3])),
              mode=int(rng.choice([0, 0, 1, 2])), n3=int(rng.integers(1, 3)), seed=int(rng.integers(0, 1000)),
              tilt=bool(rng.integers(0, 2)), beam_tilt=bool(rng.integers(0, 2)), imPot=float(rng.choice([0.0, 0.05, 0.2])),
              rect=rect, nat=int(rng.integers(1, 200)), sub=int(rng.integers(1, 3)), zfrac=float(rng.choice([0.5, 0.3, 0.15])))
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 2
    pl.close()
    out = eng.build_measurements(hp, at)["image"]
    eng.close()
    ref = oracle.build_measurements(hp, at, prec="f64" if kw["frPh"] == 0 else "f32")["image"]
    check(out, ref, None, 2e-5, f"round-3 sweep {seed}: {kw}")


@pytest.mark.parametrize("kw", [dict(m=256, m3=7, nz=2, nat=200, frPh=5, n3=2, tilt=True, beam_tilt=True),
                                dict(m=1024, m3=5, nz=3, nat=300, frPh=4), dict(m=320, m3=6, nz=2, nat=200, frPh=4, mode=2),
                                dict(m=512, m3=9, nz=1, nat=300, frPh=6, zfrac=0.2, mode=1)])
def test_gang_of_configurations_does_not_change_a_bit(oracle, kw):
    """Option "gang": the frozen-phonon configurations of a measurement run their slice loops in lockstep, every pass ONE
    launch with the configurations as grid z (own atom records, own wave; shared tables).  Same kernels, same operands per
    configuration: images and the summed exit wave are bit-identical to one configuration per launch for every gang size
    (also sizes that do not divide the number of configurations), with one and two lanes, tilt series, one to three
    species, the one-wave-per-row and the mixed-radix kernels; with skip_empty a slice is skipped only when it is empty in
    every member, so there the results agree to rounding.  The ungrouped result equals the oracle."""
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    for skip in (0, 1):
        outs = {}
        for label, opts in (("off", dict(gang=0, lanes=1)), ("2", dict(gang=2, lanes=1)), ("3", dict(gang=3, lanes=1)),
                            ("4 x 2 lanes", dict(gang=4, lanes=2)), ("8", dict(gang=8, lanes=1)), ("off, 2 lanes", dict(gang=0, lanes=2))):
            eng = fdes_amd.Engine(0, skip_empty=skip, **opts)
            r = eng.build_measurements(hp, at, want_exitwave=True)
            outs[label] = (r["image"], r["exitwave"])
            eng.close()
        # the ungrouped result against the float32 oracle (same Philox streams) at every size (round 4: also 512^2 and
        # 1024^2, at 5e-5; the gang results are bit-identical to it below)
        ref = oracle.build_measurements(hp, at, prec="f32")["image"] if skip == 0 else None
        if ref is not None:
            check(outs["off"][0], ref, None, 5e-5, f"gang off {kw}")
        for label, (img, ew) in outs.items():
            if skip == 0 and not label.endswith("lanes"):
                assert np.array_equal(img, outs["off"][0]) and np.array_equal(ew, outs["off"][1]), (label, kw)
            else:   # lanes add their partial sums in another order; skip_empty: see above
                assert relerr(img, outs["off"][0]) < 2e-6 and relerr(ew, outs["off"][1]) < 2e-6, (label, kw, skip)


@pytest.mark.parametrize("kw", [dict(m=256, m3=7, nz=2, nat=200, n3=7, tilt=True, beam_tilt=True), dict(m=320, m3=5, nz=2, nat=150, n3=5, tilt=True, pD=50.0),
                                dict(m=1024, m3=4, nz=1, nat=300, n3=4, tilt=True), dict(m=512, m3=9, nz=3, nat=300, n3=9, tilt=True, zfrac=0.2, mode=2),
                                dict(m=256, m3=5, nz=2, nat=150, n3=6, mode=1, beam_tilt=True)])
def test_gang_across_measurements_does_not_change_a_bit(oracle, kw):
    """A series with ONE configuration per measurement (tilt / defocus series without frozen phonons - the reference's own
    example, bin/dataFDES.cnf: 25 tilts): fdes_build_measurements runs `gang` measurements in lockstep, each member with its
    own tilt, incoming wave and intensity slot, the detector chain (incoherence, dose noise, MTF, crop) per slot behind the
    gang.  Images bit-identical to one measurement at a time for every gang size (also sizes that do not divide the series)
    on one lane, equal to rounding on two (skip_empty: a slice is skipped only when it is empty in every member); the
    ungrouped result equals the oracle."""
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    for skip in (0, 1):
        outs = {}
        for label, opts in (("off", dict(gang=0, lanes=1)), ("2", dict(gang=2, lanes=1)), ("3", dict(gang=3, lanes=1)),
                            ("4 x 2 lanes", dict(gang=4, lanes=2)), ("8", dict(gang=8, lanes=1)), ("auto", dict())):
            eng = fdes_amd.Engine(0, skip_empty=skip, **opts)
            outs[label] = eng.build_measurements(hp, at)["image"]
            eng.close()
        if skip == 0 and not kw.get("pD"):   # (round 4: at every size, 5e-5)
            check(outs["off"], oracle.build_measurements(hp, at, prec="f32")["image"], None, 5e-5, f"series, gang off {kw}")
        for label, img in outs.items():
            if skip == 0:
                assert np.array_equal(img, outs["off"]), (label, kw)
            else:
                assert relerr(img, outs["off"]) < 2e-6, (label, kw, skip)


def test_qsc_cell_with_vacancies_against_the_oracle(oracle, tmp_path):
    """A .qsc whose unit cell has a half-occupied and a shared site: the front-end draws the vacancies with QSTEM's ran1;
    they reach the engine as species 0 (the Kirkland fallback row, as in the reference), with their positions and
    occupancies.  Image vs the float64 oracle on the same atom list."""
    import shutil
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cfg = open(os.path.join(G, "qsc", "SrTiO3.cfg")).read()
    cfg = cfg.replace("0.5 0.5 0.5  0.4390  1.0", "0.5 0.5 0.5  0.4390  0.5").replace("0 0.5 0.5  0.7323 1.0", "0 0.5 0.5  0.7323 0.6")
    cfg = cfg.replace("Number of particles = 5", "Number of particles = 6") + "14\nN\n0 0.5 0.5  0.5 0.3\n"
    (tmp_path / "SrTiO3.cfg").write_text(cfg)
    (tmp_path / "v.qsc").write_text("mode: TEM\nfilename: SrTiO3.cfg\nNCELLX: 3\nNCELLY: 3\nNCELLZ: 4\nv0: 200\nslice-thickness: 1.9525\n"
                                    "slices: 8\nnx: 128\nCs: 0.05\nalpha: 15\ndefocus: 13.7\ncal_mode: 0\nobjective_aperture: 20e-3\n"
                                    "absorptive_potential_factor: 0.1\npixel_dose: 0\n")
    hp, at = fdes_amd.read_qsc(tmp_path / "v.qsc")
    assert at.n == 6 * 36 and 0 < (at.Z == 0).sum() < at.n and set(np.unique(at.Z)) == {0, 7, 8, 22, 38}
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    img = eng.build_measurements(hp, at)["image"]
    eng.close()
    ref = oracle.build_measurements(hp, at, prec="f64")["image"]
    check(img, ref, None, 2e-5, "SrTiO3 .qsc with vacancies (species 0)")


def test_run_measurements_and_sharded_series():
    """fdes_plan_run_measurements: an arbitrary list of measurements of a series (one configuration each) in gangs, images
    equal to the per-k calls to the bit; shard.run_sharded hands a rank's measurements to it in one call, and two ranks'
    image stacks add up to the unsharded result."""
    from fdes_amd import shard
    hp, at = S.case_tiny(m=256, m3=6, nz=2, nat=150, n3=7, tilt=True, beam_tilt=True)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, skip_empty=0)
    full = eng.build_measurements(hp, at)["image"]
    pl = eng.plan(hp, at)
    assert pl.gang() > 1
    pl.run_measurements([5, 2, 6])
    got = pl.get_images()
    assert np.array_equal(got[[5, 2, 6]], full[[5, 2, 6]]) and not got[[0, 1, 3, 4]].any()
    pl.close()
    ref = np.zeros_like(full)
    e0 = fdes_amd.Engine(0, skip_empty=0, gang=0, lanes=1)
    p0 = e0.plan(hp, at)
    for k in range(7):
        p0.begin_measurement(k)
        p0.run_config(k, 0, 1.0)
        p0.end_measurement(k)
    ref = p0.get_images()
    p0.close(); e0.close()
    assert np.array_equal(ref, full)
    stacks = []
    for rank in range(2):
        plr = eng.plan(hp, at)
        done = shard.run_sharded(plr, 7, 1, rank, 2, reduce_fn=None)
        assert done == [k for (k, _) in shard.partition(7, 1, 2, rank)]
        stacks.append(plr.get_images())
        plr.close()
    eng.close()
    assert np.array_equal(stacks[0] + stacks[1], full)
    # the in-process multi-GPU driver hands every GPU's measurements over the same way (two workers on one device here)
    multi = fdes_amd.build_measurements_multi([0, 0], hp, at)
    e1 = fdes_amd.Engine(0)          # engine defaults on both sides (skip_empty on)
    one = e1.build_measurements(hp, at)["image"]
    e1.close()
    assert relerr(multi, one) < 2e-6
