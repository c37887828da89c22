"""GPU tests added in round 2: every workgroup geometry of the fused passes against the oracle (including the
512-thread / 4096-point one of BASELINE config 5), the full C4 series, large projected phases, library teardown,
option propagation to lanes, the progress callback and physics checks that do not come from the oracle's source."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import fdes_amd
from tests import specimens as S
from tests.test_gpu_parity import check, relerr

from tests.conftest import full_only

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------------------------
# every (PRE, MID, POST) x workgroup geometry the slice loop can dispatch (fft_lds.hip: dispatch / dispatch_wg)
# ------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("m,threads", [(1024, 256), (1024, 512), (1024, 1), (2048, 256), (2048, 512), (2048, 1),
                                       (512, 256), (512, 512), (256, 512)])
@pytest.mark.parametrize("nz", [1, 2])
def test_fused_slice_loop_every_geometry(oracle, m, threads, nz):
    """pass_threads forced to 256 / 512 (two rows per thread) and 1 (one row per thread, four rows per workgroup): exit
    wave after an odd number of slices (the last pair is half empty), potential of both members of a pair, with one
    species (MID_GTAB) and two (MID_GTABN), band-limit bookkeeping on.  phaseGrating src/crystalMaker.cu:507-536,
    forwardPropagation src/multisliceSimulation.cu:538-549."""
    hp, at = S.case_tiny(m=m, m3=5, nz=nz, nat=300, tilt=True, seed=21 + nz)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    eng = fdes_amd.Engine(0, pass_threads=threads, skip_empty=0)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 2
    psi = pl.tap_wave(0, 0)
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    check(psi, ref, oracle.wave(q, at, 0, 0, prec="f32"), 1e-5, f"exit wave {m}^2 threads={threads} nz={nz}")
    xyz = oracle.config_coords(q, at, 0, -1)
    for s in (2, 3):
        V = pl.tap_potential(0, 0, s)
        check(V, oracle.phase_grating(q, at, xyz, s, "f64"), None, 1e-5, f"potential s={s} {m}^2 threads={threads} nz={nz}")
    pl.close()
    eng.close()


@pytest.mark.parametrize("kw", [dict(m=2048, rect=True, m3=4, nz=2, nat=300, tilt=True), dict(m=4096, rect=True, m3=3, nz=1, nat=300),
                                dict(m=2048, rect=True, m3=5, nz=2, nat=200, frPh=3, mode=1, beam_tilt=True)])
def test_rectangular_grids_with_padded_rows(oracle, kw):
    """2048 x 1024 and 4096 x 2048 grids: the two row lengths of the fused loop differ, rows are padded (pitch_pad 64
    from 2048 points on) and the band-limit bookkeeping is off (it assumes a square grid): exit wave / image against the
    float64 oracle, on the one-lane two-stream loop and on lanes + graph replay."""
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    assert hp.c.m1 == 2 * hp.c.m2
    eng = fdes_amd.Engine(0)
    if kw.get("frPh", 0) > 0:
        img = eng.build_measurements(hp, at)["image"]
        check(img, oracle.build_measurements(hp, at, prec="f64")["image"], None, 1e-5, f"rectangular image {kw}")
    else:
        q, _ = oracle.sub_sliced(hp)
        pl = eng.plan(hp, at)
        assert pl.fft_backend() == 2
        psi = pl.tap_wave(0, 0)
        check(psi, oracle.wave(q, at, 0, 0, prec="f64"), None, 1e-5, f"rectangular exit wave {kw}")
        pl.close()
    eng.close()


@pytest.mark.parametrize("threads,band_skip", [pytest.param(0, 1, marks=full_only), (512, 0)])   # (defaults at 4096^2: test_c5_full_specimen)
def test_c5_grid_slice_loop_against_oracle(oracle, threads, band_skip):
    """BASELINE config 5's grid (4096^2): the 512-thread / 136 KiB pass geometry (MID_ATOMS, MID_GTABN, MID_EXPIV_PAIR,
    MID_MASK, MID_MULPSI, MID_PTAB at N = 4096) through five slices with two species and 600 atoms, against the float64
    oracle: exit wave and the potential of a slice pair; also with the band-limit bookkeeping off and with the
    explicit 512-thread option."""
    hp, at = S.case_tiny(m=4096, m3=5, nz=2, nat=600, tilt=True, seed=31)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    eng = fdes_amd.Engine(0, pass_threads=threads, skip_empty=0, band_skip=band_skip)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 2
    psi = pl.tap_wave(0, 0)
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    r32 = oracle.wave(q, at, 0, 0, prec="f32") if threads == 0 else None
    check(psi, ref, r32, 1e-5, f"C5 grid exit wave, threads={threads} band_skip={band_skip}")
    if threads == 0:
        xyz = oracle.config_coords(q, at, 0, -1)
        for s in (0, 1, 4):
            V = pl.tap_potential(0, 0, s)
            check(V, oracle.phase_grating(q, at, xyz, s, "f64"), None, 1e-5, f"C5 grid potential s={s}")
    pl.close()
    eng.close()


@full_only   # (test_c5_full_specimen runs both launch sequences at 4096^2; the short cut itself: tests/test_gpu_fused.py, test_gpu_r4.py)
def test_c5_grid_engine_default_with_empty_slices(oracle):
    """4096^2 with the engine defaults (graph replay, two lanes, empty-slice runs as P^n) on a specimen that leaves
    slices empty at both ends: image of two frozen-phonon configurations against the float64 oracle."""
    hp, at = S.case_tiny(m=4096, m3=8, nz=2, nat=300, zfrac=0.2, frPh=2, seed=33)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    img = eng.build_measurements(hp, at)["image"]
    eng.close()
    ref = oracle.build_measurements(hp, at, prec="f64")["image"]
    check(img, ref, None, 1e-5, "C5 grid image, engine defaults")


def test_c5_full_specimen(oracle):
    """BASELINE config 5 itself: Au cuboctahedron k = 60 (738 221 atoms), 4096^2 wave, 512 slices, frozen phonons.  (1) the
    wave after the first 20 slices (the particle begins at slice 12) of configuration (0, 0) - the densest binning the engine sees: 1 440 atoms per slice -
    against the float32 / float64 oracle; (2) one configuration through all 512 slices on both launch sequences (every
    slice the full sequence / runs of empty slices as one P^n step): finite, normalised (no absorption: mean 1 up to the
    aperture), and equal to each other within the float32 error of 512 slices."""
    hp, at = S.case_c5()
    fdes_amd.consistent(hp)
    assert (hp.c.m1, hp.c.m3, hp.c.frPh, at.n) == (4096, 512, 16, 738221)
    q, _ = oracle.sub_sliced(hp)
    imgs = {}
    for skip in (0, 1):
        eng = fdes_amd.Engine(0, skip_empty=skip)
        pl = eng.plan(hp, at)
        if skip == 0:
            psi = pl.tap_wave(0, 0, 20)
            ref = oracle.wave(q, at, 0, 0, nslices=20, prec="f64")
            check(psi, ref, oracle.wave(q, at, 0, 0, nslices=20, prec="f32"), 1e-5, "C5 specimen, wave after 20 slices")
            assert np.abs(ref - 1).max() > 0.1   # the particle has begun
        pl.begin_measurement(0)
        pl.run_config(0, 0, 1.0)
        pl.end_measurement(0)
        imgs[skip] = pl.get_images()[0].astype(np.float64)
        pl.close()
        eng.close()
    for skip, img in imgs.items():
        assert np.isfinite(img).all()
        print(f"[parity] C5 full depth, skip_empty={skip}: image mean {img.mean():.6f} min {img.min():.4f} max {img.max():.4f}")
        assert 0.7 < img.mean() < 1.02 and img.min() >= 0
    e = relerr(imgs[1], imgs[0])
    print(f"[parity] C5 full depth: P^n runs vs full sequence on every slice: {e:.3e}")
    assert e < 1e-4
    assert imgs[0].std() / imgs[0].mean() > 0.05


def test_c4_full_series(oracle):
    """BASELINE config 4 at full series length: 64 beam tilts x 8 frozen-phonon configurations, SrTiO3 9x9x20 cells
    (8 100 atoms, three species), 1024^2 wave, 40 slices = 20 480 slice-propagations through three lanes and graph
    replay.  Checked: (1) every image finite and normalised; (2) the images of three measurements equal the same
    measurements re-run alone through the plan interface (the 512 configurations in flight do not leak into each other);
    (3) exit waves of single (k, j) configurations against the float32 / float64 oracle; (4) distinct tilts differ;
    (5) round 4: the IN-SERIES images of k = 27 and k = 63 (gangs of four configurations on two lanes, the default at
    1024^2) against the float32 oracle's images of those measurements (8 configurations x 40 slices each, the same Philox
    streams: every stream is keyed on (k, j))."""
    hp, at = S.case_c4()
    fdes_amd.consistent(hp)
    assert (hp.c.n3, hp.c.frPh, hp.c.m1, hp.c.m3, at.n) == (64, 8, 1024, 40, 8100)
    eng = fdes_amd.Engine(0)
    out = eng.build_measurements(hp, at)["image"]
    assert np.isfinite(out).all()
    means = out.reshape(64, -1).mean(axis=1)
    print("[parity] C4 full series: image means", means.min(), means.max())
    assert means.min() > 0.5 and means.max() < 1.05   # absorptive potential: a little below 1
    pl = eng.plan(hp, at)
    w = float(np.float32(1.0) / np.float32(8))
    for k in (0, 27, 63):
        pl.begin_measurement(k)
        for j in range(8):
            pl.run_config(k, j, w)
        pl.end_measurement(k)
    alone = pl.get_images()
    for k in (0, 27, 63):
        e = relerr(out[k], alone[k].astype(np.float64))
        print(f"[parity] C4 full series, measurement {k}: in the series vs alone {e:.3e}")
        assert e < 2e-6   # association order of the lane sums only
    q, _ = oracle.sub_sliced(hp)
    for (k, j) in ((27, 5), (63, 0)):
        psi = pl.tap_wave(k, j)
        ref = oracle.wave(q, at, k, j, prec="f64")
        check(psi, ref, oracle.wave(q, at, k, j, prec="f32"), 1e-4, f"C4 exit wave (k={k}, j={j})")
    assert relerr(out[0], out[63]) > 1e-3
    pl.close()
    eng.close()
    for k in (27, 63):
        ref = oracle.measurement(hp, at, k, prec="f32")
        e = relerr(out[k], ref.astype(np.float64))
        print(f"[parity] C4 full series, in-series image of measurement {k} vs the float32 oracle: {e:.3e}")
        assert e <= 5e-5
        assert np.abs(out[k] - ref).max() <= 1e-3 * ref.max()


# ------------------------------------------------------------------------------------------------------------------
# large projected phases: the fused passes' own sine / cosine against the generic path's sincosf
# ------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("occ", [3.0e2, 1.0e5])
def test_large_projected_phase(oracle, occ):
    """potential2Transmission (src/multisliceSimulation.cu:41-52) with phases far beyond what a specimen produces
    (occupancy scaled up: |sigma v_z| up to ~750 rad, still on the fast path, and ~2.5e5 rad): the fused passes switch from their Cody-Waite sine /
    cosine to a reduction in double precision above 1e3 rad.  At these magnitudes one float32 ulp of the phase is 1e-4 to
    4e-3 rad, so the yardstick is the float32 oracle's own error, not a fixed tolerance."""
    hp, at = S.case_tiny(m=256, m3=2, nz=2, nat=80, imPot=0.0, seed=41)
    at = fdes_amd.HostAtoms(at.Z, at.xyz, at.dwf, np.full(at.n, occ, np.float32))
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    xyz = oracle.config_coords(q, at, 0, -1)
    vmax = max(np.abs(oracle.phase_grating(q, at, xyz, s, "f64").real).max() for s in range(2))
    r64 = oracle.wave(q, at, 0, 0, prec="f64")
    e32 = relerr(oracle.wave(q, at, 0, 0, prec="f32"), r64)
    waves = {}
    for fft in (1, 2):
        eng = fdes_amd.Engine(0, fft=fft, skip_empty=0)
        pl = eng.plan(hp, at)
        waves[fft] = pl.tap_wave(0, 0)
        pl.close()
        eng.close()
    e_gen, e_fused = relerr(waves[1], r64), relerr(waves[2], r64)
    print(f"[parity] large phase (max |V| = {vmax:.3g} rad): E(fused)={e_fused:.3e} E(generic)={e_gen:.3e} E(cpu_f32)={e32:.3e}")
    assert (vmax > 1e4) if occ > 1e4 else (100 < vmax < 1e3)
    assert np.isfinite(waves[2]).all()
    assert e_fused <= 4 * max(e32, e_gen) + 1e-6


# ------------------------------------------------------------------------------------------------------------------
# teardown: a host that exits with plans / contexts open
# ------------------------------------------------------------------------------------------------------------------

_CHILD = """
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import fdes_amd
from tests import specimens as S
mode = sys.argv[1]
hp, at = S.case_tiny(m={m}, m3=3, nz=2, nat=40)
fdes_amd.consistent(hp)
eng = fdes_amd.Engine(0)
pl = eng.plan(hp, at)
pl.begin_measurement(0)
pl.run_config(0, 0, 1.0)
if mode == "raise":
    raise SystemExit(7)           # interpreter shutdown with live objects, finalisers in arbitrary order
if mode == "destroy_ctx_first":
    lib = fdes_amd.load_library()
    assert lib.fdes_destroy(eng.h) == 0        # takes its plans down
    assert lib.fdes_plan_destroy(pl.h) == -1   # the stale plan handle is refused, not dereferenced
    assert lib.fdes_destroy(eng.h) == -1
    eng.h = None; pl.h = None
    print("ok"); sys.exit(0)
if mode == "leak":
    import ctypes
    # drop the Python wrappers without closing: only the library's own at-exit sweep is left to clean up
    eng.close = lambda: None; pl.close = lambda: None
    print("ok"); sys.exit(0)
print("ok")
"""


@pytest.mark.parametrize("m", [64, 256])
@pytest.mark.parametrize("mode", ["plain", "raise", "destroy_ctx_first", "leak"])
def test_process_exit_with_open_plan(tmp_path, mode, m):
    """A host process that ends while a plan (rocFFT path at 64^2, fused path with lanes and graphs at 256^2) is still
    open and has work in flight must exit cleanly: plans are registered in their context, fdes_destroy takes them down,
    and whatever is left at exit is closed by the library before the HIP runtime unloads (round 1 saw exit code 135
    from a run that died with a plan open)."""
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent(_CHILD.format(root=ROOT, m=m)))
    r = subprocess.run([sys.executable, str(script), mode], capture_output=True, text=True, timeout=300, cwd=ROOT)
    print(r.stdout[-300:], r.stderr[-600:])
    assert r.returncode == (7 if mode == "raise" else 0), (r.returncode, r.stderr[-2000:])


# ------------------------------------------------------------------------------------------------------------------
# options, lanes, graph cache, progress
# ------------------------------------------------------------------------------------------------------------------

def test_options_set_after_plan_creation_reach_every_lane(oracle):
    """seed / skip_empty / band_skip changed on a context that already has a plan: the configurations dealt to lanes
    1..n must follow (round 1 copied them into the lane contexts at plan creation only)."""
    hp, at = S.case_tiny(m=256, m3=6, nz=2, nat=80, frPh=4, zfrac=0.25, seed=51)
    fdes_amd.consistent(hp)
    w = 0.25

    def run(eng, pl):
        pl.begin_measurement(0)
        for j in range(4):
            pl.run_config(0, j, w)
        pl.end_measurement(0)
        return pl.get_images()

    a = fdes_amd.Engine(0, seed=7, skip_empty=0, band_skip=0, lanes=2)   # (a job this small would get one lane of gangs)
    pa = a.plan(hp, at)
    assert pa.lanes() >= 2
    ref = run(a, pa)
    b = fdes_amd.Engine(0, lanes=2)  # defaults: seed 1, skip_empty 1, band_skip 1
    pb = b.plan(hp, at)
    first = run(b, pb)
    for k, v in (("seed", 7), ("skip_empty", 0), ("band_skip", 0)):
        b.set_option(k, v)
    late = run(b, pb)
    assert relerr(first, ref.astype(np.float64)) > 1e-4          # another seed: other displacements
    e = relerr(late, ref.astype(np.float64))
    print(f"[parity] options changed after plan creation vs set before: {e:.3e}")
    assert e < 1e-6   # the round-robin dealer continues where the first run stopped: another association of the lane sums
    for e in (pa, pb):
        e.close()
    a.close(); b.close()


@pytest.mark.parametrize("m", [256, 2048])
def test_layout_and_launch_options_do_not_change_a_bit(m):
    """Row padding (`pitch_pad`), split launches (`walk`) and the potential chain on a stream of its own (`split`) change
    where data lives and how and when a pass is launched, not what is computed: the exit wave must be bit-identical to
    the dense, whole-launch, one-stream run."""
    hp, at = S.case_tiny(m=m, m3=5, nz=2, nat=200, tilt=True, seed=71)
    fdes_amd.consistent(hp)
    waves = {}
    # (round 4: a pass launched in parts, walk > 1, runs on the multi-wave row kernels of fft_lds.hip - the one-wave-per-row
    #  kernels, the default at 1024 and 2048 points, do not implement it - so those runs are compared bit for bit with a
    #  whole-launch run on the same kernel family, pass_threads = 256, and to rounding with the default family)
    for key, opts in {"dense": dict(pitch_pad=0, walk=1, split=0), "padded": dict(pitch_pad=64, walk=1), "odd pad": dict(pitch_pad=136, walk=1),
                      "two streams": dict(split=1), "two streams, no graph": dict(split=1, graph=0), "one stream": dict(split=0),
                      "dense, multi-wave rows": dict(pitch_pad=0, walk=1, split=0, pass_threads=256),
                      "halves": dict(pitch_pad=64, walk=2, pass_threads=256), "quarters": dict(pitch_pad=0, walk=4, pass_threads=256),
                      "halves, default family asked": dict(pitch_pad=64, walk=2)}.items():
        eng = fdes_amd.Engine(0, skip_empty=0, **opts)
        pl = eng.plan(hp, at)
        waves[key] = pl.tap_wave(0, 0)
        pl.close()
        eng.close()
    for key, w in waves.items():
        walked = key in ("dense, multi-wave rows", "halves", "quarters") or (m == 2048 and key.startswith("halves, default"))
        ref = waves["dense, multi-wave rows"] if walked else waves["dense"]
        assert np.array_equal(w.view(np.float32), ref.view(np.float32)), key
    assert relerr(waves["dense, multi-wave rows"], waves["dense"].astype(np.complex128)) < 1e-6
    assert np.isfinite(waves["dense"]).all() and np.abs(waves["dense"] - 1).max() > 1e-3   # not the vacuum wave


def test_dense_specimen_stops_asking_for_empty_slices(oracle):
    """`skip_empty` asks once per configuration which slices hold no atom (one D2H and one host wait on the lane).  After
    eight configurations in a row without an empty slice the engine stops asking (and asks again every 64th): every slice
    then takes the full sequence, which is what it took anyway.  The same configuration run before and after that point
    gives the same bits, and a specimen WITH empty slices keeps being asked (its result stays that of the per-slice
    decision, equal to the full sequence within rounding)."""
    hp, at = S.case_tiny(m=256, m3=6, nz=2, frPh=24, nat=900, zfrac=0.5, seed=5)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, lanes=1, split=0, gang=0)   # (a gang asks once for all its members: below)
    pl = eng.plan(hp, at)
    first = pl.tap_wave(0, 3)
    pl.begin_measurement(0)
    for j in range(20):
        pl.run_config(0, j, 1.0 / 20)
    pl.sync()
    assert pl.empty_queries() == 8          # the tap and seven configurations; the other thirteen were not asked
    late = pl.tap_wave(0, 3)
    assert np.array_equal(first.view(np.float32), late.view(np.float32))
    assert relerr(late, oracle.wave(oracle.sub_sliced(hp)[0], at, 0, 3, prec="f64")) < 2e-5
    pl.close(); eng.close()
    eng = fdes_amd.Engine(0, lanes=1, gang=8)            # 20 configurations = gangs of 8, 8, 4: one question per gang,
    pl = eng.plan(hp, at)                                # none once eight members in a row had no empty slice
    pl.begin_measurement(0)
    for j in range(20):
        pl.run_config(0, j, 1.0 / 20)
    pl.sync()
    assert pl.gang() == 8 and pl.empty_queries() == 1
    pl.close(); eng.close()
    # a specimen with vacuum above and below: every configuration is asked, the short cut keeps being taken
    hp2, at2 = S.case_tiny(m=256, m3=12, nz=2, frPh=24, nat=300, zfrac=0.2, seed=6)
    fdes_amd.consistent(hp2)
    waves = []
    for skip in (1, 0):
        eng = fdes_amd.Engine(0, lanes=1, split=0, skip_empty=skip, gang=0)
        pl = eng.plan(hp2, at2)
        pl.begin_measurement(0)
        for j in range(12):
            pl.run_config(0, j, 1.0 / 12)
        pl.sync()
        assert pl.empty_queries() == (12 if skip else 0)
        waves.append(pl.tap_wave(0, 5))
        pl.close(); eng.close()
    assert relerr(waves[0], waves[1]) < 1e-6


@pytest.mark.parametrize("kw", [dict(m=256, m3=6, nz=2, frPh=4, nat=150, tilt=True), dict(m=256, m3=4, nz=2, mode=2, nat=80),
                                dict(m=1024, m3=5, nz=3, frPh=3, nat=300, beam_tilt=True, n3=2),
                                dict(m=320, m3=6, nz=2, frPh=3, nat=200, tilt=True), dict(m=800, m3=4, nz=2, mode=2, nat=150),
                                dict(m=2048, m3=4, nz=2, frPh=2, nat=300), dict(m=374, m3=4, nz=2, frPh=2, nat=150, fft=1),
                                dict(m=256, m3=4, nz=2, frPh=2, nat=150, fft=1)])
def test_fused_path_is_bit_reproducible(kw):
    """Two runs of the same simulation on the fused path give the same bits: the deposits go through single-wave LDS
    atomics in sorted order, lanes are dealt round-robin and folded in lane order, graphs replay fixed launch sequences,
    and the CBED probe norm is a fixed-order two-stage sum (round 1: float atomicAdd across blocks).  Round 3: the same
    for the mixed-radix grids (320, 800), the one-wave-per-row passes (2048) and the rocFFT path (fft = 1; 374 = 2 11 17
    has kernels only when they are compiled at plan creation, which this suite turns off): its deposit now adds the atoms in sorted order through an LDS tile, and the
    potential output of print_level 1 with it."""
    kw = dict(kw)
    fft = kw.pop("fft", 0)
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    outs = []
    for rep in range(3):
        eng = fdes_amd.Engine(0, fft=fft)
        pl = eng.plan(hp, at)
        assert pl.fft_backend() == (1 if fft == 1 else 2)
        pl.close()
        outs.append(eng.build_measurements(hp, at, want_exitwave=True, want_potential=True))
        eng.close()
    for o in outs[1:]:
        assert np.array_equal(o["image"].view(np.uint32), outs[0]["image"].view(np.uint32))
        assert np.array_equal(o["exitwave"].view(np.uint32), outs[0]["exitwave"].view(np.uint32))
        assert np.array_equal(o["potential"].view(np.uint32), outs[0]["potential"].view(np.uint32))


def test_fft_option_is_part_of_the_plan_cache_key():
    hp, at = S.case_tiny(m=256, m3=2, nz=1, nat=10)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    backends = []
    for fft in (2, 1, 0):
        eng.set_option("fft", fft)
        pl = eng.plan(hp, at)
        backends.append(pl.fft_backend())
        pl.close()
    eng.close()
    assert backends == [2, 1, 2]


def test_progress_callback(oracle):
    hp, at = S.case_tiny(m=256, m3=8, nz=2, nat=60, frPh=6, n3=2)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    seen = []
    eng.set_progress(lambda done, total: seen.append((done, total)), min_interval_ms=0)
    img = eng.build_measurements(hp, at)["image"]
    eng.set_progress(None)
    ref = eng.build_measurements(hp, at)["image"]
    eng.close()
    total = 2 * 6 * 8
    assert seen and seen[-1] == (total, total)
    assert all(t == total for _, t in seen)
    assert [d for d, _ in seen] == sorted(d for d, _ in seen) and all(d % 8 == 0 for d, _ in seen)
    assert np.array_equal(img, ref)     # bounding the queue depth does not change results


def test_multi_gpu_driver_on_distinct_devices_when_present(engine):
    """fdes_build_measurements_multi on different device ids whenever the box has more than one GPU (the per-device
    dynamic-LDS attribute of the pass kernels, the per-device tables); on a one-GPU box the same call with two workers
    on device 0, with the exit-wave and potential outputs."""
    import torch
    n = torch.cuda.device_count()
    devs = list(range(min(n, 4))) if n > 1 else [0, 0]
    hp, at = S.case_tiny(m=256, m3=6, nz=2, frPh=4, n3=2, tilt=True)
    fdes_amd.consistent(hp)
    ref = engine.build_measurements(hp, at, want_potential=True, want_exitwave=True)
    out = fdes_amd.build_measurements_multi(devs, hp, at, want_potential=True, want_exitwave=True)
    print(f"[parity] multi-GPU driver on devices {devs}: image {relerr(out['image'], ref['image'].astype(np.float64)):.3e}")
    assert relerr(out["image"], ref["image"].astype(np.float64)) < 2e-6
    assert relerr(out["exitwave"], ref["exitwave"].astype(np.float64)) < 2e-6
    assert np.array_equal(out["potential"], ref["potential"])


# ------------------------------------------------------------------------------------------------------------------
# physics, written from the textbook rather than from the oracle's source
# ------------------------------------------------------------------------------------------------------------------

def test_weak_phase_object(engine):
    """One light atom, one slice, no absorption: the exit wave is F^-1[P F[BL(exp(i sigma v_z))]], so to first order in the
    phase psi - psi_0 = i * (P (x) BL[sigma v_z]) with psi_0 = P(0) = 1: the imaginary part of the exit wave equals the
    Fresnel-propagated band-limited projected potential, the real part deviates from 1 only in second order.  Uses only
    the potential tap, the propagator tap and numpy's FFT."""
    m = 256
    hp, _ = S.case_tiny(m=m, m3=1, nz=1, nat=1, imPot=0.0)
    at = fdes_amd.HostAtoms([3], [[0.0, 0.0, 0.0]], 6e-21, 1.0)   # lithium: sigma v_z << 1 away from the nucleus
    fdes_amd.consistent(hp)
    pl = engine.plan(hp, at)
    V = pl.tap_potential(0, 0, 0).real.astype(np.float64)
    P = pl.tap_propagator().astype(np.complex128) * (m * m)       # the tap carries 1 / (m1 m2)
    psi = pl.tap_wave(0, 0).astype(np.complex128)
    pl.close()
    mask = (np.abs(P) > 0)
    first = np.fft.ifft2(np.fft.fft2(1j * V) * mask * P)          # i * P (x) BL[V]
    second = np.fft.ifft2(np.fft.fft2(-0.5 * V * V) * mask * P)
    vmax = np.abs(V).max()
    err1 = np.abs(psi - (1.0 + first)).max()
    err2 = np.abs(psi - (1.0 + first + second)).max()
    print(f"[physics] weak phase object: max phase {vmax:.3e} rad, residual after 1st order {err1:.3e}, after 2nd order {err2:.3e}")
    assert 1e-3 < vmax < 0.5
    assert err1 < 1.0 * vmax ** 2 + 1e-6       # second-order remainder (band-limit ringing included)
    assert err2 < 0.5 * vmax ** 3 + 3e-6       # third-order remainder + float32 rounding


def test_inversion_symmetry_of_the_fused_path(engine):
    """A specimen that is invariant under (x, y) -> (-x, -y) about the pixel the grid's own inversion maps onto itself
    (index i -> m - i: pixel m/2 is fixed) gives an exit wave with psi[i2, i1] = psi[m - i2, m - i1]: the row passes,
    their transposed stores and the band-limit bookkeeping treat +k and -k alike."""
    m = 256
    rng = np.random.default_rng(61)
    hp, _ = S.case_tiny(m=m, m3=4, nz=2, nat=2)
    d = hp.c.d1
    half = rng.uniform(-0.3 * m * d, 0.3 * m * d, (40, 3)).astype(np.float32)
    half[:, 2] = rng.uniform(-1.9e-10, 1.9e-10, 40)
    # squareAtoms_d maps x -> x / d + m/2 - 0.5 (src/crystalMaker.cu:85): pixel coordinate u and its mirror image m - u
    # correspond to x and -x + d (the -0.5 offset), so the mirror partner of an atom at x sits at d - x
    mirror = half.copy()
    mirror[:, :2] = d - half[:, :2]
    xyz = np.concatenate([half, mirror]).astype(np.float32)
    Z = np.tile(np.array([79, 14] * 20, np.int32), 2)
    at = fdes_amd.HostAtoms(Z, xyz, 6e-21, 1.0)
    fdes_amd.consistent(hp)
    pl = engine.plan(hp, at)
    assert pl.fft_backend() == 2
    psi = pl.tap_wave(0, 0).astype(np.complex128)
    pl.close()
    flipped = np.roll(psi[::-1, ::-1], (1, 1), axis=(0, 1))       # index i -> (m - i) mod m
    e = np.linalg.norm(psi - flipped) / np.linalg.norm(psi)
    print(f"[physics] inversion symmetry residual of the exit wave: {e:.3e} (contrast {np.abs(psi).std():.3e})")
    assert np.abs(psi).std() > 3e-3
    assert e < 1e-4   # float32 rounding of the mirrored coordinates (d - x) and of the transforms; an asymmetric kernel gives 1e-2 or more
