"""GPU parity tests: the HIP engine (through the C-ABI) against the CPU oracle on the same inputs.

Stated tolerance (float32 wave optics, deterministic configurations): with
E(x) = ||x - truth_f64||_2 / ||truth_f64||_2,
    E(gpu) <= 1e-5 for <= 16 slices, E(gpu) <= 1e-4 up to 512 slices,
    max |gpu - truth| <= 1e-3 * max(truth),
and E(gpu) is printed beside E(cpu_f32) of the float32 oracle.  Atom geometry (tilts, frozen-phonon
displacements, slice binning) must be BIT-EXACT against the oracle.
"""
import os

import numpy as np
import pytest

import fdes_amd
from tests import specimens as S

from tests.conftest import full_only

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def relerr(a, b):
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    return float(np.linalg.norm(np.asarray(a, dt) - np.asarray(b, dt)) / np.linalg.norm(b))


def check(gpu, f64, f32=None, tol=1e-5, what=""):
    e = relerr(gpu, f64)
    e32 = relerr(f32, f64) if f32 is not None else float("nan")
    print(f"[parity] {what}: E(gpu)={e:.3e} E(cpu_f32)={e32:.3e}")
    assert e <= tol, (what, e, e32)
    assert np.abs(np.asarray(gpu) - f64).max() <= 1e-3 * np.abs(f64).max(), what


def test_library_is_the_hip_one():
    lib = fdes_amd.load_library()
    assert lib.fdes_gpu_available() == 1
    assert lib.fdes_abi_version() == 1


def test_coordinates_bit_exact(engine, oracle):
    hp, at = S.case_tiny(m=64, m3=4, nz=3, n3=3, tilt=True, frPh=2)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(oracle.consistent(hp.copy()))
    pl = engine.plan(hp, at)
    for (k, j) in [(-1, -1), (0, -1), (2, -1), (0, 0), (1, 1), (2, 0)]:
        g = pl.tap_coords(k, j)
        o = oracle.config_coords(q, at, k, j)
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32)), (k, j)
    pl.close()


def test_params_consistent_matches_oracle(oracle):
    for E0 in (40e3, 50e3, 80e3, 100e3, 200e3, 300e3):
        a = fdes_amd.consistent(S.make_params(2, E0=E0, tiltbeam=[0, 0, 1e-3, 0]))
        b = oracle.consistent(S.make_params(2, E0=E0, tiltbeam=[0, 0, 1e-3, 0]))
        for f in ("gamma", "lambda_", "sigma", "m1", "m2", "doBeamTilt"):
            assert getattr(a.c, f) == getattr(b.c, f), (E0, f)


def test_propagator(engine, oracle):
    for kw in (dict(m=64), dict(m=96, rect=True)):
        hp, at = S.case_tiny(m3=3, nz=1, **kw)
        fdes_amd.consistent(hp)
        q, _ = oracle.sub_sliced(hp)
        pl = engine.plan(hp, at)
        P = pl.tap_propagator()
        ref = oracle.fresnel_propagator(q, "f64")
        assert np.array_equal(P == 0, ref == 0)  # same band-limit mask
        check(P, ref, oracle.fresnel_propagator(q, "f32"), 1e-6, "propagator")
        pl.close()


@pytest.mark.parametrize("kw", [dict(m=64, nz=1), dict(m=64, nz=3), dict(m=96, nz=2, rect=True), dict(m=60, nz=2),
                                dict(m=100, nz=2, rect=True)])
def test_potential_per_slice(engine, oracle, kw):
    hp, at = S.case_tiny(m3=4, tilt=True, **kw)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    xyz = oracle.config_coords(q, at, 0, -1)
    pl = engine.plan(hp, at)
    for s in range(q.c.m3):
        V = pl.tap_potential(0, 0, s)
        ref = oracle.phase_grating(q, at, xyz, s, "f64")
        if np.abs(ref).max() == 0:
            assert np.abs(V).max() == 0
            continue
        check(V, ref, oracle.phase_grating(q, at, xyz, s, "f32"), 1e-5, f"potential {kw} s={s}")
    pl.close()


@pytest.mark.parametrize("kw", [dict(m=64, m3=4, nz=2), dict(m=64, m3=6, nz=3, tilt=True, beam_tilt=True, n3=2),
                                dict(m=64, m3=3, nz=2, frPh=2, sub=3), dict(m=128, m3=16, nz=2, nat=200),
                                dict(m=60, m3=4, nz=2), dict(m=64, m3=4, nz=2, mode=2)])
def test_exit_wave(engine, oracle, kw):
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    pl = engine.plan(hp, at)
    k = q.c.n3 - 1
    j = 1 if q.c.frPh > 0 else 0
    psi = pl.tap_wave(k, j)
    ref = oracle.wave(q, at, k, j, prec="f64")
    check(psi, ref, oracle.wave(q, at, k, j, prec="f32"), 1e-5 if q.c.m3 <= 16 else 1e-4, f"exit wave {kw}")
    # intermediate depth
    psi = pl.tap_wave(k, j, 1)
    check(psi, oracle.wave(q, at, k, j, nslices=1, prec="f64"), None, 1e-5, f"wave after 1 slice {kw}")
    pl.close()


@pytest.mark.parametrize("name", list(S.GOLDEN_CASES))
def test_images_against_golden_and_oracle(engine, oracle, name):
    kw = S.GOLDEN_CASES[name]
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    out = engine.build_measurements(hp, at, want_potential=True, want_exitwave=True)
    g = np.load(os.path.join(G, "tiny_cases.npz"))
    check(out["image"], g[name + "_f64"], g[name + "_f32"], 1e-5, f"golden image {name}")
    ref = oracle.build_measurements(hp, at, prec="f64", want_potential=True, want_exitwave=True)
    check(out["image"], ref["image"], None, 1e-5, f"oracle image {name}")
    check(out["exitwave"], ref["exitwave"], None, 1e-5, f"exit wave stack {name}")
    check(out["potential"], ref["potential"], None, 1e-5, f"potential stack {name}")


def test_dose_noise_statistics(engine, oracle):
    """pixel_dose > 0: the Poisson surrogate rounds to whole counts, so float32 rounding can flip single
    pixels; deviates are bit-identical (Philox), hence almost all pixels must agree to rounding."""
    hp, at = S.case_tiny(m=64, m3=3, nz=2, pD=40.0)
    hp.set(mtfa=1.0, mtfb=0.0, mtfc=0.0, mtfd=0.0)
    fdes_amd.consistent(hp)
    img = engine.build_measurements(hp, at)["image"]
    ref = oracle.build_measurements(hp, at, prec="f32")["image"]
    close = np.abs(img - ref) < 1e-4 * ref.max()
    print("[parity] dose: fraction of pixels equal to rounding:", close.mean())
    assert close.mean() > 0.98
    assert abs(img.mean() - ref.mean()) < 2e-3 * ref.mean()


def test_sharding_equals_single_plan(engine, oracle):
    """Two 'ranks' (two plans on one GPU) run the block partition of (k, j); their partial intensity sums are
    exchanged through fdes_plan_copy_intensity, summed (what the RCCL all-reduce does) and finalised by the
    owner of k: equals the single-plan result (RNG keyed on (k, j), not on processing order)."""
    import torch
    from fdes_amd import shard
    hp, at = S.case_tiny(m=64, m3=3, nz=2, frPh=3, n3=2, tilt=True)
    fdes_amd.consistent(hp)
    full = engine.build_measurements(hp, at)["image"]
    world = 2
    plans = [engine.plan(hp, at) for _ in range(world)]
    bufs = [torch.zeros(64 * 64 * 2, device="cuda") for _ in range(world)]
    torch.cuda.synchronize()

    class Rank:
        def __init__(self, r):
            self.r, self.pl = r, plans[r]
            self.begin_measurement = self.pl.begin_measurement
            self.run_config = self.pl.run_config
            self.end_measurement = self.pl.end_measurement

    # ranks run one after the other; the "collective" is executed when the last rank arrives
    pending = {}

    def reduce_fn(rank_obj, k):
        rank_obj.pl.copy_intensity(bufs[rank_obj.r].data_ptr(), 0)
        pending.setdefault(k, []).append(rank_obj.r)

    # phase 1: every rank computes its partial sums (end_measurement deferred): drive run_sharded per rank with a
    # reduce hook that only snapshots; then sum and finalise on the owner.
    own, ranks_of = shard.owners(2, 3, world)
    w = 1.0 / 3.0
    for k in range(2):
        for r in range(world):
            plans[r].begin_measurement(k)
            for (kk, j) in shard.partition(2, 3, world, r):
                if kk == k:
                    plans[r].run_config(k, j, w)
            plans[r].copy_intensity(bufs[r].data_ptr(), 0)
        total = bufs[0] + bufs[1]
        torch.cuda.synchronize()
        plans[own[k]].copy_intensity(total.data_ptr(), 1)
        plans[own[k]].end_measurement(k)
    out = np.zeros_like(full)
    for k in range(2):
        out[k] = plans[own[k]].get_images()[k]
    for pl in plans:
        pl.close()
    print("[parity] sharded vs single:", relerr(out, full.astype(np.float64)))
    assert relerr(out, full.astype(np.float64)) < 1e-6


def test_shipped_style_cnf_through_legacy_symbol(engine, oracle, tmp_path):
    """The pyFDES.py call sequence: .cnf on disk + atoms as a flat [Z,x,y,z,DWF,occ] array -> FDES() -> buffer ==
    Measurements.bin == oracle."""
    hp, at = S.case_tiny(m=64, m3=3, nz=2, n3=2, tilt=True)
    at.occ[:] = 1.0  # the legacy entry truncates occupancies to int (src/paramStructure.cu:323)
    fdes_amd.consistent(hp)
    cnf = tmp_path / "dataFDES.cnf"
    fdes_amd.write_cnf(cnf, hp, None)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        out = np.zeros((hp.c.n3, hp.c.n2, hp.c.n1), np.float32)
        fdes_amd.run_file(str(cnf), "Measurements.bin", "results.emd", atoms=at, out=out)
        disk = np.fromfile("Measurements.bin", np.float32).reshape(out.shape)
        assert os.path.exists("dataFDES_used.cnf") and os.path.exists("config.emd")
        # the caller's atom list is echoed as testRead.txt (src/paramStructure.cu:344)
        _, at_echo = fdes_amd.read_cnf("testRead.txt", bug_compatible=False)
        assert at_echo.n == at.n and np.array_equal(at_echo.Z, at.Z)
    finally:
        os.chdir(cwd)
    assert np.array_equal(out, disk)
    hp2, _ = fdes_amd.read_cnf(cnf, skip_atoms=True)
    ref = oracle.build_measurements(hp2, at, prec="f64")["image"]
    check(out, ref, None, 1e-5, "legacy FDES() symbol")


def test_full_size_propagation_properties(engine):
    """BASELINE size (2048^2): size-independent properties of the propagation unit on device buffers:
    linearity and band-limited norm conservation (t = 1)."""
    import torch
    hp, at = S.case_c3(k=2, m3=2, frPh=0)
    fdes_amd.consistent(hp)
    pl = engine.plan(hp, at)
    m = 2048
    gen = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(m, m, 2, device="cuda", generator=gen)
    b = torch.randn(m, m, 2, device="cuda", generator=gen)
    t = torch.zeros(m, m, 2, device="cuda")
    t[..., 0] = 1.0
    torch.cuda.synchronize()

    def prop(x, n=1):
        y = x.clone()
        torch.cuda.synchronize()
        for _ in range(n):
            pl.propagate_dev(y.data_ptr(), t.data_ptr())
        pl.sync()
        return y

    pa, pb, pab = prop(a), prop(b), prop(2.0 * a - 0.5 * b)
    lin = (pab - (2.0 * pa - 0.5 * pb)).norm() / pab.norm()
    print("[parity] 2048^2 linearity residual:", float(lin))
    assert float(lin) < 5e-6
    # after one step psi is band-limited; further steps with t = 1 conserve the norm
    p2 = prop(pa, 4)
    drift = abs(float(p2.norm() / pa.norm()) - 1.0)
    print("[parity] 2048^2 norm drift over 4 free-space steps:", drift)
    assert drift < 1e-5
    pl.close()


def test_shipped_au309_example_through_the_cli(oracle, tmp_path):
    """The reference's own example (ExampleSpecimens/Au_cubeoctahedron_emd/Auparticle.emd: 309 Au atoms, 320^2 wave,
    12 slices -> 132 sub-slices, 25 specimen tilts, pixel dose 100) through the FDES command line: .emd reader,
    non-power-of-two grid (320 = 2^6 5: mixed-radix fused passes since round 3), sub-slicing, tilts, detector chain with Poisson surrogate noise,
    Measurements.bin + results.emd writers."""
    import subprocess
    src = os.path.join(G, "Auparticle_config.emd")
    exe = os.path.join(os.path.dirname(G), "..", "fdes_amd", "csrc", "FDES")
    r = subprocess.run([os.path.abspath(exe), "--input_name", src, "--image_name", "Measurements.bin", "--emd_name", "results.emd"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Wave 320 x 320, slice loop: fused LDS passes" in r.stderr  # round 3: mixed-radix passes, fft_backend == 2
    hp, at = fdes_amd.read_emd(src)
    img = np.fromfile(tmp_path / "Measurements.bin", np.float32).reshape(hp.c.n3, hp.c.n2, hp.c.n1)
    ref = oracle.build_measurements(hp, at, prec="f32")["image"]
    close = np.abs(img - ref) < 1e-4 * ref.max()
    print("[parity] Au-309 CLI: fraction of pixels equal to rounding:", close.mean(), "mean", img.mean(), ref.mean())
    assert close.mean() > 0.97
    assert abs(img.mean() - ref.mean()) < 2e-3 * ref.mean()
    print("[parity] Au-309 CLI: libhdf5 available, results.emd read back:", bool(fdes_amd.emd_available()))
    if fdes_amd.emd_available():
        hp2, at2 = fdes_amd.read_emd(tmp_path / "results.emd")
        assert hp2.c.n3 == 25 and at2.n == 309
    # an .emd input echoes its parameters as ParamsUsedEmd.txt (src/rwHdf5.cu:2565) and writes no config.emd (src/FDES.cu:229)
    used, at_used = fdes_amd.read_cnf(tmp_path / "ParamsUsedEmd.txt", bug_compatible=False)
    assert at_used.n == 309 and (used.c.n1, used.c.n3, used.c.m3) == (160, 25, 12)
    assert not os.path.exists(tmp_path / "config.emd")


def test_srtio3_qsc_through_the_cli(oracle, tmp_path):
    """SURVEY C1 as a QSTEM input: SrTiO3.cfg (the reference's bin/SrTiO3.cfg) replicated 3x3x4 by the .qsc
    front-end, nx = 128 -> 256^2 wave, 8 slices of 1.9525 A cut into 80 sub-slices (subSlTh = d3/10,
    src/rwQsc.cu:953), imaging mode with objective aperture; FDES CLI -> ParamsUsedQsc.txt, Measurements.bin."""
    import shutil
    import subprocess
    shutil.copy(os.path.join(G, "qsc", "SrTiO3.cfg"), tmp_path / "SrTiO3.cfg")
    (tmp_path / "c1.qsc").write_text(
        "mode: TEM\nfilename: SrTiO3.cfg\nNCELLX: 3\nNCELLY: 3\nNCELLZ: 4\nv0: 200\ntds: no\n"
        "slice-thickness: 1.9525\nslices: 8\nnx: 128\nCs: 0.05\nalpha: 15\ndefocus: 13.7\n"
        "cal_mode: 0\nobjective_aperture: 20e-3\nabsorptive_potential_factor: 0.1\npixel_dose: 0\n")
    exe = os.path.abspath(os.path.join(os.path.dirname(G), "..", "fdes_amd", "csrc", "FDES"))
    r = subprocess.run([exe, "--input_name", "c1.qsc", "--image_name", "Measurements.bin", "--emd_name", "results.emd"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    hp, at = fdes_amd.read_qsc(tmp_path / "c1.qsc")
    assert (hp.c.m1, hp.c.m2, hp.c.m3, at.n) == (256, 256, 8, 180)
    used, at_used = fdes_amd.read_cnf(tmp_path / "ParamsUsedQsc.txt", bug_compatible=False)
    assert (used.c.m1, used.c.m3, at_used.n) == (256, 8, 180) and abs(used.c.subSlTh / hp.c.subSlTh - 1) < 1e-6
    img = np.fromfile(tmp_path / "Measurements.bin", np.float32).reshape(1, 128, 128)
    ref = oracle.build_measurements(hp, at, prec="f64")["image"]
    err = np.abs(img - ref).max() / np.abs(ref).max()
    print("[parity] SrTiO3 .qsc CLI vs f64 oracle: max rel err", err, "contrast", ref.std() / ref.mean())
    assert ref.std() / ref.mean() > 1e-3      # a non-trivial image
    assert err < 2e-4


def test_c2_si001_1024(engine, oracle):
    """BASELINE config 2 at full size: Si[001] 19x19x4 cells = 11 552 atoms, 1024^2 wave, 64 slices, one
    configuration; image against the float64 oracle (float32 oracle beside it)."""
    hp, at = S.case_c2()
    assert at.n == 11552
    fdes_amd.consistent(hp)
    out = engine.build_measurements(hp, at)["image"]
    r64 = oracle.build_measurements(hp, at, prec="f64")["image"]
    r32 = oracle.build_measurements(hp, at, prec="f32")["image"]
    assert r64.std() / r64.mean() > 1e-2
    check(out, r64, r32, 2e-5, "C2 Si[001] 1024^2 x 64 slices image")


@full_only   # (tests/test_gpu_r2.py::test_c4_full_series runs the whole series)
def test_c4_beam_tilt_series_reduced(engine, oracle):
    """BASELINE config 4 with the series cut to 2 x 2 beam tilts x 2 frozen-phonon configurations and half the
    thickness (SrTiO3 9x9x10 cells, 1024^2 wave, 20 slices): beam tilt + Tukey window + band limit on the way in,
    phonon averaging per tilt, three lanes + graph replay on the way through."""
    hp, at = S.case_c4(n3=4, frPh=2, cells=(9, 9, 10))
    fdes_amd.consistent(hp)
    assert hp.c.doBeamTilt == 1 and (hp.c.m1, hp.c.m3, at.n) == (1024, 20, 4050)
    out = engine.build_measurements(hp, at)["image"]
    r32 = oracle.build_measurements(hp, at, prec="f32")["image"]
    for k in range(4):
        e = relerr(out[k], r32[k])
        print(f"[parity] C4 tilt {k}: E(gpu vs cpu_f32) = {e:.3e}")
        assert e < 5e-5
    assert relerr(out[0], out[3]) > 1e-3      # different tilts give different images


def test_c5_size_propagation_properties(engine):
    """BASELINE config 5 grid (4096^2): linearity and free-space norm conservation of the propagation unit."""
    import torch
    hp, at = S.case_c5(k=2, frPh=0)
    hp.set(m3=2)
    fdes_amd.consistent(hp)
    pl = engine.plan(hp, at)
    m = 4096
    gen = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(m, m, 2, device="cuda", generator=gen)
    b = torch.randn(m, m, 2, device="cuda", generator=gen)
    t = torch.zeros(m, m, 2, device="cuda")
    t[..., 0] = 1.0
    torch.cuda.synchronize()

    def prop(x, n=1):
        y = x.clone()
        torch.cuda.synchronize()
        for _ in range(n):
            pl.propagate_dev(y.data_ptr(), t.data_ptr())
        pl.sync()
        return y

    pa, pb, pab = prop(a), prop(b), prop(2.0 * a - 0.5 * b)
    lin = float((pab - (2.0 * pa - 0.5 * pb)).norm() / pab.norm())
    drift = abs(float(prop(pa, 3).norm() / pa.norm()) - 1.0)
    print("[parity] 4096^2 linearity residual:", lin, "norm drift over 3 free-space steps:", drift)
    assert lin < 5e-6 and drift < 1e-5
    pl.close()


def _edge_case(kind):
    fused = kind.startswith("fused_")      # 256^2: hand-written passes; 64^2: generic rocFFT path
    kind = kind.replace("fused_", "")
    hp, at = S.case_tiny(m=256 if fused else 64, m3=4, nz=2, nat=60, seed=11)
    xyz = at.xyz.copy()
    d, m = hp.c.d1, (256 if fused else 64)
    if kind == "outside":
        # atoms beyond the lateral window (both sides), above the first and below the last slice, and exactly on the
        # slice boundaries and on the window edge
        xyz[0] = (0.9 * m * d, 0.0, 0.0)
        xyz[1] = (-0.9 * m * d, 0.2 * m * d, 0.0)
        xyz[2] = (0.0, 0.75 * m * d, 0.0)
        xyz[3] = (0.0, 0.0, 9.0e-10)
        xyz[4] = (0.0, 0.0, -9.0e-10)
        xyz[5] = (0.5 * m * d, 0.5 * m * d, 1.0e-10)
        xyz[6] = (-0.5 * m * d, -0.5 * m * d, -1.0e-10)
        xyz[7] = (0.0, 0.0, 2.0e-10)
        xyz[8] = (0.0, 0.0, -2.0e-10)
    if kind == "stacked":
        xyz[:30, :2] = xyz[0, :2]          # thirty atoms in one column: same pixels in every slice they fall into
    at = fdes_amd.HostAtoms(at.Z, xyz, at.dwf, at.occ)
    if kind == "one_slice":
        hp.set(m3=1)
    if kind == "no_atoms":
        at = fdes_amd.HostAtoms(at.Z[:0], xyz[:0], at.dwf[:0], at.occ[:0])
    if kind == "one_atom":
        at = fdes_amd.HostAtoms(at.Z[:1], xyz[:1] * 0, at.dwf[:1], at.occ[:1])
    return hp, at


@pytest.mark.parametrize("kind", ["outside", "fused_outside", "stacked", "fused_stacked", "one_slice", "fused_one_slice", "no_atoms",
                                  "fused_no_atoms", "one_atom", "fused_one_atom"])
def test_edge_cases_match_the_oracle(engine, oracle, kind):
    """Inputs at the edges of the domain: atoms outside the window / the slice range / on boundaries, many atoms on one
    pixel (colliding deposits), a single slice, a single atom, no atom at all (free-space propagation)."""
    hp, at = _edge_case(kind)
    fdes_amd.consistent(hp)
    out = engine.build_measurements(hp, at, want_potential=True)
    ref = oracle.build_measurements(hp, at, prec="f64", want_potential=True)
    pot = out["potential"][..., 0] + 1j * out["potential"][..., 1]
    rpot = ref["potential"][..., 0] + 1j * ref["potential"][..., 1]
    if np.abs(rpot).max() > 0:
        check(pot, rpot, None, 2e-5, f"edge case {kind}: potential stack")
    else:
        assert np.abs(pot).max() == 0
    check(out["image"], ref["image"], None, 1e-5, f"edge case {kind}: image")
    assert np.isfinite(out["image"]).all()


def test_shipped_si001_example_through_the_cli(oracle, tmp_path):
    """The reference's ExampleSpecimens/Si_001_11k_cnf/dataFDES_11k.cnf (11 552 Si atoms, 1000^2 wave = 2^3 5^3 points
    per side, 205 slices of 0.1 A, 100 kV, no aperture cut) through the FDES command line and the bug-compatible
    .cnf reader: mixed-radix fused passes (rocFFT path before round 3), Measurements.bin against the float32 oracle."""
    import subprocess
    src = os.path.join(G, "dataFDES_Si001_11k.cnf")
    exe = os.path.abspath(os.path.join(os.path.dirname(G), "..", "fdes_amd", "csrc", "FDES"))
    r = subprocess.run([exe, "--input_name", src, "--image_name", "Measurements.bin", "--emd_name", "results.emd"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Wave 1000 x 1000, slice loop: fused LDS passes" in r.stderr  # round 3: mixed-radix passes, fft_backend == 2
    hp, at = fdes_amd.read_cnf(src)
    assert (hp.c.m1, hp.c.m2, hp.c.m3, hp.c.n1) == (1000, 1000, 205, 660) and at.n in (11552, 11553)  # +1: duplicated last atom quirk
    img = np.fromfile(tmp_path / "Measurements.bin", np.float32).reshape(1, 660, 660)
    ref = oracle.build_measurements(hp, at, prec="f32")["image"]
    e = relerr(img, ref)
    print("[parity] Si[001] 11k CLI (1000^2 x 205 slices): E(gpu vs cpu_f32) =", e, "contrast", ref.std() / ref.mean())
    assert ref.std() / ref.mean() > 1e-2
    assert e < 5e-5


def test_c_host_program_runs_a_simulation(engine, tmp_path):
    """The C-ABI without Python in the compute path: tests/abi_c/host_check.c (plain C99) reads a .cnf, calls
    fdes_build_measurements and writes the image stack; the result equals the same call through the ctypes binding."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(G), ".."))
    libdir = os.path.join(root, "fdes_amd", "csrc")
    exe = str(tmp_path / "host_check")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "abi_c", "host_check.c"),
                           "-o", exe, "-L", libdir, "-lFDES_SHARED_LIB", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    hp, at = S.case_tiny(m=256, m3=6, nz=2, frPh=2, n3=2, tilt=True)
    fdes_amd.consistent(hp)
    fdes_amd.write_cnf(tmp_path / "case.cnf", hp, at)
    r = subprocess.run([exe, str(tmp_path / "case.cnf"), str(tmp_path / "image.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    hq, aq = fdes_amd.read_cnf(tmp_path / "case.cnf")
    ref = engine.build_measurements(hq, aq)["image"]
    img = np.fromfile(tmp_path / "image.bin", np.float32).reshape(ref.shape)
    assert np.array_equal(img, ref)


def test_multi_gpu_driver_in_one_process(engine, tmp_path):
    """fdes_build_measurements_multi (C++ host threads, one per device): three 'GPUs' that are all device 0 must give
    the single-GPU images up to the association order of the per-measurement intensity sum; also through the CLI
    (FDES_DEVICES)."""
    import subprocess
    hp, at = S.case_tiny(m=256, m3=6, nz=2, frPh=4, n3=3, tilt=True, pD=0.0)
    fdes_amd.consistent(hp)
    ref = engine.build_measurements(hp, at)["image"]
    for devs in ([0, 0], [0, 0, 0], [0, 0, 0, 0, 0]):
        out = fdes_amd.build_measurements_multi(devs, hp, at)
        e = relerr(out, ref)
        print(f"[parity] multi-GPU driver with {len(devs)} workers vs single: {e:.3e}")
        assert e < 2e-6 and np.isfinite(out).all()
    fdes_amd.write_cnf(tmp_path / "case.cnf", hp, at)
    exe = os.path.abspath(os.path.join(os.path.dirname(G), "..", "fdes_amd", "csrc", "FDES"))
    env = dict(os.environ, FDES_DEVICES="0,0,0", FDES_STRICT_CNF="1")
    r = subprocess.run([exe, "--input_name", "case.cnf", "--image_name", "m.bin", "--emd_name", "r.emd"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "3 GPUs" in r.stderr, r.stderr[-1500:]
    img = np.fromfile(tmp_path / "m.bin", np.float32).reshape(ref.shape)
    assert relerr(img, ref) < 2e-6


@pytest.mark.parametrize("seed", list(range(14)))
def test_randomised_parameter_sweep(engine, oracle, seed):
    """Random draws over the parameter surface (mode, grid size / path, species, sub-slicing, odd and even slice counts,
    empty slices, specimen and beam tilts, several measurements, frozen phonons, rectangular grids): images against the
    float64 oracle."""
    rng = np.random.default_rng(1000 + seed)
    fused = bool(rng.integers(0, 2))
    rect = bool(rng.integers(0, 4) == 0)
    kw = dict(m=int(rng.choice([256, 512])) if fused else int(rng.choice([48, 64, 96])),
              m3=int(rng.integers(1, 8)), nz=int(rng.integers(1, 4)), frPh=int(rng.choice([0, 0, 2, 3])),
              mode=int(rng.choice([0, 0, 1, 2])), n3=int(rng.integers(1, 3)), seed=int(rng.integers(0, 1000)),
              tilt=bool(rng.integers(0, 2)), beam_tilt=bool(rng.integers(0, 2)), imPot=float(rng.choice([0.0, 0.05, 0.2])),
              rect=rect, nat=int(rng.integers(1, 120)), sub=int(rng.integers(1, 4)), zfrac=float(rng.choice([0.5, 0.3, 0.15])))
    hp, at = S.case_tiny(**kw)
    fdes_amd.consistent(hp)
    out = engine.build_measurements(hp, at)["image"]
    ref = oracle.build_measurements(hp, at, prec="f64")["image"]
    check(out, ref, None, 2e-5, f"sweep {seed}: {kw}")


def test_shipped_qsc_example_through_the_cli(oracle, tmp_path):
    """The reference's bin/test.qsc as shipped: SrTiO3 9x9x20 cells (8 100 atoms, three species), nx = 400 -> 800^2 wave
    (mixed-radix fused passes), 40 slices cut into 400 sub-slices, CBED probe (cal_mode 2), pixel dose 10 -> Poisson surrogate noise;
    FDES CLI against the float32 oracle with the same Philox streams."""
    import subprocess
    src = os.path.join(G, "qsc", "test.qsc")
    exe = os.path.abspath(os.path.join(os.path.dirname(G), "..", "fdes_amd", "csrc", "FDES"))
    r = subprocess.run([exe, "--input_name", src, "--image_name", "Measurements.bin", "--emd_name", "results.emd"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Wave 800 x 800, slice loop: fused LDS passes" in r.stderr  # round 3: mixed-radix passes, fft_backend == 2
    hp, at = fdes_amd.read_qsc(src)
    assert (hp.c.m1, hp.c.m3, hp.c.mode, at.n) == (800, 40, 2, 8100)
    img = np.fromfile(tmp_path / "Measurements.bin", np.float32).reshape(1, 400, 400)
    ref = oracle.build_measurements(hp, at, prec="f32")["image"]
    close = np.abs(img - ref) < 1e-4 * ref.max()
    print("[parity] test.qsc CLI (800^2, 400 sub-slices, CBED, dose 10): fraction of pixels equal to rounding:", close.mean(),
          "mean", img.mean(), ref.mean())
    assert close.mean() > 0.97
    assert abs(img.mean() - ref.mean()) < 2e-3 * abs(ref.mean())


def test_c3_headline_size_single_configuration(oracle):
    """The headline workload itself (C3: Au cuboctahedron, 94 611 atoms, 2048^2 wave, 256 slices; one configuration,
    frozen phonons off so that the float64 oracle is the truth): SURVEY 8c's acceptance rule E(gpu) <= 2 E(cpu_f32) and
    E(gpu) <= 1e-4, for the full sequence on every slice and for the engine default (runs of empty slices as one
    Fresnel step with P^n)."""
    hp, at = S.case_c3(frPh=0)
    fdes_amd.consistent(hp)
    outs = {}
    for skip in (0, 1):
        eng = fdes_amd.Engine(0, skip_empty=skip)
        outs[skip] = eng.build_measurements(hp, at)["image"]
        eng.close()
    r64 = oracle.build_measurements(hp, at, prec="f64")["image"]
    # E(cpu_f32): the float32 oracle's own distance from the truth is a property of the oracle and the specimen (1.84e-5 in every
    # run of rounds 2-5); the default suite uses that recorded value, FDES_GPU_SUITE=full recomputes it (20 s of CPU time)
    e32 = 1.84e-5
    if os.environ.get("FDES_GPU_SUITE") == "full":
        e32 = relerr(oracle.build_measurements(hp, at, prec="f32")["image"], r64)
    for skip in (0, 1):
        e = relerr(outs[skip], r64)
        print(f"[parity] C3 2048^2 x 256 slices, skip_empty={skip}: E(gpu)={e:.3e} E(cpu_f32)={e32:.3e}")
        assert e <= 2 * e32 and e <= 1e-4
        assert np.abs(outs[skip] - r64).max() <= 1e-3 * np.abs(r64).max()
    assert r64.std() / r64.mean() > 0.05
