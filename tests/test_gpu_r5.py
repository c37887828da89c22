"""GPU tests added in round 5."""
import numpy as np
import pytest

import fdes_amd
from tests import specimens as S
from tests.test_gpu_parity import check, relerr
from tests.conftest import full_only

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


# ------------------------------------------------------------------------------------------------------------------
# run-time compilation of the mixed-radix row passes for a grid length without compiled-in kernels (gen_jit.cpp)
# ------------------------------------------------------------------------------------------------------------------

def test_run_time_compiled_kernels_against_oracle(oracle, tmp_path, monkeypatch):
    """cufftPlan2d serves any size alike (src/paramStructure.cu:676-679).  A grid length without compiled-in kernels (1100 =
    2 nx of a .qsc with nx = 550) gets the compile-time form of its row passes from hipRTC at plan creation: both axes report
    such kernels, the slice loop (one and two species, with and without the empty-slice short cut) meets the float64 oracle,
    the code object lands in the directory cache, and the run-time-length kernels (jit = 0) give the same physics."""
    monkeypatch.setenv("FDES_JIT_CACHE", str(tmp_path / "jit"))
    import os
    sizes = ((1100, 1), (572, 2)) if os.environ.get("FDES_GPU_SUITE") == "full" else ((572, 2),)   # (1100^2: tests/test_gpu_r3.py runs its slice loop on the run-time-length kernels)
    for m, nz in sizes:
        hp, at = S.case_tiny(m=m, m3=5, nz=nz, nat=300, tilt=True, seed=51 + nz)
        fdes_amd.consistent(hp)
        q, _ = oracle.sub_sliced(hp)
        ref = oracle.wave(q, at, 0, 0, prec="f64")
        r32 = oracle.wave(q, at, 0, 0, prec="f32")
        waves = {}
        for jit in (1, 0):
            for skip in (0, 1):
                eng = fdes_amd.Engine(0, skip_empty=skip, jit=jit)
                pl = eng.plan(hp, at)
                assert pl.fft_backend() == 2 and pl.jit_kernels() == (2 if jit else 0)
                psi = pl.tap_wave(0, 0)
                check(psi, ref, r32, 1e-5, f"run-time-compiled kernels {m}^2 nz={nz} jit={jit} skip_empty={skip}")
                if skip == 0:
                    waves[jit] = psi
                    if jit:
                        xyz = oracle.config_coords(q, at, 0, -1)
                        V = pl.tap_potential(0, 0, 2)
                        check(V, oracle.phase_grating(q, at, xyz, 2, "f64"), None, 1e-5, f"run-time-compiled kernels, potential s=2 {m}^2 nz={nz}")
                pl.close()
                eng.close()
        assert relerr(waves[1], waves[0]) < 2e-6
    cached = sorted(p.name for p in (tmp_path / "jit").iterdir())
    assert len(cached) == len(sizes) and cached[-1].startswith("gpass_572_"), cached
    assert all((tmp_path / "jit" / n).read_bytes()[:8] == b"FDESJIT1" and (tmp_path / "jit" / n).read_bytes()[24:28] == b"\x7fELF" for n in cached)


def test_run_time_compiled_kernels_whole_driver_and_mixed_axes(oracle, tmp_path, monkeypatch):
    """Whole driver (lanes, graph replay, gangs, detector chain) on run-time-compiled kernels: a frozen-phonon image at 660^2
    (660 = 4 * 3 * 5 * 11) against the float32 oracle; a rectangular grid with ONE such axis (2288 x 1024: a two-row-tile length
    beside a power of two) reports one compiled axis and meets the float64 oracle; a length with compiled-in kernels (1000) and a
    power of two report none; the FFT alone at 2288 x 1716 (two compiled lengths) against numpy."""
    monkeypatch.setenv("FDES_JIT_CACHE", str(tmp_path / "jit"))
    hp, at = S.case_tiny(m=660, m3=6, nz=2, frPh=3, nat=200, tilt=True)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, jit=1)
    pl = eng.plan(hp, at)
    assert pl.jit_kernels() == 2
    pl.close()
    img = eng.build_measurements(hp, at)["image"]
    eng.close()
    e = relerr(img, oracle.build_measurements(hp, at, prec="f32")["image"])
    print(f"[parity] run-time-compiled kernels, driver 660^2 frPh=3: E = {e:.3e}")
    assert e <= 1e-5
    hp, at = S.case_tiny(m=2288, m2=1024, m3=3, nz=1, nat=200, tilt=True, seed=5)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, jit=1)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 2 and pl.jit_kernels() == 1
    pl.close()
    out = eng.build_measurements(hp, at)["image"]
    check(out, oracle.build_measurements(hp, at, prec="f64")["image"], None, 1e-5, "run-time-compiled kernels on one axis, 2288 x 1024")
    for m in (1000, 2048):
        hp2, at2 = S.case_tiny(m=m, m3=2, nz=1, nat=20)
        fdes_amd.consistent(hp2)
        pl = eng.plan(hp2, at2)
        assert pl.jit_kernels() == 0
        pl.close()
    rng = np.random.default_rng(7)
    f = (rng.standard_normal((1716, 2288)) + 1j * rng.standard_normal((1716, 2288))).astype(np.complex64)
    for inv in (False, True):
        o, used = eng.fft2(f, inv, backend=2)
        assert used == 2
        ref = np.fft.ifft2(f.astype(np.complex128)) * f.size if inv else np.fft.fft2(f.astype(np.complex128))
        assert relerr(o, ref) < 6e-7
    eng.close()


@pytest.mark.parametrize("m,nz,jit", [(750, 2, 0), pytest.param(750, 1, 1, marks=full_only), pytest.param(500, 1, 0, marks=full_only), (500, 2, 1), pytest.param(1250, 1, 1, marks=full_only),
                                      (375, 2, 1), (1001, 1, 0), pytest.param(625, 1, 1, marks=full_only), pytest.param(1125, 2, 1, marks=full_only)])   # odd lengths: 375 = 3 * 5^3, 1001 = 7 * 11 * 13, 1125 = 9 * 125
def test_grids_that_the_tile_rows_do_not_divide(oracle, tmp_path, monkeypatch, m, nz, jit):
    """A mixed-radix row length has tiles of 8 (up to 512 points), 4 (up to 2048) or 2 rows; until round 5 a grid whose other
    dimension that number does not divide left the fused loop (750^2 and 1250^2: m = 2 nx of a .qsc with an odd nx,
    src/rwQsc.cu:943-948; 500^2; every odd length).  Such grids now run smaller tiles (gen_pass_tile_rows: 750 -> 2 rows, 500 -> 4)
    or - odd row counts, which no power-of-two tile divides - the length's own tiles with a PARTIAL last one (k_gpass: rvalid; its
    missing rows are neither read behind the grid nor stored), on the run-time-length kernels resp. on kernels compiled for the
    length and the tile rows: slice loop and potential against the float64 oracle, with and without the empty-slice short cut."""
    monkeypatch.setenv("FDES_JIT_CACHE", str(tmp_path / "jit"))
    hp, at = S.case_tiny(m=m, m3=5, nz=nz, nat=300, tilt=True, seed=71 + nz)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    r32 = oracle.wave(q, at, 0, 0, prec="f32")
    assert fdes_amd.load_library().fdes_grid_backend(m, m, 0) == 2
    for skip in (0, 1):
        eng = fdes_amd.Engine(0, skip_empty=skip, jit=jit)
        pl = eng.plan(hp, at)
        assert pl.fft_backend() == 2 and pl.jit_kernels() == (2 if jit else 0)
        psi = pl.tap_wave(0, 0)
        check(psi, ref, r32, 1e-5, f"{m}^2 with a partial last tile, nz={nz} jit={jit} skip_empty={skip}")
        if skip == 0:
            xyz = oracle.config_coords(q, at, 0, -1)
            V = pl.tap_potential(0, 0, 3)
            check(V, oracle.phase_grating(q, at, xyz, 3, "f64"), None, 1e-5, f"{m}^2 with a partial last tile, potential s=3 nz={nz} jit={jit}")
        pl.close()
        eng.close()
    if jit:
        assert any(p.name.startswith(f"gpass_{m}") for p in (tmp_path / "jit").iterdir())


def test_smaller_tiles_whole_driver_and_fft(oracle, engine):
    """Frozen-phonon image through the whole driver (lanes, gangs, graph replay) on a 750^2 grid (two-row tiles) against the
    float32 oracle; the 2-D FFT alone at 1430 x 2002 (neither length's four-row tiles divide the other dimension) against numpy."""
    hp, at = S.case_tiny(m=750, m3=6, nz=2, frPh=3, nat=200, tilt=True)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0)
    img = eng.build_measurements(hp, at)["image"]
    eng.close()
    e = relerr(img, oracle.build_measurements(hp, at, prec="f32")["image"])
    print(f"[parity] driver 750^2 frPh=3 on two-row tiles: E = {e:.3e}")
    assert e <= 1e-5
    rng = np.random.default_rng(3)
    f = (rng.standard_normal((2002, 1430)) + 1j * rng.standard_normal((2002, 1430))).astype(np.complex64)
    for inv in (False, True):
        o, used = engine.fft2(f, inv, backend=2)
        assert used == 2
        ref = np.fft.ifft2(f.astype(np.complex128)) * f.size if inv else np.fft.fft2(f.astype(np.complex128))
        assert relerr(o, ref) < 6e-7


_JIT_CACHE_PROBE = r'''
import sys
sys.path.insert(0, sys.argv[1])
import fdes_amd
from tests import specimens as S
hp, at = S.case_tiny(m=572, m3=2, nz=1, nat=20)
fdes_amd.consistent(hp)
eng = fdes_amd.Engine(0, jit=1)
pl = eng.plan(hp, at)
print("axes", pl.jit_kernels())
pl.close(); eng.close()
'''


def test_run_time_compilation_cache_survives_a_damaged_file(tmp_path):
    """The code objects of the run-time compilation live in a directory cache.  A second process reads what the first one wrote
    (no compilation: the file is untouched); a file cut short (a full disk) fails its length / checksum header,
    never reaches the module loader (which does not survive a truncated code object), and is compiled and cached again."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDES_JIT_CACHE=str(tmp_path / "jit"), FDES_JIT="1")
    def run():
        r = subprocess.run([sys.executable, "-c", _JIT_CACHE_PROBE, root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "axes 2" in r.stdout, r.stdout + r.stderr
    run()
    files = list((tmp_path / "jit").iterdir())
    assert len(files) == 1 and files[0].stat().st_size > 10000
    stamp = files[0].stat().st_mtime_ns
    run()
    assert files[0].stat().st_mtime_ns == stamp          # read, not rewritten
    files[0].write_bytes(files[0].read_bytes()[:4096])
    run()
    assert files[0].stat().st_size > 10000               # dropped, compiled again, cached again


@pytest.mark.parametrize("m1,m2", [(8192, 256), pytest.param(256, 6144, marks=full_only), (4800, 500), pytest.param(5000, 5000, marks=full_only)])
def test_rows_beyond_4096_points(oracle, tmp_path, monkeypatch, m1, m2):
    """Grid lengths of 4098 ... 8192 points (even, 2^a 3^b 5^c 7^d 11^e 13^f; cufftPlan2d serves any size alike,
    src/paramStructure.cu:676-679) run the fused loop on kernels compiled at plan creation - one tile image of two rows, three
    or four stages (8192 = 8 x 16 x 16 x 4) - instead of the rocFFT + point-wise loop; with jit = 0 they take rocFFT as before.
    Rectangular grids keep the oracle cheap: 2-D FFT against numpy, slice loop and potential against the float64 oracle."""
    monkeypatch.setenv("FDES_JIT_CACHE", str(tmp_path / "jit"))
    rng = np.random.default_rng(11)
    f = (rng.standard_normal((m2, m1)) + 1j * rng.standard_normal((m2, m1))).astype(np.complex64)
    eng = fdes_amd.Engine(0, jit=1)
    for inv in (False, True):
        o, used = eng.fft2(f, inv, backend=0)
        assert used == 2
        ref = np.fft.ifft2(f.astype(np.complex128)) * f.size if inv else np.fft.fft2(f.astype(np.complex128))
        e = relerr(o, ref)
        print(f"[parity] fft {m1} x {m2} inv={inv}: rel L2 {e:.3e}")
        assert e < 7e-7
    eng.close()
    hp, at = S.case_tiny(m=m1, m2=m2, m3=3, nz=2, nat=150, tilt=True, seed=23)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    r32 = oracle.wave(q, at, 0, 0, prec="f32")
    for skip in (0, 1):
        eng = fdes_amd.Engine(0, skip_empty=skip, jit=1)
        pl = eng.plan(hp, at)
        assert pl.fft_backend() == 2 and pl.jit_kernels() >= 1
        psi = pl.tap_wave(0, 0)
        check(psi, ref, r32, 1e-5, f"rows beyond 4096 points, {m1} x {m2} skip_empty={skip}")
        if skip == 0:
            xyz = oracle.config_coords(q, at, 0, -1)
            for sl in range(q.c.m3):   # the first slice that holds atoms
                Vref = oracle.phase_grating(q, at, xyz, sl, "f64")
                if np.abs(Vref).max() > 0:
                    check(pl.tap_potential(0, 0, sl), Vref, None, 1e-5, f"rows beyond 4096 points, potential s={sl} {m1} x {m2}")
                    break
            else:
                raise AssertionError("no slice with atoms")
        pl.close()
        eng.close()
    eng = fdes_amd.Engine(0, jit=0)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 1 and pl.jit_kernels() == 0      # no kernels without the compilation: rocFFT, as every unsupported size
    pl.close()
    eng.close()


def test_multi_gpu_driver_on_run_time_compiled_kernels(oracle, tmp_path, monkeypatch):
    """fdes_build_measurements_multi with three host threads (all on GPU 0 here) on a grid length whose kernels are compiled at
    plan creation: the workers meet in the compilation (one compiles, the others take its code object and load the module - the
    per-device lock and the compilation lock are taken in that order everywhere), and the summed image equals the oracle's."""
    monkeypatch.setenv("FDES_JIT_CACHE", str(tmp_path / "jit"))
    monkeypatch.setenv("FDES_JIT", "1")
    hp, at = S.case_tiny(m=616, m3=4, nz=2, frPh=5, nat=150, n3=2, tilt=True, seed=9)   # 616 = 8 * 7 * 11: a length no other test of this process compiles
    fdes_amd.consistent(hp)
    img = fdes_amd.build_measurements_multi([0, 0, 0], hp, at)
    e = relerr(img, oracle.build_measurements(hp, at, prec="f32")["image"])
    print(f"[parity] three workers, 616^2 on run-time-compiled kernels: E = {e:.3e}")
    assert len(list((tmp_path / "jit").iterdir())) == 1   # compiled once for the three workers
    assert e <= 1e-5


_JIT_STAGES_PROBE = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import fdes_amd
from tests import specimens as S, oracle_py
from tests.test_gpu_parity import relerr
oracle_py.lib()
hp, at = S.case_tiny(m=1100, m3=3, nz=2, nat=200, tilt=True, seed=4)
fdes_amd.consistent(hp)
q, _ = oracle_py.sub_sliced(hp)
ref = oracle_py.wave(q, at, 0, 0, prec="f64")
eng = fdes_amd.Engine(0, jit=1, skip_empty=0)
pl = eng.plan(hp, at)
print("axes", pl.jit_kernels(), "E", relerr(pl.tap_wave(0, 0), ref))
pl.close(); eng.close()
'''


@pytest.mark.parametrize("stages", [pytest.param("10,10,11", marks=full_only), pytest.param("5,20,11", marks=full_only), "11,4,5,5"])
def test_run_time_compilation_with_given_stage_orders(tmp_path, stages):
    """FDES_JIT_STAGES (a tuning knob) gives the stage radices of a compilation outright: any order of supported radices whose
    product is the length - three or four stages, prime and composite radices anywhere - must give the same physics (1100^2 slice
    loop against the float64 oracle, in a process of its own: the knob is read when a length is first compiled)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDES_JIT_CACHE=str(tmp_path / "jit"), FDES_JIT="1", FDES_JIT_STAGES=stages)
    r = subprocess.run([sys.executable, "-c", _JIT_STAGES_PROBE, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "axes 2" in r.stdout, r.stdout + r.stderr
    e = float(r.stdout.split("E")[-1])
    print(f"[parity] 1100^2 slice loop with stages {stages}: E = {e:.3e}")
    assert e <= 1e-5


@pytest.mark.parametrize("m,nz", [(1088, 1), pytest.param(608, 2, marks=full_only), (475, 2), pytest.param(736, 1, marks=full_only)])   # (475 = 19 * 25: up to 512 points the stages run between two tile images - a dispatch of its own, which the stress sweep found without these radices)
def test_lengths_with_factors_17_19_23(oracle, tmp_path, monkeypatch, m, nz):
    """1088 = 64 * 17, 608 = 32 * 19, 736 = 32 * 23: radices 17, 19 and 23 exist in the compile-time kernels only, so these
    lengths run the fused loop on kernels compiled at plan creation (as the rows beyond 4096 points) and rocFFT with jit = 0:
    2-D FFT against numpy, slice loop and potential against the float64 oracle."""
    monkeypatch.setenv("FDES_JIT_CACHE", str(tmp_path / "jit"))
    rng = np.random.default_rng(17)
    f = (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))).astype(np.complex64)
    eng = fdes_amd.Engine(0, jit=1)
    for inv in (False, True):
        o, used = eng.fft2(f, inv, backend=0)
        assert used == 2
        ref = np.fft.ifft2(f.astype(np.complex128)) * f.size if inv else np.fft.fft2(f.astype(np.complex128))
        e = relerr(o, ref)
        print(f"[parity] fft {m}^2 inv={inv}: rel L2 {e:.3e}")
        assert e < 7e-7
    eng.close()
    hp, at = S.case_tiny(m=m, m3=4, nz=nz, nat=200, tilt=True, seed=19)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    r32 = oracle.wave(q, at, 0, 0, prec="f32")
    for skip in (0, 1):
        eng = fdes_amd.Engine(0, skip_empty=skip, jit=1)
        pl = eng.plan(hp, at)
        assert pl.fft_backend() == 2 and pl.jit_kernels() == 2
        check(pl.tap_wave(0, 0), ref, r32, 1e-5, f"{m}^2 (radix {17 if m % 17 == 0 else (19 if m % 19 == 0 else 23)}) skip_empty={skip}")
        if skip == 0:
            xyz = oracle.config_coords(q, at, 0, -1)
            for sl in range(q.c.m3):
                Vref = oracle.phase_grating(q, at, xyz, sl, "f64")
                if np.abs(Vref).max() > 0:
                    check(pl.tap_potential(0, 0, sl), Vref, None, 1e-5, f"{m}^2 potential s={sl}")
                    break
        pl.close()
        eng.close()
    eng = fdes_amd.Engine(0, jit=0)
    pl = eng.plan(hp, at)
    assert pl.fft_backend() == 1
    pl.close()
    eng.close()
