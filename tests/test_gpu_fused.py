"""GPU tests of the hand-written LDS FFT passes and the fused slice loop (power-of-two grids >= 256)."""
import numpy as np
import pytest

import fdes_amd
from tests import specimens as S
from tests.test_gpu_parity import check, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(256, 256), (512, 512), (1024, 1024), (2048, 2048), (4096, 4096), (256, 512), (1024, 256)])
def test_lds_fft_against_numpy(engine, shape):
    rng = np.random.default_rng(7)
    f = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    for inv in (False, True):
        out, used = engine.fft2(f, inv, backend=2)
        assert used == 2
        ref = np.fft.ifft2(f.astype(np.complex128)) * f.size if inv else np.fft.fft2(f.astype(np.complex128))
        e = relerr(out, ref)
        print(f"[parity] lds fft {shape} inv={inv}: rel L2 {e:.3e}")
        assert e < 5e-7


@pytest.mark.parametrize("kw", [dict(m=256, nz=1), dict(m=256, nz=3), dict(m=512, nz=2, rect=True)])
def test_fused_potential(engine, oracle, kw):
    hp, at = S.case_tiny(m3=3, tilt=True, nat=120, **kw)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    xyz = oracle.config_coords(q, at, 0, -1)
    pl = engine.plan(hp, at)
    for s in range(q.c.m3):
        V = pl.tap_potential(0, 0, s)
        ref = oracle.phase_grating(q, at, xyz, s, "f64")
        check(V, ref, oracle.phase_grating(q, at, xyz, s, "f32"), 1e-5, f"fused potential {kw} s={s}")
    pl.close()


@pytest.mark.parametrize("kw", [dict(m=256, m3=6, nz=1), dict(m=256, m3=4, nz=3, tilt=True, beam_tilt=True, n3=2),
                                dict(m=512, m3=4, nz=2, rect=True, frPh=2), dict(m=256, m3=3, nz=2, mode=2),
                                dict(m=1024, m3=3, nz=1, nat=400)])
def test_fused_exit_wave(engine, oracle, kw):
    hp, at = S.case_tiny(**{"nat": 150, **kw})
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    pl = engine.plan(hp, at)
    k = q.c.n3 - 1
    j = 1 if q.c.frPh > 0 else 0
    psi = pl.tap_wave(k, j)
    ref = oracle.wave(q, at, k, j, prec="f64")
    check(psi, ref, oracle.wave(q, at, k, j, prec="f32"), 1e-5, f"fused exit wave {kw}")
    pl.close()


def test_fused_equals_rocfft_path(oracle):
    """Same inputs through the generic (rocFFT + point-wise kernels) and the fused LDS-pass slice loops."""
    hp, at = S.case_tiny(m=256, m3=8, nz=2, frPh=2, nat=300, tilt=True)
    fdes_amd.consistent(hp)
    imgs = {}
    for fft in (1, 2):
        eng = fdes_amd.Engine(0, fft=fft)
        imgs[fft] = eng.build_measurements(hp, at, want_exitwave=True)
        eng.close()
    ref = oracle.build_measurements(hp, at, prec="f64", want_exitwave=True)
    check(imgs[2]["image"], ref["image"], imgs[1]["image"], 1e-5, "fused image (f32 column = rocFFT path)")
    check(imgs[2]["exitwave"], ref["exitwave"], imgs[1]["exitwave"], 1e-5, "fused exit-wave stack")


def test_c1_reference_sized_case(engine, oracle):
    """BASELINE config 0 (C1: SrTiO3, 256^2, 8 slices, 3 species): image parity."""
    hp, at = S.case_c1()
    fdes_amd.consistent(hp)
    img = engine.build_measurements(hp, at)["image"]
    ref = oracle.build_measurements(hp, at, prec="f64")["image"]
    check(img, ref, oracle.build_measurements(hp, at, prec="f32")["image"], 1e-5, "C1 image")


def test_empty_slice_fast_path(oracle):
    """Slices without atoms: with skip_empty the engine only applies the Fresnel step (t = 1 exactly); both settings
    must agree with the oracle, which always runs the full sequence."""
    hp, at = S.case_tiny(m=256, m3=24, nz=2, nat=60, zfrac=0.12, frPh=2)
    fdes_amd.consistent(hp)
    ref = oracle.build_measurements(hp, at, prec="f64", want_exitwave=True)
    outs = {}
    for skip, graph in ((0, 1), (1, 1), (1, 0)):   # runs of empty slices become one step with P^n, with and without graphs
        eng = fdes_amd.Engine(0, skip_empty=skip)
        eng.set_option("graph", graph)
        outs[(skip, graph)] = eng.build_measurements(hp, at, want_exitwave=True)
        eng.close()
        check(outs[(skip, graph)]["exitwave"], ref["exitwave"], None, 1e-5, f"exit wave, skip_empty={skip} graph={graph}")
        check(outs[(skip, graph)]["image"], ref["image"], None, 1e-5, f"image, skip_empty={skip} graph={graph}")
    print("[parity] skip vs no-skip:", relerr(outs[(1, 1)]["exitwave"], outs[(0, 1)]["exitwave"].astype(np.float64)))
    assert np.array_equal(outs[(1, 1)]["image"], outs[(1, 0)]["image"])


@pytest.mark.parametrize("m", [256, 1024])
def test_propagation_unit_against_the_oracle(engine, oracle, m):
    """The stand-alone propagation unit psi <- F^-1[P F[t psi]] (three fused row passes behind fdes_plan_propagate_dev)
    on a random wave and a random unit-modulus transmission function, against the float64 oracle
    (multiplyElementwise + convolveWithFrProp, src/multisliceSimulation.cu:546-548)."""
    import torch
    hp, at = S.case_tiny(m=m, m3=2, nz=1, nat=4)
    fdes_amd.consistent(hp)
    q, _ = oracle.sub_sliced(oracle.consistent(hp.copy()))
    rng = np.random.default_rng(5)
    psi = (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))).astype(np.complex64)
    t = np.exp(1j * rng.uniform(-np.pi, np.pi, (m, m))).astype(np.complex64)
    P = oracle.fresnel_propagator(q, prec="f64")
    ref = oracle.propagate_unit(q, psi.astype(np.complex128), t.astype(np.complex128), P, prec="f64")
    pl = engine.plan(hp, at)
    assert pl.fft_backend() == 2
    d_psi = torch.view_as_real(torch.from_numpy(psi)).contiguous().cuda()
    d_t = torch.view_as_real(torch.from_numpy(t)).contiguous().cuda()
    torch.cuda.synchronize()
    pl.propagate_dev(d_psi.data_ptr(), d_t.data_ptr())
    pl.sync()
    out = torch.view_as_complex(d_psi.cpu()).numpy()
    check(out, ref, None, 2e-6, f"propagation unit {m}^2")
    pl.close()
