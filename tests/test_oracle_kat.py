"""CPU tests that pin the oracle: reference-produced known answers, published RNG vectors, an
independent FFT, and the committed golden fixtures."""
import os

import numpy as np
import pytest

from tests import specimens as S

G = os.path.join(os.path.dirname(__file__), "golden")


def test_consistent_params_kat_200kV(oracle):
    # src/paramStructure.cu:509-512
    hp = oracle.consistent(S.make_params(1, E0=200e3))
    assert np.float32(hp.c.gamma) == np.float32(1.3913902)
    assert np.float32(hp.c.lambda_) == np.float32(2.507934e-12)
    assert np.float32(hp.c.sigma) == np.float32(7288400.5)


def test_consistent_params_kat_50kV(oracle):
    # attributes of ExampleSpecimens/Au_cubeoctahedron_emd/Auparticle.emd (/microscope/*)
    hp = oracle.consistent(S.make_params(1, E0=50e3))
    assert np.float32(hp.c.gamma) == np.float32(1.09784758)
    assert np.float32(hp.c.lambda_) == np.float32(5.35530691e-12)
    assert abs(float(hp.c.sigma) - 12279866.0) <= 1.0  # 1 ulp at this magnitude


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_normal_moments(oracle):
    x = np.array([oracle.normal(1, 0, 2, 3, i) for i in range(100000)])
    assert abs(x.mean()) < 0.02 and abs(x.var() - 1) < 0.02
    assert abs(((x - x.mean()) ** 4).mean() / x.var() ** 2 - 3) < 0.1


@pytest.mark.parametrize("shape", [(8, 8), (12, 20), (64, 64), (30, 50), (125, 81), (14, 22), (320, 320)])
def test_fft_against_numpy(oracle, shape):
    rng = np.random.default_rng(1)
    f = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    for inv in (False, True):
        ref = np.fft.ifft2(f) * f.size if inv else np.fft.fft2(f)
        assert np.abs(oracle.fft2(f, inv, "f64") - ref).max() / np.abs(ref).max() < 1e-13
        assert np.abs(oracle.fft2(f, inv, "f32") - ref).max() / np.abs(ref).max() < 2e-6


def test_kirkland_table_fixture():
    tab = np.fromfile(os.path.join(G, "kirkland_table.bin"), np.float32).reshape(104, 12)
    assert np.all(tab[0] == np.array([0, 1, 0, 1, 0, 1, 1, 0, 1, 0, 1, 0], np.float32))  # projectedPotential.cu:2985-3009
    # Z = 1 (src/projectedPotential.cu:100-124) and Z = 103 (:2956-2982) as spelled in the reference
    assert np.allclose(tab[1], [4.20298334e-3, 2.25350887e-1, 6.27762526e-2, 2.25366950e-1, 3.00907344e-2, 2.25331753e-1,
                                6.77756667e-2, 4.38853979, 3.56609235e-3, 4.03884828e-1, 2.76135821e-2, 1.44490170], rtol=1e-7)
    assert np.allclose(tab[103], [4.86738014, 1.60320511e1, 3.19974393e-1, 6.70871139e-2, 4.58872414, 5.77039361e-1,
                                  1.21482447e-1, 7.22275898e-2, 2.31639862, 1.41279736e1, 3.79258126e-1, 3.89973491e-1], rtol=1e-7)
    # the two .inc copies used by the engine and by the oracle hold the same numbers
    root = os.path.dirname(os.path.dirname(__file__))
    a = open(os.path.join(root, "oracle", "kirkland_table.inc")).read()
    b = open(os.path.join(root, "fdes_amd", "csrc", "kirkland_table.inc")).read()
    assert a == b
    assert np.all(tab[1:] [:, [1, 3, 5]] > 0)


def test_au309_generator_matches_shipped_particle():
    # fixture = atom records of ExampleSpecimens/Au_cubeoctahedron_cnf/dataFDES_Auparticle.cnf
    ref = np.load(os.path.join(G, "au309_atoms.npy"))
    gen = S.au_cuboctahedron(4)
    assert ref.shape == (309, 6) and gen.shape == (309, 3)
    key = lambda a: sorted(map(tuple, np.round(a / S.A_AU).astype(int)))
    assert key(ref[:, 1:4]) == key(gen)
    assert np.all(ref[:, 0] == 79)


def test_single_atom_potential_vs_kirkland_closed_form(oracle):
    """Independent physics check of phaseGrating: Kirkland (2009) eq. C.20 real-space projected potential."""
    from scipy.special import k0
    from fdes_amd.abi import HostAtoms
    m, d = 256, 0.1e-10
    hp = oracle.consistent(S.make_params(1, E0=200e3, n1=m - 2, n2=m - 2, dn1=1, dn2=1, d1=d, d2=d, m3=1, d3=2e-10,
                                         subSlTh=2e-10, imPot=0.0))
    at = HostAtoms([79], [[0.5 * d, 0.5 * d, 0.0]], 6e-21, 1.0)
    V = oracle.phase_grating(hp, at, at.xyz, 0, "f64").real
    t = np.fromfile(os.path.join(G, "kirkland_table.bin"), np.float32).reshape(104, 12).astype(np.float64)[79]
    a, b, c, dd = t[0:6:2], t[1:6:2], t[6:12:2], t[7:12:2]
    a0e = 0.529177 * 14.39964
    for rpx in (3, 5, 8):
        r = rpx * d * 1e10
        vz = 4 * np.pi ** 2 * a0e * sum(a[i] * k0(2 * np.pi * r * np.sqrt(b[i])) for i in range(3)) + \
            2 * np.pi ** 2 * a0e * sum(c[i] / dd[i] * np.exp(-np.pi ** 2 * r ** 2 / dd[i]) for i in range(3))
        ref = vz * hp.c.sigma * 1e-10
        assert abs(V[m // 2, m // 2 + rpx] / ref - 1) < 0.05


def test_free_space_propagation_conserves_band_limited_norm(oracle):
    hp, at = S.case_tiny(m=64, m3=3, nz=1, nat=0)
    hp = oracle.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    P = oracle.fresnel_propagator(q, "f64")
    rng = np.random.default_rng(5)
    psi = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
    # band-limit psi first, then propagate with t = 1: |psi| must be conserved
    f = np.fft.fft2(psi)
    f[np.abs(P) == 0] = 0
    psi = np.fft.ifft2(f)
    out = oracle.propagate_unit(q, psi, np.ones_like(psi), P, "f64")
    assert abs(np.linalg.norm(out) / np.linalg.norm(psi) - 1) < 1e-12


def test_oracle_f32_close_to_f64_all_modes(oracle):
    for mode in (0, 1, 2):
        hp, at = S.case_tiny(m=64, m3=4, nz=3, mode=mode, n3=2, tilt=True, beam_tilt=(mode != 2))
        hp = oracle.consistent(hp)
        a = oracle.build_measurements(hp, at, prec="f32")["image"]
        b = oracle.build_measurements(hp, at, prec="f64")["image"]
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 5e-6


def test_golden_tiny_images(oracle):
    """The committed goldens (tools/make_golden.py) are reproduced by the oracle built here."""
    g = np.load(os.path.join(G, "tiny_cases.npz"))
    for name, kw in S.GOLDEN_CASES.items():
        hp, at = S.case_tiny(**kw)
        hp = oracle.consistent(hp)
        img = oracle.build_measurements(hp, at, prec="f64")["image"]
        ref = g[name + "_f64"]
        assert np.linalg.norm(img - ref) / np.linalg.norm(ref) < 1e-10, name
