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


def _kirkland(Z):
    t = np.fromfile(os.path.join(G, "kirkland_table.bin"), np.float32).reshape(104, 12).astype(np.float64)[Z]
    return t[0:6:2], t[1:6:2], t[6:12:2], t[7:12:2]   # a_i, b_i, c_i, d_i of Kirkland (2009) table C.1


def _single_atom_potential(oracle, Z, m, d, frac=(0.0, 0.0)):
    """sigma v_z of one atom whose centre falls `frac` pixels from pixel (m/2, m/2) (squareAtoms_d maps x to
    x / d + m/2 - 0.5, src/crystalMaker.cu:85)."""
    from fdes_amd.abi import HostAtoms
    hp = oracle.consistent(S.make_params(1, E0=200e3, n1=m - 2, n2=m - 2, dn1=1, dn2=1, d1=d, d2=d, m3=1, d3=2e-10,
                                         subSlTh=2e-10, imPot=0.0))
    at = HostAtoms([Z], [[(0.5 + frac[0]) * d, (0.5 + frac[1]) * d, 0.0]], 6e-21, 1.0)
    return hp, oracle.phase_grating(hp, at, at.xyz, 0, "f64").real


def test_single_atom_potential_vs_kirkland_closed_form(oracle):
    """Independent physics check of phaseGrating: Kirkland (2009) eq. C.20, the real-space projected potential.  The grid
    carries the Fourier components up to Nyquist only and the closed form has a logarithmic singularity at r = 0, so
    this comparison cannot be made tighter than a few per cent at 0.3-0.8 A; the tight checks are the next two tests."""
    from scipy.special import k0
    m, d = 256, 0.1e-10
    hp, V = _single_atom_potential(oracle, 79, m, d)
    a, b, c, dd = _kirkland(79)
    a0e = 0.529177 * 14.39964
    for rpx in (3, 5, 8):
        r = rpx * d * 1e10
        vz = 4 * np.pi ** 2 * a0e * sum(a[i] * k0(2 * np.pi * r * np.sqrt(b[i])) for i in range(3)) + \
            2 * np.pi ** 2 * a0e * sum(c[i] / dd[i] * np.exp(-np.pi ** 2 * r ** 2 / dd[i]) for i in range(3))
        ref = vz * hp.c.sigma * 1e-10
        assert abs(V[m // 2, m // 2 + rpx] / ref - 1) < 0.05


@pytest.mark.parametrize("Z", [6, 14, 38, 79])
def test_single_atom_potential_moments_from_first_principles(oracle, Z):
    """Two numbers that neither the band limit at Nyquist nor the pixel size can change, written from the physics and
    from Kirkland's table only (no constant of the reference's source):
      integral of v_z over the plane = h^2 / (2 pi m0 e) * f_e(0),       f_e(0) = sum a_i / b_i + sum c_i   (Born)
      <r^2> of v_z                   = -(laplacian_q F)(0) / (4 pi^2 F(0)), F = f_e(q) / (sinc(pi qx d) sinc(pi qy d)):
                                       (sum a_i / b_i^2 + sum c_i d_i) / (pi^2 f_e(0))  -  d^2 / 6
    (the second term is the pixel-wide top-hat that divideBySinc, src/crystalMaker.cu:136-158, takes OUT of the
    deposit: a de-convolution narrows).  Both to 2e-5 / 1e-4, on and off the pixel centre."""
    m, d = 256, 0.1e-10
    a, b, c, dd = _kirkland(Z)
    fe0 = (a / b).sum() + c.sum()                                   # Angstrom
    h, m0, e = 6.62607015e-34, 9.1093837015e-31, 1.602176634e-19
    for frac in ((0.0, 0.0), (0.37, -0.21), (0.5, 0.5)):
        hp, V = _single_atom_potential(oracle, Z, m, d, frac)
        total = V.sum() * d * d / hp.c.sigma                        # V m^2 * m  (sigma v_z / sigma, integrated)
        born = h * h / (2 * np.pi * m0 * e) * fe0 * 1e-10
        assert abs(total / born - 1) < 2e-5, (Z, frac, total / born)
        i2, i1 = np.indices(V.shape)
        x = (i1 - (m / 2 + frac[0])) * d
        y = (i2 - (m / 2 + frac[1])) * d
        x -= np.round(x / (m * d)) * (m * d)
        y -= np.round(y / (m * d)) * (m * d)
        r2 = (V * (x * x + y * y)).sum() / V.sum()
        want = ((a / b ** 2).sum() + (c * dd).sum()) / (np.pi ** 2 * fe0) * 1e-20 - d * d / 6
        # the bilinear deposit of an off-centre atom adds the variance of its two-point weights per axis: f (1 - f) d^2
        fx, fy = abs(frac[0]), abs(frac[1])
        want += (fx * (1 - fx) + fy * (1 - fy)) * d * d
        assert abs(r2 / want - 1) < 1e-4, (Z, frac, r2, want)


def test_single_atom_potential_vs_band_limited_fourier_sum(oracle):
    """The whole potential image of one atom against a direct evaluation, in numpy float64, of what the reference's method
    defines: Kirkland's scattering factor f_e(q) (eq. C.15, written out here) on the reciprocal points the grid carries,
    times the Born constant, divided by the transform of the pixel-wide top-hat, times the phase factors of the (up to
    four) pixels the bilinear deposit touches.  Agreement to 1e-6 of the peak; the constant h^2 / (2 pi m0 e) is taken
    from CODATA, not from the reference (it differs from the reference's 4.78776452e-19 by 7e-6)."""
    m, d = 128, 0.1e-10
    a, b, c, dd = _kirkland(79)
    h, m0, e = 6.62607015e-34, 9.1093837015e-31, 1.602176634e-19
    for frac in ((0.0, 0.0), (0.3, -0.4)):
        hp, V = _single_atom_potential(oracle, 79, m, d, frac)
        q1 = np.fft.fftfreq(m, d * 1e10)                            # 1 / Angstrom
        qx, qy = np.meshgrid(q1, q1)
        q2 = qx ** 2 + qy ** 2
        fe = sum(a[i] / (q2 + b[i]) + c[i] * np.exp(-dd[i] * q2) for i in range(3))
        ax, ay = np.pi * qx * d * 1e10, np.pi * qy * d * 1e10
        sinc = lambda t: np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t))
        F = fe / (sinc(ax) * sinc(ay))
        # bilinear deposit: weight (1 - |f|) on the nearest pixel and |f| on the neighbour towards the atom, per axis
        dep = np.zeros((m, m))
        for (oy, wy) in ((0, 1 - abs(frac[1])), (int(np.sign(frac[1])), abs(frac[1]))):
            for (ox, wx) in ((0, 1 - abs(frac[0])), (int(np.sign(frac[0])), abs(frac[0]))):
                dep[m // 2 + oy, m // 2 + ox] += wx * wy
        ref = np.fft.ifft2(np.fft.fft2(dep) * F).real * (h * h / (2 * np.pi * m0 * e)) * 1e-10 * hp.c.sigma / (d * d)
        err = np.abs(V - ref).max() / np.abs(ref).max()
        assert err < 2e-5, (frac, err)


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


@pytest.mark.parametrize("kw", [dict(m=64, m3=5, nz=2, nat=60), dict(m=96, m3=4, nz=3, nat=80, rect=True, imPot=0.0),
                                dict(m=60, m3=3, nz=1, nat=30, sub=2)])
def test_slice_loop_against_an_independent_numpy_multislice(oracle, kw):
    """The reference holds no wave function, so nothing reference-made pins the oracle's slice loop.  Second
    opinion, written from the DEFINITIONS of the method (Kirkland's multislice as FDES configures it, SURVEY 8a / 8a') in
    numpy float64 and sharing no line with oracle_core.c's forward_propagation / fresnel_propagator / zero_high_freq:
        t_s   = exp(-Im V_s) exp(i Re V_s)                               (potential2Transmission)
        BL[f] = F^-1[ M F[f] ],  M = [9 (i1^2 + i2^2) <= min(m1, m2)^2]   (radial 2/3 limit in index space, i = m/2 kept as +m/2)
        psi  <- F^-1[ M exp(-i pi lambda d3 (kx^2 + ky^2)) F[ BL[t_s] psi ] ],  psi_0 = 1
    on the potentials V_s the oracle's phaseGrating returns (those are checked independently above: Born integral, second
    moment, band-limited Fourier sum).  Agreement of the exit wave to 1e-12 means the restatement has no transcription
    error in the loop (operation order, normalisations 1/m12, band limit, propagator sign and scaling)."""
    hp, at = S.case_tiny(**kw)
    hp = oracle.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    c = q.c
    m1, m2 = c.m1, c.m2
    xyz = oracle.config_coords(q, at, 0, -1)
    i1 = np.fft.fftfreq(m1) * m1
    i2 = np.fft.fftfreq(m2) * m2
    i1[m1 // 2] = m1 // 2  # iwCoordIp keeps index m/2 as +m/2 (include/coordArithmetic.h:32); it only matters squared
    i2[m2 // 2] = m2 // 2
    I1, I2 = np.meshgrid(i1, i2)
    M = (9.0 * (I1 ** 2 + I2 ** 2) <= float(min(m1, m2)) ** 2)
    kx, ky = I1 / (m1 * float(c.d1)), I2 / (m2 * float(c.d2))
    pi_ref = float(np.float32(3.141592654))  # the reference's float constant (src/multisliceSimulation.cu:259), 2.8e-8 above pi
    P = M * np.exp(-1j * pi_ref * float(c.lambda_) * float(c.d3) * (kx ** 2 + ky ** 2))
    # the reference's two normalisations are the float32 number 1.f / (float)(m1 m2) (src/multisliceSimulation.cu:557-559, 599-602):
    # exact on power-of-two grids, 6e-8 off otherwise; the float64 oracle keeps the reference's constant, so does this
    nrm = float(np.float32(1.0) / np.float32(m1 * m2)) * (m1 * m2)
    psi = np.ones((m2, m1), np.complex128)
    for s in range(c.m3):
        V = oracle.phase_grating(q, at, xyz, s, "f64")
        t = np.exp(-V.imag) * np.exp(1j * V.real)
        t = np.fft.ifft2(M * np.fft.fft2(t)) * nrm
        psi = np.fft.ifft2(P * np.fft.fft2(t * psi)) * nrm
    ref = oracle.wave(q, at, 0, 0, prec="f64")
    err = np.linalg.norm(psi - ref) / np.linalg.norm(ref)
    assert err < 1e-11, (kw, err)
    assert np.abs(ref - 1).max() > 0.05   # a non-trivial wave


def test_single_measurement_entry_equals_the_series(oracle):
    """oracle.measurement(k) (round 4: the body of buildMeasurements' k loop, src/crystalMaker.cu:324-373, for ONE k) gives the
    k-th image of the full driver bit for bit: every random stream is keyed on (k, j), nothing is carried from one
    measurement to the next (pD = 0)."""
    hp, at = S.case_tiny(m=64, m3=4, nz=2, frPh=2, n3=3, tilt=True, beam_tilt=True)
    oracle.consistent(hp)
    for prec in ("f32", "f64"):
        full = oracle.build_measurements(hp, at, prec=prec)["image"]
        for k in range(3):
            assert np.array_equal(oracle.measurement(hp, at, k, prec=prec), full[k]), (prec, k)


def test_stage_goldens_and_au309_measurement_12(oracle):
    """SURVEY 8c's remaining fixtures (tools/make_golden.py, our oracle's outputs: the reference holds no such numbers):
    per-stage goldens at 64 x 64 - potential of one sub-slice, Fresnel propagator + band-limit mask, incoming wave, one full
    slice step - and the exit-wave intensity / image of measurement k = 12 of the shipped Au-309 example.  The oracle built
    here reproduces them (float64 to 1e-10; float32 to rounding of the compiler's libm calls)."""
    import fdes_amd
    g = np.load(os.path.join(G, "stage_cases.npz"))
    hp, at = S.case_tiny(**S.GOLDEN_CASES["img_2sp"])
    oracle.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    xyz = oracle.config_coords(q, at, 0, -1)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    for prec, tol in (("f64", 1e-10), ("f32", 1e-6)):
        V = oracle.phase_grating(q, at, xyz, 2, prec)
        P = oracle.fresnel_propagator(q, prec)
        psi0 = oracle.incoming_wave(q, 0, prec)
        assert rel(V, g["potential_s2_" + prec]) < tol and rel(P, g["propagator_" + prec]) < tol
        assert rel(psi0, g["incoming_" + prec]) < tol
        assert rel(oracle.forward_propagation(q, psi0, V, prec), g["one_step_" + prec]) < tol
    # the mask is the radial 2/3 limit of zeroHighFreq (src/multisliceSimulation.cu:225-250), in index space
    m = q.c.m1
    i = np.fft.fftfreq(m) * m
    assert np.array_equal(g["band_mask"].astype(bool), (9.0 * (i[None, :] ** 2 + i[:, None] ** 2) / m ** 2) <= 1.0)
    a = np.load(os.path.join(G, "au309_k12.npz"))
    hp, at = fdes_amd.read_cnf(os.path.join(G, "dataFDES_Auparticle.cnf"), bug_compatible=False)
    hp.set(pD=0.0)
    oracle.consistent(hp)
    q, _ = oracle.sub_sliced(hp)
    assert (q.c.m1, q.c.m3, at.n) == (320, 132, 309)
    psi = oracle.wave(q, at, 12, 0, prec="f64")
    assert rel(np.abs(psi) ** 2, a["exit_intensity_f64"]) < 1e-10
    e32 = rel(a["exit_intensity_f32"].astype(np.float64), a["exit_intensity_f64"])
    print(f"[kat] Au-309 k = 12: float32 oracle vs float64 truth {e32:.2e}")
    assert e32 < 1e-4
