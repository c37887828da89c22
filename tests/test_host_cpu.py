"""CPU tests of the host side: C-ABI exports, parameter handling, the .cnf reader/writer and sharding."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import fdes_amd
from fdes_amd import abi, shard
from tests import specimens as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import glob
    hdrs = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    assert [os.path.basename(h) for h in hdrs] == ["fdes_abi.h", "fdes_abi_test.h"]
    hdr = "".join(open(h).read() for h in hdrs)
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fdes_[a-z0-9_]+|FDES)\s*\(", hdr))
    declared.discard("fdes_abi_h_")
    declared.discard("fdes_abi_test_h_")
    lib = fdes_amd.load_library()
    bound = {n for n, _, _ in abi.PROTOTYPES}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fdes_abi_version() == 1
    # ... and NOTHING else: the library is built hidden with an export list (fdes_amd/csrc/exports_product.txt for
    # fdes_abi.h, exports_hooks.txt for fdes_abi_test.h; a TEST_HOOKS=0 build drops the second list)
    out = subprocess.run(["nm", "-D", "--defined-only", abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert exported == declared, sorted(exported ^ declared)
    csrc = os.path.join(ROOT, "fdes_amd", "csrc")
    prod = set(open(os.path.join(csrc, "exports_product.txt")).read().split())
    hooks = set(open(os.path.join(csrc, "exports_hooks.txt")).read().split())
    h_prod = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fdes_abi.h")).read(), flags=re.S)
    assert prod == set(re.findall(r"\b(fdes_[a-z0-9_]+|FDES)\s*\(", h_prod)) - {"fdes_abi_h_"}
    assert prod | hooks == declared and not (prod & hooks)


def test_export_list_without_test_hooks(tmp_path):
    """`make TEST_HOOKS=0`: the version script of such a build lists the product entry points only."""
    csrc = os.path.join(ROOT, "fdes_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, f"B={tmp_path}", "TEST_HOOKS=0", f"{tmp_path}/exports.map"], check=True)
    names = set(re.findall(r"^([A-Za-z_0-9]+);", open(tmp_path / "exports.map").read(), flags=re.M))
    assert names == set(open(os.path.join(csrc, "exports_product.txt")).read().split())
    assert "fdes_bench_pass" not in names and "FDES" in names


def test_library_builds_from_clean(tmp_path):
    """Every object of the library from scratch into a scratch directory (the in-tree build reuses objects): all 17
    translation units for gfx950 (gen_jit.cpp with the text of fft_gen.hip that the Makefile writes beside the objects) and the
    link with the export list - as a `make TEST_HOOKS=0` build, the one a maintainer
    ships: it exports the product entry points only, and fdes_amd/abi.py loads it (the hook prototypes are optional there: a
    call of one fails loudly)."""
    csrc = os.path.join(ROOT, "fdes_amd", "csrc")
    lib = tmp_path / "libFDES_SHARED_LIB.so"
    r = subprocess.run(["make", "-s", "-j8", "-C", csrc, f"B={tmp_path}/build", f"LIB={lib}", "TEST_HOOKS=0", str(lib)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(os.listdir(tmp_path / "build")) == 19   # 17 objects + exports.map + gen_jit_src.inc
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib)], capture_output=True, text=True, check=True).stdout
    prod = set(open(os.path.join(csrc, "exports_product.txt")).read().split())
    assert {l.split()[-1] for l in out.splitlines()} == prod
    code = ("import sys; sys.path.insert(0, %r); from fdes_amd import abi\n"
            "lib = abi.load_library(%r)\n"
            "assert lib.fdes_abi_version() == 1\n"
            "try:\n    lib.fdes_plan_probe_ms(None, None, None)\nexcept RuntimeError as e:\n    print('hook refused:', e)\n" % (ROOT, str(lib)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "hook refused" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("n,ept", [(1100, 16), (720, 8), (3840, 16)])
def test_mixed_radix_source_compiles_under_hiprtc(n, ept):
    """gen_jit.cpp hands hipRTC the text of fft_gen.hip (+ fft_lds.h, geometry.h, fft_dev.inc) with -DFDES_GEN_JIT_N=<length> at
    plan creation: the same text with the same options must compile here (no GPU needed) and export the sixteen pass kernels
    by name - the guards around the host side of these files are part of the product."""
    csrc = os.path.join(ROOT, "fdes_amd", "csrc")
    try:
        rtc = C.CDLL("libhiprtc.so")
    except OSError:
        rtc = C.CDLL("/opt/rocm/lib/libhiprtc.so")
    src = open(os.path.join(csrc, "fft_gen.hip"), "rb").read()
    names = [b"fft_lds.h", b"geometry.h", b"fft_dev.inc"]
    hdrs = [open(os.path.join(csrc, f.decode()), "rb").read() for f in names]
    prog = C.c_void_p()
    A = C.c_char_p * 3
    assert rtc.hiprtcCreateProgram(C.byref(prog), src, b"fft_gen.hip", 3, A(*hdrs), A(*names)) == 0
    opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-munsafe-fp-atomics", b"-DFDES_TEST_HOOKS=0", b"-DFDES_GEN_JIT_N=%d" % n, b"-DFDES_GEN_JIT_EPT=%d" % ept]
    rc = rtc.hiprtcCompileProgram(prog, len(opts), (C.c_char_p * len(opts))(*opts))
    ls = C.c_size_t()
    rtc.hiprtcGetProgramLogSize(prog, C.byref(ls))
    log = C.create_string_buffer(max(ls.value, 1))
    rtc.hiprtcGetProgramLog(prog, log)
    assert rc == 0, log.value.decode(errors="replace")[-3000:]
    cs = C.c_size_t()
    assert rtc.hiprtcGetCodeSize(prog, C.byref(cs)) == 0 and cs.value > 10000
    code = C.create_string_buffer(cs.value)
    assert rtc.hiprtcGetCode(prog, code) == 0
    raw = code.raw
    for kind in ("0_0_0_0", "1_9_0_1", "2_12_1_1", "1_2_2_1", "1_8_2_1", "1_4_2_1", "2_5_1_1", "0_5_1_1", "1_6_2_1"):
        assert b"fdes_jit_gpass_" + kind.encode() in raw, kind
    rtc.hiprtcDestroyProgram(C.byref(prog))


def test_which_grids_run_the_fused_loop():
    """fdes_grid_backend (host only): powers of two and lengths 2^a 3^b 5^c 7^d 11^e 13^f in [256, 4096], odd ones included, run the
    fused loop (2) - also where the tile rows of a mixed-radix length (8 up to 512 points, 4 up to 2048) do not divide the other
    dimension: smaller tiles then (500^2; 750^2 = 2 nx of a .qsc with an odd nx), a partial last tile for odd row counts (375^2,
    1001^2); a prime factor above 13, a length below 256 or beyond 8192, or a length beside a power of two whose row group does
    not divide it stay on rocFFT (1), as does option fft = 1."""
    lib = fdes_amd.load_library()
    for m in (256, 320, 375, 500, 572, 750, 800, 1000, 1001, 1100, 1125, 1250, 1430, 2002, 2048, 2288, 3000, 3003, 3300, 4000, 4096):
        assert lib.fdes_grid_backend(m, m, 0) == 2, m
    for m in (250, 255, 374, 999, 1450, 3002, 4100, 8192):   # (374 = 2 * 11 * 17 and 2006 = 2 * 17 * 59: see below)
        assert lib.fdes_grid_backend(m, m, 0) == 1, m
    assert lib.fdes_grid_backend(450, 4096, 0) == 2 and lib.fdes_grid_backend(500, 512, 0) == 2 and lib.fdes_grid_backend(1100, 572, 0) == 2
    assert lib.fdes_grid_backend(750, 1024, 0) == 1 and lib.fdes_grid_backend(1000, 1000, 1) == 1
    # rows of 4098 ... 8192 points exist as kernels compiled at plan creation only: fused (2) where that is not turned off
    # (this suite runs with FDES_JIT=0, tests/conftest.py: the query says 1 here) and libhiprtc loads
    code = ("import sys; sys.path.insert(0, %r); import fdes_amd\n"
            "lib = fdes_amd.load_library()\n"
            "print([lib.fdes_grid_backend(m, m, 0) for m in (8192, 5000, 6144, 4100, 8194, 9000, 1088, 1216, 736, 1450)], lib.fdes_grid_backend(8192, 256, 0))\n" % ROOT)
    # (1088 = 64 * 17, 1216 = 64 * 19, 736 = 32 * 23: radices of the compile-time kernels only; 1450 = 2 * 25 * 29: rocFFT)
    for jit, want in (("0", "[1, 1, 1, 1, 1, 1, 1, 1, 1, 1] 1"), ("1", "[2, 2, 2, 1, 1, 1, 2, 2, 2, 1] 2")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FDES_JIT=jit), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and want in r.stdout, (jit, r.stdout, r.stderr)


def test_no_gpu_means_loud_failure_not_fallback():
    lib = fdes_amd.load_library()
    if lib.fdes_gpu_available():
        pytest.skip("GPU present")
    with pytest.raises(fdes_amd.FdesError):
        fdes_amd.Engine(0)


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fdes_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "oracle/" not in txt, os.path.join(dirpath, f)


def test_params_defaults_and_consistency(oracle):
    lib = fdes_amd.load_library()
    p = abi.Params()
    assert lib.fdes_params_init(C.byref(p), 3) == 0
    d = oracle.default_params(3)
    for f in ("E0", "gamma", "lambda_", "sigma", "mtfa", "ObjAp", "m1", "m2", "m3", "d1", "d2", "d3", "subSlTh", "dn1",
              "dn2", "n1", "n2", "n3", "frPh", "pD", "mode", "imPot"):
        assert getattr(p, f) == getattr(d.c, f), f
    assert p.ab.C1_0 == d.c.ab.C1_0 and p.ab.C3_0 == d.c.ab.C3_0
    assert p.sample_name == b"Empty sample" and p.user_name == b"John Smith"
    lib.fdes_params_release(C.byref(p))
    for E0 in (40e3, 50e3, 200e3, 300e3):
        a = fdes_amd.consistent(S.make_params(2, E0=E0, n1=10, dn1=3, n2=8, dn2=1, tiltbeam=[0, 0, 0, 2e-3]))
        b = oracle.consistent(S.make_params(2, E0=E0, n1=10, dn1=3, n2=8, dn2=1, tiltbeam=[0, 0, 0, 2e-3]))
        for f in ("gamma", "lambda_", "sigma", "m1", "m2", "doBeamTilt"):
            assert getattr(a.c, f) == getattr(b.c, f)
    hp = fdes_amd.consistent(S.make_params(1, E0=200e3))
    assert (np.float32(hp.c.gamma), np.float32(hp.c.lambda_), np.float32(hp.c.sigma)) == \
        (np.float32(1.3913902), np.float32(2.507934e-12), np.float32(7288400.5))  # src/paramStructure.cu:509-512


def test_sub_slices_matches_reference_example():
    # bin/dataFDES.cnf: 12 slices of 2.1 A, subpixel_size_z 0.2 A -> ratio 11, 132 sub-slices
    hp = S.make_params(1, m3=12, d3=2.1e-10, subSlTh=0.2e-10)
    q, ratio = fdes_amd.sub_sliced(hp)
    assert ratio == 11 and q.c.m3 == 132 and abs(q.c.d3 - 2.1e-10 / 11) < 1e-17
    q, ratio = fdes_amd.sub_sliced(S.make_params(1, m3=7, d3=1e-10, subSlTh=1e-10))
    assert ratio == 1 and q.c.m3 == 7


CNF = """# test file
voltage:  80000     # comment
C1:  -2e-9
A1:  1e-9  0.5
mtf_a: 0.58
mtf_b: 0.42
mode:  1
gpu_number: 3
sample_size_z:  5
pixel_size_x:  0.2e-10
pixel_size_y:  0.3e-10
pixel_size_z:  1.5e-10
border_size_x:  4
border_size_y:  6
image_size_x:  24
image_size_y:  20
image_size_z:  2
frozen_phonons: 3
pixel_dose: 12.5
subpixel_size_z: 0.5e-10
absorptive_potential_factor: 0.1
specimen_tilt_offset_x: 0.17
user_name:  Ada Lovelace
sample_name:  Test sample
specimen_tilt:  0.1  0.2
specimen_tilt:  0.3  0.4
beam_tilt:  1e-3  2e-3
beam_tilt:  3e-3  4e-3
defoci:  1e-9
defoci:  2e-9
atom: 79  1e-10  2e-10  3e-10  6e-21  1
atom: 14  -1e-10  -2e-10  -3e-10  5e-21  0.5
"""


def _write(tmp_path, text):
    p = tmp_path / "x.cnf"
    p.write_text(text)
    return p


def test_read_cnf_clean_and_bug_compatible(tmp_path):
    hp, at = fdes_amd.read_cnf(_write(tmp_path, CNF), bug_compatible=False)
    c = hp.c
    assert (c.E0, c.mode, c.m3, c.dn1, c.dn2, c.n1, c.n2, c.n3, c.frPh) == (80000.0, 1, 5, 4, 6, 24, 20, 2, 3)
    assert (c.m1, c.m2) == (32, 32) and c.doBeamTilt == 1
    assert np.float32(c.d1) == np.float32(0.2e-10) and np.float32(c.pD) == np.float32(12.5)
    assert np.float32(c.ab.A1_0) == np.float32(1e-9) and np.float32(c.ab.A1_1) == np.float32(0.5)
    assert c.user_name == b"Ada Lovelace" and c.sample_name == b"Test sample"
    assert np.allclose(hp.tiltspec, [0.1, 0.2, 0.3, 0.4]) and np.allclose(hp.tiltbeam, [1e-3, 2e-3, 3e-3, 4e-3])
    assert np.allclose(hp.defoci, [1e-9, 2e-9])
    assert at.n == 2 and list(at.Z) == [79, 14] and np.allclose(at.xyz[1], [-1e-10, -2e-10, -3e-10])
    assert np.float32(at.occ[1]) == np.float32(0.5)
    # reference quirk: file ends in '\\n' after the last atom line -> that atom is read twice
    hp, at = fdes_amd.read_cnf(_write(tmp_path, CNF), bug_compatible=True)
    assert at.n == 3 and list(at.Z) == [79, 14, 14] and np.allclose(at.xyz[2], at.xyz[1])
    # ... and not when the final newline is missing
    hp, at = fdes_amd.read_cnf(_write(tmp_path, CNF.rstrip("\n")), bug_compatible=True)
    assert at.n == 2
    # quirk: a blank line after a specimen_tilt: line advances the index (stale token)
    quirk = CNF.replace("specimen_tilt:  0.1  0.2\n", "specimen_tilt:  0.1  0.2\n\n").replace("image_size_z:  2", "image_size_z:  3")
    hp, _ = fdes_amd.read_cnf(_write(tmp_path, quirk), bug_compatible=True)
    assert np.allclose(hp.tiltspec[:6], [0.1, 0.2, 0.0, 0.0, 0.3, 0.4])
    hp, _ = fdes_amd.read_cnf(_write(tmp_path, quirk), bug_compatible=False)
    assert np.allclose(hp.tiltspec[:4], [0.1, 0.2, 0.3, 0.4])


def test_default_subslice_thickness_is_not_the_parsed_d3(tmp_path):
    # defaultParams sets subSlTh = the DEFAULT d3 (2 A) before parsing (src/paramStructure.cu:561)
    txt = "\n".join(l for l in CNF.splitlines() if not l.startswith("subpixel_size_z")) + "\n"
    hp, _ = fdes_amd.read_cnf(_write(tmp_path, txt))
    assert np.float32(hp.c.subSlTh) == np.float32(2e-10)


def test_cnf_write_read_round_trip(tmp_path):
    hp, at = S.case_tiny(m=64, m3=3, nz=3, n3=2, tilt=True, beam_tilt=True, frPh=2, pD=3.0)
    fdes_amd.consistent(hp)
    p = tmp_path / "rt.cnf"
    fdes_amd.write_cnf(p, hp, at)
    hp2, at2 = fdes_amd.read_cnf(p, bug_compatible=False)
    for f, _t in abi.Params._fields_:
        if f in ("tiltspec", "tiltbeam", "defoci", "ab", "cap", "user_name", "institution", "department", "email",
                 "comments", "sample_name", "material", "nAt"):
            continue
        a, b = getattr(hp.c, f), getattr(hp2.c, f)
        assert a == b or abs(a - b) <= 1e-7 * abs(a), f
    assert hp2.c.nAt == at.n
    assert np.allclose(hp.tiltspec, hp2.tiltspec, rtol=1e-7) and np.allclose(hp.defoci, hp2.defoci, rtol=1e-7)
    assert at2.n == at.n and np.array_equal(at.Z, at2.Z) and np.allclose(at.xyz, at2.xyz, rtol=1e-7)
    # pointers of a parsed HostParams must reference its own numpy storage
    assert C.cast(hp2.c.tiltspec, C.c_void_p).value == hp2.tiltspec.ctypes.data


def test_atoms_from_array_truncates_occupancy_like_the_reference():
    lib = fdes_amd.load_library()
    arr = np.array([[79, 1, 2, 3, 6e-21, 0.7], [8, 4, 5, 6, 5e-21, 1.9]], np.float32)
    a = abi.Atoms()
    assert lib.fdes_atoms_from_array(C.byref(a), abi.fptr(arr), 2, 1) == 0
    assert [a.occ[0], a.occ[1]] == [0.0, 1.0] and [a.Z[0], a.Z[1]] == [79, 8]  # (int) cast, src/paramStructure.cu:323
    lib.fdes_atoms_release(C.byref(a))
    assert lib.fdes_atoms_from_array(C.byref(a), abi.fptr(arr), 2, 0) == 0
    assert np.float32(a.occ[0]) == np.float32(0.7)
    lib.fdes_atoms_release(C.byref(a))


def test_partition_covers_every_configuration_once():
    for n3, count, world in [(1, 32, 8), (64, 8, 8), (3, 5, 4), (2, 3, 2), (1, 1, 4), (5, 1, 3)]:
        seen = []
        for r in range(world):
            seen += shard.partition(n3, count, world, r)
        assert seen == [(k, j) for k in range(n3) for j in range(count)]
        own, ranks_of = shard.owners(n3, count, world)
        assert set(own) == set(range(n3))


_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from fdes_amd import shard
from tests import oracle_py as O, specimens as S
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
hp, at = S.case_tiny(m=32, m3=2, nz=2, frPh=3, n3=2, tilt=True)
O.consistent(hp)
q, _ = O.sub_sliced(hp)
m12 = q.c.m1 * q.c.m2

class OraclePlan:  # stands in for fdes_amd.Plan: same driver calls, compute by the CPU oracle
    def __init__(self):
        self.I = np.zeros((q.c.m2, q.c.m1), np.complex128)
        self.img = np.zeros((q.c.n3, q.c.n2, q.c.n1))
    def begin_measurement(self, k):
        self.I[:] = 0
    def run_config(self, k, j, w):
        import ctypes as C
        psi = O.wave(q, at, k, j, prec="f64")
        buf = O._c2buf(psi, np.float64)
        O.lib().oracle_apply_lens_f64(q.ptr, k, buf.ctypes.data_as(C.POINTER(C.c_double)))
        self.I += w * (buf[..., 0] ** 2 + buf[..., 1] ** 2)
    def end_measurement(self, k):
        import ctypes as C
        buf = O._c2buf(self.I, np.float64)
        J = np.zeros((q.c.n2, q.c.n1))
        O.lib().oracle_add_noise_and_mtf_f64(q.ptr, k, buf.ctypes.data_as(C.POINTER(C.c_double)), J.ctypes.data_as(C.POINTER(C.c_double)))
        self.img[k] = J

def reduce_fn(plan, k):
    t = torch.from_numpy(np.ascontiguousarray(plan.I.real))
    dist.all_reduce(t)
    plan.I = t.numpy().astype(np.complex128)

pl = OraclePlan()
done = shard.run_sharded(pl, q.c.n3, 3, rank, world, reduce_fn)
full = O.build_measurements(hp, at, prec="f64")["image"]
for k in done:
    err = np.linalg.norm(pl.img[k] - full[k]) / np.linalg.norm(full[k])
    assert err < 1e-12, (rank, k, err)
out = torch.zeros(q.c.n3)
for k in done:
    out[k] = 1
dist.all_reduce(out)
assert bool((out == 1).all()), out  # every k finalised by exactly one rank
dist.destroy_process_group()
print("rank", rank, "ok", done)
"""


def test_sharded_driver_world_size_2_gloo(tmp_path):
    """N > 1 path on CPU: two gloo ranks run shard.run_sharded with the oracle standing in for the GPU plan;
    the reduced, finalised images equal the single-process result."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", str(script), ROOT], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


# ---------------------------------------------------------------- EMD / HDF5 (SURVEY 8 f-1, f-2 vi)
EMD_FIXTURE = os.path.join(ROOT, "tests", "golden", "Auparticle_config.emd")


@pytest.mark.skipif(not fdes_amd.emd_available(), reason="libhdf5 not loadable")
def test_read_emd_matches_the_cnf_it_was_written_from():
    """Reference-produced fixture: Auparticle.emd was written by FDES from dataFDES_Auparticle.cnf (SURVEY KAT 3)."""
    hp, at = fdes_amd.read_emd(EMD_FIXTURE)
    c = hp.c
    assert (c.m1, c.m2, c.m3, c.n1, c.n2, c.n3, c.dn1, c.dn2, c.mode, c.frPh) == (320, 320, 12, 160, 160, 25, 80, 80, 0, 0)
    assert np.float32(c.E0) == np.float32(50e3) and np.float32(c.pD) == np.float32(100.0)
    assert np.float32(c.d1) == np.float32(0.25e-10) and np.float32(c.d3) == np.float32(2.1e-10)
    assert np.float32(c.subSlTh) == np.float32(0.2e-10)
    # derived constants re-computed by us == the values the reference stored in the file
    assert np.float32(c.gamma) == np.float32(1.09784758) and np.float32(c.lambda_) == np.float32(5.35530691e-12)
    assert abs(float(c.sigma) - 12279866.0) <= 1.0
    assert c.sample_name == b"Gold cuboctahedron" and c.material == b"Au_309" and c.user_name == b"John Smith"
    ref = np.load(os.path.join(ROOT, "tests", "golden", "au309_atoms.npy"))
    assert at.n == 309 and np.array_equal(at.Z, ref[:, 0].astype(np.int32)) and np.array_equal(at.xyz, ref[:, 1:4])
    assert np.array_equal(at.dwf, ref[:, 4]) and np.array_equal(at.occ, ref[:, 5])
    assert hp.tiltspec.size == 50 and np.float32(hp.tiltspec[0]) == np.float32(-0.17453294)


@pytest.mark.skipif(not fdes_amd.emd_available(), reason="libhdf5 not loadable")
def test_write_emd_round_trip_and_layout(tmp_path):
    hp, at = S.case_tiny(m=64, m3=3, nz=3, n3=2, tilt=True, beam_tilt=True, frPh=2, pD=3.0)
    hp.set(sample_name=b"tiny", material=b"AuSiO", comments=b"round trip")
    fdes_amd.consistent(hp)
    c = hp.c
    rng = np.random.default_rng(0)
    img = rng.random((c.n3, c.n2, c.n1), np.float32)
    pot = rng.random((c.m3, c.m2, c.m1, 2), np.float32)
    ew = rng.random((c.n3, c.m2, c.m1, 2), np.float32)
    out = tmp_path / "results.emd"
    fdes_amd.write_emd(out, hp, at, img, pot, ew, print_level=2)
    hp2, at2 = fdes_amd.read_emd(out)
    for f, _t in abi.Params._fields_:
        if f in ("tiltspec", "tiltbeam", "defoci", "ab", "cap", "user_name", "institution", "department", "email", "comments",
                 "sample_name", "material", "nAt"):
            continue
        assert getattr(hp.c, f) == getattr(hp2.c, f), f
    assert hp2.c.nAt == at.n
    for f, _t in abi.Aberration._fields_:
        if f in ("C1_1", "C3_1", "C5_1"):
            continue  # round aberrations carry no angle in the schema
        assert getattr(hp.c.ab, f) == getattr(hp2.c.ab, f), f
    assert np.array_equal(hp.tiltspec, hp2.tiltspec) and np.array_equal(hp.tiltbeam, hp2.tiltbeam)
    assert np.array_equal(hp.defoci, hp2.defoci)
    assert np.array_equal(at.xyz, at2.xyz) and np.array_equal(at.Z, at2.Z) and np.array_equal(at.occ, at2.occ)
    assert hp2.c.sample_name == b"tiny" and hp2.c.comments == b"round trip"
    h5dump = "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):  # data layout: /data/images/data is (n1, n2, n3) with x slowest (src/rwHdf5.cu:413-423)
        r = subprocess.run([h5dump, "-d", "/data/images/data", "-y", "-w", "0", str(out)], capture_output=True, text=True).stdout
        body = r[r.index("DATA {") + 6:r.rindex("}")]
        vals = np.array([float(x) for x in body.replace("}", " ").replace("\n", " ").split(",") if x.strip()], np.float32)
        assert vals.size == img.size
        assert np.allclose(vals.reshape(c.n1, c.n2, c.n3), img.transpose(2, 1, 0), rtol=2e-5, atol=1e-6)  # h5dump prints 6 digits
        names = subprocess.run([h5dump, "-n", str(out)], capture_output=True, text=True).stdout
        for need in ("/data/potential_slices/data", "/data/exit_wave/dim4", "/imaging/specimen_tilt_x", "/sample/debeye_waller_factors",
                     "/microscope/aberrations"):
            assert need in names, need


@pytest.mark.skipif(not fdes_amd.emd_available(), reason="libhdf5 not loadable")
def test_config_emd_equals_the_file_the_reference_wrote(tmp_path):
    """Reference-held golden for the writer schema: ExampleSpecimens/Au_cubeoctahedron_emd/Auparticle.emd was written by the
    reference's configuration writer (src/rwHdf5.cu:1085-1944, same helpers as the results writer :27-1083) from
    dataFDES_Auparticle.cnf.  Our writer, fed the same .cnf, must produce the same HDF5 object tree: every group, dataset and
    attribute name, type (class, size, sign, byte order, string padding), shape and value."""
    from tests import h5lite
    hp, at = fdes_amd.read_cnf(os.path.join(ROOT, "tests", "golden", "dataFDES_Auparticle.cnf"), bug_compatible=True)
    fdes_amd.consistent(hp)
    out = tmp_path / "config.emd"
    fdes_amd.write_emd(out, hp, at, None, None, None, print_level=0)
    ours, ref = h5lite.describe(out), h5lite.describe(EMD_FIXTURE)
    # 1 ulp on sigma: the file holds the reference binary's value 12279866, consitentParams in float32 gives ...867 (KAT 2)
    sig = ours["/microscope"]["attrs"]["interaction_constant"]
    got, want = (np.frombuffer(d["/microscope"]["attrs"]["interaction_constant"]["data"], np.float32)[0] for d in (ours, ref))
    assert abs(float(got) - float(want)) <= 1.0
    sig["data"] = ref["/microscope"]["attrs"]["interaction_constant"]["data"]
    assert h5lite.diff(ours, ref) == []
    # the frame axis of the images is the plain index (src/rwHdf5.cu:1393-1398), x and y are centred (:1330-1335)
    assert np.array_equal(np.frombuffer(ours["/data/images/dim3"]["data"], np.float32), np.arange(25, dtype=np.float32))
    assert np.array_equal(np.frombuffer(ours["/data/images/dim1"]["data"], np.float32), np.arange(160, dtype=np.float32) - 79.5)


@pytest.mark.skipif(not fdes_amd.emd_available(), reason="libhdf5 not loadable")
def test_results_emd_axes(tmp_path):
    """Results file at print_level 2 (src/rwHdf5.cu:27-1083): images and exit wave carry the frame INDEX as dim3
    ((float) i, :330-335, 493-498), the potential slices a centred dim3 (i - (m3-1)/2, :160-165); x and y centred everywhere."""
    from tests import h5lite
    hp, at = S.case_tiny(m=64, m3=4, nz=2, n3=3, tilt=True)
    fdes_amd.consistent(hp)
    c = hp.c
    rng = np.random.default_rng(1)
    img = rng.random((c.n3, c.n2, c.n1), np.float32)
    pot = rng.random((c.m3, c.m2, c.m1, 2), np.float32)
    ew = rng.random((c.n3, c.m2, c.m1, 2), np.float32)
    out = tmp_path / "results.emd"
    fdes_amd.write_emd(out, hp, at, img, pot, ew, print_level=2)
    d = h5lite.describe(out)

    def ax(name):
        return np.frombuffer(d[name]["data"], np.float32)
    assert np.array_equal(ax("/data/images/dim3"), np.arange(c.n3, dtype=np.float32))
    assert np.array_equal(ax("/data/exit_wave/dim3"), np.arange(c.n3, dtype=np.float32))
    assert np.array_equal(ax("/data/potential_slices/dim3"), np.arange(c.m3, dtype=np.float32) - np.float32((c.m3 - 1) / 2.0))
    for g, n in (("images", c.n1), ("exit_wave", c.m1), ("potential_slices", c.m1)):
        assert np.array_equal(ax(f"/data/{g}/dim1"), (np.arange(n) - (n - 1) / 2.0).astype(np.float32)), g
        assert np.array_equal(ax(f"/data/{g}/dim2"), (np.arange(n) - (n - 1) / 2.0).astype(np.float32)), g
    assert d["/data/exit_wave/data"]["shape"] == (c.m1, c.m2, c.n3, 2) and d["/data/potential_slices/data"]["shape"] == (c.m1, c.m2, c.m3, 2)
    assert d["/data/images/data"]["shape"] == (c.n1, c.n2, c.n3)
    # data layout without h5dump: /data/images/data is (n1, n2, n3) with x slowest (src/rwHdf5.cu:413-423)
    got = np.frombuffer(d["/data/images/data"]["data"], np.float32).reshape(c.n1, c.n2, c.n3)
    assert np.array_equal(got, img.transpose(2, 1, 0))
    got = np.frombuffer(d["/data/exit_wave/data"]["data"], np.float32).reshape(c.m1, c.m2, c.n3, 2)
    assert np.array_equal(got, ew.transpose(2, 1, 0, 3))
    assert d["/data/exit_wave/dim4"]["data"] == b"realimag" and d["/data/exit_wave/dim4"]["type"][:2] == ("string", 4)


# ---------------------------------------------------------------------------------------------------
# .qsc front-end: src/rwQsc.cu + qstem-libs readparam / .cfg reader / replicateUnitCell
QSC = os.path.join(ROOT, "tests", "golden", "qsc")   # the reference's bin/test.qsc + bin/SrTiO3.cfg (data fixtures)


def _srtio3_supercell(ncx, ncy, ncz, a=3.905):
    """Independent numpy restatement of the atom list readQsc hands to buildMeasurements for an orthogonal
    cell without tilt: unit-cell atoms sorted by (z, y, x), index = (icz + icy*ncz + icx*ncy*ncz)*5 + i,
    cartesian = a*(frac + cell), minus the box corner (0), x1e-10, minus (max-min)/2 per axis."""
    cell = [(38, 0, 0, 0, 0.6214), (22, .5, .5, .5, 0.4390), (8, 0, .5, .5, 0.7323), (8, .5, 0, .5, 0.7323), (8, .5, .5, 0, 0.7323)]
    cell.sort(key=lambda t: (t[3], t[2], t[1]))
    Z = np.zeros(5 * ncx * ncy * ncz, np.int32)
    xyz = np.zeros((Z.size, 3), np.float32)
    dwf = np.zeros(Z.size, np.float32)
    for i, (z, fx, fy, fz, dw) in enumerate(cell):
        for icx in range(ncx):
            for icy in range(ncy):
                for icz in range(ncz):
                    j = (icz + icy * ncz + icx * ncy * ncz) * 5 + i
                    Z[j] = z
                    frac = np.float32([fx, fy, fz]) + np.float32([icx, icy, icz])
                    xyz[j] = (a * frac.astype(np.float64)).astype(np.float32)
                    dwf[j] = np.float32(np.float64(np.float32(dw)) * 1e-20)
    xyz = (xyz.astype(np.float64) * 1e-10).astype(np.float32)
    lo = np.minimum(xyz.min(0), np.float32(1))
    hi = np.maximum(xyz.max(0), np.float32(0))
    return Z, xyz - (hi - lo) / np.float32(2), dwf


def test_read_qsc_shipped_example():
    """bin/test.qsc: 9x9x20 SrTiO3 cells, nx = 400 -> m = 800, 40 slices of 1.9525 A -> subSlTh = d3/10, CBED."""
    hp, at = fdes_amd.read_qsc(os.path.join(QSC, "test.qsc"))
    c = hp.c
    assert (c.n1, c.n2, c.dn1, c.dn2, c.m1, c.m2, c.m3, c.n3) == (400, 400, 200, 200, 800, 800, 40, 1)
    f32 = np.float32
    assert c.d1 == f32(np.float64(f32(0.087862)) * 1e-10) and c.d2 == c.d1
    assert c.d3 == f32(np.float64(f32(1.9525)) * 1e-10)
    assert c.subSlTh == f32(np.float64(f32(1.9525)) * 1e-10 / 10)
    assert c.E0 == f32(200e3) and c.illangle == f32(np.float64(f32(15)) / 1e3)
    assert c.mode == 2 and c.pD == f32(10) and c.ObjAp == f32(20e-3) and c.imPot == f32(0.1) and c.frPh == 0
    assert c.defocspread == f32(1e-9) and (c.mtfa, c.mtfb, c.mtfc) == (1, 0, 0)
    cs_A = f32(np.float64(f32(0.05)) * 1e7)
    assert c.ab.C3_0 == f32(np.float64(cs_A) * 1e-10)
    assert c.ab.C1_0 == f32(np.float64(f32(10.0 * np.float64(f32(13.7)))) * 1e-10)
    assert c.ab.C5_0 == 0 and c.ab.A1_0 == 0 and c.ab.A1_1 == 0
    assert c.tilt_offset_x == 0 and hp.tiltbeam[0] == 0 and hp.tiltbeam[1] == 0
    assert c.material == b"SrTiO3" and c.sample_name == b"SrTiO3_CELL_09_09_20"
    assert abs(c.lambda_ - 2.5079e-12) < 1e-15          # made consistent
    Z, xyz, dwf = _srtio3_supercell(9, 9, 20)
    assert at.n == 8100 and c.nAt == 8100
    assert np.array_equal(at.Z, Z)
    assert np.array_equal(at.xyz, xyz)                   # bit-exact
    assert np.array_equal(at.dwf, dwf) and np.all(at.occ == 1)
    q, ratio = fdes_amd.sub_sliced(hp)
    assert ratio == 10 and q.c.m3 == 400


_QSC_MIN = """mode: {mode}
filename: {cfg}
NCELLX: 2
NCELLY: 3
NCELLZ: {ncz}
{extra}
nx: 32
v0: 80
Cs: 1.2
alpha: 0.5
"""


def _write_qsc(tmp_path, mode="TEM", cfg="SrTiO3.cfg", ncz="2", extra=""):
    import shutil
    shutil.copy(os.path.join(QSC, "SrTiO3.cfg"), tmp_path / "SrTiO3.cfg")
    p = tmp_path / "t.qsc"
    p.write_text(_QSC_MIN.format(mode=mode, cfg=cfg, ncz=ncz, extra=extra))
    return p


def test_read_qsc_defaults_and_quirks(tmp_path):
    # no resolution -> super-cell / nx; no slice-thickness -> c / slices; ny = nx; Scherzer defocus; "STEM" contains "TEM"
    p = _write_qsc(tmp_path, mode="STEM", extra="slices: 4\nBeam tilt X: 2 deg\nBeam tilt Y: 3\nCrystal tilt Z: 0.25")
    hp, at = fdes_amd.read_qsc(p)
    c = hp.c
    f32 = np.float32
    assert (c.n1, c.n2, c.m1, c.m3) == (32, 32, 64, 4) and at.n == 5 * 2 * 3 * 2
    # rotated about z by 0.25 rad: the bounding box of the 2x3 cell footprint sets ax, by
    w, h = 2 * 3.905, 3 * 3.905
    ax = w * np.cos(0.25) + h * np.sin(0.25)
    assert abs(c.d1 - ax / 32 * 1e-10) < 1e-17
    assert abs(c.d3 - 2 * 3.905 / 4 * 1e-10) < 1e-17 and abs(c.subSlTh - c.d3 / 10) < 1e-18
    assert hp.tiltbeam[0] == f32(np.float64(f32(2)) * (3.1415926535897 / 180.0)) and hp.tiltbeam[1] == f32(3)
    assert c.tilt_offset_z == f32(0.25) and c.tilt_offset_x == 0
    lam = 12.3984244 / np.sqrt(80.0 * (2 * 510.99906 + 80.0))
    cs_A = f32(np.float64(f32(1.2)) * 1e7)
    df = -f32(np.sqrt(1.5 * np.float64(cs_A) * lam))
    assert c.ab.C1_0 == f32(np.float64(df) * 1e-10)
    # rotation about the box centre keeps the centroid of the (centred) atom cloud on the axis
    Z0, xyz0, _ = _srtio3_supercell(2, 3, 2)
    assert np.array_equal(at.Z, Z0)
    r0 = np.hypot(*(xyz0[:, :2] - xyz0[:, :2].mean(0)).T)
    r1 = np.hypot(*(at.xyz[:, :2] - at.xyz[:, :2].mean(0)).T)
    assert np.allclose(r0, r1, atol=2e-16) and np.allclose(at.xyz[:, 2], xyz0[:, 2], atol=1e-17)
    # skip_atoms mirrors atomsFromExternal
    hp2, none = fdes_amd.read_qsc(p, skip_atoms=True)
    assert none is None and hp2.c.m1 == 64


def test_read_qsc_slices_from_thickness_and_celldiv(tmp_path):
    p = _write_qsc(tmp_path, ncz="4/2", extra="slice-thickness: 1.0\ndefocus: 5\nastigmatism: 2\nC5: 1")
    hp, at = fdes_amd.read_qsc(p)
    # slices = int(c / (cellDiv * thickness) + 0.99) with c = 4 cells.  (A "center slices:" line without a
    # "slices:" line would be picked up by the substring search for "slices:" — undefined in the reference.)
    assert hp.c.m3 == int(np.float32(4 * 3.905) / (2 * 1.0) + 0.99)
    assert hp.c.ab.C1_0 == np.float32(np.float64(np.float32(50.0)) * 1e-10)
    assert hp.c.ab.A1_0 == np.float32(np.float64(np.float32(20.0)) * 1e-9)      # the reference's unit slip, kept
    assert hp.c.ab.C5_0 == np.float32(np.float64(np.float32(np.float64(np.float32(1)) * 1e7)) * 1e-3)


@pytest.mark.parametrize("kw, code", [
    (dict(mode="CBED"), -5),
    (dict(cfg="missing.cfg"), -2), (dict(cfg="missing.cssr"), -2), (dict(cfg="cell.pdb"), -5), (dict(cfg="cell.xyz"), -5)])
def test_read_qsc_rejects_what_it_does_not_carry_over(tmp_path, kw, code):
    p = _write_qsc(tmp_path, **kw)
    with pytest.raises(fdes_amd.FdesError) as e:
        fdes_amd.read_qsc(p)
    assert e.value.code == code


_SRTIO3_CSSR = """ 3.905 3.905 3.905
 {angles} SPGR = {spgr} P 1 OPT = 1
 5 0
 SrTiO3 written as a CSSR cell
   1 Sr  0.0 0.0 0.0  0 0 0 0 0 0 0 0  0.6214
   2 Ti1 0.5 0.5 0.5  0 0 0 0 0 0 0 0  0.4390
   3 O1  0.0 0.5 0.5  0 0 0 0 0 0 0 0  0.7323
   4 O2  0.5 0.0 0.5  0 0 0 0 0 0 0 0  0.7323
   5 O3  0.5 0.5 0.0  0 0 0 0 0 0 0 0  0.7323
"""
_SRTIO3_DAT = """Number of atoms = 5
a = 3.905
b = 3.905
c = 3.905
alpha = 90
beta = 90
gamma = 90
Sr 0.0 0.0 0.0
Ti 0.5 0.5 0.5
a comment line that names no element is skipped
O  0.0 0.5 0.5
O  0.5 0.0 0.5
O  0.5 0.5 0.0
"""


def test_read_qsc_cssr_and_dat_cells(tmp_path):
    """The vendored readUnitCell's other two formats (fileio_fftw3.cpp:666-713, 785-815, 825-895, 998-1052): the same
    SrTiO3 cell as .cssr and as .dat gives the .cfg super cell, with the lattice parameters rounded to float as there
    (MULS::ax is a float), the CSSR's last column as the Debye-Waller factor and the .dat rule 0.45*28/(2 Z)."""
    f32 = np.float32
    (tmp_path / "SrTiO3.cssr").write_text(_SRTIO3_CSSR.format(angles="90 90 90", spgr=1))
    (tmp_path / "SrTiO3.dat").write_text(_SRTIO3_DAT)
    Z0, xyz0, dwf0 = _srtio3_supercell(2, 3, 2, a=float(f32(3.905)))
    p = _write_qsc(tmp_path, cfg="SrTiO3.cssr", extra="slices: 4")
    hp, at = fdes_amd.read_qsc(p)
    assert hp.c.material == b"SrTiO3" and hp.c.sample_name == b"SrTiO3_CELL_02_03_02"
    assert np.array_equal(at.Z, Z0) and np.array_equal(at.xyz, xyz0) and np.array_equal(at.dwf, dwf0)
    assert np.all(at.occ == 1)
    # without an extension the .cssr is preferred over the .cfg (rwQsc.cu:181-210)
    hp_b, at_b = fdes_amd.read_qsc(_write_qsc(tmp_path, cfg="SrTiO3", extra="slices: 4"))
    assert np.array_equal(at_b.xyz, xyz0) and hp_b.c.d1 == hp.c.d1
    _, at_cfg = fdes_amd.read_qsc(_write_qsc(tmp_path, cfg="SrTiO3.cfg", extra="slices: 4"))
    assert np.allclose(at_cfg.xyz, xyz0, rtol=0, atol=1e-16)
    hp_d, at_d = fdes_amd.read_qsc(_write_qsc(tmp_path, cfg="SrTiO3.dat", extra="slices: 4"))
    assert np.array_equal(at_d.Z, Z0) and np.array_equal(at_d.xyz, xyz0) and hp_d.c.material == b"SrTiO3"
    dw = f32(0.45 * 28.0 / (2.0 * Z0.astype(np.float64)))
    assert np.array_equal(at_d.dwf, (dw.astype(np.float64) * 1e-20).astype(f32))


def test_read_qsc_cssr_oblique_cell_and_refusals(tmp_path):
    # gamma = 120: makeCellVectMuls (matrixlib.cpp:722-747) sets b = by*(cos g, cos g, 0) - kept as it is there
    (tmp_path / "SrTiO3.cssr").write_text(_SRTIO3_CSSR.format(angles="90 90 120", spgr=1))
    p = _write_qsc(tmp_path, cfg="SrTiO3.cssr", ncz="1", extra="slices: 2")
    hp, at = fdes_amd.read_qsc(p)
    a = float(np.float32(3.905))
    d = 1.7453292519943e-2
    cg = np.cos(120 * d)
    Mm = np.array([[a, 0, 0], [a * cg, a * cg, 0], [a * np.cos(90 * d), a * (np.cos(90 * d) - np.cos(90 * d) * cg) / np.sin(120 * d),
                                                     a * np.sqrt(1 - 2 * np.cos(90 * d) ** 2 + 2 * np.cos(90 * d) ** 2 * cg) / np.sin(120 * d)]])
    corners = np.array([[i, j, k] for i in (0, 2) for j in (0, 3) for k in (0, 1)], float) @ Mm
    ext = corners.max(0) - corners.min(0)
    assert abs(hp.c.d1 * hp.c.n1 - ext[0] * 1e-10) < 1e-16 and abs(hp.c.d3 * hp.c.m3 - ext[2] * 1e-10) < 1e-16
    cell = [(0, 0, 0), (.5, .5, .5), (0, .5, .5), (.5, 0, .5), (.5, .5, 0)]
    cell.sort(key=lambda t: (t[2], t[1], t[0]))
    want = np.array([(np.array(cell[i]) + (icx, icy, 0)) @ Mm - corners.min(0)
                     for icx in range(2) for icy in range(3) for i in range(5)]) * 1e-10
    want = want.reshape(2, 3, 5, 3).reshape(-1, 3)
    lo = np.minimum(want.min(0), 1.0)
    hi = np.maximum(want.max(0), 0.0)
    assert at.n == 30 and np.allclose(at.xyz, want - (hi - lo) / 2, rtol=0, atol=2e-16)
    for text, code in ((_SRTIO3_CSSR.format(angles="90 90 90", spgr=2), -5),
                       (_SRTIO3_CSSR.format(angles="90 90 90", spgr=1).replace("SPGR =", "SPGR:"), -1),
                       (_SRTIO3_CSSR.format(angles="90 90 90", spgr=1).replace("  0 0 0 0 0 0 0 0  0.6214", ""), -1),
                       (_SRTIO3_CSSR.format(angles="90 90 90", spgr=1).replace(" 5 0\n", " 6 0\n"), -1)):
        (tmp_path / "SrTiO3.cssr").write_text(text)
        with pytest.raises(fdes_amd.FdesError) as e:
            fdes_amd.read_qsc(p)
        assert e.value.code == code
    (tmp_path / "SrTiO3.dat").write_text(_SRTIO3_DAT.replace("Number of atoms = 5", "Number of atoms = 6"))
    with pytest.raises(fdes_amd.FdesError) as e:
        fdes_amd.read_qsc(_write_qsc(tmp_path, cfg="SrTiO3.dat", extra="slices: 4"))
    assert e.value.code == -1


@pytest.mark.parametrize("tilt, off", [((0.0, 0.0, 0.0), (0.0, 0.0)), ((0.1, -0.2, 0.25), (0.0, 0.0)), ((0.0, 0.3, 0.0), (1.5, -0.7))])
def test_read_qsc_boxed_super_cell(tmp_path, tilt, off):
    """`Cube:` (tiltBoxed, fileio_fftw3.cpp:1661-1925): the crystal, tilted about the origin, cut to the box [0, cube]
    (borders included) after the x / y offset; lattice range from the box corners through inv(R M); atoms in the order
    site (sorted z, y, x), ix, iy, iz; ax, by, c become the box; slices divide the box height.  Against an independent
    numpy restatement."""
    cube = (13.0, 11.0, 9.5)
    extra = f"Cube: {cube[0]} {cube[1]} {cube[2]}\nslices: 5\nCrystal tilt X: {tilt[0]}\nCrystal tilt Y: {tilt[1]}\nCrystal tilt Z: {tilt[2]}\n" \
            f"xOffset: {off[0]}\nyOffset: {off[1]}"
    p = _write_qsc(tmp_path, extra=extra)
    hp, at = fdes_amd.read_qsc(p)
    f32 = np.float32
    a = 3.905
    px, py, pz = (float(f32(t)) for t in tilt)
    R = np.eye(3)
    if any(tilt):
        R = np.array([[np.cos(pz) * np.cos(py), np.cos(pz) * np.sin(py) * np.sin(px) - np.sin(pz) * np.cos(px), np.cos(pz) * np.sin(py) * np.cos(px) + np.sin(pz) * np.sin(px)],
                      [np.sin(pz) * np.cos(py), np.sin(pz) * np.sin(py) * np.sin(px) + np.cos(pz) * np.cos(px), np.sin(pz) * np.sin(py) * np.cos(px) - np.cos(pz) * np.sin(px)],
                      [-np.sin(py), np.cos(py) * np.sin(px), np.cos(py) * np.cos(px)]])
    M = R @ (a * np.eye(3))
    Minv = np.linalg.inv(M)
    d = np.array([float(f32(off[0])), float(f32(off[1])), 0.0])
    cb = np.array([float(f32(c)) for c in cube])
    lo = np.floor(Minv @ np.zeros(3) - d).astype(int)
    hi = lo.copy()
    for ix in (0, 1):
        for iy in (0, 1):
            for iz in (0, 1):
                b = Minv @ (np.array([ix, iy, iz]) * cb - d)
                lo = np.minimum(lo, np.floor(b).astype(int))
                hi = np.maximum(hi, np.ceil(b).astype(int))
    cell = [(38, (0, 0, 0), 0.6214), (22, (.5, .5, .5), 0.4390), (8, (0, .5, .5), 0.7323), (8, (.5, 0, .5), 0.7323), (8, (.5, .5, 0), 0.7323)]
    cell.sort(key=lambda t: (t[1][2], t[1][1], t[1][0]))
    Z, xyz = [], []
    for z, fr, dw in cell:
        for ix in range(lo[0], hi[0] + 1):
            for iy in range(lo[1], hi[1] + 1):
                for iz in range(lo[2], hi[2] + 1):
                    pos = M @ (np.array([ix, iy, iz], float) + np.array(fr)) + d
                    if np.all(pos >= 0) and np.all(pos <= cb):
                        Z.append(z)
                        xyz.append(pos)
    Z = np.array(Z, np.int32)
    xyz = (np.array(xyz).astype(f32).astype(np.float64) * 1e-10).astype(f32)
    assert at.n == Z.size > 20 and np.array_equal(at.Z, Z)
    shift = (np.maximum(xyz.max(0), f32(0)) - np.minimum(xyz.min(0), f32(1))) / f32(2)
    assert np.allclose(at.xyz, xyz - shift, rtol=0, atol=3e-16)
    c = hp.c
    assert abs(c.d1 * c.n1 - cb[0] * 1e-10) < 1e-16 and c.m3 == 5 and abs(c.d3 - cb[2] / 5 * 1e-10) < 1e-17


def test_read_qsc_tds_applies_einstein_displacements(tmp_path):
    """`tds: yes` (phononDisplacement, Einstein branch, fileio_fftw3.cpp:367-600): every site of every cell is displaced by
    a Gaussian with <u_i^2> = (T / 300) dw / (8 pi^2) / 3 per cartesian component.  The reference seeds the generator from the
    clock; here the seed is fixed: two reads agree to the bit, the displacements have the right size per species and scale
    with the temperature, and atoms of one cell move independently."""
    p0 = _write_qsc(tmp_path, extra="slices: 4")
    hp0, a0 = fdes_amd.read_qsc(p0)
    (tmp_path / "hot").mkdir()
    outs = {}
    for T in (300, 1200):
        p = _write_qsc(tmp_path, extra=f"slices: 4\ntds: yes\ntemperature: {T}")
        p = p.rename(tmp_path / f"t{T}.qsc")
        hp, at = fdes_amd.read_qsc(p)
        hp_b, at_b = fdes_amd.read_qsc(p)
        assert np.array_equal(at.xyz, at_b.xyz) and np.array_equal(at.Z, a0.Z)
        outs[T] = at.xyz.astype(np.float64)
    # undo the centring shift of readQsc (it moves with the extreme atoms): compare displacements about their mean per axis
    for T, xyz in outs.items():
        d = xyz - a0.xyz.astype(np.float64)
        d -= d.mean(0)
        for Z, dw in ((38, 0.6214), (22, 0.4390), (8, 0.7323)):
            sel = a0.Z == Z
            want = np.sqrt(T / 300.0 * dw / (8 * np.pi ** 2) / 3.0) * 1e-10
            got = d[sel].std()
            assert abs(got / want - 1) < (0.35 if sel.sum() < 20 else 0.25), (T, Z, got, want)
    d300 = outs[300] - a0.xyz
    assert abs(np.corrcoef(d300[:-1, 0], d300[1:, 0])[0, 1]) < 0.4      # neighbours in the list move independently


def test_read_qsc_needs_the_keys_the_reference_exits_on(tmp_path):
    for key in ("nx:", "v0:", "Cs:", "alpha:", "filename:", "mode:"):
        p = _write_qsc(tmp_path)
        p.write_text("".join(l for l in p.read_text().splitlines(True) if not l.startswith(key)))
        with pytest.raises(fdes_amd.FdesError):
            fdes_amd.read_qsc(p)


def _ran1_sequence(n):
    """ran1 of Numerical Recipes as the vendored QSTEM library has it (fileio_fftw3.cpp:2559-2599), seed -1."""
    IA, IM, IQ, IR, NTAB = 16807, 2147483647, 127773, 2836, 32
    NDIV = 1 + (IM - 1) // NTAB
    idum, iv = 1, [0] * NTAB
    for j in range(NTAB + 7, -1, -1):
        k = idum // IQ
        idum = IA * (idum - k * IQ) - IR * k
        if idum < 0:
            idum += IM
        if j < NTAB:
            iv[j] = idum
    iy, out = iv[0], []
    for _ in range(n):
        k = idum // IQ
        idum = IA * (idum - k * IQ) - IR * k
        if idum < 0:
            idum += IM
        j = iy // NDIV
        iy = iv[j]
        iv[j] = idum
        out.append(min(iy / IM, 1.0 - 1.2e-7))
    return out


def test_cfg_reader_partial_and_shared_occupancy_draws_vacancies_like_qstem(tmp_path):
    """replicateUnitCell with handleVacancies (fileio_fftw3.cpp:1188-1306): the Ti site half occupied, and the first O
    site shared by O (0.6) and N (0.3).  Sites are visited from the last sorted atom backwards, cells from the last to
    the first, one ran1 deviate (seed -1) per cell and site; the atom whose occupancy interval holds the deviate stays,
    the others become species 0 with their position and occupancy kept."""
    p = _write_qsc(tmp_path, extra="slices: 4")
    cfg = (tmp_path / "SrTiO3.cfg").read_text()
    cfg = cfg.replace("0.5 0.5 0.5  0.4390  1.0", "0.5 0.5 0.5  0.4390  0.5")
    cfg = cfg.replace("0 0.5 0.5  0.7323 1.0", "0 0.5 0.5  0.7323 0.6").replace("Number of particles = 5", "Number of particles = 6")
    cfg += "14\nN\n0 0.5 0.5  0.5 0.3\n"
    (tmp_path / "SrTiO3.cfg").write_text(cfg)
    hp, at = fdes_amd.read_qsc(p)
    ncx, ncy, ncz, nc = 2, 3, 2, 6
    # the unit cell as it is sorted (z, y, x; file order reversed first, stable): index, Z, occupancy
    cell = [(38, (0, 0, 0), 1.0), (22, (.5, .5, .5), 0.5), (8, (0, .5, .5), 0.6), (8, (.5, 0, .5), 1.0), (8, (.5, .5, 0), 1.0), (7, (0, .5, .5), 0.3)]
    order = sorted(range(nc - 1, -1, -1), key=lambda i: (cell[i][1][2], cell[i][1][1], cell[i][1][0]))
    uc = [cell[i] for i in order]
    rnd = iter(_ran1_sequence(4 * ncx * ncy * ncz))
    Z = np.zeros(nc * ncx * ncy * ncz, np.int32)
    i = nc - 1
    while i >= 0:
        jeq, tot = i - 1, uc[i][2]
        while jeq >= 0 and uc[jeq][1] == uc[i][1]:
            tot += uc[jeq][2]
            jeq -= 1
        for icx in range(ncx - 1, -1, -1):
            for icy in range(ncy - 1, -1, -1):
                for icz in range(ncz - 1, -1, -1):
                    jc = (icz + icy * ncz + icx * ncy * ncz) * nc
                    for i2 in range(i, jeq, -1):
                        Z[jc + i2] = uc[i2][0]
                    if tot < 1 or jeq < i - 1:
                        choice = next(rnd) if tot < 1.0 else tot * next(rnd)
                        last = 0.0
                        for i2 in range(i, jeq, -1):
                            occ = float(np.float32(uc[i2][2]))
                            if choice < last or choice >= last + occ:
                                Z[jc + i2] = 0
                            last += occ
        i = jeq
    assert at.n == Z.size and np.array_equal(at.Z, Z)
    kept_ti = (at.Z == 22).sum()
    assert 0 < kept_ti < ncx * ncy * ncz and (at.Z == 0).sum() > 0 and (at.Z == 7).sum() > 0
    occ = at.occ.reshape(-1, nc)
    assert np.allclose(occ, np.float32([u[2] for u in uc])[None, :])     # vacancies keep their occupancy


def test_c_abi_from_plain_c(tmp_path):
    """include/fdes_abi.h compiles as C99 and as C++17, and a C program linked against libFDES_SHARED_LIB.so reads the
    reference's Au-309 .cnf through the C-ABI (no Python, no GPU)."""
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "fdes_amd", "csrc")
    src = os.path.join(ROOT, "tests", "abi_c", "host_check.c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, src])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c++", "-I", inc, src])
    exe = str(tmp_path / "host_check")
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, src, "-o", exe, "-L", libdir, "-lFDES_SHARED_LIB",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    cnf = tmp_path / "au.cnf"
    hp, at = fdes_amd.read_emd(EMD_FIXTURE) if fdes_amd.emd_available() else (None, None)
    if hp is None:
        hp, at = S.case_tiny(m=64, m3=4, nz=2)
        fdes_amd.consistent(hp)
    fdes_amd.write_cnf(cnf, hp, at)
    out = subprocess.run([exe, str(cnf)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    kv = dict(t.split("=") for t in out.stdout.split())
    # the bug-compatible reader duplicates the last atom when the file ends in a newline
    assert int(kv["nAt"]) in (at.n, at.n + 1) and int(kv["m1"]) == hp.c.m1 and int(kv["m3_sub"]) >= int(kv["m3"])
    assert abs(float(kv["lambda"]) / hp.c.lambda_ - 1) < 1e-6


@pytest.mark.parametrize("name", ["dataFDES_bin.cnf", "dataFDES_Auparticle.cnf"])
def test_shipped_cnf_files_parse(name):
    """The reference's own .cnf inputs (bin/dataFDES.cnf, ExampleSpecimens/Au_cubeoctahedron_cnf): 309 Au atoms, 25
    measurements, 320^2 wave; the bug-compatible reader reproduces getParams' duplicated last atom when the file ends
    in a newline, the clean reader does not; against the shipped Auparticle.emd they describe the same simulation."""
    path = os.path.join(ROOT, "tests", "golden", name)
    hp, at = fdes_amd.read_cnf(path, bug_compatible=False)
    hb, ab = fdes_amd.read_cnf(path, bug_compatible=True)
    c = hp.c
    assert at.n == 309 and ab.n in (309, 310) and np.all(at.Z == 79)
    assert (c.m1, c.m2, c.m3, c.n1, c.n2, c.n3, c.dn1) == (320, 320, 12, 160, 160, 25, 80)
    assert np.array_equal(ab.xyz[:309], at.xyz) and (ab.n == 309 or np.array_equal(ab.xyz[309], at.xyz[308]))
    q, ratio = fdes_amd.sub_sliced(hp)
    assert ratio == 11 and q.c.m3 == 132                      # 2.1 A slices cut to ~0.2 A
    if fdes_amd.emd_available():
        he, ae = fdes_amd.read_emd(EMD_FIXTURE)
        assert ae.n == 309 and (he.c.m1, he.c.m3, he.c.n3) == (c.m1, c.m3, c.n3)
        assert np.allclose(ae.xyz, at.xyz, rtol=0, atol=1e-16) and abs(he.c.pD - c.pD) < 1e-6
        assert np.allclose(he.tiltspec[:50], hp.tiltspec[:50], atol=1e-7)


def _build_and_run(tmp_path, name, sources, flags, args=(), env=None):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O1", "-g", *flags, "-I", os.path.join(ROOT, "include"), "-o", exe,
           *[os.path.join(ROOT, s) for s in sources], "-ldl", "-lpthread"]
    subprocess.check_call(cmd)
    return subprocess.run([exe, *args], capture_output=True, text=True, errors="replace", timeout=600, env=dict(os.environ, **(env or {})))


def test_multi_gpu_driver_under_thread_sanitizer(tmp_path):
    """fdes_amd/csrc/multi.cpp (host threads, barriers, partition, ownership, failure of one worker) linked against stub
    plans that keep their sums in unlocked host memory, under ThreadSanitizer: any pair of calls the driver's barriers
    do not order is a reported race; the images / exit waves / potential must equal the serial sums."""
    r = _build_and_run(tmp_path, "multi_tsan", ["tests/host_cpp/multi_tsan.cpp", "fdes_amd/csrc/multi.cpp"], ["-fsanitize=thread"],
                       env={"TSAN_OPTIONS": "halt_on_error=1 exitcode=66"})
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0 and "all ok" in r.stdout and "ThreadSanitizer" not in r.stderr


def test_host_parsers_under_address_sanitizer(tmp_path):
    """The .cnf / .qsc / .cfg / .emd readers and writers (no GPU code) under AddressSanitizer + UBSan: the shipped inputs
    in every reader mode, write -> read round trips, and damaged inputs (truncated, over-long lines, missing values,
    garbage)."""
    r = _build_and_run(tmp_path, "parsers_asan",
                       ["tests/host_cpp/parsers_asan.cpp", "fdes_amd/csrc/cnf.cpp", "fdes_amd/csrc/qsc.cpp", "fdes_amd/csrc/emd.cpp",
                        "fdes_amd/csrc/params.cpp"], ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                       args=[os.path.join(ROOT, "tests", "golden"), str(tmp_path)], env={"ASAN_OPTIONS": "detect_leaks=1"})
    print(r.stdout[-1500:], r.stderr[-3000:])
    assert r.returncode == 0 and "all ok" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_bench_deals_every_configuration_once_and_starts_its_own_ranks():
    """bench.py: rank r runs configuration j = r + world * s in step s (src/crystalMaker.cu:324-367: the (k, j) loops are
    independent), so the ranks' timed steps cover j = 0 .. world * steps - 1 exactly once; and `python bench.py --gpus N`
    with no launcher starts N ranks itself (FDES_BENCH_DRYRUN: each rank prints what it would run and exits before any
    GPU call)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 3, 8):
        for steps in (1, 4, 7):
            js = sorted(bench.deal(r, world, s) for r in range(world) for s in range(steps))
            assert js == list(range(world * steps))
    env = dict(os.environ, FDES_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "5", "--warmup", "2"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1, 2]
    assert all(l["world"] == 3 and l["gpus_flag"] == 3 and l["local_rank"] == l["rank"] for l in lines)
    assert sorted(j for l in lines for j in l["timed_j"]) == list(range(15))
    # under a launcher (WORLD_SIZE set) the same script is one rank and starts nothing
    env.update(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                         capture_output=True, text=True, timeout=120)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["rank"] == 1 and lines[0]["timed_j"] == [1, 3]


def test_bench_launcher_takes_the_other_ranks_down_when_one_dies():
    """`python bench.py --gpus N` without a launcher: a rank that exits with an error before the rendezvous must not leave
    the parent waiting on its siblings (they would block in init_process_group / a barrier): the parent polls all ranks,
    terminates the rest on the first failure and returns that exit code; a deadline covers ranks that hang together."""
    import time
    env = dict(os.environ, FDES_BENCH_DRYRUN="fail:1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, (out.returncode, out.stderr[-500:])
    assert time.monotonic() - t0 < 60
    # every rank hangs: the deadline ends the run with 124
    env.update(FDES_BENCH_DRYRUN="fail:99", FDES_BENCH_DEADLINE_S="3")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 124, (out.returncode, out.stderr[-500:])
    assert time.monotonic() - t0 < 60


def test_pipelined_kernels_never_read_a_landing_register_early():
    """fft_wave.hip, pass_threads = 65: look-ahead operands are loaded into accumulation registers by inline assembly that
    the compiler's s_waitcnt insertion does not see; tools/check_acc_landing.py compiles the file to assembly and verifies
    that no compiler-generated instruction reads such a register and that these kernels have no scratch traffic."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_acc_landing.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok   k_wpass<") >= 20
