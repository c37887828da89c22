import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# rocFFT compiles kernels for sizes outside its prebuilt set at plan time (tens of seconds each on a fresh
# box); keep its cache in-tree (git-ignored, travels with the gpurun snapshot like the built .so files).
os.environ.setdefault("ROCFFT_RTC_CACHE_PATH", os.path.join(ROOT, "fdes_amd", "csrc", "build", "rocfft_rtc_cache.db"))
# Grid lengths without compiled-in mixed-radix kernels get theirs compiled by hipRTC at plan creation (gen_jit.cpp: seconds per
# length).  The suite keeps that OFF by default, so that the run-time-length kernels - the fallback of every such length - stay
# covered and twenty lengths do not cost two minutes of compilation; the tests of the run-time compilation ask for it (jit = 1).
os.environ.setdefault("FDES_JIT", "0")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# GPU tests that take several seconds and only repeat, at another size or on a non-default kernel geometry, what a test of
# the default suite already covers: they run with FDES_GPU_SUITE=full (every BASELINE configuration keeps its full-size oracle
# test in the default suite; VERDICT r4: the driver's `-m gpu` run has a time limit)
full_only = pytest.mark.skipif(os.environ.get("FDES_GPU_SUITE") != "full", reason="repeats the coverage of a default-suite test (FDES_GPU_SUITE=full runs it)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def engine():
    import fdes_amd
    if not fdes_amd.gpu_available():
        pytest.fail("gpu-marked test but the HIP library sees no GPU")
    eng = fdes_amd.Engine(0)
    yield eng
    eng.close()
