import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# rocFFT compiles kernels for sizes outside its prebuilt set at plan time (tens of seconds each on a fresh
# box); keep its cache in-tree (git-ignored, travels with the gpurun snapshot like the built .so files).
os.environ.setdefault("ROCFFT_RTC_CACHE_PATH", os.path.join(ROOT, "fdes_amd", "csrc", "build", "rocfft_rtc_cache.db"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
