"""fdes_amd — MI355X-native forward multislice engine behind the FDES input/output surface.

The product is the HIP shared library fdes_amd/csrc/libFDES_SHARED_LIB.so (C-ABI: include/fdes_abi.h);
this package is its ctypes host binding.  Nothing here computes: if the library is not built, every
entry point raises.
"""
from .abi import HostAtoms, HostParams, load_library  # noqa: F401
from .api import (Engine, FdesError, Plan, build_measurements_multi, consistent, emd_available, gpu_available, read_cnf, read_emd, read_qsc, run_file,  # noqa: F401
                  sub_sliced, write_cnf, write_emd)
