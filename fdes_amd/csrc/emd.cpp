// emd.cpp — EMD/HDF5 results writer (src/rwHdf5.cu:27-1083).  libhdf5 is resolved at run time
// with dlopen so that the engine has no link-time dependency on it (the GPU box may not have it).
#include <dlfcn.h>

#include <cstdio>

#include "fdes_internal.h"

extern "C" int fdes_write_emd(const char* file, const fdes_params* p, const fdes_atoms* atoms, const float* image,
                              const float* potential, const float* exitwave, int print_level)
{
    (void)file; (void)p; (void)atoms; (void)image; (void)potential; (void)exitwave; (void)print_level;
    return FDES_EUNSUPPORTED; // schema writer lands with SURVEY 8(f-1); Measurements.bin is always written
}
