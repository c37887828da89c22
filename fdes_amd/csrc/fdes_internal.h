// fdes_internal.h — shared declarations of the engine's translation units (not installed).
#ifndef FDES_INTERNAL_H_
#define FDES_INTERNAL_H_
#include "../../include/fdes_abi_test.h"

int fdes_params_clone(fdes_params* dst, const fdes_params* src);
extern "C" int fdes_atoms_alloc(fdes_atoms* a, int n);

#endif
