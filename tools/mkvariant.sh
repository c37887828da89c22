#!/bin/bash
# usage: tools/mkvariant.sh name [hipcc flags]  -> fdes_amd/csrc/build/variants/lib_<name>.so: the in-tree library with
# fft_lds.hip rebuilt under the extra flags (A/B runs: FDES_LIB=<that file>, tools/ab_libs.sh, tools/bench_one.py)
set -e
name=$1; shift
C=$(dirname "$0")/../fdes_amd/csrc
V=$C/build/variants
mkdir -p $V
OBJ=${OBJ:-fft_lds}   # the translation unit that is rebuilt (OBJ=fft_wave for the one-wave-per-row passes)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-result -munsafe-fp-atomics "$@" -I$C -c ${SRC:-$C/$OBJ.hip} -o $V/${OBJ}_$name.o
objs=$(ls $C/build/*.o | grep -v "/$OBJ.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $V/lib_$name.so $objs $V/${OBJ}_$name.o -L/opt/rocm/lib -lrocfft -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
echo built $V/lib_$name.so
