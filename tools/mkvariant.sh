#!/bin/bash
# usage: tools/mkvariant.sh name [hipcc flags]  -> fdes_amd/csrc/build/variants/lib_<name>.so: the in-tree library with
# fft_lds.hip rebuilt under the extra flags (A/B runs: FDES_LIB=<that file>, tools/ab_libs.sh, tools/bench_one.py)
# OBJ="fft_wave" (or a list: OBJ="fft_lds fft_wave fft_gen") selects the translation unit(s) that are rebuilt.
set -e
name=$1; shift
C=$(dirname "$0")/../fdes_amd/csrc
V=$C/build/variants
mkdir -p $V
OBJ=${OBJ:-fft_lds}
objs=$(ls $C/build/*.o)
new=""
for o in $OBJ; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-result -munsafe-fp-atomics "$@" -I$C -c $C/$o.hip -o $V/${o}_$name.o &
  objs=$(echo "$objs" | grep -v "/$o.o")
  new="$new $V/${o}_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $V/lib_$name.so $objs $new -L/opt/rocm/lib -lrocfft -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
echo built $V/lib_$name.so
