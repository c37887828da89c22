#!/bin/bash
# usage: tools/kres.sh file.hip [extra flags]  -> one line per kernel: name VGPRs spills occupancy
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics "$@" -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
 awk '/Function Name/ {n=$NF} /VGPRs:/ {v=$NF} /VGPRs Spill/ {s=$NF} /Occupancy/ {o=$NF} /LDS Size/ {print n, "vgpr", v, "spill", s, "occ", o}' | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | c++filt | sed 's/fdes::(anonymous namespace):://; s/(fdes::PassArgs)//'
