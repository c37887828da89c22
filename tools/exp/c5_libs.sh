#!/bin/bash
# usage: tools/exp/c5_libs.sh "tree name1 name2" [rounds]: BASELINE config 5 (tools/run_c5.py 4) with the in-tree library and with
# fdes_amd/csrc/build/variants/lib_<name>.so in turn, one summary line per run
V=$PWD/fdes_amd/csrc/build/variants
for r in $(seq 1 ${2:-2}); do
  for n in $1; do
    echo -n "$n: "
    if [ "$n" = tree ]; then python3 tools/run_c5.py 4 2>&1 | grep "^C5:" | cut -c1-60
    else FDES_LIB=$V/lib_$n.so python3 tools/run_c5.py 4 2>&1 | grep "^C5:" | cut -c1-60; fi
  done
done
