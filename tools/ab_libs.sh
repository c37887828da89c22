#!/bin/bash
# usage: tools/ab_libs.sh "name1 name2 ..." [rounds]: bench.py with fdes_amd/csrc/build/variants/lib_<name>.so ("tree" = in-tree library)
R=${2:-2}
for i in $(seq 1 $R); do
  for v in $1; do
    if [ $v = tree ]; then unset FDES_LIB; else export FDES_LIB=$PWD/fdes_amd/csrc/build/variants/lib_$v.so; fi
    timeout -k 10 200 python bench.py --cpu-baseline 0 --extra-skip-run 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'P5 alone us', d['roofline']['launch_us'])"
  done
done
