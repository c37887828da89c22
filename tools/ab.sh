#!/bin/bash
# usage: tools/ab.sh [rounds]   alternates bench.py between the in-tree library (B) and fdes_amd/csrc/build/variants/libA.so (A)
R=${1:-3}
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then export FDES_LIB=$PWD/fdes_amd/csrc/build/variants/libA.so; else unset FDES_LIB; fi
    timeout -k 10 200 python bench.py --cpu-baseline 0 --extra-skip-run 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'P5 alone us', d['roofline']['launch_us'])"
  done
done
