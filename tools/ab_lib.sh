#!/bin/bash
# usage: tools/ab_lib.sh OUT name [rounds] ["bench flags"]: bench.py headline with the in-tree library and with build/variants/lib_<name>.so in turn
OUT=$1; NAME=$2; R=${3:-3}; FLAGS=${4:-}
for i in $(seq 1 $R); do
  for v in tree $NAME; do
    if [ $v = tree ]; then unset FDES_LIB; else export FDES_LIB=$PWD/fdes_amd/csrc/build/variants/lib_$v.so; fi
    timeout -k 10 200 python bench.py --cpu-baseline 0 --extra-skip-run 0 --extras 0 --hbm-cold 0 $FLAGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'P5 alone us', d['roofline']['launch_us'])" >> $OUT
  done
done
