#!/bin/bash
# usage: tools/ab_opt.sh OUT "flagsA" "flagsB" [rounds]   bench.py headline with two sets of flags in turn (A/B on one box)
OUT=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq 1 $R); do
  for f in "$A" "$B"; do
    timeout -k 10 200 python bench.py --cpu-baseline 0 --extra-skip-run 0 --extras 0 --hbm-cold 0 $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', d['value'], 'P5 alone us', d['roofline']['launch_us'])" >> $OUT
  done
done
