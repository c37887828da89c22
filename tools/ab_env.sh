#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." [rounds]   bench.py with environment variable VAR set to each value in turn
VAR=$1; VALS=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 200 python bench.py --cpu-baseline 0 --extra-skip-run 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], 'P5 alone us', d['roofline']['launch_us'])"
  done
done
