#!/bin/bash
# usage: tools/ab_small.sh "size slices" [rounds]: bench.py at a small grid with library A (build/variants/libA.so) and B (in tree)
set -- $1 ${2:-2}
for i in $(seq 1 $3); do
  for v in A B; do
    if [ $v = A ]; then export FDES_LIB=$PWD/fdes_amd/csrc/build/variants/libA.so; else unset FDES_LIB; fi
    timeout -k 10 200 python bench.py --size $1 --slices $2 --cpu-baseline 0 --extra-skip-run 0 --probe-stride 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'size', $1, d['value'])"
  done
done
