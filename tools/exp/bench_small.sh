#!/bin/bash
# usage: tools/bench_small.sh "size slices lanes wg" ...
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --size $1 --slices $2 --lanes $3 --pass-threads $4 --cpu-baseline 0 --extra-skip-run 0 --probe-stride 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('size',$1,'slices',$2,'lanes',$3,'wg',$4,'value',d['value'])"
done
