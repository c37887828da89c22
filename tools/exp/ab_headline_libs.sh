#!/bin/bash
# usage: tools/exp/ab_headline_libs.sh "tree name1 ..." [rounds] ["bench flags"]: bench.py (no extras) with the in-tree library and
# with fdes_amd/csrc/build/variants/lib_<name>.so in turn: value per run
V=$PWD/fdes_amd/csrc/build/variants
B="python3 bench.py --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 --extra-skip-run 0 --hbm-cold 0 $3"
for r in $(seq 1 ${2:-3}); do
  for n in $1; do
    echo -n "$n: "
    if [ "$n" = tree ]; then $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
    else FDES_LIB=$V/lib_$n.so $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; fi
  done
done
