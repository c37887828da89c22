#!/bin/bash
# usage: tools/bench_variants.sh "lanes wg" ...   (runs bench.py for each variant, prints one summary line each)
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --lanes $1 --pass-threads $2 --cpu-baseline 0 2>/dev/null | tail -1 > /tmp/bv.json
  python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open('/tmp/bv.json'))
print("lanes",sys.argv[1],"wg",sys.argv[2],"value",d["value"],"ms/step",d["ms_per_step"],"P5 us",d["roofline"]["launch_us"])
PY
done
