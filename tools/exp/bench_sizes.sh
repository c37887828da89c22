#!/bin/bash
# usage: tools/exp/bench_sizes.sh "1280 2000 2560 3000 3600 4000" [extra bench.py flags]: slice-propagations/s of the C3 specimen on each grid size
# (32 slices, 6 timed configurations, every slice the full sequence), one line per size with the per-pass launch times
for s in $1; do
  python3 bench.py --size $s --slices 32 --steps 6 --warmup 2 --cpu-baseline 0 --extras 0 --extra-skip-run 0 --hbm-cold 0 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['roofline'].get('passes') or []
print('$s', d['value'], 'lanes', d['lanes'], d['slice_loop'], ' '.join('P%d %.1f/%s' % (r['pass'], r['launch_us'], r['launch_us_lanes']) for r in p))
"
done
