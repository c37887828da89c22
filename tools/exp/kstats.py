#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run: tools/exp/kstats.py <dir> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', round(tot / 1e6, 2), 'launches', sum(int(r['Calls']) for r in rows))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{r['Name'][-64:]:64s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.1f} us  {100 * float(r['TotalDurationNs']) / tot:5.1f} %")
