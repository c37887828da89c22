#!/usr/bin/env python3
"""Timeline figures from a rocprofv3 rocpd SQLite file: how much of the wall time has 0 / 1 / 2+ kernels running, gaps
between consecutive kernels of one queue, per-kernel averages inside the busiest window.
   python3 tools/timeline.py gpurun_out/prof/x_results.db [t_lo_frac t_hi_frac]"""
import re
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
print("# table", kt, "columns", cols)
qcol = next((q for q in ("queue_id", "queue", "stream_id", "stream") if q in cols), None)
rows = c.execute(f"select name, start, end{', ' + qcol if qcol else ''} from {kt} order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
rows = [r for r in rows if r[1] >= a and r[2] <= b]
print(f"# {len(rows)} kernels in window of {(b - a) / 1e6:.2f} ms")
ev = []
for r in rows:
    ev.append((r[1], 1))
    ev.append((r[2], -1))
ev.sort()
level, last, hist = 0, rows[0][1], defaultdict(int)
for t, d in ev:
    hist[level] += t - last
    last = t
    level += d
span = rows[-1][2] - rows[0][1]
tot = sum(r[2] - r[1] for r in rows)
print(f"span {span / 1e6:.3f} ms, sum of kernel durations {tot / 1e6:.3f} ms, mean concurrency {tot / span:.2f}")
for k in sorted(hist):
    print(f"  {k} kernels running: {hist[k] / 1e6:8.3f} ms  {100.0 * hist[k] / span:5.1f} %")
if qcol:
    byq = defaultdict(list)
    for r in rows:
        byq[r[3]].append(r)
    for q, rs in byq.items():
        gaps = [rs[i + 1][1] - rs[i][2] for i in range(len(rs) - 1)]
        gaps = [g for g in gaps if g < 1e6]
        if gaps:
            gaps.sort()
            print(f"  queue {q}: {len(rs)} kernels, gap to next kernel: median {gaps[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, p90 {gaps[int(0.9 * len(gaps))] / 1e3:.2f} us")
agg = defaultdict(lambda: [0, 0])
for r in rows:
    n = re.sub(r"\s+", " ", r[0]).replace("void fdes::(anonymous namespace)::", "").replace("(fdes::PassArgs)", "")
    agg[n][0] += 1
    agg[n][1] += r[2] - r[1]
for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print(f"  {n[:60]:60s} calls {k:6d} avg {t / k / 1e3:8.2f} us  share {100.0 * t / tot:5.1f} %")
