#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd SQLite file -> CSV on stdout.
   python3 tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else cols[0]
rows = c.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                 f"from kernels group by {name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for n, k, t, a, lo, hi in rows:
    n = re.sub(r"\s+", " ", n)
    print(f"\"{n}\",{k},{t},{a:.1f},{lo},{hi},{100.0 * t / tot:.2f}")
