#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv): the queues the pass kernels ran on, kernels per queue, busy time per queue and
the average number of pass kernels in flight."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [r for r in rows if "k_wpass" in r["Kernel_Name"] or "k_pass" in r["Kernel_Name"] or "k_gpass" in r["Kernel_Name"]]
if not ks:
    sys.exit("no pass kernels")
t0 = min(int(r["Start_Timestamp"]) for r in ks); t1 = max(int(r["End_Timestamp"]) for r in ks)
perq = collections.defaultdict(lambda: [0, 0])
for r in ks:
    q = r["Queue_Id"]; perq[q][0] += 1; perq[q][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(v[1] for v in perq.values())
print(f"{len(ks)} pass kernels over {(t1 - t0) / 1e3:.0f} us on {len(perq)} queues; mean in flight {busy / (t1 - t0):.2f}")
for q, (n, b) in sorted(perq.items()):
    print(f"  queue {q}: {n} kernels, busy {b / 1e3:.0f} us ({b / (t1 - t0):.2f})")
# mean duration per kernel name (short)
d = collections.defaultdict(list)
for r in ks:
    d[r["Kernel_Name"][-40:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print(f"  {k}: n {len(v)} mean {sum(v) / len(v) / 1e3:.1f} us")
