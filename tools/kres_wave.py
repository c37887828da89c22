#!/usr/bin/env python3
"""Register / spill table of the one-wave-per-row kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks on stdin."""
import re, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: \S*k_wpassILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)", line)
    if m:
        cur = {"k": "N=%s pre=%s mid=%s post=%s T=%s pipe=%s" % m.groups()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(\w[\w ]*?)(?: \[bytes/lane\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
for r in rows:
    print(f"{r['k']:46s} VGPR {r.get('VGPRs', 0):4d} AGPR {r.get('AGPRs', 0):4d} vspill {r.get('VGPRs Spill', 0):4d} sspill {r.get('SGPRs Spill', 0):4d} scratch {r.get('ScratchSize', 0):5d} occ {r.get('Occupancy', 0)}")
