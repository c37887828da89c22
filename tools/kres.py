#!/usr/bin/env python3
"""usage: tools/kres.py file.hip [extra hipcc flags] -> one line per kernel: VGPRs, spills, occupancy"""
import re, subprocess, sys
f = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", *sys.argv[2:], "-c", f,
       "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    if "error" in line and "remark" not in line:
        print(line)
    m = re.search(r"remark:\s+(Function Name|VGPRs|VGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("fdes::(anonymous namespace)::", "").replace("(fdes::PassArgs)", "").replace("void ", "")
    print(f"{n[:70]:70s} vgpr {r.get('VGPRs','?'):>4s} spill {r.get('VGPRs Spill','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} occ {r.get('Occupancy [waves/SIMD]','?')}")
