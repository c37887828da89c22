#!/usr/bin/env python3
"""Instruction mix of the one-wave-per-row kernels from a device assembly listing:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o x.s fft_wave.hip; tools/instr_mix.py x.s [N]"""
import collections, re, sys
txt = open(sys.argv[1]).read()
n = sys.argv[2] if len(sys.argv) > 2 else "2048"
pat = re.compile(r"^_ZN4fdes\S*k_wpassILi%sELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb0E\S*: .*?\n(.*?)s_endpgm" % n, re.S | re.M)
for m in pat.finditer(txt):
    c = collections.Counter()
    for line in m.group(5).splitlines():
        line = line.strip()
        if not line or line[0] in ";." or line.endswith(":"):
            continue
        c[line.split()[0]] += 1
    cls = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print(f"pre={m.group(1)} mid={m.group(2)} post={m.group(3)} T={m.group(4)}: vector {cls('v_')} (packed {cls('v_pk_')}) lds {cls('ds_')} global {cls('global_')} scalar {cls('s_')}")
    print("    ", ", ".join(f"{k} {v}" for k, v in c.most_common(16)))
