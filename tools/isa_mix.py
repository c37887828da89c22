#!/usr/bin/env python3
"""Instruction mix per k_pass instance from the device assembly:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ifdes_amd/csrc -Iinclude --cuda-device-only -S fdes_amd/csrc/fft_lds.hip -o /tmp/fft_lds.s
   python3 tools/isa_mix.py /tmp/fft_lds.s [N] [WG]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
N0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
WG0 = int(sys.argv[3]) if len(sys.argv) > 3 else 256
for f in re.split(r'\n(?=_ZN4fdes\S+:\s)', txt):
    m = re.match(r'_ZN4fdes12_GLOBAL__N_16k_passILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E', f)
    if not m:
        continue
    N, WG, PRE, MID, POST, ST = map(int, m.groups())
    if N != N0 or WG != WG0:
        continue
    body = f.split('s_endpgm')[0]
    c = collections.Counter()
    for line in body.split('\n'):
        w = line.strip().split()
        if not line.startswith('\t') or not w or w[0][0] in '.;':
            continue
        i = w[0]
        if i.startswith('v_pk'):
            c['vpk'] += 1
        elif i.startswith('v_'):
            c['v'] += 1
        elif i.startswith('ds_'):
            c['ds_' + ('w' if 'write' in i else 'r')] += 1
        elif i.startswith(('global_', 'buffer_')):
            c['g_' + ('st' if 'store' in i else 'ld')] += 1
        elif i.startswith('s_waitcnt'):
            c['wait'] += 1
        elif i.startswith('s_barrier'):
            c['bar'] += 1
        elif i.startswith('s_'):
            c['s'] += 1
    print((PRE, MID, POST, ST), dict(c), 'valu', c['v'] + c['vpk'])
