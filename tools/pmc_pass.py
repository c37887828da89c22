#!/usr/bin/env python3
"""Run a few LDS passes once each (for rocprofv3 --pmc): python3 tools/pmc_pass.py [n] [band]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdes_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
band = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", int(os.environ.get("WG", 256 if n <= 2048 else 512)))   # WG=1: one row per thread, WG=64: one wave per row
for key, bb in [((0, 0, 0, 1), 0), ((1, 0, 0, 1), 0), ((1, 4, 2, 1), 1), ((2, 5, 1, 1), 6), ((1, 6, 2, 1), 1), ((2, 3, 1, 1), 4)]:
    eng.set_option("bench_band", bb if band else 0)
    print(key, eng.bench_pass(n, key[0], key[1], key[2], key[3], 3, 1))
