#!/usr/bin/env python3
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hp, at = fdes_amd.read_cnf(os.path.join(ROOT, "tests", "golden", "dataFDES_bin.cnf"))
eng = fdes_amd.Engine(0, graph=int(os.environ.get("GRAPH", "1")))
for rep in range(4):
    t0 = time.perf_counter()
    eng.build_measurements(hp, at)
    print(f"call {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
eng.close()
