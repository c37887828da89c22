#!/usr/bin/env python3
"""Where a small boundary call spends its time: plan creation, first gang (graph capture), later gangs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hp, at = fdes_amd.read_cnf(os.path.join(ROOT, "tests", "golden", "dataFDES_bin.cnf"))
for graph in (1, 0):
    eng = fdes_amd.Engine(0, graph=graph)
    eng.build_measurements(hp, at)
    for rep in range(2):
        t0 = time.perf_counter()
        pl = eng.plan(hp, at)
        pl.sync()
        t1 = time.perf_counter()
        pl.close()
        t2 = time.perf_counter()
        img = eng.build_measurements(hp, at)["image"]
        t3 = time.perf_counter()
        print(f"graph {graph}: plan create {1e3 * (t1 - t0):6.2f} ms, destroy {1e3 * (t2 - t1):6.2f} ms, whole call {1e3 * (t3 - t2):6.2f} ms (lanes {pl.lanes() if False else '-'})", flush=True)
    eng.close()
