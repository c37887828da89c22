#!/usr/bin/env python3
"""SURVEY 8d micro-benchmark: the stand-alone propagation unit psi <- F^-1[P F[t psi]] on device-resident random
inputs, m in {1024, 2048, 4096}, batches of independent waves (one transmission function each), 256 units timed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdes_amd
from tests import specimens as S

for m in (1024, 2048, 4096):
    hp, at = S.case_c3(k=2, n=m // 2, dn=m // 4, m3=2, frPh=0)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(0, lanes=1)
    pl = eng.plan(hp, at)
    for batch in (1, 4):
        g = torch.Generator(device="cuda").manual_seed(0)
        psi = torch.randn(batch, m, m, 2, device="cuda", generator=g)
        ph = (torch.rand(batch, m, m, device="cuda", generator=g) * 2 - 1) * 3.14159265
        t = torch.stack([torch.cos(ph), torch.sin(ph)], -1).contiguous()
        torch.cuda.synchronize()
        reps = max(256 // batch, 1)
        for _ in range(3):
            pl.propagate_dev(psi.data_ptr(), t.data_ptr(), batch, True)
        pl.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            pl.propagate_dev(psi.data_ptr(), t.data_ptr(), batch, True)
        pl.sync()
        dt = time.perf_counter() - t0
        rate = reps * batch / dt
        print(f"m={m} batch={batch}: {rate:9.0f} units/s  {dt / (reps * batch) * 1e6:7.1f} us/unit  "
              f"{80 * m * m * rate / 1e9:7.0f} GB/s on the 80 B/px model ({80 * m * m * rate / 8e12:.3f} of 8 TB/s)")
    pl.close()
