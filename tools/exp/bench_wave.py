#!/usr/bin/env python3
"""One-wave-per-row passes (pass_threads = 64, fft_wave.hip) against the two-rows-per-thread ones, pass by pass:
mean launch time [us] on one and two streams, and a sweep of the staggered start.  Run on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd

PASSES = {"copy T": (0, 0, 0, 1), "FFT T": (1, 0, 0, 1), "P2 gtab": (1, 2, 2, 1), "P3 pair": (2, 12, 1, 1), "P4 mask": (1, 4, 2, 1),
          "P5 mulpsi": (2, 5, 1, 1), "P6 ptab": (1, 6, 2, 1)}
sizes = [int(x) for x in (sys.argv[1:] or ["2048"])]
BAND = {4: 1, 6: 1, 5: 6, 12: 4}  # band-limit bookkeeping per pass as the engine sets it (bit 0 live rows, 1 dead loads, 2 dead stores)
for n in sizes:
    for name, key in PASSES.items():
        row = f"n={n:5d} {name:10s}"
        for wg in ((1 if n <= 1024 else (256 if n <= 2048 else 512)), 64) + ((65,) if n >= 2048 else ()):
            for stg in ((0,) if (wg != 64 or n <= 1024) else (0, 8, 16)):
                eng = fdes_amd.Engine(0, pass_threads=wg, stagger=stg, bench_band=BAND.get(key[1], 0),
                                      bench_pitch=0 if n <= 1024 else (32 if n == 2048 else 64))
                res = []
                for ns in ((1, 2) if n > 1024 else (1, 2, 3)):
                    try:
                        res.append(f"{eng.bench_pass(n, key[0], key[1], key[2], key[3], 200, ns):6.2f}")
                    except Exception as e:
                        res.append("   n/a")
                eng.close()
                row += f" | wg{wg}" + (f" s{stg}" if wg == 64 else "") + ": " + "/".join(res)
        print(row, flush=True)
