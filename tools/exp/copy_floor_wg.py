"""Transposed-store copy floor at 2048^2 for 4-row (256 threads, 32-byte segments) and 8-row (512 threads, 64-byte segments) workgroups."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
eng = fdes_amd.Engine(0)
n = 2048
for pitch in (0, 32, 64):
    eng.set_option("bench_pitch", pitch)
    for wg in (256, 512):
        eng.set_option("pass_threads", wg)
        for band in (0, 6):
            eng.set_option("bench_band", band)
            for st in (0, 1):
                us = [eng.bench_pass(n, 0, 0, 0, st, 300, ns) for ns in (1, 2)]
                mb = n * n * 16 / 1e6 * ((2 / 3) if band else 1.0)
                print(f"pitch {pitch:3d} wg={wg} band={band} store {'transposed' if st else 'natural   '}: " + "  ".join(f"x{i+1} {u:6.2f} us ({mb / u:5.2f} TB/s)" for i, u in enumerate(us)), flush=True)
