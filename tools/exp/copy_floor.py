"""Memory floor of the pass shape: copy with natural and with transposed stores (no transform), one and two streams."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdes_amd
eng = fdes_amd.Engine(0)
for n, wg in ((2048, 256), (4096, 512)):
    eng.set_option("pass_threads", wg)
    for band in (0, 6):
        eng.set_option("bench_band", band)
        for st in (0, 1):
            us = [eng.bench_pass(n, 0, 0, 0, st, 200, ns) for ns in (1, 2, 3)]
            mb = n * n * 16 / 1e6 * ((2 / 3) if band else 1.0)
            print(f"n={n} band={band} store {'transposed' if st else 'natural'}: " + "  ".join(f"x{i+1} {u:6.2f} us ({mb / u:5.2f} TB/s)" for i, u in enumerate(us)), flush=True)
