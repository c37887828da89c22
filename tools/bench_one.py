#!/usr/bin/env python3
"""One pass, a few settings, one line: python3 tools/bench_one.py n pre mid post [band] [wg]   (FDES_LIB selects the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdes_amd
n, pre, mid, post = (int(x) for x in sys.argv[1:5])
band = int(sys.argv[5]) if len(sys.argv) > 5 else 0
wg = int(sys.argv[6]) if len(sys.argv) > 6 else (256 if n <= 2048 else 512)
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", wg)
eng.set_option("bench_band", band)
eng.set_option("walk", int(os.environ.get("WALK", "1")))
out = []
for rep in range(3):
    out.append("/".join(f"{eng.bench_pass(n, pre, mid, post, 1, 200, ns):6.2f}" for ns in (1, 2)))
print(os.path.basename(os.environ.get("FDES_LIB", "default")), f"n={n} ({pre},{mid},{post}) band={band} wg={wg} walk={os.environ.get('WALK', '1')}: x1/x2 us:", "  ".join(out), flush=True)
