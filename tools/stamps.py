#!/usr/bin/env python3
"""Phase timeline of one LDS pass from in-kernel shader-clock stamps (library built with EXTRA=-DFDES_STAMPS, loaded
through FDES_LIB).  usage: tools/stamps.py n pre mid post [wg] [band] [streams]
Prints, over all waves of the last launch: when each phase boundary was passed relative to the kernel's first stamp
(median / p10 / p90 in us at the measured shader clock) and the mean duration of each phase."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f = tempfile.mktemp(suffix=".stamps")
os.environ["FDES_STAMP_FILE"] = f
import fdes_amd
n, pre, mid, post = (int(x) for x in sys.argv[1:5])
wg = int(sys.argv[5]) if len(sys.argv) > 5 else 256
band = int(sys.argv[6]) if len(sys.argv) > 6 else 0
streams = int(sys.argv[7]) if len(sys.argv) > 7 else 1
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", wg)
eng.set_option("bench_band", band)
eng.set_option("walk", int(os.environ.get("WALK", "1")))
us = eng.bench_pass(n, pre, mid, post, 1, 20, streams)
d = np.fromfile(f, np.uint64).reshape(-1, 16).astype(np.float64)
os.unlink(f)
d = d[d[:, 0] > 0]
names = {0: "entry", 1: "loads requested", 2: "operand 1 landed", 3: "transform 1 done", 4: "operand 2 landed", 5: "transform 2 done",
         6: "point-wise done", 7: "last transform done", 10: "tile staged", 8: "stores issued", 9: "stores drained"}
# The shader clock counters of the eight XCDs have different origins: durations come from a wave's own stamps, the
# wave's start relative to the launch from the 100 MHz wall clock (slot 15, common to the chip, 10 ns steps).
ghz = 2.4
print(f"pass ({pre},{mid},{post}) n={n} wg={wg} band={band} streams={streams}: host-timed {us:.2f} us per launch; {len(d)} waves")
entry = (d[:, 15] - d[:, 15].min()) * 0.01   # us
order = [0, 1, 2, 3, 4, 5, 6, 7, 10, 8, 9]
have = [s for s in order if (d[:, s] > 0).all()]
end = entry + (d[:, have[-1]] - d[:, 0]) / (ghz * 1e3)
print(f"  wave start after the first wave: median {np.median(entry):5.2f} us, p90 {np.percentile(entry, 90):5.2f}, max {entry.max():5.2f};"
      f"  wave lifetime: median {np.median(end - entry):5.2f} us;  last wave ends at {end.max():5.2f} us (= kernel span at 2.4 GHz)")
prev = None
for s in have:
    rel = entry + (d[:, s] - d[:, 0]) / (ghz * 1e3)
    line = f"  {names[s]:22s} at {np.median(rel):6.2f} us (p10 {np.percentile(rel, 10):6.2f}, p90 {np.percentile(rel, 90):6.2f}, max {rel.max():6.2f})"
    if prev is not None:
        line += f"   phase {np.mean(d[:, s] - d[:, prev]) / (ghz * 1e3):6.2f} us"
    print(line)
    prev = s
