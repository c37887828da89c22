import sys, os
sys.path.insert(0, os.getcwd())
import fdes_amd
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", 256)
for name, key, bb in (("P4", (1, 4, 2, 1), 1), ("P6", (1, 6, 2, 1), 1), ("P5", (2, 5, 1, 1), 6), ("P3", (2, 3, 1, 1), 4)):
    eng.set_option("bench_band", bb)
    print(name, " ".join(f"x{s}: {eng.bench_pass(2048, *key, 100, s):6.2f}" for s in (1, 2, 3, 4)))
