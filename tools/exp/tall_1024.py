import sys, os
sys.path.insert(0, "/root/repo")
import fdes_amd
eng = fdes_amd.Engine(0)
eng.set_option("pass_threads", 1)
PASSES = {"P4": (1, 4, 2, 1), "P5": (2, 5, 1, 6), "P6": (1, 6, 2, 1), "P2": (1, 2, 2, 0)}
for n in (1024, 512):
    for name, key in PASSES.items():
        eng.set_option("bench_band", key[3])
        row = f"n={n} {name}: us per configuration-pass"
        for tall, ns in ((1, 1), (1, 3), (2, 1), (2, 2), (4, 1), (4, 2), (3, 2)):
            eng.set_option("bench_tall", tall)
            us = eng.bench_pass(n, key[0], key[1], key[2], 1, 200, ns)
            row += f" | tall{tall} x{ns}: {us / tall:6.2f}"
        print(row, flush=True)
