import sys, os
sys.path.insert(0, "/root/repo")
import fdes_amd
eng = fdes_amd.Engine(0)
PASSES = {"copy T": (0, 0, 0, 1, 0), "P4": (1, 4, 2, 1, 1), "P5": (2, 5, 1, 1, 6), "P6": (1, 6, 2, 1, 1), "P2": (1, 2, 2, 1, 0)}
for wg in (512, 256):
    for pitch in (64, 136, 8):
        eng.set_option("pass_threads", wg)
        eng.set_option("bench_pitch", pitch)
        for name, key in PASSES.items():
            eng.set_option("bench_band", key[4])
            cells = [f"{eng.bench_pass(4096, key[0], key[1], key[2], key[3], 100, ns):7.1f}" for ns in (1, 2)]
            print(f"4096 wg={wg} pitch={pitch} {name:7s} x1/x2: " + "/".join(cells), flush=True)
