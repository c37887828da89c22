#!/usr/bin/env python3
"""bench.py — slice-propagations/s of the FDES forward path on MI355X (BASELINE.json metric).

Workload (config C3 of SURVEY.md 8d', BASELINE.json configs[2]): Au cuboctahedron (94 611 atoms, generator
tests/specimens.py), 2048 x 2048 wave, 256 slices, frozen-phonon configurations.  One STEP = the whole slice
loop (projected-potential build + band-limited transmission + Fresnel propagation, 256 slice-propagations)
plus exit-wave post-processing of ONE frozen-phonon configuration per rank, inputs resident in HBM.  Ranks
take different configurations j (weak scaling, no data-path collective); after the timed region the partial
intensity sums are all-reduced once over RCCL (the "trivial gather" of the north star).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel class (the 2-D FFT, 6 per slice),
timed live with HIP events on the engine's stream; `cpu_baseline` is the CPU oracle (our restatement; the
reference has no CPU path) on a bounded sample, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=2048, help="wave size m (m x m), default the headline 2048")
    ap.add_argument("--slices", type=int, default=256)
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--fft", type=int, default=0, help="engine option fft: 0 auto, 1 rocFFT, 2 hand-written")
    ap.add_argument("--probe-stride", type=int, default=8)
    ap.add_argument("--lanes", type=int, default=0, help="configurations in flight per GPU (engine option lanes)")
    ap.add_argument("--pass-threads", type=int, default=0)
    ap.add_argument("--split", type=int, default=-1, help="engine option split: potential chain on a stream of its own (-1: engine default)")
    ap.add_argument("--batch", type=int, default=-1, help="engine option batch: slice pairs per launch of the potential chain (-1: engine default)")
    ap.add_argument("--walk", type=int, default=1, help="engine option walk: row groups per pass workgroup")
    ap.add_argument("--pitch-pad", type=int, default=-1, help="engine option pitch_pad (-1: by grid size)")
    ap.add_argument("--graph", type=int, default=1, help="engine option graph: replay the slice loop as a hipGraph")
    ap.add_argument("--extra-skip-run", type=int, default=1, help="also time the engine default (empty-slice short cut)")
    ap.add_argument("--extras", type=int, default=1,
                    help="rank 0, N = 1: also time BASELINE config 5 (4096^2 x 512 slices) and the propagation-unit micro-benchmark "
                         "(SURVEY 8d) so that they are driver-timed figures; reported under `extras`, never as the headline")
    ap.add_argument("--hbm-cold", type=int, default=1,
                    help="also time the roofline kernel with its operands in HBM only (8 buffer sets round-robin; roofline.hbm_cold)")
    ap.add_argument("--skip-empty", type=int, default=0,
                    help="1: slices without atoms only get the Fresnel step (engine default); 0 (bench default): every "
                         "slice runs the full potential/transmission/propagation sequence like the reference")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as `python bench.py --gpus N` without a launcher: this process becomes the launcher (it has not touched
        # the GPU: no torch, no HIP call) and starts one fresh child per GPU, as torch.distributed.run would
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FDES_BENCH_DRYRUN", "").startswith("fail:"):
        # launcher rehearsal of a rank that dies before the rendezvous: that rank exits 3 at once, the others would block
        if rank == int(os.environ["FDES_BENCH_DRYRUN"][5:]):
            sys.exit(3)
        import time as _t   # (a plain `import time` here would make the name local to main() and break its closures)
        _t.sleep(300)
        return
    if os.environ.get("FDES_BENCH_DRYRUN"):
        # launcher rehearsal without a GPU (tests/test_host_cpu.py): every rank reports what it WOULD run
        print(json.dumps({"dryrun": True, "rank": rank, "world": world, "local_rank": local, "gpus_flag": args.gpus,
                          "warmup_j": [1000 + deal(rank, world, w) for w in range(args.warmup)],
                          "timed_j": [deal(rank, world, s) for s in range(args.steps)]}), flush=True)
        return

    import numpy as np
    import torch
    import torch.distributed as dist

    # FDES_BENCH_BACKEND=gloo rehearses the multi-rank path on fewer GPUs than ranks (ranks share devices, reductions go
    # through host memory); the default is RCCL ("nccl"), one rank per GPU.
    backend = os.environ.get("FDES_BENCH_BACKEND", "nccl")
    ngpu = max(torch.cuda.device_count(), 1)
    if backend != "nccl":
        local = local % ngpu
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)

    import fdes_amd
    from tests import specimens

    m = args.size
    k_au = 30 if m >= 2048 else max(2, int(30 * m / 2048))
    hp, atoms = specimens.case_c3(k=k_au, n=m - 2 * (m // 4), dn=m // 4, m3=args.slices, frPh=32)   # (n + 2 dn = m also where four does not divide m)
    fdes_amd.consistent(hp)
    probe_lanes = {"ms": 0.0, "n": 0, "passes": None}

    def timed_run(skip_empty):
        """K timed steps on a fresh engine/plan; returns (seconds, plan, engine, loop_ms, loop_slices, fft_ms, fft_n, finite)."""
        eng = fdes_amd.Engine(local, fft=args.fft, probe_stride=0, lanes=args.lanes,
                              pass_threads=args.pass_threads, skip_empty=skip_empty, pitch_pad=args.pitch_pad, split=args.split)
        eng.set_option("graph", args.graph)
        eng.set_option("walk", args.walk)
        if args.batch >= 0:
            eng.set_option("batch", args.batch)
        plan = eng.plan(hp, atoms)

        def barrier():
            plan.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        plan.begin_measurement(0)
        # plan preparation (not a step): one configuration per lane so that every lane has captured and instantiated its
        # slice-loop graph whatever W is
        for lane in range(plan.lanes()):
            plan.run_config(0, 3000 + lane, 0.0)
        for w in range(args.warmup):
            plan.run_config(0, 1000 + deal(rank, world, w), 0.0)  # untimed, weight 0: does not touch the sum
        plan.sync()
        plan.slice_loop_ms()
        plan.probe_ms()
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            plan.run_config(0, deal(rank, world, s), weight)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        loop_ms, loop_slices = plan.slice_loop_ms()
        # roofline leg (probe_passes below), once per pass class P1' ... P6: one more configuration on lane 0 with the other
        # lanes idle, every `probe_stride`-th launch of that pass bracketed by the start / stop events of the dispatch itself (hipExtLaunchKernelGGL: kernel
        # begin / end as a profiler sees them).  Before it the same with one configuration per lane, the lanes sharing the
        # chip (`launch_us_lanes`): issued directly instead of replayed as graphs, a kernel's begin-to-end span then also
        # holds the time its workgroups wait for the other lane's to retire (26 us against 21.6 us under rocprofv3 for the
        # replayed loop and 22.5-23 us alone), so the lane-0-alone figure is the one `frac` is quoted on.
        passes = probe_passes(eng, plan, rank, args.probe_stride) if plan.fft_backend() == 2 else None
        if passes is not None:
            fft_ms, fft_n, lanes_ms, lanes_n = passes[5]
        else:   # rocFFT path: every probe_stride-th 2-D FFT
            plan.probe_ms()
            eng.set_option("probe_stride", args.probe_stride)
            eng.set_option("lanes_active", 1)
            plan.run_config(0, 2000 + rank, 0.0)
            plan.sync()
            fft_ms, fft_n = plan.probe_ms()
            lanes_ms, lanes_n = 0.0, 0
            eng.set_option("probe_stride", 0)
            eng.set_option("lanes_active", 0)
        plan.slice_loop_ms()
        # after the timed region: one all-reduce of the partial intensity sums, then the detector chain
        if world > 1:
            # the intensity sum is real: the collective moves its float view (16 MiB at 2048^2), not the float2 grid
            buf = torch.empty(m * m, device="cuda", dtype=torch.float32)
            plan.copy_intensity_real(buf.data_ptr(), 0)
            if backend == "nccl":
                dist.all_reduce(buf)
            else:
                hb = buf.cpu()
                dist.all_reduce(hb)
                buf.copy_(hb)
            torch.cuda.synchronize()
            plan.copy_intensity_real(buf.data_ptr(), 1)
        plan.end_measurement(0)
        img = plan.get_images()
        probe_lanes["ms"], probe_lanes["n"], probe_lanes["passes"] = lanes_ms, lanes_n, passes
        return dt, plan, eng, loop_ms, loop_slices, fft_ms, fft_n, bool(np.isfinite(img).all()), img

    weight = 1.0 / 32.0
    # headline: EVERY slice runs the full potential / transmission / propagation sequence (what the reference does)
    dt, plan, eng, loop_ms, loop_slices, fft_ms, fft_n, finite, img0 = timed_run(args.skip_empty)
    lanes_ms, lanes_n, pass_probe = probe_lanes["ms"], probe_lanes["n"], probe_lanes["passes"]
    m3 = plan.m3

    total_slices = world * args.steps * m3
    value = total_slices / dt
    px = m * m
    roof = None
    fused = plan.fft_backend() == 2
    if fft_n > 0:
        per_launch_s = fft_ms / fft_n * 1e-3
        table = None
        if fused:
            # Per-pass table (DESIGN 4.1's bytes per launch; dead band-limit rows / columns are neither loaded nor stored).
            # The line's roofline is quoted on the pass that LOSES most time against the roof: per-slice time x (1 - frac).
            table = pass_table(m, int(np.unique(atoms.Z).size), pass_probe, pmc_file_traffic(m))
            top = max(table, key=lambda r: r["us_per_slice"] * (1.0 - r["frac"]))
            per_launch_s = top["launch_us"] * 1e-6
            fft_n = top["launches_timed"]
            lanes_ms, lanes_n = (top["launch_us_lanes"] or 0.0) * 1e-3 * top["launches_timed_lanes"], top["launches_timed_lanes"]
            kname = f"{top['kernel']} ({top['name']} of 6 passes/slice), {live_columns(m)} of {m} kx columns live"
            alg_bytes = top["algorithmic_bytes"]
            roof_pass = top["pass"]
        else:
            # one rocFFT 2-D C2C = 2 passes x (8 B read + 8 B write) per pixel ("FFT pass 16 B/px")
            kname, alg_bytes = f"rocFFT 2-D C2C {m}x{m} (row + column kernels)", 32.0 * px
            roof_pass = None
        ach = alg_bytes / per_launch_s / 1e9
        traffic, stale = pmc_traffic(m, roof_pass) if fused else (None, False)
        cold = None
        if fused and args.hbm_cold and m in (2048, 4096) and roof_pass != 1:   # (P1' builds its rows from the atom records: no operand grid to evict)
            try:
                cold = hbm_cold_launch(m, local, roof_pass)
            except Exception:
                cold = None
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic, "traffic_stale": stale,
                "launch_us": round(per_launch_s * 1e6, 2), "launches_timed": int(fft_n),
                "launch_us_lanes": (round(lanes_ms / lanes_n * 1e3, 2) if fused and lanes_n else None),
                # the same kernel with its operands in HBM only (8 buffer sets round-robin on one stream, 1.2 GB at 2048^2:
                # beyond the 256 MiB Infinity Cache, whose hits no rocprofv3 counter of this box exposes); `frac` above is
                # measured inside the slice loop, where part of the traffic is served by that cache
                "hbm_cold": (None if cold is None else {"launch_us": round(cold, 2), "achieved": round(alg_bytes / cold / 1e3, 1),
                                                         "hbm_frac": round(alg_bytes / (cold * 1e-6) / 8e12, 4)}),
                "algorithmic_bytes_per_launch": alg_bytes,
                "passes": table,
                "timed": "HIP start/stop events of the dispatch itself (hipExtLaunchKernelGGL) on every %d-th launch during one "
                         "configuration run on lane 0 right after the timed steps (other lanes idle), once per pass class; launch_us_lanes: "
                         "the same with one configuration per lane sharing the chip, issued directly; `kernel` = the pass with the "
                         "largest per-slice time x (1 - frac)" % args.probe_stride}
    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline:
        cpu = cpu_baseline(hp, atoms, m)

    extra = None
    if args.extra_skip_run and not args.skip_empty:
        # the engine's default additionally short-cuts slices that hold no atom (t = 1 exactly): reported beside
        # the headline, never as the headline
        lanes_main, fused_main = plan.lanes(), plan.fft_backend() == 2
        plan.close(); eng.close()
        dt2, plan, eng, _, _, _, _, fin2, img1 = timed_run(1)
        rel = float(np.linalg.norm(img1 - img0) / max(np.linalg.norm(img0), 1e-30))
        extra = {"value": round(world * args.steps * m3 / dt2, 2), "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                 "image_rel_diff_vs_headline_run": rel, "finite": fin2}
    lanes_used = plan.lanes()
    plan.close()
    eng.close()
    extras = None
    if rank == 0 and world == 1 and args.extras and m == 2048:
        extras = run_extras(local)
    if roof is not None and fused:
        # the whole slice step on the engine's own byte count: P1'/2 + P2/2 + P3/2 + P4 + P5 + P6 =
        # 4 + 10 + 9.33 + 10.67 + 16 + 10.67 B per pixel and slice with the dead band-limit rows / columns not counted
        bpp = ENGINE_BYTES_PER_PX_SLICE
        roof["whole_step"] = {"bytes_per_px_slice": bpp, "achieved": round(bpp * px * (value / world) / 1e9, 1), "unit": "GB/s",
                              "frac": round(bpp * px * (value / world) / 8e12, 4),
                              "note": "all six passes of a slice at the measured rate; SURVEY 8d's fixed models beside it: "
                                      "full step (176 + 56 nZ) B/px and propagation unit 80 B/px",
                              "survey_full_step_model_GBps": round((176 + 56 * 1) * px * (value / world) / 1e9, 1),
                              "survey_propagation_unit_model_GBps": round(80 * px * (value / world) / 1e9, 1)}
    if rank == 0:
        out = {
            "metric": "slice-propagations/sec", "value": round(value, 2), "unit": "slice-propagations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3 Au cuboctahedron k={k_au} ({atoms.n} atoms), {m}x{m} wave, {m3} slices, "
                                   f"1 frozen-phonon configuration per step per GPU (of 32), mode 0",
                       "wave": [m, m], "slices": m3, "atoms": atoms.n, "configs_per_step": world,
                       "parallelism": f"configs sharded over {world} GPU(s)"},
            "lanes": lanes_used, "graph": args.graph, "skip_empty": args.skip_empty,
            "lane_slice_loop_ms_per_slice": round(loop_ms / max(loop_slices, 1), 5),
            "slice_loop": "fused LDS passes" if fused else "rocFFT + point-wise kernels",
            "roofline": roof, "cpu_baseline": cpu, "finite": finite,
            "with_empty_slice_shortcut": extra,
            "extras": extras,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


ENGINE_BYTES_PER_PX_SLICE = 60.67


def deal(rank, world, step):
    """Frozen-phonon configuration j that `rank` of `world` runs in its step `step`: rank + world * step, so that the
    ranks' steps together cover j = 0 .. world * steps - 1 exactly once (tests/test_host_cpu.py)."""
    return rank + world * step


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher: start N child processes of this script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, one GPU each) and return the worst exit code.  The
    parent never initialises the GPU; rank 0's JSON line reaches stdout through the inherited descriptor."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return _reap(procs, float(os.environ.get("FDES_BENCH_DEADLINE_S", "1500")))


def _reap(procs, deadline_s, grace_s=10.0, poll_s=0.2):
    """Wait for the ranks together (what torch.distributed.run does for its workers): the first rank that exits with an
    error, or the deadline, takes the others down - terminate(), then kill() after a grace period - so that a rank that
    died before the rendezvous does not leave its siblings blocked in init_process_group or a barrier.  Returns that first
    non-zero exit code (124 for the deadline), 0 when every rank ended cleanly."""
    import time
    t_end = time.monotonic() + deadline_s
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            return 0
        if time.monotonic() > t_end:
            print(f"bench.py: ranks still running after {deadline_s:.0f} s - stopping them", file=sys.stderr)
            rc = 124
            break
        time.sleep(poll_s)
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_kill = time.monotonic() + grace_s
    for p in procs:
        try:
            p.wait(timeout=max(0.0, t_kill - time.monotonic()))
        except Exception:
            p.kill()
            p.wait()
    return rc


def _blob_hash(path):
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


PMC_FILE = "r05_pmc.json"
# (PRE, MID, POST, transposed) of the six passes as the kernel names spell them (fft_lds.h: XfKind, MidKind)
PASS_KEYS = {1: ", 1, 9, 0, true", 2: ", 1, 2, 2, true", 3: ", 2, 12, 1, true", 4: ", 1, 4, 2, true", 5: ", 2, 5, 1, true", 6: ", 1, 6, 2, true"}
PASS_NAMES = {1: "P1' deposit + FFT_x (per slice pair)", 2: "P2 FFT_y, filter, IFFT_y (per slice pair)", 3: "P3 IFFT_x, transmission of two slices, FFT_x (per slice pair)",
              4: "P4 FFT_y, band limit, IFFT_y", 5: "P5 IFFT_x of t and psi, product, FFT_x", 6: "P6 FFT_y, propagator, IFFT_y"}
PASS_MID = {1: (1, 9, 0), 2: (1, 2, 2), 3: (2, 12, 1), 4: (1, 4, 2), 5: (2, 5, 1), 6: (1, 6, 2)}
PASS_BAND = {1: 0, 2: 0, 3: 4, 4: 1, 5: 6, 6: 1}   # fdes_bench_pass's band flags as the slice loop sets them


def live_columns(m):
    """Columns (or rows) of an m-point frequency axis inside the radial 2/3 band limit: 2 L + 1 with L the largest i, 9 i^2 <= m^2."""
    L = m // 3
    while 9 * (L + 1) ** 2 <= m * m:
        L += 1
    while 9 * L * L > m * m:
        L -= 1
    return min(m, 2 * L + 1)


def pass_bytes(m, nz):
    """Algorithmic bytes per LAUNCH of the six passes on an m x m grid with nz species: dead band-limit rows /
    columns are neither loaded nor stored.  P1', P2, P3 are launched once per slice PAIR."""
    px, lv = float(m) * m, float(m) * live_columns(m)
    return {1: 8.0 * nz * px,                      # x spectra of the deposit rows, one grid per species
            2: (8.0 + 4.0) * nz * px + 8.0 * px,  # spectra + filter table in, packed pair potential out
            3: 8.0 * px + 16.0 * lv,               # pair potential in, two transmission spectra out (live kx only)
            4: 16.0 * lv,                          # live kx rows in and out
            5: 24.0 * lv,                          # t-hat and psi-hat (live kx columns) in, product spectrum (live kx) out
            6: 16.0 * lv}


def kernel_family(m):
    return "k_wpass" if m in (1024, 2048) else ("k_pass / k_wpass" if m == 4096 else ("k_pass" if m & (m - 1) == 0 else "k_gpass"))


def probe_passes(eng, plan, rank, stride):
    """{pass class: (ms alone, launches alone, ms with every lane active, launches)}: per class one configuration per lane sharing
    the chip, then one configuration on lane 0 alone, every `stride`-th launch of that class bracketed by the dispatch's own events."""
    res = {}
    for cls in range(1, 7):
        eng.set_option("probe_pass", cls)
        plan.probe_ms()
        eng.set_option("probe_stride", stride)
        for l in range(plan.lanes()):
            plan.run_config(0, 3000 + rank + 100 * l, 0.0)
        plan.sync()
        lanes_ms, lanes_n = plan.probe_ms()
        eng.set_option("lanes_active", 1)
        plan.run_config(0, 2000 + rank, 0.0)
        plan.sync()
        ms, n = plan.probe_ms()
        eng.set_option("probe_stride", 0)
        eng.set_option("lanes_active", 0)
        res[cls] = (ms, n, lanes_ms, lanes_n)
    eng.set_option("probe_pass", 5)
    return res


def pass_table(m, nz, probe, traffic):
    """roofline.passes: one row per pass class from the probe's dispatch events."""
    by = pass_bytes(m, nz)
    rows = []
    for cls in range(1, 7):
        ms, n, lms, ln = probe[cls]
        if n <= 0:
            continue
        us = ms / n * 1e3
        per_slice = us * (0.5 if cls <= 3 else 1.0)
        rows.append({"pass": cls, "name": PASS_NAMES[cls], "kernel": f"{kernel_family(m)}<{m}{PASS_KEYS[cls]}>",
                     "launch_us": round(us, 2), "launches_timed": int(n),
                     "launch_us_lanes": (round(lms / ln * 1e3, 2) if ln else None), "launches_timed_lanes": int(ln),
                     "us_per_slice": round(per_slice, 2), "algorithmic_bytes": by[cls],
                     "achieved": round(by[cls] / (us * 1e-6) / 1e9, 1), "frac": round(by[cls] / (us * 1e-6) / 8e12, 4),
                     "traffic": traffic.get(cls)})
    return rows


def pmc_file_traffic(m):
    """{pass class: HBM bytes per launch} from the committed rocprofv3 PMC passes (profiles/r05_pmc.json, tools/profile_round.sh:
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE) - only while the git blob hashes of the sources recorded there
    (kernels, shared arithmetic, pass arguments, the engine) equal this build's; otherwise {} (stale).  PMC counters cannot be
    collected from inside this process."""
    out = {}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        for src, h in d.get("source_hash", {}).items():
            if h != _blob_hash(os.path.join(ROOT, src)):
                return {"stale": True}
        for wl in d["workloads"].values():
            for k, v in wl.items():
                for cls, key in PASS_KEYS.items():
                    if k.startswith((f"k_pass<{m},", f"k_wpass<{m},", f"k_gpass<{m},")) and key in k and "hbm_bytes_per_launch_corrected" in v:
                        out[cls] = v["hbm_bytes_per_launch_corrected"]
    except Exception:
        pass
    return out


def pmc_traffic(m, cls):
    """(HBM bytes per launch of pass `cls`, stale) - see pmc_file_traffic."""
    t = pmc_file_traffic(m)
    if t.get("stale"):
        return None, True
    return t.get(cls), False


def hbm_cold_launch(m, device, cls=5):
    """Mean launch time [us] of pass `cls` (same band bookkeeping, row padding and workgroup geometry as the slice loop) over 8
    buffer sets used round-robin on ONE stream: every launch finds its operands in HBM only (tools/bench_mall.py)."""
    import fdes_amd
    eng = fdes_amd.Engine(device, bench_band=PASS_BAND[cls], bench_pitch=32 if m == 2048 else 64, bench_serial=1,
                          pass_threads=64 if m == 2048 else 512)
    try:
        pre, mid, post = PASS_MID[cls]
        return eng.bench_pass(m, pre, mid, post, 1, 60, 8)
    finally:
        eng.close()


def run_extras(device):
    """Driver-timed figures beside the headline (C5 and the SURVEY 8d micro-benchmark as bench keys)."""
    import numpy as np
    import torch
    import fdes_amd
    from tests import specimens
    out = {}
    # BASELINE config 5 at full size: Au cuboctahedron k = 60 (738 221 atoms), 4096^2 wave, 512 slices, every slice the
    # full sequence; 2 untimed + 4 timed frozen-phonon configurations of the 16
    hp, at = specimens.case_c5()
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(device, skip_empty=0)
    pl = eng.plan(hp, at)
    pl.begin_measurement(0)
    for j in range(2):
        pl.run_config(0, 100 + j, 0.0)
    pl.sync()
    torch.cuda.synchronize()
    n = 4
    rates = []
    for rep in range(5):   # min / median / max inside one process: bounds the run-to-run spread of this figure
        t0 = time.perf_counter()
        for j in range(n):
            pl.run_config(0, j, 1.0 / 80)
        pl.sync()
        torch.cuda.synchronize()
        rates.append(n * pl.m3 / (time.perf_counter() - t0))
    table = pass_table(4096, 1, probe_passes(eng, pl, 0, 8), pmc_file_traffic(4096))
    pl.end_measurement(0)
    img = pl.get_images()
    rate = float(np.median(rates))
    px = 4096 * 4096
    out["c5"] = {"workload": f"C5 Au cuboctahedron k=60 ({at.n} atoms), 4096x4096 wave, {pl.m3} slices, {n} configurations per repetition, 5 repetitions",
                 "value": round(rate, 1), "min": round(min(rates), 1), "median": round(rate, 1), "max": round(max(rates), 1),
                 "unit": "slice-propagations/s", "ms_per_step": round(pl.m3 / rate * 1e3, 2),
                 "lanes": pl.lanes(), "finite": bool(np.isfinite(img).all()),
                 "whole_step": {"bytes_per_px_slice": ENGINE_BYTES_PER_PX_SLICE,
                                "achieved": round(ENGINE_BYTES_PER_PX_SLICE * px * rate / 1e9, 1), "unit": "GB/s",
                                "frac": round(ENGINE_BYTES_PER_PX_SLICE * px * rate / 8e12, 4)},
                 "passes": table}
    pl.close()
    eng.close()
    # Grid lengths that are not powers of two: m = 2 nx of a .qsc (src/rwQsc.cu:943-948).  nx = 1500 -> 3000 points (round 4: on the
    # fused mixed-radix passes with compiled-in kernels; until round 3: rocFFT + point-wise kernels); nx = 1144 -> 2288 = 16 x 11 x 13
    # points (round 5: no compiled-in kernels - the row passes are compiled for the length by hipRTC when the plan is created,
    # gen_jit.cpp; `plan_creation_s` holds that compilation, or the read from the directory cache).  C3 specimen, 32 slices,
    # 2 untimed + 6 timed configurations, every slice the full sequence.  nx = 2500 -> 5000 points: rows beyond 4096 points exist on the
    # fused loop as run-time-compiled kernels only (round 5)
    def grid_extra(nx):
        hp, at = specimens.case_c3(k=30, n=nx, dn=nx // 2, m3=32, frPh=32)
        fdes_amd.consistent(hp)
        eng = fdes_amd.Engine(device, skip_empty=0)
        t0 = time.perf_counter()
        pl = eng.plan(hp, at)
        t_plan = time.perf_counter() - t0
        pl.begin_measurement(0)
        for j in range(2):
            pl.run_config(0, 100 + j, 0.0)
        pl.sync()
        torch.cuda.synchronize()
        n = 6
        rates = []
        for rep in range(5):
            t0 = time.perf_counter()
            for j in range(n):
                pl.run_config(0, j, 1.0 / 160)
            pl.sync()
            torch.cuda.synchronize()
            rates.append(n * pl.m3 / (time.perf_counter() - t0))
        pl.end_measurement(0)
        img = pl.get_images()
        rate = float(np.median(rates))
        m = 2 * nx
        r = {"workload": f"C3 specimen ({at.n} atoms) on a {m}x{m} wave (m = 2 nx, nx = {nx}), {pl.m3} slices, {n} configurations per repetition, 5 repetitions",
             "value": round(rate, 1), "min": round(min(rates), 1), "median": round(rate, 1), "max": round(max(rates), 1),
             "unit": "slice-propagations/s", "lanes": pl.lanes(),
             "slice_loop": "fused LDS passes" if pl.fft_backend() == 2 else "rocFFT + point-wise kernels",
             "run_time_compiled_axes": pl.jit_kernels(), "plan_creation_s": round(t_plan, 2),
             "whole_step_frac": round(ENGINE_BYTES_PER_PX_SLICE * float(m) * float(m) * rate / 8e12, 4),
             "finite": bool(np.isfinite(img).all())}
        pl.close()
        eng.close()
        return r
    out["qsc_sized_grid"] = grid_extra(1500)
    out["run_time_compiled_grid"] = grid_extra(1144)
    out["grid_beyond_4096"] = grid_extra(2500)   # 5000 = 10 x 25 x 20 points: one two-row tile per CU, kernels compiled at plan creation (rocFFT loop before: 0.20 k)
    # BASELINE config 4 at full size: SrTiO3 beam-tilt series, 64 tilts x 8 frozen-phonon configurations, 1024^2 wave, 40
    # slices, every slice the full sequence; engine defaults (lanes of gangs); one untimed job, one timed job
    hp, at = specimens.case_c4()
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(device, skip_empty=0)
    pl = eng.plan(hp, at)
    count = max(hp.c.frPh, 1)
    def c4_job():
        for k in range(hp.c.n3):
            pl.begin_measurement(k)
            for j in range(count):
                pl.run_config(k, j, 1.0 / count)
            pl.end_measurement(k)
        pl.sync()
        torch.cuda.synchronize()
    c4_job()
    secs = []
    for rep in range(5):
        t0 = time.perf_counter()
        c4_job()
        secs.append(time.perf_counter() - t0)
    img = pl.get_images()
    tot = hp.c.n3 * count * pl.m3
    dt = float(np.median(secs))
    out["c4"] = {"workload": f"C4 SrTiO3 9x9x20 cells ({at.n} atoms, 3 species), 1024x1024 wave, {pl.m3} slices, {hp.c.n3} tilts x {count} configurations, 5 repetitions of the whole job",
                 "value": round(tot / dt, 1), "min": round(tot / max(secs), 1), "median": round(tot / dt, 1), "max": round(tot / min(secs), 1),
                 "unit": "slice-propagations/s", "seconds": round(dt, 4),
                 "lanes": pl.lanes(), "gang": pl.gang(), "finite": bool(np.isfinite(img).all())}
    pl.close()
    eng.close()
    # the reference's own shipped example (bin/dataFDES.cnf, a data fixture under tests/golden: Au-309, 320^2 wave, 25 specimen
    # tilts, 12 slices -> 132 sub-slices, dose noise) through the boundary call with host buffers: what the CLI runs
    ex = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "dataFDES_bin.cnf")
    if os.path.exists(ex):
        hp, at = fdes_amd.read_cnf(ex)
        q, _ = fdes_amd.sub_sliced(hp)
        eng = fdes_amd.Engine(device)
        eng.build_measurements(hp, at)
        t0 = time.perf_counter()
        img = eng.build_measurements(hp, at)["image"]
        dt = time.perf_counter() - t0
        eng.close()
        out["reference_example"] = {"workload": f"bin/dataFDES.cnf: Au-309, 320x320 wave, {hp.c.n3} tilts x {q.c.m3} sub-slices, fdes_build_measurements with host pointers",
                                    "seconds": round(dt, 4), "value": round(hp.c.n3 * q.c.m3 / dt, 1), "unit": "slice-propagations/s",
                                    "finite": bool(np.isfinite(img).all())}
    # the boundary call itself with HOST buffers (fdes_build_measurements: atoms in over PCIe, plan creation, tables,
    # 8 frozen-phonon configurations of the headline specimen, detector chain, image out over PCIe): the PCIe- and
    # setup-inclusive rate, never the headline
    hp, at = specimens.case_c3(frPh=8)
    fdes_amd.consistent(hp)
    eng = fdes_amd.Engine(device, skip_empty=0)
    eng.build_measurements(hp, at)   # first call: code objects, rocPRIM temporaries
    t0 = time.perf_counter()
    img = eng.build_measurements(hp, at)["image"]
    dt = time.perf_counter() - t0
    eng.close()
    out["end_to_end_host_buffers"] = {"workload": f"fdes_build_measurements, C3 specimen ({at.n} atoms), 2048x2048 wave, 256 slices, 8 configurations, "
                                                  "host pointers in and out", "seconds": round(dt, 4),
                                      "value": round(8 * 256 / dt, 1), "unit": "slice-propagations/s", "finite": bool(np.isfinite(img).all())}
    # SURVEY 8d micro-benchmark: psi <- F^-1[P F[t psi]] on device-resident random psi and unit-modulus t, one stream,
    # 256 units; priced with the survey's fixed 80 B/px model
    out["propagation_unit"] = {}
    for m in (2048, 4096):
        hp, at = specimens.case_c3(k=2, n=m // 2, dn=m // 4, m3=2, frPh=0)
        fdes_amd.consistent(hp)
        eng = fdes_amd.Engine(device, lanes=1)
        pl = eng.plan(hp, at)
        g = torch.Generator(device="cuda").manual_seed(0)
        psi = torch.randn(1, m, m, 2, device="cuda", generator=g)
        ph = (torch.rand(1, m, m, device="cuda", generator=g) * 2 - 1) * 3.14159265
        t = torch.stack([torch.cos(ph), torch.sin(ph)], -1).contiguous()
        torch.cuda.synchronize()
        for _ in range(8):
            pl.propagate_dev(psi.data_ptr(), t.data_ptr(), 1, True)
        pl.sync()
        reps = 256
        t0 = time.perf_counter()
        for _ in range(reps):
            pl.propagate_dev(psi.data_ptr(), t.data_ptr(), 1, True)
        pl.sync()
        dt = time.perf_counter() - t0
        r = reps / dt
        # bytes the three passes of the unit really move (dense caller grids in and out, dead kx columns of the two inner grids
        # skipped): t psi -> F: 16 r + 8 f w; F -> E: 16 f; E -> psi: 8 f r + 8 w, f = live / m  =>  (24 + 32 f) B/px
        bpp = 24.0 + 32.0 * live_columns(m) / m
        out["propagation_unit"][str(m)] = {"units_per_s": round(r, 1), "us_per_unit": round(dt / reps * 1e6, 2),
                                           "engine_bytes_per_px": round(bpp, 2), "achieved_GBps": round(bpp * m * m * r / 1e9, 1),
                                           "frac": round(bpp * m * m * r / 8e12, 4),
                                           "survey_80B_model_GBps": round(80 * m * m * r / 1e9, 1),
                                           "note": "frac = the unit's own bytes / 8 TB/s; survey_80B_model_GBps prices the unit with SURVEY 8d's fixed "
                                                   "80 B/px (a model for comparing implementations, NOT a roofline fraction: the engine moves fewer bytes)"}
        pl.close()
        eng.close()
    return out


def cpu_baseline(hp, atoms, m):
    """CPU oracle (kind "port": our C restatement of the reference's CUDA algorithm, float32, OpenMP) on a
    bounded sample of the same workload: the first `ns` slices of configuration (0, 0)."""
    from tests import oracle_py
    q, _ = oracle_py.sub_sliced(hp)
    cores = oracle_py.get_threads()
    # about 15 s of CPU work: the oracle runs ~13 slices/s at 2048^2 and ~3 slices/s at 4096^2 on 16 threads
    ns = min(int(q.c.m3), 48 if m >= 4096 else (192 if m >= 2048 else 512))
    oracle_py.wave(q, atoms, 0, 0, nslices=1)  # warm (FFT plans, page faults)
    t0 = time.perf_counter()
    oracle_py.wave(q, atoms, 0, 0, nslices=ns)
    dt = time.perf_counter() - t0
    return {"value": round(ns / dt, 3), "unit": "slice-propagations/s", "cores": cores, "kind": "port",
            "sample": f"first {ns} slices of one {m}x{m} configuration ({dt:.1f} s incl. jitter + incoming wave)"}


if __name__ == "__main__":
    main()
