"""Sharding of the independent (k = measurement, j = frozen-phonon configuration) wave propagations
of buildMeasurements (src/crystalMaker.cu:324-367) over ranks, one process per GPU.

The only coupling between configurations is the sum over j into I per k (:353,359,365).  The flattened
(k, j) index space is block-partitioned over ranks; every rank accumulates its share of I[k] with the
global weight 1/count, the partial sums of a k that spans several ranks are sum-reduced
(`reduce_fn`, RCCL all-reduce in bench.py, gloo in the CPU tests), and the rank that owns k applies
addNoiseAndMtf (:372).  RNG is keyed on (k, j), so results do not depend on the partition.
"""
import numpy as np


def partition(n3, count, world, rank):
    """Contiguous block of the flattened (k, j) list for `rank`: [(k, j), ...] in reference order."""
    total = n3 * count
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return [(i // count, i % count) for i in range(lo, hi)]


def owners(n3, count, world):
    """owner[k] = rank holding configuration (k, 0); ranks_of[k] = set of ranks holding any (k, j)."""
    own, ranks_of = {}, {k: set() for k in range(n3)}
    for r in range(world):
        for (k, j) in partition(n3, count, world, r):
            ranks_of[k].add(r)
            if j == 0:
                own[k] = r
    return own, ranks_of


def run_sharded(plan, n3, count, rank=0, world=1, reduce_fn=None):
    """Drive `plan` (begin_measurement / run_config / end_measurement) over this rank's share.
    reduce_fn(plan, k) must sum-reduce the plan's running intensity over all ranks (called by every
    rank, for every k that is split across ranks, in ascending k).  Returns the list of k whose
    images this rank finalised."""
    mine = partition(n3, count, world, rank)
    own, ranks_of = owners(n3, count, world)
    if count == 1 and hasattr(plan, "run_measurements"):
        # one configuration per measurement: no k is split, this rank's measurements are complete by themselves and go
        # through the engine in one call (gangs of measurements)
        ks = [k for (k, _) in mine]
        if ks:
            plan.run_measurements(ks)
        return ks
    weight = float(np.float32(1.0) / np.float32(count))  # alpha of src/crystalMaker.cu:302-304 is a float32
    done = []
    by_k = {}
    for (k, j) in mine:
        by_k.setdefault(k, []).append(j)
    for k in range(n3):
        split = len(ranks_of[k]) > 1
        if k in by_k:
            plan.begin_measurement(k)
            for j in by_k[k]:
                plan.run_config(k, j, weight)
        elif split and world > 1:
            plan.begin_measurement(k)  # contributes zeros to the reduction
        if split and world > 1:
            reduce_fn(plan, k)
        if own[k] == rank:
            plan.end_measurement(k)
            done.append(k)
    return done
