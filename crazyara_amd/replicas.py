"""Multi-GPU = independent replicas: one process per GPU, no data-path collective.

The only communication is the reduction of the per-rank statistics at the end of a run -- the MI355X restatement of how the
reference aggregates its process-per-GPU self-play farm (engine/src/rl/rl_loop.py:60, selfplay.cpp:339-351 write per-device
files that a driver sums).  Works on any torch.distributed backend (RCCL on the GPUs, gloo in the CPU tests)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import torch


@dataclass
class ReplicaStats:
    units: float            # work units this rank processed (evaluations, nodes, ...); may be a vector via `extra`
    seconds: float          # wall time of this rank's timed region
    extra: Sequence[float] = ()


def shard_items(n_items: int, rank: int, world: int):
    """Static round-robin partition of independent work items (games, opening positions): item g -> rank g % world."""
    return list(range(rank, n_items, world))


def reduce_stats(local: ReplicaStats, dist=None, device: Optional[torch.device] = None):
    """Whole-job aggregate: SUM of units (and extras) over ranks, MAX of seconds.  Returns (units, seconds, extras)."""
    vec = torch.tensor([local.units, *local.extra], dtype=torch.float64, device=device)
    sec = torch.tensor([local.seconds], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(sec, op=dist.ReduceOp.MAX)
    vec = vec.cpu()
    return float(vec[0]), float(sec.cpu()[0]), [float(v) for v in vec[1:]]


def throughput(local: ReplicaStats, dist=None, device: Optional[torch.device] = None) -> float:
    units, seconds, _ = reduce_stats(local, dist, device)
    return units / seconds if seconds > 0 else 0.0


def available_cpus() -> int:
    """CPUs this process can really use: the affinity mask, cut down by a cgroup CPU quota when there is one (containers report
    the host's count in os.cpu_count()).  The search pool's workers spin while a search runs, so asking for more threads than
    this starves the thread that drives the GPU."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:     # cgroup v1
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def cgroup_throttled_usec():
    """Microseconds this container has spent throttled by its CPU quota so far (cgroup v2 cpu.stat), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            for line in f:
                k, _, v = line.partition(" ")
                if k == "throttled_usec":
                    return int(v)
    except (OSError, ValueError):
        pass
    return None
