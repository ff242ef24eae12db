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


def reduce_game_leg(local: Optional[dict], dist=None, device: Optional[torch.device] = None, world: int = 1) -> dict:
    """One game leg (self-play / arena) of a sharded run: this rank's {games, moves, seconds, mcts_nodes_per_sec} -> the whole job's
    games/min and nodes/sec (SUM of games, moves and nodes, MAX of seconds: one all_reduce pair, rl_loop.py:60 / selfplay.cpp:339-351
    sum their per-device files the same way) with every rank's own games/min beside it (one all_gather of two scalars).  `local` None:
    a rank that played nothing (a rehearsal rank) still takes part in both collectives."""
    r = local or {"games": 0, "moves": 0, "seconds": 1e-9, "mcts_nodes_per_sec": 0.0}
    nodes = float(r["mcts_nodes_per_sec"]) * float(r["seconds"])
    games_t, sec_t, ex = reduce_stats(ReplicaStats(units=float(r["games"]), seconds=float(r["seconds"]), extra=(float(r["moves"]), nodes)), dist, device)
    mine = torch.tensor([r["games"] / r["seconds"] * 60 if r["games"] else 0.0, float(r["games"])], dtype=torch.float64, device=device)
    allr = [mine]
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    return {"games_per_min": round(games_t / sec_t * 60, 1), "games": int(games_t), "moves": int(ex[0]), "seconds": round(sec_t, 3),
            "mcts_nodes_per_sec": round(ex[1] / sec_t, 1),
            "per_rank_games_per_min": [round(float(v[0]), 1) for v in allr], "per_rank_games": [int(v[1]) for v in allr]}


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


# ---- host budget of a rank: the search pool of one GPU wants its collector threads on cores near that GPU ------------------------
def _parse_cpulist(text: str):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def gpu_numa_cpus(device_index: int):
    """CPUs of the NUMA node the GPU hangs off (sysfs: the PCI device's numa_node -> that node's cpulist), or None if unknown."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            return _parse_cpulist(f.read())
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None


def plan_rank_cpus(allowed, world: int, local_rank: int, numa_cpus_of_rank=None):
    """The CPUs rank `local_rank` of `world` ranks on this node should run its threads on: an equal, disjoint slice of the allowed
    set.  numa_cpus_of_rank[r] = the CPUs of the NUMA node of rank r's GPU (every rank sees all GPUs of the node, so every rank
    computes the same table without talking): ranks whose GPUs share a node split that node's allowed CPUs in rank order, as long as
    this gives nobody less than the plain even split would."""
    allowed = sorted(allowed)
    if world <= 1:
        return allowed
    per = max(1, len(allowed) // world)
    if numa_cpus_of_rank and len(numa_cpus_of_rank) == world and all(n for n in numa_cpus_of_rank):
        groups = {}
        for r, n in enumerate(numa_cpus_of_rank):
            groups.setdefault(frozenset(n), []).append(r)
        ok = True
        plan = {}
        for node, ranks in groups.items():
            near = [c for c in allowed if c in node]
            share = len(near) // len(ranks)
            if share < per:
                ok = False
                break
            for k, r in enumerate(ranks):
                plan[r] = near[k * share:(k + 1) * share]
        if ok:
            return plan[local_rank]
    return allowed[local_rank * per:(local_rank + 1) * per] or allowed[-per:]


def pin_rank_to_cpus(world: int, local_rank: int, use_gpu_topology: bool = True):
    """Restrict this process (and the threads it starts later) to its slice of the node's CPUs.  Returns (cpus, threads) where
    `threads` is what the search pool should be given; warns on stderr when a rank gets fewer than 8."""
    import os
    import sys
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return None, available_cpus()
    budget = available_cpus()                                   # a cgroup quota may be tighter than the affinity mask
    numa = None
    if use_gpu_topology and world > 1 and torch.cuda.is_available() and torch.cuda.device_count() >= world:
        numa = [gpu_numa_cpus(r) for r in range(world)]
    cpus = plan_rank_cpus(allowed, world, local_rank, numa)
    if world > 1:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    threads = max(1, min(len(cpus), budget // max(1, world)))
    if threads < 8:
        print(f"[crazyara_amd] rank {local_rank}: only {threads} host threads for the search pool of this GPU "
              f"({len(allowed)} CPUs allowed, quota {budget}, {world} ranks): nodes/sec will be host-bound", file=sys.stderr)
    return cpus, threads
