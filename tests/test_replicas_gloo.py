"""N > 1 path on CPU: world-size-2 gloo run of the replica statistics reduction bench.py uses on RCCL."""
import os

import pytest
import socket

import torch
import torch.multiprocessing as mp

from crazyara_amd import replicas


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = replicas.shard_items(11, rank, world)
    local = replicas.ReplicaStats(units=100.0 * (rank + 1), seconds=0.5 + 0.25 * rank, extra=(float(len(items)), 7.0 * rank))
    units, seconds, extra = replicas.reduce_stats(local, dist)
    tput = replicas.throughput(local, dist)
    dist.barrier()
    dist.destroy_process_group()
    out[rank] = (units, seconds, extra, tput, items)


def test_two_rank_reduction_and_sharding():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        units, seconds, extra, tput, _ = out[r]
        assert units == 300.0 and seconds == 0.75          # SUM of units, MAX of seconds: same on every rank
        assert extra == [11.0, 7.0]                        # every item owned by exactly one rank
        assert abs(tput - 400.0) < 1e-9
    assert sorted(out[0][4] + out[1][4]) == list(range(11)) and not set(out[0][4]) & set(out[1][4])


def _game_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # configs 4 and 5 of a sharded bench run (bench.py --gpus N): rank 0 played 8 self-play games in 2 s, rank 1 8 games in 4 s; in the arena
    # leg rank 1 is a rehearsal rank that played nothing (it must still take part in the collectives)
    c4 = {"games": 8, "moves": 400 + 40 * rank, "seconds": 2.0 * (rank + 1), "mcts_nodes_per_sec": 1000.0 * (rank + 1)}
    c5 = {"games": 32, "moves": 1600, "seconds": 3.0, "mcts_nodes_per_sec": 5000.0} if rank == 0 else None
    r4 = replicas.reduce_game_leg(c4, dist, None, world)
    r5 = replicas.reduce_game_leg(c5, dist, None, world)
    dist.barrier()
    dist.destroy_process_group()
    out[rank] = (r4, r5)


def test_two_ranks_reduce_the_sharded_game_legs():
    """bench.py --gpus N runs BASELINE configs 4 (chess960 self-play) and 5 (3check / KOTH arena) sharded over the ranks; the whole-job
    record is SUM of games / moves / nodes over MAX of seconds, with every rank's own games/min beside it."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_game_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        r4, r5 = out[r]
        assert r4["games"] == 16 and r4["moves"] == 840 and r4["seconds"] == 4.0
        assert r4["games_per_min"] == 240.0                              # 16 games in the slowest rank's 4 s
        assert r4["mcts_nodes_per_sec"] == (1000.0 * 2.0 + 2000.0 * 4.0) / 4.0
        assert r4["per_rank_games_per_min"] == [240.0, 120.0] and r4["per_rank_games"] == [8, 8]
        assert r5["games"] == 32 and r5["per_rank_games"] == [32, 0] and r5["per_rank_games_per_min"] == [640.0, 0.0]
        assert r5["games_per_min"] == 640.0 and r5["mcts_nodes_per_sec"] == 5000.0
    # one rank: the identity
    solo = replicas.reduce_game_leg({"games": 4, "moves": 10, "seconds": 2.0, "mcts_nodes_per_sec": 50.0})
    assert solo["games_per_min"] == 120.0 and solo["per_rank_games"] == [4] and solo["mcts_nodes_per_sec"] == 50.0


def test_single_rank_is_identity():
    local = replicas.ReplicaStats(units=42.0, seconds=2.0, extra=(1.0,))
    assert replicas.reduce_stats(local) == (42.0, 2.0, [1.0])
    assert replicas.throughput(local) == 21.0


def test_host_cpu_probes():
    """bench.py sizes the search pool by the CPUs the process may really use (affinity mask cut by a cgroup quota) and reports how long
    the container was throttled during the search; both probes must degrade to sane values on any host."""
    from crazyara_amd import replicas
    n = replicas.available_cpus()
    assert isinstance(n, int) and 1 <= n <= (os.cpu_count() or 1)
    t = replicas.cgroup_throttled_usec()
    assert t is None or (isinstance(t, int) and t >= 0)


def test_rank_cpu_plan_is_a_disjoint_partition():
    """Host budget per rank (bench.py pins each rank's search threads): equal disjoint slices of the allowed CPUs; with the GPUs' NUMA
    nodes known, each rank's slice lies on its GPU's node as long as nobody gets less than the plain split."""
    allowed = list(range(256))
    n0 = set(range(0, 64)) | set(range(128, 192))
    n1 = set(range(64, 128)) | set(range(192, 256))
    numa = [n0] * 4 + [n1] * 4
    plans = [replicas.plan_rank_cpus(allowed, 8, r, numa) for r in range(8)]
    assert all(len(p) == 32 for p in plans)
    assert len(set().union(*map(set, plans))) == 256                                  # disjoint and complete
    assert all(set(plans[r]) <= numa[r] for r in range(8))                            # near the rank's GPU
    flat = [replicas.plan_rank_cpus(range(16), 8, r) for r in range(8)]               # the 16-CPU slice of a GPU box: 2 per rank
    assert [len(p) for p in flat] == [2] * 8 and sorted(sum(flat, [])) == list(range(16))
    tight = [replicas.plan_rank_cpus(range(16), 8, r, numa) for r in range(8)]        # topology known but no room on one node: even split
    assert sorted(sum(tight, [])) == list(range(16))
    assert replicas.plan_rank_cpus(range(10), 1, 0) == list(range(10))


def _pin_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    before = sorted(os.sched_getaffinity(0))
    cpus, threads = replicas.pin_rank_to_cpus(world, rank, use_gpu_topology=False)
    after = sorted(os.sched_getaffinity(0))
    mine = torch.tensor([float(threads), 1000.0 * (rank + 1)], dtype=torch.float64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)                                    # the per-rank nodes/sec report of bench.py --gpus N
    dist.barrier()
    dist.destroy_process_group()
    out[rank] = (before, list(cpus), after, threads, [list(map(float, v)) for v in allr])


def test_two_ranks_split_the_host_threads_and_report_per_rank():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_pin_worker, args=(world, port, out), nprocs=world, join=True)
    b0, c0, a0, t0, g0 = out[0]
    b1, c1, a1, t1, g1 = out[1]
    if len(b0) >= 2:
        assert a0 == sorted(c0) and a1 == sorted(c1) and not set(c0) & set(c1)        # each rank is pinned to its own slice
        assert set(c0) | set(c1) <= set(b0)
    assert t0 >= 1 and t1 >= 1 and t0 <= max(1, len(b0) // 2)
    assert g0 == g1 and [v[1] for v in g0] == [1000.0, 2000.0]


def test_bench_started_plainly_with_gpus_2_becomes_two_ranks():
    """`python bench.py --gpus 2` (no launcher around it: how the driver starts the N > 1 runs) re-executes itself under
    torch.distributed.run with one rank per GPU.  On this GPU-less host both ranks must come up and refuse loudly -- the hot path has
    no CPU fallback -- instead of the launch failing before any rank exists."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU box: the ranks would run")
    assert r.returncode != 0
    # the launcher ends the other rank (SIGTERM) as soon as the first one has failed: one refusal is certain, the second rank's may be
    # cut off -- but the launcher's report names both ranks
    assert 1 <= r.stdout.count("bench.py needs a GPU") <= 2, r.stdout[-2000:]
    assert "local_rank: 0" in r.stdout and "local_rank: 1" in r.stdout, r.stdout[-2000:]
