"""N > 1 path on CPU: world-size-2 gloo run of the replica statistics reduction bench.py uses on RCCL."""
import os
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
