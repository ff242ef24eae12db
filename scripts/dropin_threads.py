"""Why does the reference's own search on HipAPI nets lose going from 4 to 8 SearchThreads?  

The reference's MCTSAgent + SearchThreads (oracle/_ref/libcrazyara_ref_hip_release.so, compiled from /root/reference with its Release
flags) on HipAPI nets, Batch_Size 256, crazyhouse openings: `Threads` x `go simulations N`.  Per cell: nodes/s, host CPU seconds per wall
second (user + system of the process) and the cgroup's throttled time (the reference always evaluates whole batches: a predict() costs one
forward whatever it holds, and Threads x Batch_Size leaves are in flight on the ONE tree of the go).
A MEASUREMENT script like bench.py's dropin leg: the product path never runs this code.   python scripts/dropin_threads.py [precision]
"""
import json
import os
import resource
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crazyara_amd import netfile, openings, replicas, rise_config, search  # noqa: E402
from oracle import ref_mcts  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "float16p8"
ref_mcts.load_hip(release=True)
cfg = rise_config.rise_v2_config(19, 34, 81)
sd = rise_config.make_state_dict(cfg, seed=2024, stress=True)
d = tempfile.mkdtemp(prefix="cra_dropin_")
netfile.export_rise(os.path.join(d, f"{cfg.name}-v1.0.cranet"), cfg, sd, input_version="1.0")
fens = openings.crazyhouse_opening_set()[::7][:10]
st = search.default_settings(mode=0, version_major=1, batch_size=256)
out = {"precision": precision, "batch_size": 256, "host_cpus": replicas.available_cpus(), "cells": {}}
for sims in (1600, 6400, 25600):
    for th in (1, 2, 4, 8):
        agent = ref_mcts.RefAgent(st, hip_model_dir=d, device_id=0, precision=precision, threads=th, release=True)
        agent.set_position(fens[0], False, "crazyhouse")
        agent.go(simulations=400)
        ru0, thr0, t0 = resource.getrusage(resource.RUSAGE_SELF), replicas.cgroup_throttled_usec(), time.perf_counter()
        nodes = 0
        use = fens if sims <= 6400 else fens[:4]
        for f in use:
            agent.set_position(f, False, "crazyhouse")
            agent.go(simulations=sims)
            nodes += agent.root_info()["node_count"]
        el = time.perf_counter() - t0
        ru1, thr1 = resource.getrusage(resource.RUSAGE_SELF), replicas.cgroup_throttled_usec()
        agent.close()
        cell = {"nodes_per_sec": round(nodes / el, 1), "leaves_in_flight": th * 256, "simulations_per_go": sims,
                "host_cpu_seconds_per_wall_second": round(((ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)) / el, 2),
                "cgroup_throttled_ms": None if thr0 is None or thr1 is None else round((thr1 - thr0) / 1e3, 1)}
        out["cells"][f"sims{sims}_threads{th}"] = cell
        print(f"simulations {sims:6d} Threads {th}: {cell}", flush=True)
print("RESULT " + json.dumps(out))
