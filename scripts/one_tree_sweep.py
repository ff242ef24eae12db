"""BASELINE config 2 read literally -- ONE position, one `go` -- how full do the batches get, and which lane shape serves one tree best?
(VERDICT r04 #4.)  One shared tree, L lanes x batch B, k collectors per lane (B / k leaves each), 1600 and 6400 simulations, crazyhouse
opening set one position at a time; the many-trees figure (32 trees, 2 lanes x 256) on the same box beside it.
  python scripts/one_tree_sweep.py [precision] [quick]
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crazyara_amd import netfile, openings, replicas, rise_config, search, searchbench  # noqa: E402
from crazyara_amd.neuralnetapi import HipAPI  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "float16p8"
quick = len(sys.argv) > 2
cfg = rise_config.rise_v2_config(19, 34, 81)
sd = rise_config.make_state_dict(cfg, seed=31, stress=True)
d = tempfile.mkdtemp(prefix="cra_onetree_")
netfile.export_rise(os.path.join(d, f"{cfg.name}-v1.0.cranet"), cfg, sd, input_version="1.0")
cz = [(f, False, "crazyhouse") for f in openings.crazyhouse_opening_set()]
threads = min(16, replicas.available_cpus())
rows = []


def leg(name, batch, lanes, quota, sims, trees, shared):
    st = search.default_settings(mode=0, version_major=1, batch_size=quota)
    nets = [HipAPI(0, batch, d, precision) for _ in range(lanes)]
    th = min(threads, shared * lanes) if shared else min(threads, max(1, trees // max(1, lanes)))
    r = searchbench.timed_search_leg(st, nets, cz, trees, sims, max(1, th), min_seconds=0.7, repeats=3, shared_collectors=shared)
    for n in nets:
        n.close()
    row = {"leg": name, "lanes": lanes, "batch": batch, "collectors_per_lane": shared or None, "leaves_per_collector": quota, "simulations": sims,
           "nodes_per_sec": r["mcts_nodes_per_sec"], "fill": r["avg_batch_fill"], "evals_per_sec": r["mcts_nn_evals_per_sec"], "host_threads": th}
    rows.append(row)
    print(row, flush=True)


leg("many trees (32 x 1600, 2 lanes x 256)", 256, 2, 16, 1600, 32, 0)
shapes = [(2, 256, 8), (3, 256, 8), (4, 128, 4), (6, 128, 4), (4, 128, 8), (8, 64, 2), (8, 64, 4), (12, 64, 2), (16, 32, 2)]
if quick:
    shapes = [(2, 256, 8), (4, 128, 4), (8, 64, 2)]
for sims in (1600, 6400):
    for lanes, batch, k in shapes:
        leg(f"one tree, {lanes} lanes x {batch}, {k} collectors per lane", batch, lanes, batch // k, sims, 1, k)
print("RESULT " + json.dumps({"precision": precision, "rows": rows}))
