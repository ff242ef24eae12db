"""Round 6: single-`go` search legs on small-batch nets (the split-board forward): BASELINE config 1 (RISEv2-7, batch 8, ONE tree, 800 simulations)
with 1 / 2 lanes, and RISEv2-19 at Batch_Size 16 (the reference's default, optionsuci.cpp:69-81) -- nodes/s per precision.
usage: python scripts/small_batch_search.py [precisions=float16x3,float16x3-1wg,float16]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crazyara_amd import netfile, openings, rise_config, search, searchbench  # noqa: E402
from crazyara_amd.neuralnetapi import HipAPI  # noqa: E402

precs = (sys.argv[1] if len(sys.argv) > 1 else "float16x3,float16x3-1wg,float16").split(",")
cz = [(f, False, "crazyhouse") for f in openings.crazyhouse_opening_set()]
for blocks, batch, sims, lanes_list in ((7, 8, 800, (1, 2)), (19, 16, 1600, (1, 2)), (19, 1, 200, (1,))):
    cfg = rise_config.rise_v2_config(blocks, 34, 81)
    sd = rise_config.make_state_dict(cfg, seed=31, stress=True)
    d = tempfile.mkdtemp()
    netfile.export_rise(os.path.join(d, f"{cfg.name}-v1.0.cranet"), cfg, sd, input_version="1.0")
    for prec in precs:
        for lanes in lanes_list:
            nets = [HipAPI(0, batch, d, prec) for _ in range(lanes)]
            st = search.default_settings(mode=0, version_major=1, batch_size=batch)
            r = searchbench.timed_search_leg(st, nets, cz, 1, sims, 1 if lanes == 1 else 2, min_seconds=1.0, repeats=3, shared_collectors=1 if lanes > 1 else 0)
            for n in nets:
                n.close()
            print(f"RISEv2-{blocks} batch {batch} {sims} simulations, {lanes} lane(s), {prec}: {r['mcts_nodes_per_sec']:.0f} nodes/s "
                  f"(min {r['nodes_per_sec_min']:.0f} max {r['nodes_per_sec_max']:.0f}), fill {r.get('avg_batch_fill')}", flush=True)
