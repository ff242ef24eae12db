"""Timed search legs for bench.py: nodes/sec of the leaf collector + evaluator lanes on a fixed position set.

The metric is the reference's: nps = (nodes - nodesPreSearch) / elapsed with nodes = root visits - free visits
(engine/src/evalinfo.cpp:73-80, node.cpp:1303-1306), summed over the trees of the pool.  One *round* = every tree restarts from
its next position of the set and is searched to the simulation limit (one `go` per tree); rounds repeat until the timed region
(the sum of the pool's own run times; resets between rounds are not timed) reaches `min_seconds`, and the whole leg is repeated
`repeats` times so that a scheduler hiccup shows up as spread instead of as the number."""
from __future__ import annotations

import statistics
from typing import List, Sequence, Tuple

from . import replicas, search


def timed_search_leg(st, nets: Sequence, positions: List[Tuple[str, bool, str]], trees: int, simulations: int, threads: int,
                     min_seconds: float = 1.0, repeats: int = 3, warmup_simulations: int = 200, offset: int = 0,
                     shared_collectors: int = 0) -> dict:
    """positions: (fen, is960, variant) triples; tree i of round r starts from positions[(offset + r * trees + i) % len].
    shared_collectors >= 1: every tree is searched by that many collectors in every lane (SearchThreads sharing a tree)."""
    pool = search.SearchPool(st, net_a=nets[0], net_b=nets[1] if len(nets) > 1 else None)
    for n in nets[2:]:
        pool.add_lane(n)
    for i in range(trees):
        f, is960, variant = positions[(offset + i) % len(positions)]
        pool.add_position(f, is960, variant)
    if shared_collectors:
        pool.set_shared_collectors(shared_collectors)
    pool.run(simulations=min(warmup_simulations, simulations), threads=threads)       # untimed: worker threads, allocator, clocks
    cursor = offset
    reps = []
    thr0 = replicas.cgroup_throttled_usec()
    for _ in range(repeats):
        nodes = evals = sims = batches = rounds = 0
        seconds = 0.0
        depth_max = 0
        depth_w = 0.0
        while seconds < min_seconds or rounds == 0:
            for i in range(trees):
                f, is960, variant = positions[(cursor + i) % len(positions)]
                pool.reset_position(i, f, is960, variant)
            cursor += trees
            s = pool.run(simulations=simulations, threads=threads)
            nodes, evals, sims, batches = nodes + s.nodes, evals + s.nn_evals, sims + s.simulations, batches + s.batches
            seconds += s.seconds
            depth_w += s.depth_avg * s.simulations
            depth_max = max(depth_max, int(s.depth_max))
            rounds += 1
        reps.append(dict(nodes=nodes, evals=evals, simulations=sims, batches=batches, seconds=seconds, rounds=rounds,
                         depth_avg=depth_w / max(1, sims), depth_max=depth_max))
    thr1 = replicas.cgroup_throttled_usec()
    pool.close()
    nps = [r["nodes"] / r["seconds"] for r in reps]
    med = sorted(range(len(reps)), key=lambda k: nps[k])[len(reps) // 2]
    r = reps[med]
    batch = nets[0].get_batch_size()
    return {"mcts_nodes_per_sec": round(nps[med], 1), "nodes_per_sec_min": round(min(nps), 1), "nodes_per_sec_max": round(max(nps), 1),
            "nodes_per_sec_repeats": [round(v, 1) for v in nps], "statistic": f"median of {len(reps)} repeats",
            "mcts_nn_evals_per_sec": round(r["evals"] / r["seconds"], 1), "simulations_per_sec": round(r["simulations"] / r["seconds"], 1),
            "seconds": round(r["seconds"], 3), "rounds": r["rounds"], "trees_per_gpu": trees, "simulations_per_tree": simulations,
            "lanes": len(nets), "batch": batch, "host_threads_per_gpu": threads, "collectors_per_tree_and_lane": shared_collectors or None,
            "avg_batch_fill": round(r["evals"] / max(1, r["batches"]) / batch, 3), "depth_avg": round(r["depth_avg"], 2),
            "depth_max": r["depth_max"],
            "host_cgroup_throttled_ms_during_search": None if thr0 is None or thr1 is None else round((thr1 - thr0) / 1e3, 2),
            "_median_totals": (r["nodes"], r["evals"], r["simulations"], r["seconds"]),
            "_spread": statistics.pstdev(nps) / max(1e-9, statistics.mean(nps))}


def variant_positions(variant: str, base_variant: str = "chess", max_positions: int = 0, is960: bool = False):
    """(fen, is960, variant) triples: the positions along the reference's calibration games of `base_variant`
    (crazyara_amd/openings.py), replayed under `variant`'s rules (3check / KOTH reach their own terminals earlier: the replay of a
    game stops there)."""
    from . import env, openings
    out, seen = [], set()
    for g in openings.games(base_variant):
        p = env.Position("", False, variant)
        for mv in [None] + g:
            if mv is not None and not p.push_uci(mv):
                break
            if p.terminal() != env.TERMINAL_NONE:
                break
            f = p.fen()
            if f not in seen:
                seen.add(f)
                out.append((f, is960, variant))
    return out[:max_positions] if max_positions else out


def benchmark_positions_leg(st, nets: Sequence, simulations: int, threads: int, shared_collectors: int = 8) -> dict:
    """`CrazyAra::benchmark` (engine/src/uci/crazyara.cpp:287-330) on its own position table (engine/tests/benchmarkpositions.cpp:31-49):
    one `go` per position on ONE tree, and the command's summary -- passed (best move != the table's blunder move), NPS average and
    median over the positions (EvalInfo::calculate_nps, evalinfo.cpp:73-80), average PV depth.  The reference limits each `go` by
    movetime; a simulation limit makes the leg repeatable.  With random-init weights "passed" says nothing about strength -- it is
    reported because the command reports it."""
    from . import openings
    table = openings.benchmark_positions()
    pool = search.SearchPool(st, net_a=nets[0], net_b=nets[1] if len(nets) > 1 else None)
    pool.add_position(table[0]["fen"], False, "crazyhouse")
    if shared_collectors:
        pool.set_shared_collectors(shared_collectors)
    pool.run(simulations=min(200, simulations), threads=threads)
    nps, depth, passed, alt, nodes, seconds = [], [], 0, 0, 0, 0.0
    for t in table:
        pool.reset_position(0, t["fen"], False, "crazyhouse")
        s = pool.run(simulations=simulations, threads=threads)
        nps.append(s.nodes / max(s.seconds, 1e-9))
        nodes, seconds = nodes + s.nodes, seconds + s.seconds
        depth.append(len(pool.pv(0)["pv"]))
        best = pool.best_move(0)
        passed += best != t["blunder"]
        alt += best == t["alternative"]
    pool.close()
    return {"positions": len(table), "passed": int(passed), "alternative_played": int(alt), "nps_avg": round(sum(nps) / len(nps), 1),
            "nps_median": round(sorted(nps)[len(nps) // 2], 1), "pv_depth_avg": round(sum(depth) / len(depth), 2),
            "mcts_nodes_per_sec": round(nodes / max(seconds, 1e-9), 1), "simulations_per_go": simulations,
            "collectors_per_tree_and_lane": shared_collectors or None, "lanes": len(nets), "batch": nets[0].get_batch_size(),
            "workload": "CrazyAra::benchmark: the crazyhouse blunder-check positions of engine/tests/benchmarkpositions.cpp, one go each on ONE tree"}
