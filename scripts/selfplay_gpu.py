"""Self-play on the GPU(s): G concurrent games per GPU share the evaluator batches of one search pool (BASELINE configs 4 / 5).

  python scripts/selfplay_gpu.py --variant chess --chess960 --games 16 --concurrent 16 --simulations 200
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/selfplay_gpu.py --games 64 ...

Games are independent, so ranks shard them (replicas.shard_items) and only the final counters are reduced (SUM games / moves /
nodes, MAX seconds).  Random-init weights (no trained weights ship): the games are legal, not good.  Prints one JSON line."""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from crazyara_amd import build, netfile, replicas, rise_config, search, selfplay  # noqa: E402
from crazyara_amd.neuralnetapi import HipAPI  # noqa: E402

LICHESS = (2, 3, 80, 84, "3.0")        # MODE_LICHESS tables: 80-channel v3 planes, 84 policy channels
MODES = {"crazyhouse": (0, 1, 34, 81, "1.0"), "chess": (1, 3, 52, 76, "3.0"), "3check": LICHESS, "kingofthehill": LICHESS,
         "antichess": LICHESS, "atomic": LICHESS, "horde": LICHESS, "racingkings": LICHESS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="crazyhouse", choices=sorted(MODES))
    ap.add_argument("--chess960", action="store_true")
    ap.add_argument("--games", type=int, default=16, help="whole job, sharded over the ranks")
    ap.add_argument("--concurrent", type=int, default=16, help="games in flight per GPU")
    ap.add_argument("--simulations", type=int, default=200)
    ap.add_argument("--blocks", type=int, default=13)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--max-plies", type=int, default=300)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--pgn", default="")
    ap.add_argument("--precision", default="float16", help="float16 | fp8 | float32")
    ap.add_argument("--adaptive-quota", type=int, default=32, help="SearchPool.set_adaptive_quota: leaves per tree and batch may grow to this")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    build.build()

    mode, ver, cin, cp, vstr = MODES[args.variant]
    cfg = rise_config.rise_v2_config(args.blocks, cin, cp)
    sd = rise_config.make_state_dict(cfg, seed=1)
    d = tempfile.mkdtemp(prefix="cra_selfplay_")
    netfile.export_rise(os.path.join(d, f"{cfg.name}-v{vstr}.cranet"), cfg, sd, input_version=vstr, variant=args.variant)
    nets = [HipAPI(local_rank, args.batch, d, args.precision) for _ in range(2)]

    my_games = replicas.shard_items(args.games, rank, world)
    quota = max(1, args.batch // max(1, (args.concurrent + 1) // 2))
    st = search.default_settings(mode=mode, version_major=ver, batch_size=quota, seed=1 + rank)
    pool = search.SearchPool(st, net_a=nets[0], net_b=nets[1])
    pool.set_adaptive_quota(args.adaptive_quota)
    s = selfplay.SelfPlaySettings(variant=args.variant, is960=args.chess960, simulations=args.simulations, max_plies=args.max_plies,
                                  mean_init_ply=4.0, raw_policy_prob_temperature=0.05, init_temperature=0.8, temperature_moves=8,
                                  temperature_decay=0.9, quantile_clipping=0.25, seed=100 + rank)

    def start_fen(i):                                       # chess960: a deterministic stand-in for the reference's random start
        if not args.chess960:
            return ""
        from crazyara_amd import _capi
        return _capi.load().mi_chess960_start_fen((my_games[i % max(1, len(my_games))] * 37 + 11) % 960).decode()

    # the game loop runs inside the library (csrc/rl/selfplay.cpp); the opening plies' raw policy comes from the pool's own lanes
    loop = selfplay.SelfPlay(pool, s, min(args.concurrent, max(1, len(my_games))), start_fen=start_fen)
    games = loop.play(len(my_games), threads=args.threads)
    stt = loop.stats
    if args.pgn:
        with open(f"{args.pgn}.rank{rank}", "w") as f:
            for g in games:
                f.write(g.pgn())
    tot_games, sec, ex = replicas.reduce_stats(replicas.ReplicaStats(units=float(len(games)), seconds=stt["seconds"],
                                                                     extra=(float(stt["moves"]), float(stt["nodes"]), float(stt["nn_evals"]))),
                                               dist, torch.device("cuda", local_rank))
    if rank == 0:
        res = {1: 0, 0: 0, -1: 0}
        for g in games:
            res[g.result] += 1
        print(json.dumps({"metric": "selfplay_games_per_min", "value": round(tot_games / sec * 60, 2), "n_gpus": world,
                          "games": int(tot_games), "moves": int(ex[0]), "seconds": round(sec, 2),
                          "mcts_nodes_per_sec": round(ex[1] / sec, 1), "nn_evals_per_sec": round(ex[2] / sec, 1),
                          "config": {"variant": args.variant + ("960" if args.chess960 else ""), "net": cfg.name, "batch": args.batch,
                                     "concurrent_games_per_gpu": args.concurrent, "simulations_per_move": args.simulations,
                                     "precision": args.precision, "game_loop": "native (mi_selfplay_*)", "adaptive_quota": args.adaptive_quota},
                          "rank0_results": {"white": res[1], "draw": res[0], "black": res[-1]},
                          "rank0_kept_subtrees": stt["kept_subtrees"], "rank0_restarts": stt["restarts"],
                          "rank0_seconds_in_search": round(stt["run_seconds"], 3), "rank0_seconds_in_move_step": round(stt["move_seconds"], 3)}))
    pool.close()
    for n in nets:
        n.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
