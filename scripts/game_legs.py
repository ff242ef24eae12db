"""bench.py's game legs alone (chess960 self-play, 3check / KOTH arena on one GPU), e.g. to A/B CRA_ARENA_SERIAL=1."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="float16")
ap.add_argument("--threads", type=int, default=16)
a = ap.parse_args()
from crazyara_amd import build, replicas  # noqa: E402
build.build()
_, budget = replicas.pin_rank_to_cpus(1, 0)
print(json.dumps(bench.config_game_legs(argparse.Namespace(precision=a.precision), 0, max(1, min(a.threads, budget)))))
