"""Transcribes the reference's fixed position sets (SURVEY 8d "Fixed opening set") into crazyara_amd/data/ -- run in the build container,
where /root/reference exists; the JSON files are committed (bench / test INPUT data, like opening_games.json).

  * engine/tests/benchmarkpositions.cpp:31-49 -- the crazyhouse blunder-check positions of `CrazyAra::benchmark` (crazyara.cpp:287-330):
    FEN (both pocket dialects: "[QNbpp]" and a 9th slash field), the blunder move, the alternative move
  * etc/media/wiki/Strength_Evaluation/v0.3.1/zh-50_startpos.pgn -- 50 crazyhouse openings as SAN move lists
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "crazyara_amd", "data")


def benchmark_positions():
    src = open(os.path.join(REF, "engine/tests/benchmarkpositions.cpp")).read()
    out = []
    for line in src.splitlines():
        m = re.match(r'\s*TestPosition\("([^"]+)",\s*"([^"]+)",\s*"([^"]+)"\)', line)     # commented-out entries do not match
        if m:
            out.append({"fen": m.group(1), "blunder": m.group(2), "alternative": m.group(3).rstrip(")")})
    return out


def zh50():
    t = open(os.path.join(REF, "etc/media/wiki/Strength_Evaluation/v0.3.1/zh-50_startpos.pgn")).read()
    games = []
    for g in re.split(r"\n\n(?=\[Event)", t):
        body = re.sub(r"\{[^}]*\}", "", g.split("]\n")[-1])
        body = re.sub(r"\([^)]*\)", "", body)            # engine variations "(e7e6 b1c3 ...)" are not part of the opening line
        toks = [re.sub(r"^\d+\.+", "", x) for x in body.split() if x not in ("*", "1-0", "0-1", "1/2-1/2")]
        games.append([x for x in toks if x])
    return games


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    with open(os.path.join(OUT, "benchmark_positions.json"), "w") as f:
        json.dump({"source": "engine/tests/benchmarkpositions.cpp:31-49 (TestPosition(fen, blunderMove, alternativeMove))",
                   "variant": "crazyhouse", "positions": benchmark_positions()}, f, indent=1)
    with open(os.path.join(OUT, "zh50_startpos.json"), "w") as f:
        json.dump({"source": "etc/media/wiki/Strength_Evaluation/v0.3.1/zh-50_startpos.pgn (SAN)", "variant": "crazyhouse",
                   "games": zh50()}, f, indent=1)
    print("wrote", OUT)
