"""Fixed position sets for benchmarks and tests (SURVEY.md 8d 'Fixed opening set'): every position along the reference's
calibration games (engine/src/environments/chess_related/chessbatchstream.cpp:44-94), the 50 crazyhouse openings of
etc/media/wiki/Strength_Evaluation/v0.3.1/zh-50_startpos.pgn and the blunder-check positions of `CrazyAra::benchmark`
(engine/tests/benchmarkpositions.cpp:31-49).  The data files are transcribed by scripts/make_position_sets.py."""
from __future__ import annotations

import json
import os
from typing import List

from . import env

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "opening_games.json")


def games(variant: str) -> List[List[str]]:
    with open(_DATA) as f:
        return json.load(f)[variant]


def position_fens(variant: str, max_positions: int = 0, skip_terminal: bool = True) -> List[str]:
    """FEN after every ply of every game (start position included once)."""
    fens, seen = [], set()
    for g in games(variant):
        p = env.Position("", False, variant)
        plies = [None] + g
        for mv in plies:
            if mv is not None and not p.push_uci(mv):
                raise ValueError(f"illegal fixture move {mv} in {p.fen()}")
            if skip_terminal and p.terminal() != env.TERMINAL_NONE:
                continue
            f = p.fen()
            if f not in seen:
                seen.add(f)
                fens.append(f)
    return fens[:max_positions] if max_positions else fens


def _data(name: str):
    with open(os.path.join(os.path.dirname(_DATA), name)) as f:
        return json.load(f)


def benchmark_positions() -> List[dict]:
    """[{fen, blunder, alternative}] of engine/tests/benchmarkpositions.cpp (crazyhouse; both pocket dialects of the FEN)."""
    return _data("benchmark_positions.json")["positions"]


def zh50_fens() -> List[str]:
    """The position at the end of each of the 50 crazyhouse openings (SAN move lists replayed from the start position)."""
    fens = []
    for g in _data("zh50_startpos.json")["games"]:
        p = env.Position("", False, "crazyhouse")
        for san in g:
            if not p.push_san(san):
                raise ValueError(f"opening move {san} does not match a legal move in {p.fen()}")
        fens.append(p.fen())
    return fens


def crazyhouse_opening_set() -> List[str]:
    """The fixed crazyhouse set of the search legs: the 50 openings, then every position along the two calibration games."""
    out, seen = [], set()
    for f in zh50_fens() + position_fens("crazyhouse"):
        if f not in seen:
            seen.add(f)
            out.append(f)
    return out
