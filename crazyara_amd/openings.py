"""Fixed position sets for benchmarks and tests (SURVEY.md 8d 'Fixed opening set'): every position along the reference's
calibration games (engine/src/environments/chess_related/chessbatchstream.cpp:44-94)."""
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
