"""Pins oracle/chess_oracle.py against the reference's own golden vectors (CPU only):
plane statistics + FEN strings + rule outcomes of engine/tests/tests.cpp (tests/golden/planes_goldens.json),
the frozen label list (legacyconstants.h) and FLAT_PLANE_IDX tables (policymaprepresentation.h) in policy_tables.npz,
and published perft counts."""
import json
import os

import numpy as np
import pytest

from oracle import chess_oracle as co

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "planes_goldens.json")))
TABLES = np.load(os.path.join(os.path.dirname(__file__), "golden", "policy_tables.npz"))
TERM = {"loss": co.TERMINAL_LOSS, "draw": co.TERMINAL_DRAW, "win": co.TERMINAL_WIN, "none": co.TERMINAL_NONE}


def board_for(case):
    b = co.Board(case["fen"] or None, case.get("is960", False), case["variant"])
    for m in case["moves"]:
        b.push_uci(m, checked=not case.get("unchecked_moves", False))
    return b


def check_stats(case, x):
    s, mx, key, arg = co.plane_statistics(x)
    if "nb_values" in case:
        assert x.size == case["nb_values"]
    if "sum_range" in case:
        assert case["sum_range"][0] < s < case["sum_range"][1]
        assert case["key_range"][0] < key < case["key_range"][1]
        assert case["max_range"][0] < mx < case["max_range"][1]
        return
    if "rel" in case:   # Catch::Matchers::WithinRel
        assert abs(s - case["sum"]) <= case["rel"] * abs(case["sum"])
        assert abs(key - case["key"]) <= case["rel"] * abs(case["key"])
    else:               # exact == on doubles in the reference test
        assert s == case["sum"] and key == case["key"]
    assert mx == case["max"]
    if case["argmax"] is not None:
        assert arg == case["argmax"]


@pytest.mark.parametrize("case", G["cases"], ids=[c["src"] for c in G["cases"]])
def test_oracle_planes_match_reference_goldens(case):
    b = board_for(case)
    x = co.board_to_planes(b, case["mode"], case["version"], case["normalize"])
    check_stats(case, x)
    if "fen_after" in case:
        assert b.fen() == case["fen_after"]


@pytest.mark.parametrize("case", G["rules"], ids=[c["src"] for c in G["rules"]])
def test_oracle_rules_match_reference_tests(case):
    b = board_for(case)
    if "fen_after" in case:
        assert b.fen() == case["fen_after"]
    legal = set(b.legal_uci())
    for m in case.get("legal", []):
        assert m in legal
    for m in case.get("illegal", []):
        assert m not in legal
    if "terminal" in case:
        assert b.terminal() == TERM[case["terminal"]]


@pytest.mark.parametrize("case", G["perft"], ids=[c["src"] for c in G["perft"]])
def test_oracle_perft_published_counts(case):
    # the pure-python oracle only walks to depth <= 3 (published shallower counts of the same positions)
    depth, nodes = case.get("oracle_depth", case["depth"]), case.get("oracle_nodes", case["nodes"])
    b = co.Board(case["fen"] or None, case.get("is960", False), case["variant"])
    assert b.perft(depth) == nodes


@pytest.mark.parametrize("mode,name", [(co.MODE_CRAZYHOUSE, "crazyhouse"), (co.MODE_LICHESS, "lichess"), (co.MODE_CHESS, "chess")])
def test_oracle_labels_and_flat_plane_idx_equal_reference_tables(mode, name):
    pm = co.PolicyMap(mode)
    assert pm.labels == [str(s) for s in TABLES[f"labels_{name}"]]            # tests.cpp:569-579 "LABELS equality"
    assert pm.flat == [int(v) for v in TABLES[f"flat_{name}"]]                # policymaprepresentation.h table
    assert max(pm.flat) < {0: 81, 1: 76, 2: 84}[mode] * 64
    if mode != co.MODE_LICHESS:   # the shipped lichess table maps drops onto the king-promotion planes (quirk kept)
        assert len(set(pm.flat)) == len(pm.flat)


def test_oracle_en_passant_candidates():
    # tests.cpp:158-161 + sfutil.cpp:109-140: the diagonal rank5->6 / rank4->3 pawn captures share labels with normal moves
    pm = co.PolicyMap(co.MODE_CHESS)
    for f in range(8):
        for df in (-1, 1):
            if 0 <= f + df < 8:
                assert co.FILES[f] + "5" + co.FILES[f + df] + "6" in pm.idx
                assert co.FILES[f] + "4" + co.FILES[f + df] + "3" in pm.idx
