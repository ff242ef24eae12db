"""Product environment (C++ Position / planes / policy map through the C ABI, CPU only) vs the oracle and the
reference goldens.  Bit-exact: FEN strings, legal-move sets, plane tensors, policy indices."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

from crazyara_amd import env
from oracle import chess_oracle as co

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "planes_goldens.json")))
TABLES = np.load(os.path.join(os.path.dirname(__file__), "golden", "policy_tables.npz"))
TERM = {"loss": 0, "draw": 1, "win": 2, "none": 4}


def position_for(case):
    p = env.Position(case["fen"], case.get("is960", False), case["variant"])
    for m in case["moves"]:
        assert p.push_uci(m), m
    return p


@pytest.mark.parametrize("case", [c for c in G["cases"] if not c.get("unchecked_moves")], ids=lambda c: c["src"])
def test_product_planes_match_reference_goldens_and_oracle(case, hip_lib):
    p = position_for(case)
    x = p.planes(case["mode"], case["version"], case["normalize"])
    b = co.Board(case["fen"] or None, case.get("is960", False), case["variant"])
    for m in case["moves"]:
        b.push_uci(m)
    xo = co.board_to_planes(b, case["mode"], case["version"], case["normalize"])
    assert x.shape == xo.shape and np.array_equal(x, xo)          # bit-exact, normalised planes included
    s, mx, key, arg = co.plane_statistics(x)
    if "rel" not in case and "sum_range" not in case:
        assert (s, mx, key) == (case["sum"], case["max"], case["key"])
        if case["argmax"] is not None:
            assert arg == case["argmax"]
    if "fen_after" in case:
        assert p.fen() == case["fen_after"]


@pytest.mark.parametrize("case", G["rules"], ids=lambda c: c["src"])
def test_product_rules_match_reference_tests(case, hip_lib):
    p = position_for(case)
    if "fen_after" in case:
        assert p.fen() == case["fen_after"]
    legal = set(p.legal_uci())
    for m in case.get("legal", []):
        assert m in legal
    for m in case.get("illegal", []):
        assert m not in legal
    if "terminal" in case:
        assert p.terminal() == TERM[case["terminal"]]


@pytest.mark.parametrize("case", G["perft"] + G["perft_deep"], ids=lambda c: f'{c["variant"]}-{c["fen"][:20]}-d{c["depth"]}')
def test_product_perft_published_counts(case, hip_lib):
    p = env.Position(case["fen"], case.get("is960", False), case["variant"])
    assert p.perft(case["depth"]) == case["nodes"]


@pytest.mark.parametrize("mode,name", [(0, "crazyhouse"), (2, "lichess"), (1, "chess")])
def test_product_labels_and_flat_plane_idx_equal_reference_tables(mode, name, hip_lib):
    labels, mirrored, flat = env.policy_tables(mode)
    assert labels == [str(s) for s in TABLES[f"labels_{name}"]]
    assert flat == [int(v) for v in TABLES[f"flat_{name}"]]
    assert mirrored == [co.mirror_label(l) for l in labels]


PLAYOUTS = [
    # (variant, is960, fen, mode, versions, n games, max plies)
    ("crazyhouse", False, "", 0, (1, 2, 3), 6, 90),
    ("chess", False, "", 1, (1, 3, "2.7", "2.8"), 4, 80),
    ("chess", True, "bnnrkbrq/pppppppp/8/8/8/8/PPPPPPPP/BNNRKBRQ w GDgd - 0 1", 1, (3, "2.8"), 3, 70),
    ("chess", True, "nrbbqnkr/pppppppp/8/8/8/8/PPPPPPPP/NRBBQNKR w HBhb - 0 1", 1, (3,), 2, 60),
    ("3check", False, "", 2, (2, 3), 3, 70),
    ("kingofthehill", False, "", 2, (2, 3), 3, 70),
    ("crazyhouse", False, "", 2, (2, 3), 2, 70),
]


@pytest.mark.parametrize("variant,is960,fen,mode,versions,games,plies", PLAYOUTS)
def test_random_playouts_bit_exact_against_oracle(variant, is960, fen, mode, versions, games, plies, hip_lib):
    """Seeded random games: at every ply the legal-move set (as UCI strings), FEN, terminal type, every plane layout and
    the policy index of every legal move must equal the oracle's."""
    rng = random.Random(hash((variant, is960, fen)) & 0xFFFF)
    pm = co.PolicyMap(mode)
    for g in range(games):
        p = env.Position(fen, is960, variant)
        b = co.Board(fen or None, is960, variant)
        for ply in range(plies):
            moves = b.legal_moves()
            ucis = sorted(b.move_uci(m) for m in moves)
            assert p.legal_uci() == ucis, (b.fen(), ply)
            assert p.fen() == b.fen()
            assert p.terminal() == b.terminal()
            if p.terminal() != 4 or not moves:
                break
            for v in versions:
                for norm in (True, False):
                    assert np.array_equal(p.planes(mode, v, norm), co.board_to_planes(b, mode, v, norm)), (b.fen(), v, norm)
            for m in moves:
                u = b.move_uci(m)
                for is_map in (True, False):
                    assert p.policy_index(u, mode, is_map) == pm.index(b, m, is_map), (b.fen(), u)
            mv = rng.choice(moves)
            assert p.push_uci(b.move_uci(mv))
            b.push(mv)


def test_descriptor_is_192_bytes_and_round_trips(hip_lib):
    p = env.Position("5r2/ppp2pkp/3p4/2bP4/2Pnp1N1/3P2pP/PP2n1P1/R2Q1R1K[PBRQnbb] w - - 0 28", False, "crazyhouse")
    for m in ("Q@f6", "g7g8", "R@h8"):
        p.push_uci(m)
    d = p.desc()
    assert len(d) == 192
    bb = np.frombuffer(d, dtype=np.uint64, count=14)
    assert bin(int(bb[0])).count("1") == 7 and int(d[122]) == 1   # 7 white pawns on board; black to move


@pytest.mark.parametrize("variant", ["crazyhouse", "chess"])
def test_reference_calibration_games_replay_identically(variant, hip_lib):
    """Every move of the reference's hard-coded games (chessbatchstream.cpp:44-94) is legal in product and oracle and both
    reach identical FENs / planes along the way."""
    from crazyara_amd import openings
    for game in openings.games(variant):
        p = env.Position("", False, variant)
        b = co.Board(None, False, variant)
        for i, mv in enumerate(game):
            assert mv in b.legal_uci(), (mv, b.fen())
            assert p.push_uci(mv), (mv, p.fen())
            b.push_uci(mv)
            assert p.fen() == b.fen()
            if i % 7 == 0:
                mode = 0 if variant == "crazyhouse" else 1
                assert np.array_equal(p.planes(mode, 3, True), co.board_to_planes(b, mode, 3, True))
    assert len(openings.position_fens("crazyhouse")) > 200


def test_host_planes_from_descriptors_equal_position_planes(hip_lib):
    """mi_planes_from_descs_host (the CPU evaluator's input builder) == board_to_planes of the positions, all layouts of a mode."""
    fens = [("r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] b KQkq - 0 8", "crazyhouse", 0, (1, 2, 3)),
            ("r3k2r/pppq1ppp/2npbn2/2b1p3/2B1P3/2NPBN2/PPPQ1PPP/R3K2R w KQkq - 4 8", "chess", 1, (3, "2.7", "2.8")),
            ("r1br2k1/p4ppp/2p2n2/Q1b1p3/8/NP3N1P/P1P1BPP1/R1B1K2R b KQ - 0 12", "chess", 1, ("2.7", "2.8")),
            ("1r4k1/1p2bp1p/3p2p1/PprPp2n/1R2PPq1/3Q4/1P1B1NPP/5RK1 b - - 1+1 2 22", "3check", 2, (1, 3))]
    lib = env._capi.load()
    for fen, variant, mode, versions in fens:
        p = env.Position(fen, False, variant)
        p.push(p.legal_moves()[0])
        q = p.clone()
        q.push(q.legal_moves()[-1])
        for v in versions:
            layout = env.planes_layout(mode, v)
            got = env.planes_from_descs_host(p.desc(layout) + q.desc(layout), 2, layout, True)
            assert np.array_equal(got[0].reshape(-1), p.planes(mode, v, True).reshape(-1))
            assert np.array_equal(got[1].reshape(-1), q.planes(mode, v, True).reshape(-1))


INSUFFICIENT = [  # engine/tests/tests.cpp:203-252 "Draw_by_insufficient_material"
    ("8/8/2k5/8/8/4K3/8/8 w - - 0 1", "chess", True), ("8/8/2k5/8/5B2/4K3/8/8 w - - 0 1", "chess", True),
    ("8/8/2k5/8/5N2/4K3/8/8 w - - 0 1", "chess", True), ("8/8/2k5/8/8/3NKN2/8/8 w - - 0 1", "chess", True),
    ("kn6/8/NK6/8/8/8/8/8 w - - 0 2", "chess", False), ("rnbqkb1r/pp2pppp/3p1n2/8/3NP3/8/PPP2PPP/RNBQKB1R w KQkq - 1 5", "chess", False),
    ("8/8/2k5/8/8/4K3/8/8 w - - 0 1", "kingofthehill", False), ("8/8/2k5/8/5B2/4K3/8/8 w - - 0 1", "racingkings", False),
    ("8/8/2k5/8/5N2/4K3/8/8 w - - 0 1", "antichess", False), ("8/8/2k5/8/8/3NKN2/8/8 w - - 0 1", "horde", False),
    ("8/8/3k4/8/4P3/8/8/8 w - - 0 1", "horde", False),
]


@pytest.mark.parametrize("fen,variant,expected", INSUFFICIENT)
def test_draw_by_insufficient_material_reference_cases(hip_lib, fen, variant, expected):
    p = env.Position(fen, False, variant)
    assert p.insufficient_material() == expected
    if variant == "chess":                                              # the oracle folds the rule into its terminal verdict
        assert (co.Board(fen, False, variant).terminal() == 1) == expected and (p.terminal() == 1) == expected


@pytest.mark.parametrize("variant,seed", [("crazyhouse", 42), ("chess", 543), ("crazyhouse", 1048), ("atomic", 7), ("antichess", 8)])
def test_state_tests_of_the_reference_random_games(hip_lib, variant, seed):
    """engine/tests/tests.cpp:641-740 "State: steps_from_null / Reach terminal state / check_result / clone": random games count their
    plies, end within 10000 moves, a finished game has a result that agrees with the terminal type, and a clone has the same FEN."""
    rng = random.Random(seed)
    p = env.Position("", False, variant)
    assert p.steps_from_null() == 0
    applied = 0
    while applied < 10000:
        assert p.steps_from_null() == applied
        moves = p.legal_moves()
        if p.terminal() != 4:
            break
        p.push(rng.choice(moves))
        applied += 1
        if applied == 7:
            assert p.clone().fen() == p.fen()
    t = p.terminal()
    assert t != 4                                                       # reached a terminal state
    b = co.Board(p.fen(), False, variant)
    assert b.terminal() == t                                            # the oracle agrees on the verdict of the final position
    if t == 0:
        assert not moves or variant in ("atomic", "antichess", "horde", "racingkings", "kingofthehill", "3check")   # LOSS: mated or variant rule


def test_reference_position_sets_load():
    """The fixed sets SURVEY 8d names: the blunder-check FENs of engine/tests/benchmarkpositions.cpp (pockets as "[QNbpp]" and as a 9th
    slash field) parse, their blunder / alternative moves are legal moves of the position (so pockets, side and castling were read as the
    reference means them), and the 50 SAN openings of zh-50_startpos.pgn replay from the start position."""
    from crazyara_amd import env, openings
    pos = openings.benchmark_positions()
    assert len(pos) == 15
    for t in pos:
        p = env.Position(t["fen"], False, "crazyhouse")
        legal = set(p.legal_uci())
        assert t["blunder"] in legal, (t["fen"], t["blunder"])
        assert t["alternative"] in legal, (t["fen"], t["alternative"])
        b = co.Board(t["fen"], False, "crazyhouse")
        assert sorted(b.move_uci(m) for m in b.legal_moves()) == sorted(legal)
    fens = openings.zh50_fens()
    assert len(fens) == 50 and len(set(fens)) >= 49          # two of the file's lines end in the same position
    assert fens[0].startswith("rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR")
    assert len(openings.crazyhouse_opening_set()) > 200
