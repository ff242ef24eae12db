"""Rules of the lichess variants of the MultiAra build (antichess, atomic, horde, racing kings): the reference's own known-answer tests
(engine/tests/tests.cpp: Variants_Horde :1000-1137, Racing_Kings :1186-1218, Atomic :1220-1295, Antichess :1408-1448) against the product Position
(through the C ABI) AND the oracle board, plus random playouts product == oracle."""
import random

import numpy as np
import pytest

from crazyara_amd import env
from oracle import chess_oracle as co

WHITE_WIN, BLACK_WIN, DRAW, NO_RESULT = "1-0", "0-1", "1/2-1/2", "*"


class Prod:
    def __init__(self, fen, variant):
        self.p = env.Position(fen, False, variant)

    def result(self):
        t, white = self.p.terminal(), self.p.side_to_move() == 0
        if t == env.TERMINAL_NONE:
            return NO_RESULT
        if t == env.TERMINAL_DRAW:
            return DRAW
        return (WHITE_WIN if white else BLACK_WIN) if t == env.TERMINAL_WIN else (BLACK_WIN if white else WHITE_WIN)

    def legal(self):
        return sorted(self.p.legal_uci())

    def san(self):
        return sorted(self.p.move_san(m) for m in self.p.legal_moves())

    def push(self, uci):
        assert self.p.push_uci(uci), uci

    def fen(self):
        return self.p.fen()


class Orac:
    def __init__(self, fen, variant):
        self.b = co.Board(fen or None, False, variant)

    def result(self):
        t, white = self.b.terminal(), self.b.stm == 0
        if t == co.TERMINAL_NONE:
            return NO_RESULT
        if t == co.TERMINAL_DRAW:
            return DRAW
        return (WHITE_WIN if white else BLACK_WIN) if t == co.TERMINAL_WIN else (BLACK_WIN if white else WHITE_WIN)

    def legal(self):
        return self.b.legal_uci()

    def san(self):
        return None                                        # SAN is a product feature (Position::move_to_san)

    def push(self, uci):
        self.b.push_uci(uci)

    def fen(self):
        return self.b.fen()


IMPLS = [Prod, Orac]


@pytest.mark.parametrize("impl", IMPLS)
def test_horde_reference_cases(hip_lib, impl):
    v = "horde"
    # the pieces win by capturing all the pawns
    for fen in ("8/8/1p4k1/8/4q3/8/8/8 w - - 0 76", "6r1/8/4k3/5q2/p7/8/8/8 w - - 0 65", "8/4k3/8/4q3/8/8/8/8 w - - 0 63"):
        assert impl(fen, v).result() == BLACK_WIN
    # the pawns win by checkmating the king, also with promoted pieces
    assert impl("rnbqkbnr/1ppp1P1p/3PP3/2P5/PP5P/P1PPPPPP/PPpPPPPP/PPPPPPPP b kq - 0 10", v).result() == WHITE_WIN
    assert impl("8/8/R7/6P1/8/PP1P4/k1P5/Q3QPP1 b - - 3 69", v).result() == WHITE_WIN
    # stalemate
    assert impl("6k1/6P1/7q/8/8/8/8/8 w - - 0 1", v).result() == DRAW
    assert impl("1k6/3R4/2Q5/8/2P5/3P4/8/8 b - - 0 1", v).result() == DRAW
    # 50-move rule
    assert impl("6k1/3R4/8/8/8/8/8/8 b - - 99 85", v).result() == NO_RESULT
    assert impl("6k1/3R4/8/8/8/8/8/8 b - - 100 85", v).result() == DRAW
    # pawns on the first rank can move one or two squares
    x = impl("3k4/8/8/8/8/8/8/PPPPPPPP w - - 0 1", v)
    assert x.legal() == sorted([f"{f}1{f}2" for f in "abcdefgh"] + [f"{f}1{f}3" for f in "abcdefgh"])
    if x.san() is not None:
        assert x.san() == sorted([f"{f}3" for f in "abcdefgh"] + [f"{f}2" for f in "abcdefgh"])
    # a first-rank double step cannot be captured en passant
    x = impl("6k1/8/8/8/8/1p1p4/8/2P5 w - - 0 1", v)
    x.push("c1c3")
    assert "b3c2" not in x.legal() and "d3c2" not in x.legal()
    # the normal en passant from the second rank exists
    x = impl("1kb3nr/8/8/8/3p1pP1/8/1P2P3/P6P w - - 0 1", v)
    x.push("e2e4")
    assert "d4e3" in x.legal() and "f4e3" in x.legal()
    # a pawn that stepped from the first to the second rank may then advance two squares
    x = impl("1kb3nr/8/8/8/3p1pP1/8/4P3/PP5P w - - 0 1", v)
    x.push("b1b2")
    x.push("h8h1")
    assert "b2b4" in x.legal()
    # no draw by insufficient material; threefold repetition is a draw
    assert impl("4k3/8/8/6P1/8/8/8/8 w - - 0 1", v).result() == NO_RESULT
    x = impl("4k3/6R1/8/8/8/8/8/8 w - - 0 1", v)
    for _ in range(3):
        for u in ("g7g8", "e8e7", "g8g7", "e7e8"):
            x.push(u)
    assert x.result() == DRAW


@pytest.mark.parametrize("impl", IMPLS)
def test_racing_kings_reference_cases(hip_lib, impl):
    v = "racingkings"
    assert impl("", v).fen() == "8/8/8/8/8/8/krbnNBRK/qrbnNBRQ w - - 0 1"
    # checks are forbidden
    x = impl("8/8/8/8/8/8/krbnNBRK/qrbnNBRQ w - - 0 1", v)
    assert not {"e2c3", "e2a3"} & set(x.legal())
    x = impl("R2R4/4Q3/8/2r5/1q6/bk3N1K/2b5/8 b - - 6 13", v)
    assert not {"c2f5", "b4g4", "b4h4", "c5h5"} & set(x.legal())
    x = impl("R2R4/4Q3/8/2r5/1q6/1k3N1K/2b5/2b5 w - - 7 14", v)
    assert not {"f3d2", "f3d4", "e7e6", "e7f7", "e7e3", "d8d3", "a8a3"} & set(x.legal())
    # a king on the eighth rank wins
    assert impl("1bk1q3/8/8/6K1/8/8/8/R7 w - - 2 47", v).result() == BLACK_WIN
    assert impl("6K1/8/8/6Q1/8/8/n1k5/b7 b - - 2 25", v).result() == WHITE_WIN
    # draw if White reaches the eighth rank and Black follows at once
    x = impl("2r2NK1/kn2R3/8/8/8/8/8/8 b - - 8 26", v)
    assert x.result() == NO_RESULT
    x.push("a7b8")
    assert x.result() == DRAW
    assert impl("1k3K2/2qQ4/8/8/8/8/8/8 w - - 30 26", v).result() == DRAW


@pytest.mark.parametrize("impl", IMPLS)
def test_antichess_reference_cases(hip_lib, impl):
    v = "antichess"
    assert impl("", v).fen() == "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w - - 0 1"
    # captures are compulsory; with several captures there is a choice
    assert impl("rnb1kbnr/pp1ppppp/8/q1p5/8/2P1P3/PP1PNPPP/RNBQKB1R b - - 0 3", v).legal() == ["a5a2", "a5c3"]
    # no check, no mate: pieces move while the king is attacked
    x = impl("2Q1kb1r/3ppp1p/r4np1/p7/8/P1P1P1P1/4NP1P/RNB1KB1R b - - 0 11", v)
    assert x.result() == NO_RESULT and "a6a8" in x.legal()
    # the king can be captured
    x = impl("r1Q1kb1r/3ppp1p/5np1/p7/8/P1P1P1P1/4NP1P/RNB1KB1R w - - 1 12", v)
    assert "c8e8" in x.legal()
    x.push("c8e8")
    assert x.result() == NO_RESULT
    # who has lost all pieces has won; who cannot move has won
    assert impl("8/8/6p1/7q/8/8/8/8 w - - 0 39", v).result() == WHITE_WIN
    assert impl("5b2/4p3/1p3p2/1P5p/8/8/8/1r6 w - - 0 36", v).result() == WHITE_WIN
    # promotion to a king; no castling
    assert "c2c1k" in impl("2Q5/8/8/8/R6P/2B5/2pP4/8 b - - 1 35", v).legal()
    assert "e8c8" not in impl("r3kbnr/p2pp1pp/bp3p2/8/3P4/P1P5/1B1P1PPP/RN1QK2R b - - 0 9", v).legal()


@pytest.mark.parametrize("impl", IMPLS)
def test_atomic_reference_cases(hip_lib, impl):
    """engine/tests/tests.cpp:1220-1295"""
    v = "atomic"
    assert impl("", v).fen() == "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
    # capturer and captured always die; everything within one square explodes, except pawns
    for fen, mv, after in (
            ("rn1qkb1r/p1p3pp/b3pp1n/3pP3/1P1P1P2/7P/P5P1/RNBQKBNR w KQkq - 1 7", "f1a6", "rn1qkb1r/p1p3pp/4pp1n/3pP3/1P1P1P2/7P/P5P1/RNBQK1NR b KQkq - 0 7"),
            ("1r1qk2r/2p3pp/p1n1pp1n/1P1pP3/1b1P1P2/P6P/4Q1P1/RNB1K1NR w KQk - 1 11", "a3b4", "1r1qk2r/2p3pp/p1n1pp1n/1P1pP3/3P1P2/7P/4Q1P1/RNB1K1NR b KQk - 0 11"),
            ("1r2k2r/R1p3pp/7n/2npp3/1B1P1PP1/1q5P/2Q4R/1N3KN1 w k - 2 22", "c2b3", "1r2k2r/R1p3pp/7n/2npp3/3P1PP1/7P/7R/1N3KN1 b k - 0 22"),
            ("1r3rk1/R1p3p1/6Pp/2nppn2/3P1P2/7P/7R/1N3KN1 w - - 0 25", "a7c7", "5rk1/6p1/6Pp/2nppn2/3P1P2/7P/7R/1N3KN1 b - - 0 25"),
            ("2r3k1/6p1/6Pp/3pp3/3P1P2/2N3nP/1n5R/2K3N1 b - - 8 29", "e5d4", "2r3k1/6p1/6Pp/3p4/5P2/6nP/1n5R/2K3N1 w - - 0 30"),
            ("6k1/6p1/6Pp/3p4/5P2/6nP/Kn5R/2r3N1 b - - 3 31", "c1g1", "6k1/6p1/6Pp/3p4/5P2/6nP/Kn6/8 w - - 0 32")):
        x = impl(fen, v)
        x.push(mv)
        assert x.fen() == after, (fen, mv)
    # nuking the opposite king wins at once ...
    x = impl("6k1/5Kp1/2q3P1/5n1p/5P1P/8/1n6/8 b - - 1 40", v)
    x.push("c6g6")
    assert x.result() == BLACK_WIN
    # ... overriding checks ...
    x = impl("8/1q6/8/8/8/5k2/1R4n1/1K6 w - - 0 1", v)
    assert "b2g2" in x.legal()
    x.push("b2g2")
    assert x.result() == WHITE_WIN
    # ... and checkmates
    x = impl("8/1q6/r7/8/8/5k2/R5n1/K7 w - - 0 1", v)
    assert "a2g2" in x.legal()
    x.push("a2g2")
    assert x.result() == WHITE_WIN
    # checkmate: the mating piece cannot be taken by the king, nor by others if the own king would explode
    assert impl("3q4/6Qk/4r3/p7/6Pp/7P/8/1R2R1K1 b - - 8 29", v).result() == WHITE_WIN
    assert impl("8/kQ3r2/6p1/2P3Pp/7P/4p3/1K2B3/n7 b - - 3 39", v).result() == WHITE_WIN
    # kings may stand next to each other but never capture
    assert "e6f7" in impl("6k1/6p1/4K1Pp/5n2/5P2/7P/1n6/3q4 w - - 0 37", v).legal()
    assert "g8f8" not in impl("5Kk1/6p1/2q3Pp/5n2/5P1P/8/1n6/8 b - - 2 39", v).legal()


@pytest.mark.parametrize("variant,seed", [("antichess", 1), ("horde", 2), ("racingkings", 3), ("antichess", 4), ("horde", 5), ("racingkings", 6),
                                          ("atomic", 7), ("atomic", 8), ("atomic", 9)])
def test_random_playouts_product_equals_oracle(hip_lib, variant, seed):
    """Same legal move sets, FENs and terminal verdicts along seeded random games; lichess input planes (v1 and v3) and the policy
    index of every legal move agree as well."""
    rng = random.Random(seed)
    for game in range(3):
        p, b = env.Position("", False, variant), co.Board(None, False, variant)
        for ply in range(120):
            assert p.fen() == b.fen()
            legal = b.legal_uci()
            assert sorted(p.legal_uci()) == legal, (variant, b.fen())
            assert p.terminal() == b.terminal(), (variant, b.fen())
            for ver in (1, 3):
                assert np.array_equal(p.planes(2, ver, True), co.board_to_planes(b, 2, ver, True)), (variant, b.fen(), ver)
            pm = co.PolicyMap(2)
            for m in b.legal_moves():
                assert p.policy_index(p.uci_to_move(b.move_uci(m)), 2, True) == pm.index(b, m, True)
            if b.terminal() != co.TERMINAL_NONE or not legal:
                break
            u = rng.choice(legal)
            assert p.push_uci(u)
            b.push_uci(u)
