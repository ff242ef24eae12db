"""Host-side MCTS (C++ Tree / SearchPool through the C ABI, CPU only) vs the oracle restatement of the reference's
select / virtual-loss / backup arithmetic.  A deterministic pseudo-network (pure function of the position) plays the
evaluator on both sides, so visit counts and float32 Q / prior values must be bit-identical."""
import struct
import zlib

import numpy as np
import pytest

from crazyara_amd import env, search
from oracle import chess_oracle as co
from oracle import mcts_oracle as mo

NB_POLICY = {0: 5184, 1: 4864, 2: 5376}


def _pseudo_net(key: bytes, nb_policy: int):
    rs = np.random.Generator(np.random.PCG64(zlib.crc32(key)))
    p = rs.random(nb_policy, dtype=np.float32) ** np.float32(6.0)      # peaked like a policy
    p = p / p.sum(dtype=np.float32)
    v = np.float32(rs.random() * 1.6 - 0.8)
    return v, p


def key_from_desc(d: bytes) -> bytes:
    return d[0:96] + d[112:123]          # 12 bitboards + pockets[10] + side to move (struct BoardDesc, planes.h)


def key_from_board(b: co.Board) -> bytes:
    bbs = []
    for ch in "PNBRQKpnbrqk":
        v = 0
        for s in range(64):
            if b.b[s] == ch:
                v |= 1 << s
        bbs.append(v)
    pockets = bytes(b.pocket[c] for c in "PNBRQpnbrq")
    return struct.pack("<12Q", *bbs) + pockets + bytes([b.stm])


def test_descriptor_key_matches_oracle_board_key(hip_lib):
    for fen, variant in (("", "crazyhouse"), ("5r2/ppp2pkp/3p4/2bP4/2Pnp1N1/3P2pP/PP2n1P1/R2Q1R1K[PBRQnbb] b - - 0 28", "crazyhouse")):
        assert key_from_desc(env.Position(fen, False, variant).desc()) == key_from_board(co.Board(fen or None, False, variant))


# ---- hand-computed arithmetic (the reference has no unit test for these; worked examples from its own comments) ----
def test_oracle_arithmetic_worked_examples():
    s = mo.Settings(virtual_style=mo.VIRTUAL_LOSS)
    # get_current_cput: log((N + base + 1) / base) + init  (node.cpp:1243-1246)
    assert abs(float(mo.get_current_cput(0, s)) - (np.log(19653.0 / 19652.0) + 2.5)) < 1e-6
    assert abs(float(mo.get_current_cput(19652, s)) - (np.log(39305.0 / 19652.0) + 2.5)) < 1e-6
    # node.h:180-196 worked example: Q_1 = -0.25 with n_1 = 2 after one virtual loss on (Q=0.5, n=1); update with 0.7 -> 0.6
    t = mo.Tree.__new__(mo.Tree)
    t.s = s
    n = mo.Node.__new__(mo.Node)
    n.child_visits, n.q, n.vl, n.visit_sum, n.value_sum, n.real_visits, n.free_visits = [1], [np.float32(0.5)], [0], 1, 0.0, 0, 0
    t.apply_virtual_loss(n, 0)
    assert n.child_visits == [2] and abs(float(n.q[0]) - (-0.25)) < 1e-7 and n.vl == [1]
    t.revert_virtual_loss_and_update(n, 0, 0.7, False)
    assert abs(float(n.q[0]) - 0.6) < 1e-6 and n.vl == [0] and n.child_visits == [2]
    # collision revert restores Q exactly: (Q*n + 1)/(n - 1)  (node.cpp:661-679)
    n.child_visits, n.q, n.vl, n.visit_sum = [3], [np.float32(0.2)], [0], 3
    t.apply_virtual_loss(n, 0)
    t.revert_virtual_loss(n, 0)
    assert n.child_visits == [3] and abs(float(n.q[0]) - 0.2) < 1e-6
    # VIRTUAL_VISIT: Q untouched by the virtual visit; update = running mean over REAL visits (node.h:221-224)
    t.s = mo.Settings(virtual_style=mo.VIRTUAL_VISIT)
    n.child_visits, n.q, n.vl, n.visit_sum = [2], [np.float32(0.4)], [0], 2
    t.apply_virtual_loss(n, 0)
    assert float(n.q[0]) == np.float32(0.4)
    t.revert_virtual_loss_and_update(n, 0, 1.0, False)
    assert abs(float(n.q[0]) - (0.4 * 2 + 1.0) / 3) < 1e-6
    # first real visit overwrites Q_INIT (node.h:207-211)
    n.child_visits, n.q, n.vl = [0], [np.float32(-1.0)], [0]
    t.apply_virtual_loss(n, 0)
    t.revert_virtual_loss_and_update(n, 0, 0.3, False)
    assert float(n.q[0]) == np.float32(0.3)
    # VIRTUAL_MIX switches at the threshold (node.h:87-95)
    mix = mo.Settings(virtual_style=mo.VIRTUAL_MIX, virtual_mix_threshold=1000)
    assert mo.virtual_style(mix, 1000) == mo.VIRTUAL_VISIT and mo.virtual_style(mix, 1001) == mo.VIRTUAL_LOSS


def _fake_node(node_type=mo.NT_UNSOLVED, end_in_ply=0, n_moves=0):
    n = mo.Node.__new__(mo.Node)
    n.has_data, n.sorted, n.node_type, n.end_in_ply, n.checkmate_idx = True, True, node_type, end_in_ply, -1
    n.moves = [None] * n_moves
    n.unsolved_children = n_moves
    n.priors = [np.float32(1.0 / max(n_moves, 1))] * n_moves
    n.child_visits, n.q, n.vl = [1] * n_moves, [np.float32(0.1)] * n_moves, [0] * n_moves
    n.child, n.child_types = [None] * n_moves, [mo.NT_UNSOLVED] * n_moves
    n.value_sum, n.real_visits, n.visit_sum, n.free_visits, n.no_visit_idx = 0.0, 0, n_moves, 0, n_moves
    return n


def test_oracle_solver_worked_examples():
    """Node::solve_for_terminal (node.cpp:365-453) on hand-built nodes: a child's LOSS proves a WIN at once; a LOSS needs every
    child to be a WIN and takes the longest line; a DRAW needs every child WIN or DRAW with one DRAW; a child's WIN is disabled."""
    t = mo.Tree.__new__(mo.Tree)
    t.s = mo.Settings()
    # WIN: the second child is a mated position (terminal LOSS, 0 plies to the end) -> mate in 1, remembered as checkmate_idx
    n = _fake_node(n_moves=3)
    n.child = [_fake_node(), _fake_node(mo.NT_LOSS, 0), None]
    assert not t.solve_for_terminal(n, 0) and n.node_type == mo.NT_UNSOLVED and n.unsolved_children == 3
    assert not t.solve_for_terminal(n, 2)                                # unexpanded child: not a playout node
    assert t.solve_for_terminal(n, 1)
    assert (n.node_type, n.end_in_ply, n.checkmate_idx, n.unsolved_children) == (mo.NT_WIN, 1, 1, 2)
    assert float(n.q[1]) == 1.0 and float(n.value()) == 1.0 and n.real_visits == 1
    assert t.select_child(n) == 1 and not t.solve_for_terminal(n, 1)     # forced line; solved nodes are final
    # LOSS: both replies are proven WINs for the opponent (mate in 1 and mate in 3) -> mated in 4 plies along the longer line
    n = _fake_node(n_moves=2)
    n.child = [_fake_node(mo.NT_WIN, 1), _fake_node(mo.NT_WIN, 3)]
    assert not t.solve_for_terminal(n, 0)
    assert n.unsolved_children == 1 and float(n.priors[0]) == 0.0 and float(n.q[0]) == -2147483648.0   # disable_action
    assert t.solve_for_terminal(n, 1)
    assert (n.node_type, n.end_in_ply, n.checkmate_idx) == (mo.NT_LOSS, 4, -1) and float(n.q[1]) == -1.0
    assert t.best_action_index_fast(n) == 1                              # delay the mate
    # DRAW: one reply loses (child WIN), the other is a dead draw
    n = _fake_node(n_moves=2)
    n.child = [_fake_node(mo.NT_DRAW, 0), _fake_node(mo.NT_WIN, 1)]
    assert not t.solve_for_terminal(n, 0)
    assert t.solve_for_terminal(n, 1)
    assert (n.node_type, n.end_in_ply) == (mo.NT_DRAW, 0) and float(n.q[1]) == 0.0 and float(n.value()) == 0.0
    # not a draw while a sibling is still unsolved, and never through a child that loses for the opponent
    n = _fake_node(n_moves=3)
    n.child = [_fake_node(mo.NT_DRAW, 0), _fake_node(mo.NT_WIN, 1), _fake_node()]
    assert not t.solve_for_terminal(n, 0) and not t.solve_for_terminal(n, 1) and n.node_type == mo.NT_UNSOLVED


SOLVER_CASES = [
    # variant, fen, mode, expected root verdict (node_type, end_in_ply) or None, expected best move or None
    ("chess", "6k1/5ppp/8/8/8/8/8/R3K3 w - - 0 1", 1, (mo.NT_WIN, 1), "a1a8"),          # back-rank mate in 1
    ("antichess", "8/8/8/8/8/4p3/5P1q/8 b - - 0 1", 2, (mo.NT_LOSS, 1), None),           # both (forced) captures take White's last piece: White wins
    ("chess", "k7/8/1K6/8/8/8/8/7R w - - 0 1", 1, (mo.NT_WIN, 1), "h1h8"),
    ("crazyhouse", "4R2b/1N3rkb/1p2P1pp/p2P4/2P1P3/8/PP4Q1/3R3K[QRBBNNNPPPPpp] w - - 2 53", 0, None, None),
    ("crazyhouse", "r1b2rk1/pppp1Npp/8/8/8/8/PPPPPPPP/RNBQKB1R[Qq] w KQ - 0 1", 0, None, None),
]


@pytest.mark.parametrize("variant,fen,mode,verdict,best", SOLVER_CASES)
def test_mcts_solver_equals_oracle(hip_lib, variant, fen, mode, verdict, best):
    nbp, sims, quota = NB_POLICY[mode], 600, 8
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota)
    assert st.mcts_solver == 1                                            # MCTS_Solver defaults to on (optionsuci.cpp:129)

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    def eval_boards(boards):
        out = [_pseudo_net(key_from_board(b), nbp) for b in boards]
        return [o[0] for o in out], [o[1] for o in out]

    pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, False, variant)
    pool.run(simulations=sims, threads=1)
    moves, visits, q, _ = pool.root_children(t)
    info, solved = pool.tree_info(t), pool.root_solved(t)

    tree = mo.Tree(co.Board(fen, False, variant), mo.Settings(mode=mode, is_policy_map=True, batch_size=quota))
    mo.run_search(tree, eval_boards, sims, quota)
    r = tree.root
    assert visits == r.child_visits
    assert np.array_equal(q, np.array(r.q, np.float32))
    assert info["root_visits"] == r.visit_sum and info["node_count"] == tree.node_count()
    assert (solved["node_type"], solved["end_in_ply"], solved["checkmate_idx"]) == (r.node_type, r.end_in_ply, r.checkmate_idx)
    assert pool.best_move(t) == tree.best_move()[0]
    if verdict is not None:
        assert solved["node_type"] == verdict[0]
        if verdict[1] is not None:
            assert solved["end_in_ply"] == verdict[1]
        assert info["root_visits"] < sims + quota                        # a proven root ends the search early
        assert best is None or pool.best_move(t) == best
    # solver off: same positions keep searching to the limit
    st0 = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota, mcts_solver=0)
    pool0 = search.SearchPool(st0, eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    t0 = pool0.add_position(fen, False, variant)
    pool0.run(simulations=sims, threads=1)
    assert pool0.root_solved(t0)["node_type"] == mo.NT_UNSOLVED and pool0.tree_info(t0)["root_visits"] >= sims
    tree0 = mo.Tree(co.Board(fen, False, variant), mo.Settings(mode=mode, is_policy_map=True, batch_size=quota, mcts_solver=False))
    mo.run_search(tree0, eval_boards, sims, quota)
    assert pool0.root_children(t0)[1] == tree0.root.child_visits
    pool0.close()
    pool.close()


@pytest.mark.parametrize("variant,fen,mode,eps,alpha,seed", [
    ("crazyhouse", "", 0, 0.25, 0.2, 5),                    # the RL build's defaults (Centi_Dirichlet_Epsilon 25, Centi_Dirichlet_Alpha 20)
    ("chess", "r3k2r/pppq1ppp/2npbn2/2b1p3/2B1P3/2NPBN2/PPPQ1PPP/R3K2R w KQkq - 4 8", 1, 0.4, 0.6, 77),
])
def test_dirichlet_noise_at_the_root_equals_oracle(hip_lib, variant, fen, mode, eps, alpha, seed):
    """mctsagent.cpp:311-316: noise on the root priors at the start of every search, then the root is fully expanded.  Both sides
    draw the gamma variates from the C++ standard library with the same seed."""
    nbp, sims, quota = NB_POLICY[mode], 200, 8
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota,
                                 dirichlet_epsilon=eps, dirichlet_alpha=alpha, seed=seed)
    assert search.default_settings().dirichlet_epsilon == 0.0       # play builds: off (optionsuci.cpp:86)

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    def eval_boards(boards):
        out = [_pseudo_net(key_from_board(b), nbp) for b in boards]
        return [o[0] for o in out], [o[1] for o in out]

    pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, False, variant)
    tree = mo.Tree(co.Board(fen or None, False, variant), mo.Settings(mode=mode, is_policy_map=True, batch_size=quota, seed=seed,
                                                                      dirichlet_epsilon=np.float32(eps), dirichlet_alpha=np.float32(alpha)))
    n_legal = len(tree.root.moves)
    for go in range(2):                                     # the second `go` re-noises the already noised priors of the kept root
        pool.run(simulations=sims, threads=1)
        mo.run_search(tree, eval_boards, sims, quota)
        moves, visits, q, pri = pool.root_children(t)
        r = tree.root
        assert len(moves) == n_legal == r.no_visit_idx      # fully expanded: every legal move has its slot
        assert visits == r.child_visits
        assert np.array_equal(q, np.array(r.q, np.float32))
        assert np.allclose(pri, np.array(r.priors, np.float32), rtol=4e-7, atol=0)
        assert abs(float(np.sum(pri)) - 1.0) < 1e-5
        assert pool.best_move(t) == tree.best_move()[0]
    # the noise changed the search: without it the root is not fully expanded after so few simulations
    pool0 = search.SearchPool(search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota),
                              eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    t0 = pool0.add_position(fen, False, variant)
    pool0.run(simulations=sims, threads=1)
    assert len(pool0.root_children(t0)[0]) < n_legal
    pool0.close()
    pool.close()


CASES = [
    # variant, is960, fen, mode, sims, quota(batch), virtual style, temperature
    ("crazyhouse", False, "", 0, 300, 8, 3, 1.7),
    ("crazyhouse", False, "", 0, 200, 16, 0, 1.0),
    ("crazyhouse", False, "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] w KQkq - 0 8", 0, 250, 8, 1, 1.7),
    ("crazyhouse", False, "5r2/ppp2pkp/3p4/2bP4/2Pnp1N1/3P2pP/PP2n1P1/R2Q1R1K[PBRQnbb] w - - 0 28", 0, 300, 8, 3, 1.7),
    ("crazyhouse", False, "4R2b/1N3rkb/1p2P1pp/p2P4/2P1P3/8/PP4Q1/3R3K[QRBBNNNPPPPpp] w - - 2 53", 0, 200, 8, 2, 1.7),   # mates in tree
    ("chess", False, "", 1, 200, 8, 3, 1.7),
    ("chess", True, "bnnrkbrq/pppppppp/8/8/8/8/PPPPPPPP/BNNRKBRQ w GDgd - 0 1", 1, 150, 4, 3, 1.3),
    ("3check", False, "1r4k1/1p2bp1p/3p2p1/PprPp2n/1R2PPq1/3Q4/1P1B1NPP/5RK1 b - - 1+1 2 22", 2, 200, 8, 3, 1.7),
    ("kingofthehill", False, "rnbq1bnr/pppp1ppp/4k3/8/4P3/3K4/PPPP1PPP/RNBQ1BNR w - - 4 5", 2, 150, 8, 3, 1.7),
    # the other lichess variants of the MultiAra build: their terminals (stalemate = win, blown-up king, eighth rank) inside the tree
    ("antichess", False, "rnb1kbnr/pp1ppppp/8/q1p5/8/2P1P3/PP1PNPPP/RNBQKB1R b - - 0 3", 2, 200, 8, 3, 1.7),
    ("antichess", False, "8/8/6p1/7q/8/8/5P2/8 b - - 0 39", 2, 120, 8, 3, 1.7),
    ("atomic", False, "rn1qkb1r/p1p3pp/b3pp1n/3pP3/1P1P1P2/7P/P5P1/RNBQKBNR w KQkq - 1 7", 2, 200, 8, 3, 1.7),
    ("atomic", False, "8/1q6/8/8/8/5k2/1R4n1/1K6 w - - 0 1", 2, 150, 8, 3, 1.7),
    ("horde", False, "", 2, 150, 8, 3, 1.7),
    ("racingkings", False, "", 2, 200, 8, 3, 1.7),
    ("racingkings", False, "2r2N2/kn2R1K1/8/8/8/8/8/8 w - - 8 26", 2, 150, 8, 3, 1.7),
]


@pytest.mark.parametrize("variant,is960,fen,mode,sims,quota,vstyle,temp", CASES)
def test_product_tree_equals_oracle_tree(hip_lib, variant, is960, fen, mode, sims, quota, vstyle, temp):
    nbp = NB_POLICY[mode]
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, virtual_style=vstyle,
                                 node_policy_temperature=temp, batch_size=quota)

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, is960, variant)
    stats = pool.run(simulations=sims, threads=1)
    moves, visits, q, pri = pool.root_children(t)
    info = pool.tree_info(t)

    os_ = mo.Settings(mode=mode, is_policy_map=True, virtual_style=vstyle, node_policy_temperature=np.float32(temp), batch_size=quota)
    board = co.Board(fen or None, is960, variant)
    tree = mo.Tree(board, os_)

    def eval_boards(boards):
        out = [_pseudo_net(key_from_board(b), nbp) for b in boards]
        return [o[0] for o in out], [o[1] for o in out]

    mo.run_search(tree, eval_boards, sims, quota)
    r = tree.root
    p = env.Position(fen, is960, variant)
    assert [p.move_uci(m) for m in moves] == r.uci[:len(moves)]
    assert len(moves) == len(r.child_visits)
    assert visits == r.child_visits
    assert np.array_equal(q, np.array(r.q, np.float32))                       # bit-exact float32 Q values
    # priors: gather is exact; the temperature renormalisation sums float32 terms in move-generation order (the reference:
    # Stockfish's order, blaze's SIMD reduction) -> implementations may differ in the last ulp of that sum
    assert np.allclose(pri, np.array(r.priors[:len(moves)], np.float32), rtol=4e-7, atol=0)
    assert info["root_visits"] == r.visit_sum and info["node_count"] == tree.node_count()
    assert stats.simulations == r.visit_sum and stats.nodes == tree.node_count()
    assert pool.best_move(t) == tree.best_move()[0]
    pool.close()


@pytest.mark.parametrize("variant,fen,mode,greedy,checks,seed", [
    ("crazyhouse", "", 0, 20, 100, 7),                      # the reference's UCI defaults (Centi_Epsilon_Greedy 5, Centi_Epsilon_Checks 1)
    ("crazyhouse", "r1bq1rk1/ppp2ppp/2np1n2/2b1p3/2B1P3/2NP1N2/PPP2PPP/R1BQ1RK1[] w - - 0 7", 0, 5, 0, 11),
    ("chess", "r3k2r/pppq1ppp/2npbn2/2b1p3/2B1P3/2NPBN2/PPPQ1PPP/R3K2R w KQkq - 4 8", 1, 0, 3, 3),
    ("3check", "1r4k1/1p2bp1p/3p2p1/PprPp2n/1R2PPq1/3Q4/1P1B1NPP/5RK1 b - - 1+1 2 22", 2, 4, 6, 99),
])
def test_epsilon_exploration_equals_oracle(hip_lib, variant, fen, mode, greedy, checks, seed):
    """epsilon-greedy / epsilon-checks with the seeded generator: same random playouts, same checking-move probes, same tree."""
    nbp, sims, quota = NB_POLICY[mode], 300, 8
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota,
                                 epsilon_greedy_counter=greedy, epsilon_checks_counter=checks, seed=seed)

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, False, variant)
    pool.run(simulations=sims, threads=1)
    moves, visits, q, _ = pool.root_children(t)
    info = pool.tree_info(t)

    os_ = mo.Settings(mode=mode, is_policy_map=True, batch_size=quota, epsilon_greedy_counter=greedy, epsilon_checks_counter=checks, seed=seed)
    tree = mo.Tree(co.Board(fen or None, False, variant), os_)

    def eval_boards(boards):
        out = [_pseudo_net(key_from_board(b), nbp) for b in boards]
        return [o[0] for o in out], [o[1] for o in out]

    mo.run_search(tree, eval_boards, sims, quota)
    r = tree.root
    p = env.Position(fen, False, variant)
    assert [p.move_uci(m) for m in moves] == r.uci[:len(moves)]
    assert visits == r.child_visits
    assert np.array_equal(q, np.array(r.q, np.float32))
    assert info["root_visits"] == r.visit_sum and info["node_count"] == tree.node_count()
    assert pool.best_move(t) == tree.best_move()[0]
    # the exploration must actually have changed the search: a run without it differs
    st0 = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota)
    pool0 = search.SearchPool(st0, eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    t0 = pool0.add_position(fen, False, variant)
    pool0.run(simulations=sims, threads=1)
    assert pool0.root_children(t0)[1] != visits
    pool0.close()
    pool.close()


def test_pool_many_trees_two_lanes_matches_single_tree_runs(hip_lib):
    """Trees are independent: a pooled, two-lane, multi-threaded run must give each tree exactly the statistics it gets
    when searched alone with the same per-tree quota."""
    nbp = NB_POLICY[0]
    fens = ["", "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] w KQkq - 0 8",
            "5r2/ppp2pkp/3p4/2bP4/2Pnp1N1/3P2pP/PP2n1P1/R2Q1R1K[PBRQnbb] w - - 0 28",
            "r2qk3/1pP2r1n/p1nP4/8/3P1Bb1/2Pp1PP1/PPp2PP1/3q1K1R[Bbnnppr] w - - 2 29"]

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    st = search.default_settings(mode=0, version_major=1)
    # pooled: 4 trees, batch 16 per lane, 2 lanes -> 2 trees per lane -> quota 8 each
    import ctypes as C
    lib = __import__("crazyara_amd._capi", fromlist=["x"]).load()
    pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=16, fn_nb_policy=nbp)
    for f in fens:
        pool.add_position(f, False, "crazyhouse")
    # single lane in callback mode, so 4 trees share 16 slots -> quota 4
    stats = pool.run(simulations=120, threads=3)
    pooled = [pool.root_children(i) for i in range(4)]
    # the last position has a single legal move: "Only single move available -> early stopping" (mctsagent.cpp:303-307), not searched
    assert [len(p[0]) > 0 and sum(p[1]) >= 120 for p in pooled] == [True, True, True, False] and sum(pooled[3][1]) == 0
    assert stats.simulations >= 3 * 120
    for i, f in enumerate(fens):
        solo = search.SearchPool(st, eval_fn=eval_descs, fn_batch=4, fn_nb_policy=nbp)
        solo.add_position(f, False, "crazyhouse")
        solo.run(simulations=120, threads=1)
        m, v, q, pr = solo.root_children(0)
        assert m == pooled[i][0] and v == pooled[i][1] and np.array_equal(q, pooled[i][2])
        solo.close()
    pool.close()


def test_adaptive_quota_shares_the_batch_among_the_running_trees(hip_lib):
    """SearchPool.set_adaptive_quota (throughput setting for self-play): trees that need very different numbers of simulations (absolute
    limits after tree reuse) finish in fewer, fuller batches; every tree still reaches its limit and none overshoots it."""
    nbp = NB_POLICY[0]
    fens = ["", "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] w KQkq - 0 8",
            "5r2/ppp2pkp/3p4/2bP4/2Pnp1N1/3P2pP/PP2n1P1/R2Q1R1K[PBRQnbb] w - - 0 28",
            "r1bqkbnr/pppp1ppp/2n5/4p3/4P3/5N2/PPPP1PPP/RNBQKB1R[] w KQkq - 2 3"]

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    def run(cap):
        st = search.default_settings(mode=0, version_major=1)
        pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=32, fn_nb_policy=nbp)
        for f in fens:
            pool.add_position(f, False, "crazyhouse")
        pool.set_adaptive_quota(cap)
        for t in (1, 2, 3):                               # trees 1..3 are far ahead of tree 0: pause tree 0 for a first run
            pool.set_active(0, False)
        pool.run(simulations=150, threads=2)
        pool.set_active(0, True)
        st2 = pool.run(simulations=200, threads=2)        # tree 0 needs 200, the others 50
        visits = [pool.tree_info(i)["root_visits"] for i in range(4)]
        pool.close()
        return st2.batches, visits

    fixed_batches, fixed_visits = run(0)
    adaptive_batches, adaptive_visits = run(64)
    assert all(v >= 200 for v in fixed_visits) and all(v >= 200 for v in adaptive_visits)
    assert all(v == 200 for v in adaptive_visits), adaptive_visits           # capped by what the tree still needs: no overshoot
    assert adaptive_batches < fixed_batches, (adaptive_batches, fixed_batches)


def test_search_limits_and_terminal_root(hip_lib):
    nbp = NB_POLICY[0]

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    st = search.default_settings(mode=0, version_major=1)
    pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=8, fn_nb_policy=nbp)
    t0 = pool.add_position("", False, "crazyhouse")
    # checkmated root: terminal, never searched
    t1 = pool.add_position("4R2b/1N3rkb/1p2P1pp/p2P3N/2P1P3/8/PP4Q1/3R3K[QRBBNNPPPPpp] b - - 3 53", False, "crazyhouse")
    stats = pool.run(nodes=64, threads=2)
    i0, i1 = pool.tree_info(t0), pool.tree_info(t1)
    assert i0["node_count"] >= 64 and i1["root_visits"] == 0
    assert stats.nodes == i0["node_count"]
    with pytest.raises(RuntimeError):
        pool.run(0, 0, 1)
    pool.close()


@pytest.mark.parametrize("variant,is960,fen,mode,plies,sims", [
    ("crazyhouse", False, "", 0, 10, 120),
    ("chess", True, "bnnrkbrq/pppppppp/8/8/8/8/PPPPPPPP/BNNRKBRQ w GDgd - 0 1", 1, 8, 100),
    ("3check", False, "1r4k1/1p2bp1p/3p2p1/PprPp2n/1R2PPq1/3Q4/1P1B1NPP/5RK1 b - - 1+1 2 22", 2, 8, 100),
])
def test_tree_reuse_across_played_moves_equals_oracle(hip_lib, variant, is960, fen, mode, plies, sims):
    """A game fragment: search, play the chosen move (mi_search_apply_move keeps the subtree below it), search again from the
    kept subtree ... -- the statistics must stay bit-identical to the oracle's tree, which simply follows its child pointer.
    Every second move is an unsearched "opponent surprise" (the least visited legal move), which forces a restart."""
    nbp, quota = NB_POLICY[mode], 8
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota)

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    def eval_boards(boards):
        out = [_pseudo_net(key_from_board(b), nbp) for b in boards]
        return [o[0] for o in out], [o[1] for o in out]

    pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, is960, variant)
    tree = mo.Tree(co.Board(fen or None, is960, variant), mo.Settings(mode=mode, is_policy_map=True, batch_size=quota))
    kept_any = restarted_any = False
    for ply in range(plies):
        pool.run(simulations=sims, threads=1)
        mo.run_search(tree, eval_boards, sims, quota)
        if tree.root.terminal:
            break
        moves, visits, q, _ = pool.root_children(t)
        r = tree.root
        assert visits == r.child_visits and np.array_equal(q, np.array(r.q, np.float32))
        assert pool.tree_info(t)["root_visits"] == r.visit_sum and pool.tree_info(t)["node_count"] == tree.node_count()
        best = pool.best_move(t)
        assert best == tree.best_move()[0]
        if ply % 3 == 2:                                   # a reply the search never expanded
            legal = env.Position(pool.fen(t), is960, variant).legal_uci()
            expanded = set(r.uci[:r.no_visit_idx])
            surprise = [u for u in legal if u not in expanded]
            best = surprise[0] if surprise else best
        kept = pool.apply_move(t, best)
        assert kept == tree.apply_move(best)
        kept_any |= kept
        restarted_any |= not kept
        assert pool.fen(t) == env.Position(tree.root_board.fen(), is960, variant).fen()
        if kept:                                           # the kept subtree starts the next search with its visits
            assert pool.tree_info(t)["root_visits"] == tree.root.visit_sum > 0
    assert kept_any and restarted_any
    with pytest.raises(ValueError):
        pool.apply_move(t, "a1a1")
    pool.close()


def _fen_from_desc(d: bytes) -> str:
    """The descriptor holds the whole position (struct BoardDesc, planes.h): rebuild the FEN of a standard-chess board."""
    bbs = struct.unpack("<12Q", d[0:96])
    rows = []
    for r in range(7, -1, -1):
        row, empty = "", 0
        for f in range(8):
            ch = next(("PNBRQKpnbrqk"[i] for i in range(12) if bbs[i] >> (r * 8 + f) & 1), None)
            if ch is None:
                empty += 1
            else:
                row += (str(empty) if empty else "") + ch
                empty = 0
        rows.append(row + (str(empty) if empty else ""))
    castle = "".join(c for i, c in enumerate("KQkq") if d[123] >> i & 1) or "-"
    ep = "-" if d[124] >= 64 else "abcdefgh"[d[124] & 7] + str((d[124] >> 3) + 1)
    rule50, fullmove = struct.unpack("<HH", d[148:152])
    return f'{"/".join(rows)} {"wb"[d[122]]} {castle} {ep} {rule50} {fullmove}'


def test_chess_v28_leaf_descriptors_carry_the_move_features(hip_lib):
    """Chess input representation 2.7 / 2.8 reads the legal moves (check-giving moves, mobility; inputrepresentation.cpp:382-398).
    The collector fills those descriptor fields from the move list the new node already holds: every descriptor it hands to the
    evaluator must equal the one built from scratch for the same position, and its planes the oracle's."""
    nbp, quota = NB_POLICY[1], 8
    st = search.default_settings(mode=1, version_major=2, version_minor=8, is_policy_map=1, batch_size=quota)
    layout = env.planes_layout(1, "2.8")
    seen = []

    def eval_descs(descs):
        seen.extend(descs)
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    pool.add_position("r1br2k1/p4ppp/2p2n2/Q1b1p3/8/NP3N1P/P1P1BPP1/R1B1K2R b KQ - 0 12", False, "chess")
    pool.run(simulations=150, threads=1)
    assert len(seen) > 100
    with_checks = 0
    for d in seen:
        p = env.Position(_fen_from_desc(d), False, "chess")
        fresh = p.desc(layout)
        assert d[152:169] == fresh[152:169]                               # check_from, check_to, mobility
        assert d[168] == len(p.legal_moves())
        with_checks += d[152:160] != bytes(8)
    assert with_checks > 10
    d = seen[-1]
    b = co.Board(_fen_from_desc(d), False, "chess")
    x = env.planes_from_descs_host(d, 1, layout, True)[0]
    xo = co.board_to_planes(b, 1, "2.8", True)
    keep = [c for c in range(38) if c not in (17, 18)]                    # the FEN does not carry the last move
    assert np.array_equal(x.reshape(38, 64)[keep], xo.reshape(38, 64)[keep])
    # other layouts leave the fields empty (nothing to pay for)
    st3 = search.default_settings(mode=1, version_major=3, is_policy_map=1, batch_size=quota)
    seen.clear()
    pool3 = search.SearchPool(st3, eval_fn=eval_descs, fn_batch=quota, fn_nb_policy=nbp)
    pool3.add_position("", False, "chess")
    pool3.run(simulations=40, threads=1)
    assert seen and all(d[152:169] == bytes(17) for d in seen)


def test_first_and_second_max_reference_cases():
    """engine/tests/tests.cpp:626-646 "Blaze: first_and_second_max()"."""
    assert mo.first_and_second_max([3, 42, 1, 3, 99, 8, 7]) == (99, 42, 4, 1)
    assert mo.first_and_second_max([99, 3, 1, 3, 42, 8, 7]) == (99, 42, 0, 4)


def test_gathered_priors_and_whole_vectors_give_the_same_trees(hip_lib, monkeypatch):
    """The pool asks its lanes for the priors of the new nodes' legal moves only (gathered next to the network output; a batch
    that does not fit the gather buffer falls back to whole probability vectors).  Three settings -- default room, room for 4
    entries per slot (nearly every batch falls back), gather off -- must build bit-identical trees."""
    nbp, quota, sims = NB_POLICY[0], 8, 300
    fen = "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] b KQkq - 0 8"

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    results = []
    for setting in (None, "4", "0"):
        if setting is None:
            monkeypatch.delenv("CRA_GATHER_PER_SLOT", raising=False)
        else:
            monkeypatch.setenv("CRA_GATHER_PER_SLOT", setting)
        st = search.default_settings(mode=0, version_major=1, is_policy_map=1, batch_size=quota, epsilon_greedy_counter=13)
        pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=2 * quota, fn_nb_policy=nbp)
        ids = [pool.add_position(fen, False, "crazyhouse"), pool.add_position("", False, "crazyhouse")]
        pool.run(simulations=sims, threads=2)
        results.append([(pool.root_children(t)[0], pool.root_children(t)[1], pool.root_children(t)[2].tolist(), pool.tree_info(t)) for t in ids])
        pool.close()
    assert results[0] == results[1] == results[2]


@pytest.mark.parametrize("n_trees", [3, 7])
def test_fused_lane_step_equals_two_step(hip_lib, monkeypatch, n_trees):
    """A lane whose trees all fit into its batch finishes the batch in flight and collects the next one in a single fork/join
    (every tree keeps its slots); with more trees than slots (7 trees, 2 lanes of 2 slots) the pool rotates and goes the long way.
    Both, and the forced two-step order, must produce the same trees; trees finish at different times (different limits reached)."""
    nbp, quota = NB_POLICY[0], 8
    fens = ["", "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] b KQkq - 0 8",
            "4R2b/1N3rkb/1p2P1pp/p2P4/2P1P3/8/PP4Q1/3R3K[QRBBNNNPPPPpp] w - - 2 53", "r1b2rk1/pppp1Npp/8/8/8/8/PPPPPPPP/RNBQKB1R[Qq] w KQ - 0 1",
            "", "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] b KQkq - 0 8", ""]

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    results = []
    for two_step in (False, True):
        if two_step:
            monkeypatch.setenv("CRA_POOL_TWO_STEP", "1")
        else:
            monkeypatch.delenv("CRA_POOL_TWO_STEP", raising=False)
        st = search.default_settings(mode=0, version_major=1, is_policy_map=1, batch_size=quota)
        pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=2 * quota, fn_nb_policy=nbp)
        ids = [pool.add_position(f, False, "crazyhouse") for f in fens[:n_trees]]
        pool.run(simulations=260, threads=3)
        pool.run(simulations=90, threads=3)                                   # a second go on the kept trees
        results.append([(pool.root_children(t)[0], pool.root_children(t)[1], pool.root_children(t)[2].tolist(), pool.tree_info(t)) for t in ids])
        pool.close()
    assert results[0] == results[1]


def test_run_survives_a_failing_evaluator(hip_lib, capsys):
    """A failure in the middle of mi_search_run (here: the evaluator callback) leaves no half-applied batch behind: the call
    reports the error, the tree restarts from its position, later searches and moves work and equal a fresh tree's."""
    nbp, calls = NB_POLICY[0], [0]

    def flaky(descs):
        calls[0] += 1
        if calls[0] == 5:
            raise ValueError("boom")
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    st = search.default_settings(mode=0, version_major=1, batch_size=8)
    pool = search.SearchPool(st, eval_fn=flaky, fn_batch=8, fn_nb_policy=nbp)
    t = pool.add_position("", False, "crazyhouse")
    with pytest.raises(RuntimeError):
        pool.run(simulations=200, threads=1)
    capsys.readouterr()
    stats = pool.run(simulations=200, threads=1)                   # the tree restarted: a full search from the root position
    assert pool.tree_info(t)["root_visits"] >= 200 and stats.depth_avg > 0
    fresh = search.SearchPool(st, eval_fn=lambda d: flaky(d), fn_batch=8, fn_nb_policy=nbp)
    fresh.add_position("", False, "crazyhouse")
    fresh.run(simulations=200, threads=1)
    assert pool.root_children(t)[1] == fresh.root_children(0)[1]
    pool.apply_move(t, pool.best_move(t))                          # no "batch in flight" left over
    s2 = pool.run(simulations=260, threads=1)
    assert s2.depth_avg > 0 and s2.depth_max >= 1                  # depth statistics are per run
    pool.close()
    fresh.close()


def _check_tree_invariants(words, allow_virtual=False):
    """Walks a mi_search_tree_dump: per node the children's visits add up to the node's visit counter, no virtual loss is left,
    every record is reachable exactly once.  Returns (records, total visits below the root)."""
    i, records = 0, 0
    root_visits = None
    while i < len(words):
        m, visit_sum, real_visits = int(words[i]), int(words[i + 1]), int(words[i + 2])
        child = words[i + 8:i + 8 + 6 * m].reshape(m, 6)
        assert int(child[:, 1].sum()) == visit_sum, (records, child[:, 1], visit_sum)
        if not allow_virtual:
            assert not child[:, 2].any()                               # virtual-loss counters are back to zero
        assert set(np.unique(child[:, 5])) <= {0, 1, 2}
        assert ((child[:, 5] == 0) <= (child[:, 1] == 0)).all()        # no node yet -> no visits through that move
        if root_visits is None:
            root_visits = visit_sum
        records += 1
        i += 8 + 6 * m
    assert i == len(words)
    return records, root_visits


@pytest.mark.parametrize("k,threads", [(2, 1), (4, 4), (8, 8)])
def test_shared_tree_many_collectors_keeps_the_tree_consistent(hip_lib, k, threads):
    """One tree, k collectors (the reference's Threads SearchThreads on one tree, crazyara.cpp:555-561): per-node locks and virtual
    loss; after the run no virtual loss is left, every node's child visits add up, the limit is met, terminals and mates are found."""
    nbp = NB_POLICY[0]

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    st = search.default_settings(mode=0, version_major=1, batch_size=8)
    for fen in ("", "4R2b/1N3rkb/1p2P1pp/p2P4/2P1P3/8/PP4Q1/3R3K[QRBBNNNPPPPpp] w - - 2 53"):
        pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=32, fn_nb_policy=nbp)
        t = pool.add_position(fen, False, "crazyhouse")
        pool.set_shared_collectors(k)
        stats = pool.run(simulations=600, threads=threads)
        info = pool.tree_info(t)
        records, root_visits = _check_tree_invariants(pool.tree_dump(t))
        assert root_visits == info["root_visits"] >= 600 or pool.root_solved(t)["node_type"] != 6
        assert stats.simulations == info["root_visits"] and stats.nodes == info["node_count"] and records > 50
        assert pool.best_move(t) in env.Position(fen, False, "crazyhouse").legal_uci()
        # a second go on the kept tree, then a played move with tree reuse: the shared tree goes through the same life cycle
        pool.run(simulations=900, threads=threads)
        _check_tree_invariants(pool.tree_dump(t))
        pool.apply_move(t, pool.best_move(t))
        pool.run(simulations=400, threads=threads)
        _check_tree_invariants(pool.tree_dump(t))
        pool.close()


def test_shared_tree_with_one_collector_is_the_plain_tree(hip_lib):
    nbp = NB_POLICY[0]

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    st = search.default_settings(mode=0, version_major=1, batch_size=8)
    dumps = []
    for k in (None, 0):
        pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=8, fn_nb_policy=nbp)
        t = pool.add_position("", False, "crazyhouse")
        if k is not None:
            pool.set_shared_collectors(2)
            pool.set_shared_collectors(0)                           # back to one collector: locks off, same arithmetic
        pool.run(simulations=300, threads=2)
        dumps.append(pool.tree_dump(t))
        pool.close()
    assert np.array_equal(dumps[0], dumps[1])


def test_shared_tree_finds_the_mate_with_the_solver(hip_lib):
    nbp = NB_POLICY[1]

    def eval_descs(descs):
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]

    st = search.default_settings(mode=1, version_major=3, batch_size=8)
    pool = search.SearchPool(st, eval_fn=eval_descs, fn_batch=32, fn_nb_policy=nbp)
    t = pool.add_position("6k1/5ppp/8/8/8/8/8/R3K3 w - - 0 1", False, "chess")
    pool.set_shared_collectors(4)
    pool.run(simulations=2000, threads=4)
    assert pool.root_solved(t)["node_type"] == mo.NT_WIN and pool.best_move(t) == "a1a8"
    _check_tree_invariants(pool.tree_dump(t), allow_virtual=True)       # a proven root ends the search with batches still applied
    pool.close()


def _slow_eval(nbp, delay):
    import time

    def fn(descs):
        time.sleep(delay)
        out = [_pseudo_net(key_from_desc(d), nbp) for d in descs]
        return [o[0] for o in out], [o[1] for o in out]
    return fn


@pytest.mark.parametrize("lanes_threads", [(1, 1), (2, 3)])
def test_movetime_ends_the_search_with_consistent_trees(hip_lib, lanes_threads):
    """SearchLimits::movetime (the timer of ThreadManager::stop_search_based_on_limits, threadmanager.cpp:69-97): a run with no
    simulation / node limit returns a little after the movetime, batches in flight applied -- no virtual loss left, visit counters
    add up -- and the trees carry on in the next run."""
    import time
    nbp = NB_POLICY[0]
    st = search.default_settings(mode=0, version_major=1, batch_size=8)
    pool = search.SearchPool(st, eval_fn=_slow_eval(nbp, 0.002), fn_batch=16, fn_nb_policy=nbp)
    ids = [pool.add_position("", False, "crazyhouse"), pool.add_position("r1bqkbnr/pppp1ppp/2n5/4p3/4P3/5N2/PPPP1PPP/RNBQKB1R[] w KQkq - 2 3", False, "crazyhouse")]
    t0 = time.time()
    stats = pool.run(movetime_ms=250, threads=lanes_threads[1])
    dt = time.time() - t0
    assert 0.24 <= dt < 1.5, dt
    assert stats.simulations > 20
    visits = []
    for t in ids:
        _, rv = _check_tree_invariants(pool.tree_dump(t))
        visits.append(rv)
        assert rv == pool.tree_info(t)["root_visits"] - 1 or rv == pool.tree_info(t)["root_visits"]
    stats2 = pool.run(simulations=max(visits) + 64, threads=lanes_threads[1])      # the kept trees search on
    assert all(pool.tree_info(t)["root_visits"] >= max(visits) + 64 for t in ids)
    # the stricter of movetime and the simulation limit wins
    pool.reset_position(ids[0]); pool.reset_position(ids[1], "r1bqkbnr/pppp1ppp/2n5/4p3/4P3/5N2/PPPP1PPP/RNBQKB1R[] w KQkq - 2 3")
    t0 = time.time()
    pool.run(simulations=40, movetime_ms=20000, threads=lanes_threads[1])
    assert time.time() - t0 < 5.0
    with pytest.raises(RuntimeError, match="limit"):
        pool.run(threads=1)
    pool.close()


def test_stop_from_another_thread_ends_the_run(hip_lib):
    """SearchThread::stop / MCTSAgent::stop (searchthread.cpp:109-112, mctsagent.cpp:364-373): `go infinite` is a run with a limit far
    away and a stop from the UCI thread.  The run returns promptly with consistent trees; a stop without a run is a no-op."""
    import threading
    import time
    nbp = NB_POLICY[0]
    st = search.default_settings(mode=0, version_major=1, batch_size=8)
    pool = search.SearchPool(st, eval_fn=_slow_eval(nbp, 0.002), fn_batch=8, fn_nb_policy=nbp)
    t = pool.add_position("", False, "crazyhouse")
    pool.stop()                                                        # nothing running: ignored, the next run is a full one
    s0 = pool.run(simulations=64, threads=1)
    assert pool.tree_info(t)["root_visits"] >= 64 and s0.simulations >= 60
    result = {}

    def go():
        result["stats"] = pool.run(simulations=50_000_000, threads=2)
    th = threading.Thread(target=go)
    t0 = time.time()
    th.start()
    time.sleep(0.3)
    pool.stop()
    th.join(timeout=10)
    assert not th.is_alive() and time.time() - t0 < 3.0
    assert result["stats"].simulations > 20
    _check_tree_invariants(pool.tree_dump(t))
    assert pool.best_move(t)
    pool.apply_move(t, pool.best_move(t))                              # no batch left in flight
    pool.run(simulations=32, threads=1)
    pool.close()


def test_go_announced_before_the_previous_run_returned_keeps_its_stop(hip_lib):
    """UCI `stop` followed directly by `go` (or ponderhit) without joining the search thread: the next go is announced while the
    previous run() is still on its way out.  Round 4's protocol kept one shared state word that the old run's exit reset, which wiped
    the announcement and let a stop for the NEW search get lost; generations have no such word.  Here: search 1 runs, is
    stopped, search 2 is announced and stopped before search 1's run() has returned -- search 2 must end at once when it is entered, and
    search 3 must run to its limit."""
    import threading
    import time
    nbp = NB_POLICY[0]
    st = search.default_settings(mode=0, version_major=1, batch_size=8)
    pool = search.SearchPool(st, eval_fn=_slow_eval(nbp, 0.05), fn_batch=8, fn_nb_policy=nbp)       # 50 ms per batch: a slow exit
    t = pool.add_position("", False, "crazyhouse")
    result = {}

    def first():
        result["s1"] = pool.run(simulations=50_000_000, threads=1)
    pool.announce_go()
    th = threading.Thread(target=first)
    th.start()
    time.sleep(0.3)                                                    # search 1 is inside run(), a batch in flight
    pool.stop()                                                        # names search 1
    pool.announce_go()                                                 # `go`: search 2, announced while run 1 has not returned yet
    pool.stop()                                                        # ... and stopped again before anybody entered run() for it
    th.join(timeout=10)
    assert not th.is_alive()
    v1 = pool.tree_info(t)["root_visits"]
    t0 = time.time()
    s2 = pool.run(simulations=50_000_000, threads=1)                   # adopts search 2: already stopped
    assert time.time() - t0 < 2.0 and s2.simulations < 64
    assert pool.tree_info(t)["root_visits"] - v1 < 64
    s3 = pool.run(simulations=v1 + 300, threads=1)                     # search 3 was never stopped
    assert pool.tree_info(t)["root_visits"] >= v1 + 300 and s3.simulations > 100
    _check_tree_invariants(pool.tree_dump(t))
    pool.close()


@pytest.mark.parametrize("variant,is960,fen,mode,version,eps", [
    ("crazyhouse", False, "", 0, 1, 0), ("crazyhouse", False, "", 0, 3, 0),          # v3 planes carry last moves: kept subtrees drop their states
    ("crazyhouse", False, "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] w KQkq - 0 8", 0, 2, 20),
    ("chess", False, "r3k2r/pppq1ppp/2npbn2/2b1p3/2B1P3/2NPBN2/PPPQ1PPP/R3K2R b KQkq - 4 8", 1, 3, 0),
    ("chess", True, "bnnrkbrq/pppppppp/8/8/8/8/PPPPPPPP/BNNRKBRQ w GDgd - 0 1", 1, 3, 20),
    ("3check", False, "", 2, 3, 0), ("kingofthehill", False, "", 2, 1, 10),
])
def test_stored_leaf_states_grow_the_same_trees(hip_lib, variant, is960, fen, mode, version, eps):
    """mi_search_set_state_budget (the reference's MCTS_STORE_STATES, searchthread.cpp:198-213): with every new node keeping its position
    an expansion starts from its parent's state instead of a root clone + path replay.  The trees must be the same bit for bit -- over
    two searches, a played move with tree reuse between them (kept subtrees of a crazyhouse tree with last-move planes drop their
    states: their move lists reach back beyond the new root), epsilon exploration (whose steps need the position on the way down), and
    with a budget that runs out in the middle of the search (nodes beyond it replay from the nearest ancestor that has a state)."""
    nbp = NB_POLICY[mode]
    dumps = []
    for budget in (0, 1 << 20, 150):
        st = search.default_settings(mode=mode, version_major=version, is_policy_map=1, batch_size=8, epsilon_greedy_counter=eps,
                                     epsilon_checks_counter=eps * 2, seed=9)
        pool = search.SearchPool(st, eval_fn=_slow_eval(nbp, 0.0), fn_batch=8, fn_nb_policy=nbp)
        pool.set_state_budget(budget)
        t = pool.add_position(fen, is960, variant)
        pool.run(simulations=400, threads=1)
        first = pool.tree_dump(t).copy()
        best = pool.best_move(t)
        kept = pool.apply_move(t, best)
        pool.run(simulations=600, threads=1)
        dumps.append((first, pool.tree_dump(t).copy(), best, kept, pool.best_move(t)))
        _check_tree_invariants(pool.tree_dump(t))
        pool.close()
    for d in dumps[1:]:
        assert np.array_equal(d[0], dumps[0][0]) and np.array_equal(d[1], dumps[0][1]) and d[2:] == dumps[0][2:]


def test_an_announced_go_that_is_never_run_does_not_hand_its_stop_to_the_next_search(hip_lib):
    """`go` is announced, stopped, and the commanding thread never enters run() for it.  The withdrawn announcement
    (mi_search_cancel_go) must not stay the oldest un-run generation: the next announced search runs to its limit, also when another
    idle `stop` arrived in between."""
    nbp = NB_POLICY[0]
    st = search.default_settings(mode=0, version_major=1, batch_size=8)
    pool = search.SearchPool(st, eval_fn=_slow_eval(nbp, 0.0), fn_batch=8, fn_nb_policy=nbp)
    t = pool.add_position("", False, "crazyhouse")
    assert not pool.cancel_go()                                        # nothing announced
    pool.announce_go()
    pool.stop()
    assert pool.cancel_go()                                            # the dropped search: no run() for it
    assert not pool.cancel_go()
    pool.stop()                                                        # an idle stop: names only generations that are over
    pool.announce_go()
    s1 = pool.run(simulations=300, threads=1)
    assert pool.tree_info(t)["root_visits"] >= 300 and s1.simulations > 100
    # two announcements waiting, the first one dropped: the run belongs to the second and ignores the first one's stop
    pool.announce_go()
    pool.stop()
    pool.announce_go()
    assert pool.cancel_go()
    s2 = pool.run(simulations=600, threads=1)
    assert pool.tree_info(t)["root_visits"] >= 600 and s2.simulations > 100
    pool.close()


def test_stop_sent_before_the_search_thread_entered_run_is_not_lost(hip_lib):
    """The reference's stop is sticky (SearchThread::stop sets isRunning = false, searchthread.cpp:109-112; the search loop tests it
    before every mini-batch): `go` announces the search on the commanding thread (mi_search_announce_go), a `stop` that arrives before
    the search thread has entered mi_search_run ends that search at once -- and only that one."""
    import threading
    import time
    nbp = NB_POLICY[0]
    st = search.default_settings(mode=0, version_major=1, batch_size=8)
    pool = search.SearchPool(st, eval_fn=_slow_eval(nbp, 0.002), fn_batch=8, fn_nb_policy=nbp)
    t = pool.add_position("", False, "crazyhouse")
    result = {}

    def go():
        time.sleep(0.2)                                                # the search thread is slow to start
        result["stats"] = pool.run(simulations=50_000_000, threads=2)
    pool.announce_go()
    th = threading.Thread(target=go)
    t0 = time.time()
    th.start()
    pool.stop()                                                        # before run() was entered
    th.join(timeout=10)
    assert not th.is_alive() and time.time() - t0 < 2.0
    assert result["stats"].simulations < 64                            # the root evaluation and at most a batch or two
    assert pool.best_move(t)                                           # ... but a move exists
    _check_tree_invariants(pool.tree_dump(t))
    # the stop belonged to that search: the next one (announced or not) runs to its limit
    s1 = pool.run(simulations=200, threads=1)
    assert pool.tree_info(t)["root_visits"] >= 200 and s1.simulations > 100
    pool.announce_go()
    s2 = pool.run(simulations=400, threads=1)
    assert pool.tree_info(t)["root_visits"] >= 400 and s2.simulations > 100
    # a stop with nothing announced or running is ignored (MCTSAgent::stop: `if (!isRunning) return`)
    pool.stop()
    s3 = pool.run(simulations=600, threads=1)
    assert pool.tree_info(t)["root_visits"] >= 600 and s3.simulations > 100
    pool.close()
