"""The product's leaf collector (csrc/search/mcts.cpp through `mi_search_*`) against THE REFERENCE'S OWN search code.

oracle/_ref/libcrazyara_ref.so is built by oracle/ref/build_ref.py from /root/reference/engine/src (MCTSAgent, SearchThread, Node,
NodeData, EvalInfo, blazeutil, NeuralNetAPI -- unmodified sources; the environment behind the State interface and the blaze
stand-in are ours, see oracle/ref/shim/).  Both sides search the same position with the same settings struct and one evaluator
callback (a deterministic pseudo-network keyed on the position), single-threaded, so every float of every node must agree:
SURVEY 8a rows M1-M10 pinned to the reference's code, not to a restatement of it.

Not comparable, by the reference's own doing:
  * MCTS_Virtual_Style "virtual_offset": revert_virtual_loss_and_update reads an uninitialised `childRealVisit` in that branch
    (node.h:225-229) -- undefined behaviour, the product implements the evidently intended formula (real visits).
  * std::sort tie order of equal priors (node.cpp:464-470): the product's sort is stable; the pseudo-network has no ties.
"""
import zlib

import numpy as np
import pytest

from crazyara_amd import env, search
from oracle import ref_mcts

pytestmark = pytest.mark.skipif(not ref_mcts.available(), reason="oracle/_ref not built and /root/reference absent")

NB_POLICY = {0: 5184, 1: 4864, 2: 5376}
REF_UNSOLVED, PRODUCT_UNSOLVED = 3, 6            # NodeType without / with the tablebase entries (nodedata.h:40-52)


def _pseudo_net(key: bytes, nb_policy: int):
    rs = np.random.Generator(np.random.PCG64(zlib.crc32(key)))
    p = rs.random(nb_policy, dtype=np.float32) ** np.float32(6.0)
    p = p / p.sum(dtype=np.float32)
    return np.float32(rs.random() * 1.6 - 0.8), p


def _evaluator(nbp, log=None):
    def eval_descs(descs):
        if log is not None:
            log.append(len(descs))
        out = [_pseudo_net(d[0:96] + d[112:123], nbp) for d in descs]     # bitboards + pockets + side to move (struct BoardDesc)
        return [o[0] for o in out], [o[1] for o in out]
    return eval_descs


def _normalise_ref_dump(words: np.ndarray) -> np.ndarray:
    """node_type field of every record: the reference build (no tablebases) numbers UNSOLVED 3, the product 6"""
    w = words.copy()
    i = 0
    while i < len(w):
        m = int(w[i])
        if w[i + 4] == REF_UNSOLVED:
            w[i + 4] = PRODUCT_UNSOLVED
        i += 8 + 6 * m
    assert i == len(w)
    return w


def _compare(pool, t, ra, check_best=True, centi_base=None):
    moves, visits, q, pri = pool.root_children(t)
    m2, v2, q2, p2 = ra.root_children()
    assert moves == m2                                         # same moves in the same (prior-sorted) order
    assert visits == v2
    assert np.array_equal(q, q2)                               # float32 bit equality
    assert np.array_equal(pri, p2)
    info, rinfo = pool.tree_info(t), ra.root_info()
    assert info["root_visits"] == rinfo["root_visits"] and info["node_count"] == rinfo["node_count"]
    assert np.float32(info["root_value"]) == np.float32(rinfo["root_value"])
    solved = pool.root_solved(t)
    assert solved["node_type"] == (PRODUCT_UNSOLVED if rinfo["node_type"] == REF_UNSOLVED else rinfo["node_type"])
    assert solved["end_in_ply"] == rinfo["end_in_ply"] and solved["checkmate_idx"] == rinfo["checkmate_idx"]
    # the whole tree, node by node
    assert np.array_equal(pool.tree_dump(t), _normalise_ref_dump(ra.tree_dump()))
    if check_best:
        ev = ra.eval_info()
        assert pool.best_move(t) == ev["best_move"]
        pol, best_q = pool.root_policy(t)
        assert np.array_equal(pol, ev["policy"][:len(pol)]) and not ev["policy"][len(pol):].any()   # EvalInfo pads unexpanded moves with 0
        assert np.float32(best_q) == np.float32(ev["best_q"]) or rinfo["node_type"] != REF_UNSOLVED
        assert ev["nodes"] == info["node_count"]
        # what the UCI front end prints: principal variation, pseudo-centipawns, mate distance (EvalInfo pv / centipawns / movesToMate)
        mine, ref = pool.pv(t), ra.pv()
        assert mine["pv"] == ref["pv"] and mine["moves_to_mate"] == ref["moves_to_mate"]
        if centi_base is None:
            assert mine["centipawns"] == ref["centipawns"]
        elif ref["moves_to_mate"] == 0:
            # VALUE_TO_CENTI_PARAM is a compile-time constant of the build flavour (constants.h:89-93: 1.4 with MODE_CHESS, else 1.2) and
            # oracle/_ref is one binary (1.2): for the chess flavour the reference's formula is re-applied to ITS bestMoveQ with 1.4
            q = np.float32(ev["best_q"])
            want = int(np.sign(q)) * 9999 if abs(q) >= 1 else int(np.float32(-(np.float32(np.sign(q)) * np.log(np.float32(1) - np.abs(q)) / np.log(np.float32(centi_base))) * np.float32(100)))
            assert mine["centipawns"] == want, (mine["centipawns"], want, ref["centipawns"])


CASES = [
    # variant, is960, fen, mode, simulations, batch, temperature
    ("crazyhouse", False, "", 0, 400, 8, 1.7),
    ("crazyhouse", False, "", 0, 300, 16, 1.0),
    ("crazyhouse", False, "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] w KQkq - 0 8", 0, 300, 8, 1.7),
    ("crazyhouse", False, "5r2/ppp2pkp/3p4/2bP4/2Pnp1N1/3P2pP/PP2n1P1/R2Q1R1K[PBRQnbb] w - - 0 28", 0, 300, 8, 1.7),
    ("crazyhouse", False, "4R2b/1N3rkb/1p2P1pp/p2P4/2P1P3/8/PP4Q1/3R3K[QRBBNNNPPPPpp] w - - 2 53", 0, 250, 8, 1.7),     # mates in the tree
    ("chess", False, "", 1, 300, 8, 1.7),
    ("chess", False, "r3k2r/pppq1ppp/2npbn2/2b1p3/2B1P3/2NPBN2/PPPQ1PPP/R3K2R b KQkq - 4 8", 1, 300, 8, 1.7),         # black to move: mirrored policy
    ("chess", True, "bnnrkbrq/pppppppp/8/8/8/8/PPPPPPPP/BNNRKBRQ w GDgd - 0 1", 1, 200, 4, 1.3),
    ("3check", False, "1r4k1/1p2bp1p/3p2p1/PprPp2n/1R2PPq1/3Q4/1P1B1NPP/5RK1 b - - 1+1 2 22", 2, 250, 8, 1.7),
    ("kingofthehill", False, "rnbq1bnr/pppp1ppp/4k3/8/4P3/3K4/PPPP1PPP/RNBQ1BNR w - - 4 5", 2, 200, 8, 1.7),
    ("racingkings", False, "", 2, 200, 8, 1.7),                                                                     # never mirrored
    ("antichess", False, "rnb1kbnr/pp1ppppp/8/q1p5/8/2P1P3/PP1PNPPP/RNBQKB1R b - - 0 3", 2, 200, 8, 1.7),
]


@pytest.mark.parametrize("vstyle", [3, 0, 1], ids=["mix", "loss", "visit"])
@pytest.mark.parametrize("variant,is960,fen,mode,sims,quota,temp", CASES)
def test_product_tree_equals_reference_build(hip_lib, variant, is960, fen, mode, sims, quota, temp, vstyle):
    nbp = NB_POLICY[mode]
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, virtual_style=vstyle,
                                 node_policy_temperature=temp, batch_size=quota)
    if vstyle == 3:
        st.virtual_mix_threshold = 40                           # so that both branches of VIRTUAL_MIX run (default 1000)
    plog, rlog = [], []
    pool = search.SearchPool(st, eval_fn=_evaluator(nbp, plog), fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, is960, variant)
    pool.run(simulations=sims, threads=1)
    ra = ref_mcts.RefAgent(st, _evaluator(nbp, rlog), nbp)
    ra.set_position(fen, is960, variant)
    ra.go(simulations=sims)
    assert plog == rlog                                         # same batches: root alone, then the same leaf counts per mini-batch
    _compare(pool, t, ra, centi_base=1.4 if mode == 1 else None)
    pool.close()
    ra.close()


MCGS_CASES = [
    # positions whose trees hold many transpositions (move-order permutations reach the same position inside a few hundred simulations)
    ("chess", False, "", 1, 1200, 8, 1.7),
    ("chess", False, "8/2k5/8/8/8/8/3K4/R6r w - - 0 1", 1, 800, 8, 1.7),                 # two kings, two rooks: nearly every node is reached twice
    ("crazyhouse", False, "", 0, 800, 16, 1.7),
    ("crazyhouse", False, "8/2k5/8/8/8/8/3K4/R6r[Pp] w - - 0 1", 0, 600, 8, 1.7),         # drops and rook moves permute freely
    ("kingofthehill", False, "", 2, 1000, 8, 1.7),
]


@pytest.mark.parametrize("variant,is960,fen,mode,sims,quota,temp", MCGS_CASES)
def test_mcgs_flag_searches_the_same_tree(hip_lib, variant, is960, fen, mode, sims, quota, temp):
    """`mcts.h` says "mcgs and mcts search the same tree" -- the reference's transposition link
    (Node::add_new_node_to_tree, node.cpp:722-762) takes its candidate from the child slot that is still empty when SearchThread calls it
    (searchthread.cpp:194-211), so useMCGS only fills a hash table nobody reads a node from.  Pinned here on the compiled reference itself:
    the same searches with SearchSettings::useMCGS = true (the UCI default, Search_Type mcgs) and = false dump identical trees, a second `go`
    on the kept tree included, and both equal the product's tree; the positions are transposition-rich (the dump is checked to hold
    positions that occur more than once, i.e. the table had hits to offer)."""
    nbp = NB_POLICY[mode]
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, node_policy_temperature=temp, batch_size=quota)
    seen = []

    def evaluator(descs):
        seen.extend(d[0:96] + d[112:123] for d in descs)
        return _evaluator(nbp)(descs)
    dumps = []
    for mcgs in (True, False):
        ref_mcts.set_use_mcgs(mcgs)
        try:
            ra = ref_mcts.RefAgent(st, evaluator if mcgs else _evaluator(nbp), nbp)
        finally:
            ref_mcts.set_use_mcgs(False)
        ra.set_position(fen, is960, variant)
        ra.go(simulations=sims)
        first = ra.tree_dump().copy()
        ra.go(simulations=2 * sims)                               # "reuse the full tree": the table of the first go is still there
        dumps.append((first, ra.tree_dump().copy(), ra.root_children(), ra.eval_info()["best_move"]))
        if not mcgs:
            pool = search.SearchPool(st, eval_fn=_evaluator(nbp), fn_batch=quota, fn_nb_policy=nbp)
            t = pool.add_position(fen, is960, variant)
            pool.run(simulations=sims, threads=1)
            pool.run(simulations=2 * sims, threads=1)
            _compare(pool, t, ra, centi_base=1.4 if mode == 1 else None)
            pool.close()
        ra.close()
    assert len(set(seen)) < len(seen)                             # some position was expanded at two places of the tree: transpositions exist
    assert np.array_equal(dumps[0][0], dumps[1][0]) and np.array_equal(dumps[0][1], dumps[1][1])
    assert dumps[0][2][0] == dumps[1][2][0] and dumps[0][2][1] == dumps[1][2][1] and dumps[0][3] == dumps[1][3]


@pytest.mark.parametrize("variant,fen,mode,verdict,best", [
    ("chess", "6k1/5ppp/8/8/8/8/8/R3K3 w - - 0 1", 1, (0, 1), "a1a8"),                  # back-rank mate in 1: WIN in 1
    ("antichess", "8/8/8/8/8/4p3/5P1q/8 b - - 0 1", 2, (2, 1), None),                    # both forced captures take White's last piece: LOSS in 1
    ("chess", "7k/R7/1R6/7p/8/8/8/K7 b - - 0 1", 1, None, None),                        # two moves, both answered by Rb8#
    ("chess", "k7/8/1K6/8/8/8/8/7R w - - 0 1", 1, (0, 1), "h1h8"),
    ("crazyhouse", "4R2b/1N3rkb/1p2P1pp/p2P4/2P1P3/8/PP4Q1/3R3K[QRBBNNNPPPPpp] w - - 2 53", 0, None, None),
    ("crazyhouse", "r1b2rk1/pppp1Npp/8/8/8/8/PPPPPPPP/RNBQKB1R[Qq] w KQ - 0 1", 0, None, None),
])
@pytest.mark.parametrize("solver", [1, 0])
def test_solver_equals_reference_build(hip_lib, variant, fen, mode, verdict, best, solver):
    """Node::solve_for_terminal and the solved branches of get_mcts_policy / get_best_action_index (node.cpp:299-453,1070-1148)"""
    nbp, sims, quota = NB_POLICY[mode], 600, 8
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota, mcts_solver=solver)
    pool = search.SearchPool(st, eval_fn=_evaluator(nbp), fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, False, variant)
    pool.run(simulations=sims, threads=1)
    ra = ref_mcts.RefAgent(st, _evaluator(nbp), nbp)
    ra.set_position(fen, False, variant)
    ra.go(simulations=sims)
    _compare(pool, t, ra, centi_base=1.4 if mode == 1 else None)
    rinfo = ra.root_info()
    if solver and verdict is not None:
        assert rinfo["node_type"] == verdict[0] and (verdict[1] is None or rinfo["end_in_ply"] == verdict[1])
        assert rinfo["root_visits"] < sims + quota and best in (None, ra.eval_info()["best_move"])   # a proven root ends the search
    if not solver:
        assert rinfo["node_type"] == REF_UNSOLVED and rinfo["root_visits"] >= sims
    pool.close()
    ra.close()


@pytest.mark.parametrize("variant,fen,mode,eps,alpha,seed", [
    ("crazyhouse", "", 0, 0.25, 0.2, 5),                        # RL defaults (Centi_Dirichlet_Epsilon 25, Centi_Dirichlet_Alpha 20)
    ("chess", "r3k2r/pppq1ppp/2npbn2/2b1p3/2B1P3/2NPBN2/PPPQ1PPP/R3K2R w KQkq - 4 8", 1, 0.4, 0.6, 77),
])
def test_dirichlet_noise_equals_reference_build(hip_lib, variant, fen, mode, eps, alpha, seed):
    """mctsagent.cpp:311-316 + node.cpp:950-954 + blazeutil.h:113-124: noise at the start of every go, root fully expanded; the
    reference's generator (std::default_random_engine of node.cpp's translation unit) is seeded like the product tree's."""
    nbp, sims, quota = NB_POLICY[mode], 200, 8
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota,
                                 dirichlet_epsilon=eps, dirichlet_alpha=alpha, seed=seed)
    pool = search.SearchPool(st, eval_fn=_evaluator(nbp), fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, False, variant)
    ra = ref_mcts.RefAgent(st, _evaluator(nbp), nbp)
    ra.set_position(fen, False, variant)
    for go in range(2):                                         # the second go re-noises the kept root ("reuse the full tree")
        pool.run(simulations=sims, threads=1)
        ra.go(simulations=sims)
        _compare(pool, t, ra, centi_base=1.4 if mode == 1 else None)
        assert len(ra.root_children()[0]) == ra.root_info()["n_legal"]
    pool.close()
    ra.close()


@pytest.mark.parametrize("variant,fen,mode,greedy,checks,seed", [
    ("crazyhouse", "", 0, 20, 100, 7),                          # UCI defaults (Centi_Epsilon_Greedy 5, Centi_Epsilon_Checks 1)
    ("crazyhouse", "r1bq1rk1/ppp2ppp/2np1n2/2b1p3/2B1P3/2NP1N2/PPP2PPP/R1BQ1RK1[] w - - 0 7", 0, 5, 0, 11),
    ("chess", "r3k2r/pppq1ppp/2npbn2/2b1p3/2B1P3/2NPBN2/PPPQ1PPP/R3K2R w KQkq - 4 8", 1, 0, 3, 3),
    ("3check", "1r4k1/1p2bp1p/3p2p1/PprPp2n/1R2PPq1/3Q4/1P1B1NPP/5RK1 b - - 1+1 2 22", 2, 4, 6, 99),
])
def test_epsilon_exploration_equals_reference_build(hip_lib, variant, fen, mode, greedy, checks, seed):
    """searchthread.cpp:124-185,451-501 with rand() bound to the generator every product tree owns (oracle/ref/ref_driver.cpp)"""
    nbp, sims, quota = NB_POLICY[mode], 400, 8
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota,
                                 epsilon_greedy_counter=greedy, epsilon_checks_counter=checks, seed=seed)
    pool = search.SearchPool(st, eval_fn=_evaluator(nbp), fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, False, variant)
    pool.run(simulations=sims, threads=1)
    ra = ref_mcts.RefAgent(st, _evaluator(nbp), nbp)
    ra.set_position(fen, False, variant)
    ra.go(simulations=sims)
    _compare(pool, t, ra, centi_base=1.4 if mode == 1 else None)
    pool.close()
    ra.close()


@pytest.mark.parametrize("variant,is960,fen,mode", [
    ("crazyhouse", False, "", 0),
    ("chess", True, "bnnrkbrq/pppppppp/8/8/8/8/PPPPPPPP/BNNRKBRQ w GDgd - 0 1", 1),
    ("3check", False, "", 2),
])
def test_tree_reuse_across_played_moves_equals_reference_build(hip_lib, variant, is960, fen, mode):
    """MCTSAgent::apply_move_to_tree + get_root_node_from_tree (mctsagent.cpp:130-164,230-247): one engine plays both sides of a
    game fragment; after every go the best move is played and the subtree below it is the next root (or the tree restarts)."""
    nbp, sims, quota = NB_POLICY[mode], 120, 8
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=quota)
    pool = search.SearchPool(st, eval_fn=_evaluator(nbp), fn_batch=quota, fn_nb_policy=nbp)
    t = pool.add_position(fen, is960, variant)
    ra = ref_mcts.RefAgent(st, _evaluator(nbp), nbp)
    ra.set_position(fen, is960, variant)
    kept_any = False
    for ply in range(8):
        pool.run(simulations=sims, threads=1)
        ra.go(simulations=sims)
        _compare(pool, t, ra, centi_base=1.4 if mode == 1 else None)
        mv = ra.eval_info()["best_move"]
        kept_any |= pool.apply_move(t, mv)
        ra.apply_move(mv)
        assert pool.fen(t) == ra.fen()
    assert kept_any
    pool.close()
    ra.close()


def test_reference_helpers_worked_examples():
    """Single functions of the compiled reference against hand-computed values and the reference's own test
    (first_and_second_max, engine/tests/tests.cpp:626-646)."""
    import ctypes as C
    lib = ref_mcts.load()
    assert abs(lib.ref_get_current_cput(0.0, 2.5, 19652.0) - (np.log(19653.0 / 19652.0) + 2.5)) < 1e-6      # node.cpp:1243-1246
    v = np.array([5, 3, 7, 1, 9, 2], np.float32)               # tests.cpp:629-636 shape: first 9 at 4, second 7 at 2
    f, s, fa, sa = C.c_float(), C.c_float(), C.c_int(), C.c_int()
    lib.ref_first_and_second_max(v.ctypes.data_as(C.POINTER(C.c_float)), 6, 6, C.byref(f), C.byref(s), C.byref(fa), C.byref(sa))
    assert (f.value, s.value, fa.value, sa.value) == (9.0, 7.0, 4, 2)
    # get_quantile (blazeutil.h:188-212): sorts, returns 0 if the smallest entry already reaches the quantile, else accumulates from
    # the SECOND smallest and returns the previous entry + FLT_EPSILON
    p = np.array([0.1, 0.2, 0.7], np.float64)
    assert lib.ref_get_quantile(p.ctypes.data_as(C.POINTER(C.c_double)), 3, 0.05) == 0.0
    q = lib.ref_get_quantile(p.ctypes.data_as(C.POINTER(C.c_double)), 3, 0.25)
    assert abs(q - (0.2 + np.finfo(np.float32).eps)) < 1e-7     # sum: 0.2 (<0.25), 0.9 (>=0.25) -> previous entry 0.2
    lib.ref_apply_quantile_clipping(p.ctypes.data_as(C.POINTER(C.c_double)), 3, 0.25)
    assert np.allclose(p, [0.0, 0.0, 1.0])
    assert lib.ref_value_to_centipawn(1.0) == 9999 and lib.ref_value_to_centipawn(0.0) == 0
    # the product's ports of the two -- the library's (csrc/rl/selfplay.cpp, used by the native game loops) and the numpy restatement
    # (crazyara_amd/selfplay.py) -- against the compiled functions
    from crazyara_amd import _capi, selfplay
    hip = _capi.load()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rng = np.random.default_rng(3)
    for trial in range(200):
        n = int(rng.integers(2, 40))
        p = rng.random(n) ** 3
        p /= p.sum()
        quant = float(rng.choice([0.05, 0.25, 0.5, 0.9]))
        if 1.0 - p.min() < quant + 1e-3:
            continue                                            # the compiled function would hit its assert(false), blazeutil.h:211
        ref_q = lib.ref_get_quantile(p.ctypes.data_as(C.POINTER(C.c_double)), n, quant)
        assert np.float32(selfplay.get_quantile(p, quant)) == np.float32(ref_q)
        clipped = p.copy()
        lib.ref_apply_quantile_clipping(clipped.ctypes.data_as(C.POINTER(C.c_double)), n, quant)
        assert np.array_equal(selfplay.apply_quantile_clipping(quant, p), clipped)
        assert np.float32(hip.mi_policy_get_quantile(dp(p), n, quant)) == np.float32(ref_q)
        mine = p.copy()
        hip.mi_policy_apply_quantile_clipping(dp(mine), n, quant)
        assert np.array_equal(mine, clipped)
        # sharpen_distribution (blazeutil.h:94-105): the exported policy of self-play under lowPolicyClipThreshold
        thresh = float(rng.choice([0.0, 0.01, 0.03, 0.1, 0.9]))
        sharp_ref, sharp = p.copy(), p.copy()
        lib.ref_sharpen_distribution(dp(sharp_ref), n, thresh)
        hip.mi_policy_sharpen_distribution(dp(sharp), n, thresh)
        assert np.array_equal(sharp, sharp_ref)
        temp = float(rng.choice([0.5, 1.0, 2.0, 10.0]))
        tp = p.copy()
        hip.mi_policy_apply_temperature(dp(tp), n, temp)
        assert np.allclose(tp, selfplay.apply_temperature(p.copy(), temp), rtol=1e-14, atol=0)


def test_time_for_move_equals_the_reference_time_manager():
    """mi_time_for_move against TimeManager::get_time_for_move of the reference build (timemanager.cpp:50-103, randomMoveFactor 0) over
    the branches of the function: infinite, node / simulation / depth limits with and without movetime, movestogo, sudden death before
    and after the proportional threshold (move 35), increments, tiny clocks, no limits at all (1000 ms), both sides."""
    rng = np.random.default_rng(5)
    cases = []
    for _ in range(400):
        kind = rng.integers(0, 7)
        kw = dict(move_overhead=int(rng.choice([0, 20, 100, 500])))
        if kind == 0:
            kw.update(infinite=True, wtime=int(rng.integers(0, 60000)))
        elif kind == 1:
            kw.update(movetime=int(rng.integers(1, 5000)))
        elif kind == 2:
            kw.update(nodes=int(rng.integers(0, 2) * 800), simulations=int(rng.integers(0, 2) * 400), depth=int(rng.integers(0, 2) * 5),
                      movetime=int(rng.integers(0, 2) * rng.integers(1, 3000)), wtime=int(rng.integers(0, 2) * 30000), btime=20000)
        elif kind == 3:
            kw.update(movestogo=int(rng.integers(1, 40)), wtime=int(rng.integers(1, 300000)), btime=int(rng.integers(1, 300000)),
                      winc=int(rng.integers(0, 3000)), binc=int(rng.integers(0, 3000)))
        elif kind == 4:
            kw.update(wtime=int(rng.integers(1, 600000)), btime=int(rng.integers(1, 600000)), winc=int(rng.integers(0, 5000)),
                      binc=int(rng.integers(0, 5000)))
        elif kind == 5:
            kw.update(wtime=int(rng.integers(1, 700)), btime=int(rng.integers(1, 700)))               # less than the safety buffer
        cases.append((int(rng.integers(0, 2)), int(rng.integers(0, 80)), kw))
    for side, move_number, kw in cases:
        lim = search.GoLimitsC(kw.get("movetime", 0), kw.get("nodes", 0), kw.get("simulations", 0), kw.get("movestogo", 0), kw.get("depth", 0),
                               (search.C.c_int * 2)(kw.get("wtime", 0), kw.get("btime", 0)), (search.C.c_int * 2)(kw.get("winc", 0), kw.get("binc", 0)),
                               kw.get("move_overhead", 20), int(kw.get("infinite", False)))
        want = ref_mcts.time_for_move(lim, side, move_number)
        got = search.time_for_move(side, move_number, **kw)
        assert got == want, (side, move_number, kw, got, want)
    with pytest.raises(ValueError):
        search.time_for_move(2, 0)


@pytest.mark.parametrize("variant,is960,fen,mode,sims", [
    ("crazyhouse", False, "", 0, 400),
    ("crazyhouse", False, "4R2b/1N3rkb/1p2P1pp/p2P4/2P1P3/8/PP4Q1/3R3K[QRBBNNNPPPPpp] w - - 2 53", 0, 300),     # proven lines among the top moves
    ("chess", False, "r3k2r/pppq1ppp/2npbn2/2b1p3/2B1P3/2NPBN2/PPPQ1PPP/R3K2R b KQkq - 4 8", 1, 300),
    ("3check", False, "1r4k1/1p2bp1p/3p2p1/PprPp2n/1R2PPq1/3Q4/1P1B1NPP/5RK1 b - - 1+1 2 22", 2, 300),
])
def test_multipv_lines_equal_the_reference_build(hip_lib, variant, is960, fen, mode, sims):
    """Multi_PV = 4: the lines update_eval_info writes (root move of rank idx in the MCTS policy, Node::get_principal_variation below it,
    bestMoveQ / centipawns / movesToMate per line), product against the reference build.  Lines whose policy entry ties with a neighbour's
    are left out of the comparison (the reference orders them with an unstable std::sort)."""
    nbp = NB_POLICY[mode]
    st = search.default_settings(mode=mode, version_major=1 if mode == 0 else 3, is_policy_map=1, batch_size=8)
    pool = search.SearchPool(st, eval_fn=_evaluator(nbp), fn_batch=8, fn_nb_policy=nbp)
    t = pool.add_position(fen, is960, variant)
    pool.run(simulations=sims, threads=1)
    ra = ref_mcts.RefAgent(st, _evaluator(nbp), nbp)
    ra.set_position(fen, is960, variant)
    ra.set_multipv(4)
    ra.go(simulations=sims)
    mine, ref = pool.pv_multi(t, 4), ra.pv_multi(4)
    assert len(mine) == len(ref) >= 2
    pol, _ = pool.root_policy(t)
    ranked = np.sort(np.asarray(pol, np.float32))[::-1]
    compared = 0
    for idx, (a, b) in enumerate(zip(mine, ref)):
        tie = idx > 0 and ((idx + 1 < len(ranked) and ranked[idx] == ranked[idx + 1]) or ranked[idx] == ranked[idx - 1])
        if tie:
            continue
        assert a["pv"] == b["pv"] and a["moves_to_mate"] == b["moves_to_mate"], (idx, a, b)
        assert np.float32(a["best_move_q"]) == np.float32(b["best_move_q"]), (idx, a, b)
        if mode != 1 or b["moves_to_mate"] != 0:
            assert a["centipawns"] == b["centipawns"]                    # (the chess flavour's constant: see _compare)
        compared += 1
    assert compared >= 2
    assert mine[0]["pv"] == pool.pv(t)["pv"]                              # line 0 is the single-PV line
    assert pool.pv_multi(t, 1) == mine[:1]
    pool.close()
