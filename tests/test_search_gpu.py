"""GPU: the whole hot path end to end -- position -> 192-byte descriptor -> GPU plane builder -> RISE forward -> policy
gather in the leaf collector -- against the oracle chain (oracle planes -> oracle NN -> oracle policy index)."""
import numpy as np
import pytest
import torch

import nn_cases
from crazyara_amd import env, openings, search
from crazyara_amd.neuralnetapi import HipAPI
from oracle import chess_oracle as co
from oracle import rise_oracle as ro

pytestmark = pytest.mark.gpu


def test_root_priors_and_value_match_oracle_chain(tmp_path, hip_lib):
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    fens = openings.position_fens("crazyhouse")[::23][:8]
    net = HipAPI(0, 8, d, "float32")
    st = search.default_settings(mode=0, version_major=1, node_policy_temperature=1.0)
    pool = search.SearchPool(st, net_a=net)
    for f in fens:
        pool.add_position(f, False, "crazyhouse")
    stats = pool.run(simulations=1, threads=2)          # evaluates the roots (+1 simulation each)
    assert stats.nn_evals >= len(fens)
    pm = co.PolicyMap(co.MODE_CRAZYHOUSE)
    for i, f in enumerate(fens):
        b = co.Board(f, False, "crazyhouse")
        x = torch.from_numpy(co.board_to_planes(b, co.MODE_CRAZYHOUSE, 1, True)[None])
        v, p, _ = ro.predict(cfg, sd, x)
        exp = {b.move_uci(m): float(p[0, pm.index(b, m, True)]) for m in b.legal_moves()}
        # root node keeps every legal move with its gathered prior (T = 1: no renormalisation)
        lib = __import__("crazyara_amd._capi", fromlist=["x"]).load()
        info = pool.tree_info(i)
        assert abs(info["root_value"] - float(v[0])) < 1e-4 or info["root_visits"] > 0
        moves, visits, q, pri = pool.root_children(i)
        pos = env.Position(f, False, "crazyhouse")
        for m, pr in zip(moves, pri):
            assert abs(exp[pos.move_uci(m)] - float(pr)) < 1e-6
        # children are sorted by descending prior, so the first expanded child is the oracle's argmax move
        best = max(exp, key=exp.get)
        assert abs(exp[pos.move_uci(moves[0])] - exp[best]) < 1e-7
    pool.close()
    net.close()


def test_two_lane_pool_on_opening_set(tmp_path, hip_lib):
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    a, b = HipAPI(0, 64, d, "float16"), HipAPI(0, 64, d, "float16")
    st = search.default_settings(mode=0, version_major=1)
    pool = search.SearchPool(st, net_a=a, net_b=b)
    fens = openings.position_fens("crazyhouse")[:16]
    for f in fens:
        pool.add_position(f, False, "crazyhouse")
    stats = pool.run(simulations=200, threads=4)
    assert stats.simulations >= 16 * 200 and stats.nodes > 0 and stats.nn_evals > 16 and stats.seconds > 0
    for i, f in enumerate(fens):
        info = pool.tree_info(i)
        assert info["root_visits"] >= 200 and info["node_count"] <= info["root_visits"]
        assert pool.best_move(i) in env.Position(f, False, "crazyhouse").legal_uci()
    # second `go` on the same pool reuses the trees (tree reuse, mctsagent.cpp:136-164); the limit is absolute on the root's visits
    # (nodes_limits_ok, searchthread.cpp:326-331): a go to 50 finds every root beyond it, a go to 250 adds the missing visits
    assert pool.run(simulations=50, threads=4).simulations == 0
    stats2 = pool.run(simulations=250, threads=4)
    assert stats2.simulations >= 16 * 40 and all(pool.tree_info(i)["root_visits"] >= 250 for i in range(16))
    pool.close()
    a.close()
    b.close()


def test_lane_count_does_not_change_the_trees(tmp_path, hip_lib):
    """A tree only ever sees its own fixed quota of every batch, so the searches of the first four trees are the same whether
    the pool runs 1, 2 or 3 evaluator lanes (mi_search_add_lane; 4 trees per lane -> quota 16 each).  Solver and root Dirichlet
    noise are switched on so that those paths run against the real network as well."""
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    fens = openings.position_fens("crazyhouse")[5:17]
    results = []
    for lanes in (1, 2, 3):
        nets = [HipAPI(0, 64, d, "float32") for _ in range(lanes)]
        st = search.default_settings(mode=0, version_major=1, batch_size=16, dirichlet_epsilon=0.25, seed=9)
        pool = search.SearchPool(st, net_a=nets[0], net_b=nets[1] if lanes > 1 else None)
        for n in nets[2:]:
            pool.add_lane(n)
        for f in fens[:4 * lanes]:
            pool.add_position(f, False, "crazyhouse")
        pool.run(simulations=160, threads=3)
        results.append([(pool.root_children(i)[1], pool.best_move(i)) for i in range(4)])
        pool.close()
        for n in nets:
            n.close()
    assert results[0] == results[1] == results[2]
    for (visits, bm), f in zip(results[0], fens):
        assert sum(visits) >= 160 and bm in env.Position(f, False, "crazyhouse").legal_uci()
        assert len(visits) == len(env.Position(f, False, "crazyhouse").legal_uci())      # Dirichlet: root fully expanded


def test_lanes_on_two_devices_search_the_same_trees(tmp_path, hip_lib):
    """The reference's other multi-GPU mode: evaluator lanes of ONE pool whose nets sit on different devices (Threads x (Last_Device_ID -
    First_Device_ID + 1) SearchThreads, uci/crazyara.cpp:555-561,734).  Every mi_net call selects its own device, so a pool with a lane on
    device 0 and a lane on device 1 searches exactly the trees of two lanes on device 0.  Needs two visible devices: skipped on the one-GPU
    boxes this repository is developed on (never run so far -- DESIGN 8)."""
    from crazyara_amd import _capi
    if _capi.load().mi_device_count() < 2:
        pytest.skip("one visible device")
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    fens = openings.position_fens("crazyhouse")[5:13]
    results = []
    for devices in ((0, 0), (0, 1)):
        nets = [HipAPI(dev, 64, d, "float32") for dev in devices]
        st = search.default_settings(mode=0, version_major=1, batch_size=16, seed=9)
        pool = search.SearchPool(st, net_a=nets[0], net_b=nets[1])
        for f in fens:
            pool.add_position(f, False, "crazyhouse")
        pool.run(simulations=160, threads=3)
        results.append([(pool.root_children(i)[1], pool.best_move(i)) for i in range(len(fens))])
        pool.close()
        for n in nets:
            n.close()
    assert results[0] == results[1]


@pytest.mark.parametrize("precision", ["float16", "fp8", "float16x3", "float16p8"])
def test_priors_gathered_on_the_gpu_equal_whole_probability_vectors(tmp_path, hip_lib, monkeypatch, precision):
    """The HIP lanes bring back only the probabilities of the new nodes' legal moves (gather kernel behind the forward, ~40 KB per
    batch instead of 5.3 MB).  The same searches with the gather switched off, and with room for 4 entries per slot (fallback on
    nearly every batch), must give the same trees down to the last Q bit."""
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    fens = openings.position_fens("crazyhouse")[20:28]
    results = []
    # a lane step is one launch at this batch size (the forward kernel builds the planes from the descriptors and writes the gathered
    # priors itself), two launches for large batches (plane builder, then the forward whose head writes the priors) -- and three
    # launches (plane builder, forward, gather kernel) when forced or when the forward is not one kernel (float16x3): the same trees
    for setting, launches in ((None, None), ("0", None), ("4", None), (None, "1"), (None, "2"), (None, "3")):
        if setting is None:
            monkeypatch.delenv("CRA_GATHER_PER_SLOT", raising=False)
        else:
            monkeypatch.setenv("CRA_GATHER_PER_SLOT", setting)
        if launches is None:
            monkeypatch.delenv("CRA_LANE_LAUNCHES", raising=False)
        else:
            monkeypatch.setenv("CRA_LANE_LAUNCHES", launches)
        nets = [HipAPI(0, 64, d, precision) for _ in range(2)]
        st = search.default_settings(mode=0, version_major=1, batch_size=16, seed=3)
        pool = search.SearchPool(st, net_a=nets[0], net_b=nets[1])
        for f in fens:
            pool.add_position(f, False, "crazyhouse")
        pool.run(simulations=240, threads=4)
        results.append([(pool.root_children(i)[1], pool.root_children(i)[2].tolist(), pool.tree_info(i)) for i in range(len(fens))])
        pool.close()
        for n in nets:
            n.close()
    assert all(r == results[0] for r in results[1:])
    assert all(sum(r[0]) >= 239 for r in results[0])


def test_selfplay_loop_on_the_gpu_chess960(tmp_path, hip_lib):
    """BASELINE config 4 in miniature: concurrent chess960 self-play games on one GPU with a real (random-init) net: raw-policy
    opening plies, temperature sampling, tree reuse; every recorded move is legal and every game ends."""
    from crazyara_amd import _capi, selfplay
    cfg, sd, _ = nn_cases.make_case("risev33")
    d = nn_cases.export_case(tmp_path, "risev33", cfg, sd, version="3.0")
    a, b, raw = (HipAPI(0, 32, d, "float16") for _ in range(3))
    st = search.default_settings(mode=1, version_major=3, batch_size=8, seed=4)
    pool = search.SearchPool(st, net_a=a, net_b=b)
    s = selfplay.SelfPlaySettings(variant="chess", is960=True, simulations=40, max_plies=40, mean_init_ply=3.0,
                                  init_temperature=0.8, temperature_moves=6, temperature_decay=0.9, seed=2)
    lib = _capi.load()
    loop = selfplay.SelfPlay(pool, s, 8, start_fen=lambda i: lib.mi_chess960_start_fen((i * 91 + 5) % 960).decode(),
                             raw_policy=selfplay.net_raw_policy(raw, 1, 3))
    games = loop.play(10, threads=4)
    assert len(games) == 10 and loop.stats["kept_subtrees"] > 0 and loop.stats["nn_evals"] > 0
    assert len({g.start_fen for g in games}) > 1
    for g in games:
        p = env.Position(g.start_fen, True, "chess")
        for u in g.uci:
            assert p.push_uci(u), (g.start_fen, g.uci, u)
        assert g.result in (1, 0, -1) and g.variant == "chess960" and (p.terminal() != env.TERMINAL_NONE) == (g.termination == "terminal")
    pool.close()
    for n in (a, b, raw):
        n.close()


def test_one_tree_many_collectors_on_the_gpu(tmp_path, hip_lib):
    """The single-`go` mode on real lanes: ONE tree, two nets in flight, 4 collectors per lane collecting in parallel under per-node
    locks (the reference: Threads SearchThreads on one tree, crazyara.cpp:555-561).  The tree stays consistent (no virtual loss left,
    child visits add up), the limit is met and the move is legal; one collector per lane on one thread is still a valid search."""
    from test_mcts import _check_tree_invariants
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    a, b = HipAPI(0, 64, d, "float16"), HipAPI(0, 64, d, "float16")
    st = search.default_settings(mode=0, version_major=1)
    for k, threads in ((4, 8), (1, 1)):
        pool = search.SearchPool(st, net_a=a, net_b=b)
        t = pool.add_position("", False, "crazyhouse")
        pool.set_shared_collectors(k)
        stats = pool.run(simulations=3000, threads=threads)
        info = pool.tree_info(t)
        records, root_visits = _check_tree_invariants(pool.tree_dump(t))
        assert root_visits == info["root_visits"] >= 3000 and stats.nodes == info["node_count"] and records > 200
        assert stats.nn_evals / max(1, stats.batches) > (32 if k == 4 else 8)          # batches are filled from one tree
        assert pool.best_move(t) in env.Position("", False, "crazyhouse").legal_uci()
        pool.apply_move(t, pool.best_move(t))
        pool.run(simulations=3000, threads=threads)
        _check_tree_invariants(pool.tree_dump(t))
        pool.close()
    a.close()
    b.close()


@pytest.mark.parametrize("precision", ["float16x3", "float16p8"])
def test_few_boards_on_a_net_made_for_many_take_the_small_forward(tmp_path, hip_lib, precision, monkeypatch):
    """Round 6: a float16x3 / float16p8 net made for 256 boards that is handed n <= 64 (the root of a `go`, the first batches of one tree)
    runs a forward of n boards on its companion net (the split-board launches with the boards of THIS call) instead of the whole batch's
    tower: the same numbers within float16x3's bounds whatever n, bit-identical from call to call, and faster; CRA_NO_SMALL_PATH=1 is the old
    behaviour (the A/B and the reference for the numbers)."""
    import ctypes as C
    import time
    from crazyara_amd import _capi, env, openings
    from crazyara_amd.neuralnetapi import HipAPI
    from oracle import rise_oracle as ro
    cfg, sd, _ = nn_cases.make_case("risev2-7")
    d = nn_cases.export_case(tmp_path, "risev2-7", cfg, sd)
    lib = _capi.load()
    B = 256
    fens = openings.position_fens("crazyhouse")
    fens = [fens[i % len(fens)] for i in range(B)]
    pos = [env.Position(f, False, "crazyhouse") for f in fens]
    descs = b"".join(p.desc() for p in pos)
    planes = torch.from_numpy(np.stack([p.planes(0, 1, True) for p in pos]).astype(np.float32))
    o_value, o_logits, _ = ro.forward(cfg, sd, planes)
    o_probs = torch.softmax(o_logits, 1).numpy()
    layout = lib.mi_planes_layout(0, 1)
    dbuf = lib.mi_host_alloc(len(descs)); C.memmove(dbuf, descs, len(descs))
    v = lib.mi_host_alloc(4 * B); p = lib.mi_host_alloc(4 * B * cfg.nb_policy)
    va = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_float)), shape=(B,))
    pa = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(B, cfg.nb_policy))
    tol = 1e-4 if precision == "float16x3" else 3e-4
    times = {}
    for small in (True, False):
        if small:
            monkeypatch.delenv("CRA_NO_SMALL_PATH", raising=False)
        else:
            monkeypatch.setenv("CRA_NO_SMALL_PATH", "1")
        net = HipAPI(0, B, d, precision)
        for n in (1, 5, 33, 64, 65, 256):
            outs = []
            for rep in range(3):
                va[:] = 7.0; pa[:] = 7.0
                t0 = time.perf_counter()
                assert lib.mi_net_submit_boards(net._h, dbuf, n, layout, v, p, None) == 0, _capi.last_error()
                net.wait()
                dt = time.perf_counter() - t0
                outs.append((va[:n].copy(), pa[:n].copy()))
                if rep == 2:
                    times[(small, n)] = dt
            assert np.abs(outs[0][0] - o_value.numpy().reshape(-1)[:n]).max() < tol
            assert np.abs(outs[0][1] - o_probs[:n]).max() < 1e-5
            for a, b_ in outs[1:]:
                assert np.array_equal(a, outs[0][0]) and np.array_equal(b_, outs[0][1])
        net.close()
    print("seconds per call (small path, old path):", {n: (round(times[(True, n)] * 1e3, 3), round(times[(False, n)] * 1e3, 3)) for n in (1, 5, 33, 64, 65, 256)})
    assert times[(True, 1)] < 0.9 * times[(False, 1)]
    lib.mi_host_free(dbuf); lib.mi_host_free(v); lib.mi_host_free(p)


def test_nets_of_one_process_work_on_different_hardware_queues(tmp_path, hip_lib, monkeypatch):
    """Round 6: nets take their stream from a per-device set the library keeps (one per hardware queue, never destroyed), the one unused
    the longest -- not a stream created per net, whose hardware queue depended on how many idle nets the process still held
    (scripts/ubench/stream_queues.hip).  Four nets open at once work in four different streams; a fifth shares the one that has been
    idle the longest; a net that is closed frees nothing that matters; every net still computes the same numbers in whichever stream."""
    from crazyara_amd import _capi
    cfg, sd, _ = nn_cases.make_case("risev2-7")
    d = nn_cases.export_case(tmp_path, "risev2-7", cfg, sd)
    lib = _capi.load()
    monkeypatch.delenv("CRA_OWN_STREAM_PER_NET", raising=False)
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    x = np.ascontiguousarray(torch.rand(8, cfg.nb_input_channels, 8, 8).numpy())

    def run(n):
        v = np.full(8, 7.0, np.float32)
        p = np.full(8 * cfg.nb_policy, 7.0, np.float32)
        n.predict(x, v, p, np.full(8 * 4, 7.0, np.float32) if cfg.nb_aux else None)
        return v, p

    nets = [HipAPI(0, 8, d, "float16") for _ in range(4)]
    streams = [lib.mi_net_stream(n._h) for n in nets]
    assert len(set(streams)) == 4
    ref = run(nets[0])
    for n in nets[1:]:
        out = run(n)
        assert all(np.array_equal(a, b) for a, b in zip(ref, out))
    # nets[0] predicted first of the four: its stream is the one unused the longest now
    fifth = HipAPI(0, 8, d, "float16")
    assert lib.mi_net_stream(fifth._h) == streams[0]
    assert all(np.array_equal(a, b) for a, b in zip(ref, run(fifth)))
    nets[1].close()
    sixth = HipAPI(0, 8, d, "float16x3")          # a float16x3 net made for 8 boards (several launches, a captured graph)
    assert lib.mi_net_stream(sixth._h) in streams
    o_value, o_logits, _ = ro.forward(cfg, sd, torch.from_numpy(x))
    v6 = run(sixth)[0]
    assert np.abs(v6 - o_value.numpy().reshape(-1)).max() < 1e-4
    for n in (nets[0], nets[2], nets[3], fifth, sixth):
        n.close()
    monkeypatch.setenv("CRA_OWN_STREAM_PER_NET", "1")
    own = HipAPI(0, 8, d, "float16")
    assert lib.mi_net_stream(own._h) not in streams
    assert all(np.array_equal(a, b) for a, b in zip(ref, run(own)))
    own.close()
