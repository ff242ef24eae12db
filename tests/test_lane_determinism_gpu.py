"""Two evaluator lanes must give reproducible searches (profiles/NOTES.md round 4).

A tree of the many-trees pool lives in ONE lane, so its result is a pure function of the network outputs it consumed: equal searches
that end in different trees got different numbers from the GPU.  The scenario that exposed the round-3 fault: 8 crazyhouse trees, two
lanes (two float16x3 nets on two streams, batch 64, so the two forwards overlap freely on the chip), 4 host threads, 240 simulations
-- repeated 200 times for both forms of the float16x3 value head.  Every batch of every run is also recorded and replayed alone on the
device (mi_search_debug_replay): a differing word names the batch, the slot and the output that was not reproducible.

Round 5: the fault was v_pk_fma_f32 in the value head's FC1 beside the MFMA waves of the other lane's policy conv (profiles/NOTES.md);
the kernel runs on v_fmac_f32 since and WITHOUT round 4's LDS fence -- this test is what holds that the unfenced kernel reproduces.
"""
import numpy as np
import pytest

import nn_cases
from crazyara_amd import openings, search
from crazyara_amd.neuralnetapi import HipAPI

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision,value_head", [("float16x3", "one"), ("float16x3", "three"), ("float16p8", "one")])
def test_two_lane_float16x3_searches_reproduce_200_times(tmp_path, hip_lib, monkeypatch, precision, value_head):
    monkeypatch.setenv("CRA_X3_VALUE_HEAD", value_head)          # read when a net is built
    monkeypatch.setenv("CRA_LANE_RECORD", "1")                   # read when a pool's lanes are made
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    fens = openings.position_fens("crazyhouse")[20:28]
    nets = [HipAPI(0, 64, d, precision) for _ in range(2)]
    first, differing_runs, replay_words, reports = None, [], 0, []
    for run in range(200):
        st = search.default_settings(mode=0, version_major=1, batch_size=16, seed=3)
        pool = search.SearchPool(st, net_a=nets[0], net_b=nets[1])
        for f in fens:
            pool.add_position(f, False, "crazyhouse")
        pool.run(simulations=240, threads=4)
        dumps = [pool.tree_dump(i).tobytes() for i in range(len(fens))]
        bad, text = pool.debug_replay()
        replay_words += bad
        if bad and len(reports) < 3:
            reports.append(text[-800:])
        pool.close()
        if first is None:
            first = dumps
        elif dumps != first:
            differing_runs.append(run)
    for n in nets:
        n.close()
    assert replay_words == 0, reports
    assert not differing_runs, differing_runs[:10]


def test_concurrent_predicts_of_two_nets_reproduce(tmp_path, hip_lib, monkeypatch):
    """The same at the boundary the reference uses (two SearchThreads, each blocking in predict() on its own net, crazyara.cpp:548-563):
    two host threads, zero-copy predict on pinned buffers, fixed inputs -- every output of 3000 predicts per net equals the net alone."""
    import threading
    from crazyara_amd.neuralnetapi import NeuralNetAPIUser
    monkeypatch.setenv("CRA_X3_VALUE_HEAD", "one")
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    nets = [HipAPI(0, 64, d, "float16x3") for _ in range(2)]
    users = [NeuralNetAPIUser([n]) for n in nets]
    rng = np.random.default_rng(5)
    ref = []
    for n, u in zip(nets, users):
        u.input_planes[:] = (rng.random(u.input_planes.shape) < 0.1).astype(np.float32)
        n.predict(u.input_planes, u.value_outputs, u.prob_outputs)
        ref.append((u.value_outputs.copy(), u.prob_outputs.copy()))
    bad = [0, 0]

    def loop(i):
        n, u = nets[i], users[i]
        for _ in range(3000):
            u.value_outputs[:] = np.nan
            n.predict(u.input_planes, u.value_outputs, u.prob_outputs)
            if not (np.array_equal(u.value_outputs.view(np.uint32), ref[i][0].view(np.uint32))
                    and np.array_equal(u.prob_outputs.view(np.uint32), ref[i][1].view(np.uint32))):
                bad[i] += 1
    th = [threading.Thread(target=loop, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for u in users:
        u.close()
    for n in nets:
        n.close()
    assert bad == [0, 0]
