"""integration/hipapi.h -- the reference-side binding of the product library -- compiled against the REFERENCE'S OWN
nn/neuralnetapi.{h,cpp}, nn/neuralnetapiuser.cpp and agents/mctsagent.cpp (oracle/_ref/libcrazyara_ref_hip.so, built by
oracle/ref/build_ref.py in the container where /root/reference exists; the GPU box loads the prebuilt file) and run on the GPU:

  * HipAPI::HipAPI -> NeuralNetAPI::initialize() -> the four private virtuals; every base-class getter; validate_neural_network
  * NeuralNetAPIUser::run_inference (the reference's `inference` command, crazyara.cpp:156-181) through HipAPI::predict
  * a whole MCTSAgent search (one SearchThread) with HipAPI nets == the product's own search pool on the same net, tree for tree
"""
import os

import numpy as np
import pytest
import torch

import nn_cases
from oracle import ref_mcts
from oracle import rise_oracle as ro

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_mcts.hip_available(), reason="oracle/_ref/libcrazyara_ref_hip.so not built")]


def _export(tmp_path, name, version, dirname="model"):
    """The model directory's LAST character is the game phase of the net when it is a digit (read_game_phase_from_string,
    neuralnetapi.cpp:229-239), and NeuralNetAPIUser asserts phase < number of nets: single-net directories must not end in a digit."""
    from crazyara_amd import netfile
    cfg, sd, x = nn_cases.make_case(name)
    d = os.path.join(str(tmp_path), dirname)
    os.makedirs(d, exist_ok=True)
    netfile.export_rise(os.path.join(d, f"{cfg.name}-v{version}.cranet"), cfg, sd, input_version=version)
    return cfg, sd, x, d


@pytest.mark.parametrize("name,mode,version,precision", [
    ("risev2-7", 0, "1.0", "float16"), ("risev2-7", 0, "1.0", "float32"),
    ("risev33-wdlp", 1, "3.0", "float16"), ("risev2-13-lichess", 2, "3.0", "float16"), ("risev2-3-flat", 0, "1.0", "float16"),
])
def test_hipapi_behind_the_reference_base_class(tmp_path, hip_lib, name, mode, version, precision):
    from crazyara_amd.neuralnetapi import HipAPI, make_version
    cfg, sd, x, d = _export(tmp_path, name, version)
    B = x.shape[0]
    net = ref_mcts.RefHipAPI(d, 0, B, precision, mode)
    info = net.info()
    major, minor = (int(v) for v in version.split("."))
    assert info["version"] == make_version(major, minor)                       # read_version_from_string on HipAPI's modelName
    assert net.model_name().endswith(f"-v{version}.cranet")
    assert info["batch_size"] == B and info["nb_input_values_total"] == cfg.nb_input_channels * 64
    assert info["nb_policy_values"] == cfg.nb_policy and info["nb_auxiliary_outputs"] == cfg.nb_aux
    assert info["has_auxiliary_outputs"] == (1 if cfg.nb_aux else 0) and info["game_phase"] == 0
    assert info["is_policy_map"] == (1 if cfg.select_policy_from_plane else 0)   # policyLen != NB_LABELS (tensorrtapi.cpp:157)
    assert net.validate() == 0
    value, probs, aux, _ = net.run_inference(x.numpy(), iterations=3)
    o_value, o_logits, o_aux = ro.forward(cfg, sd, x)
    tol_v, tol_p = (1e-4, 1e-6) if precision == "float32" else (1e-3, 1e-5)
    assert np.abs(value - o_value.numpy().reshape(-1)).max() < tol_v
    assert np.abs(probs - torch.softmax(o_logits, 1).numpy()).max() < tol_p
    if cfg.nb_aux:
        assert np.abs(aux[:B * 4].reshape(B, 4) - o_aux.numpy()).max() < 1e-3
    # the Python twin of the shim (crazyara_amd/neuralnetapi.py) gives the same bits
    twin = HipAPI(0, B, d, precision)
    v2, p2 = np.zeros(B, np.float32), np.zeros(B * cfg.nb_policy, np.float32)
    twin.predict(np.ascontiguousarray(x.numpy()), v2, p2, np.zeros(B * 4, np.float32) if cfg.nb_aux else None)
    twin.close()
    assert np.array_equal(v2, value) and np.array_equal(p2.reshape(B, -1), probs)
    net.close()


def test_game_phase_comes_from_the_directory_name(tmp_path, hip_lib):
    """read_game_phase_from_string (neuralnetapi.cpp:229-239) runs in the base class on HipAPI's modelDir; the library's own
    mi_net_design reports the same phase."""
    cfg, sd, x, d = _export(tmp_path, "risev2-3", "1.0", dirname="phase2")
    net = ref_mcts.RefHipAPI(d, 0, 4, "float16", 0)
    assert net.info()["game_phase"] == 2
    net.close()
    from crazyara_amd.neuralnetapi import HipAPI
    twin = HipAPI(0, 4, d, "float16")
    assert twin.get_game_phase() == 2
    twin.close()


def test_hipapi_constructor_errors_throw_like_the_other_back_ends(tmp_path, hip_lib):
    with pytest.raises(RuntimeError, match="HipAPI"):
        ref_mcts.RefHipAPI(str(tmp_path / "missing"), 0, 8, "float16", 0)       # neuralnetapi.cpp:65-70: invalid directory throws
    cfg, sd, x, d = _export(tmp_path, "risev2-3", "1.0")
    with pytest.raises(RuntimeError, match="HipAPI"):
        ref_mcts.RefHipAPI(d, 0, 4, "int4", 0)


@pytest.mark.parametrize("variant,fen,sims,quota", [
    ("crazyhouse", "", 400, 8),
    ("crazyhouse", "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] b KQkq - 0 8", 300, 16),
])
def test_reference_mctsagent_on_hipapi_equals_the_product_search_pool(tmp_path, hip_lib, variant, fen, sims, quota):
    """The drop-in, end to end: the reference's MCTSAgent + SearchThread (float planes built on the host, blocking predict through
    integration/hipapi.h) and the product's pool (192-byte descriptors, planes built on the GPU, priors gathered on the GPU) search
    the same position on the same network and must grow the same tree, bit for bit."""
    from crazyara_amd import search
    from crazyara_amd.neuralnetapi import HipAPI
    from test_mcts_reference_build import _normalise_ref_dump
    cfg, sd, _, d = _export(tmp_path, "risev2-7", "1.0")
    st = search.default_settings(mode=0, version_major=1, is_policy_map=1, batch_size=quota)
    net = HipAPI(0, quota, d, "float16")
    pool = search.SearchPool(st, net_a=net)
    t = pool.add_position(fen, False, variant)
    stats = pool.run(simulations=sims, threads=1)
    ra = ref_mcts.RefAgent(st, hip_model_dir=d, precision="float16")
    ra.set_position(fen, False, variant)
    ra.go(simulations=sims)
    moves, visits, q, pri = pool.root_children(t)
    m2, v2, q2, p2 = ra.root_children()
    assert moves == m2 and visits == v2
    assert np.array_equal(q, q2) and np.array_equal(pri, p2)
    assert np.array_equal(pool.tree_dump(t), _normalise_ref_dump(ra.tree_dump()))
    ev = ra.eval_info()
    assert pool.best_move(t) == ev["best_move"] and ev["nodes"] == pool.tree_info(t)["node_count"] == stats.nodes
    pool.close()
    net.close()
    ra.close()


@pytest.mark.parametrize("variant,fen,sims,quota", [
    ("crazyhouse", "", 400, 8), ("crazyhouse", "", 700, 64),
    ("crazyhouse", "r1b1k2r/ppp2ppp/2n5/3qp3/1b1P4/2N1PN2/PP3PPP/R1BQKB1R[Pn] w KQkq - 0 8", 500, 16),
    ("crazyhouse", "5r2/ppp2pkp/3p4/2bP4/2Pnp1N1/3P2pP/PP2n1P1/R2Q1R1K[PBRQnbb] w - - 0 28", 400, 8),
])
def test_patched_reference_searchthread_grows_the_same_tree(tmp_path, hip_lib, variant, fen, sims, quota):
    """integration/searchthread_hip.patch: the reference's SearchThread under HIP_BACKEND hands every new leaf over as
    its 192-byte descriptor (planes built on the GPU) and takes back only the priors of its legal moves (mi_net_submit_boards_gathered,
    scattered into the slots Node::set_probabilities_for_moves reads) instead of building float planes on the host and copying whole
    probability vectors.  oracle/ref/build_ref.py applies the patch to a build-time copy of searchthread.cpp; the patched and the unpatched
    reference, both on HipAPI nets of the same model, must grow the same tree bit for bit (and that tree is the product pool's)."""
    if not ref_mcts.hip_patched_available():
        pytest.skip("oracle/_ref/libcrazyara_ref_hip_patched.so not built")
    from crazyara_amd import search
    from test_mcts_reference_build import _normalise_ref_dump
    cfg, sd, _, d = _export(tmp_path, "risev2-7", "1.0")
    st = search.default_settings(mode=0, version_major=1, is_policy_map=1, batch_size=quota)
    dumps = []
    for patched in (False, True):
        ra = ref_mcts.RefAgent(st, hip_model_dir=d, precision="float16", patched=patched)
        ra.set_position(fen, False, variant)
        ra.go(simulations=sims)
        first = ra.tree_dump().copy()
        ra.go(simulations=2 * sims)                                   # a second go on the kept tree
        dumps.append((first, ra.tree_dump().copy(), ra.root_children(), ra.eval_info()["best_move"]))
        ra.close()
    assert np.array_equal(dumps[0][0], dumps[1][0]) and np.array_equal(dumps[0][1], dumps[1][1])
    assert dumps[0][2][0] == dumps[1][2][0] and dumps[0][2][1] == dumps[1][2][1] and dumps[0][3] == dumps[1][3]
    assert np.array_equal(dumps[0][2][2], dumps[1][2][2]) and np.array_equal(dumps[0][2][3], dumps[1][2][3])
