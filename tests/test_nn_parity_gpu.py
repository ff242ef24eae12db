"""GPU parity of the HIP NN path (through the C ABI) against the oracle and the committed reference goldens.

Tolerances (BASELINE.json north_star: "value/policy logits within 1e-3 fp32"):
  Precision float32 (exact-f32 MFMA): |logit| err < 1e-4, |value| err < 1e-4, |prob| err < 1e-6 (measured on an MI355X:
      5.0e-6 / 4.9e-7 / 1.2e-8, profiles/r02/a_error_scan.txt) -- THE mode that satisfies north_star's 1e-3 on the logits.
  Precision float16x3 (split-operand f16 MFMAs, x3.hip): the same bounds as float32 -- the mode's own emulation
      (oracle/rise_oracle.forward_x3) differs from fp32 by 1e-6 ... 6e-6 on the logits of these cases, i.e. by f32 round-off.  THE FAST
      mode that satisfies north_star's 1e-3 on the logits; the headline of bench.py.
  Precision float16 (f16 MFMA operands, f32 accumulate -- the reference TensorRT default): predict() outputs
      |value| err < 1e-3 (measured <= 6.9e-4), |prob| err < 1e-5 (measured <= 3.1e-6).  The LOGITS do not meet 1e-3 with f16
      operands: rounding the weights alone to f16 moves them by 1.6e-3 (scripts/error_budget.py, profiles/r02/a_error_budget.txt),
      the whole path measures 1.07e-3 ... 1.73e-3 x max|logit| over the cases of this file (worst: the 256-board headline test,
      3.31e-3 at max|logit| 1.91).  The bound here is 1.25 x that: 2.2e-3 x max|logit| of the case, never above 4.8e-3.
"""
import os

import numpy as np
import pytest
import torch

import nn_cases
from oracle import rise_oracle as ro

pytestmark = pytest.mark.gpu

TOL = {"float32": dict(logit=1e-4, logit_rel=None, value=1e-4, prob=1e-6, aux=1e-4),
       "float16": dict(logit=4.8e-3, logit_rel=2.2e-3, value=1e-3, prob=1e-5, aux=1e-3)}


def logit_tol(tol, ref_logits):
    """absolute bound on the logit error of one case: the mode's cap, scaled down by the size of the case's logits"""
    if tol["logit_rel"] is None:
        return tol["logit"]
    return min(tol["logit"], max(2e-4, tol["logit_rel"] * float(np.abs(np.asarray(ref_logits)).max())))


# float16 runs the residual-tower kernel (runs of 3x3 blocks in one launch); "-perblock" = one fused launch per bottleneck
# block, "-unfused" = layer-granular kernels (conv GEMM / depthwise / project as separate launches): three implementations
TOL["float16x3"] = TOL["float32"]
TOL["float16x3-perblock"] = TOL["float32"]  # one launch per 3x3 block (block_x3_kernel); plain float16x3 runs them in one launch (tower_x3_kernel)
TOL["float16x3-unfused"] = TOL["float32"]   # every block on the layer kernels (conv GEMM x3 / float depthwise), as the 5x5 blocks always are
# float16x3 whose one-launch tower takes the cross terms of both 1x1 GEMMs through e5m2 MFMAs (truncated activation bytes): emulated
# 5e-5 ... 1.3e-4 on the logits, 3e-5 on the value (scripts/studies/p8_format_study.py); bounds at twice that, a third of north_star's 1e-3
TOL["float16p8"] = dict(logit=3e-4, logit_rel=None, value=1e-4, prob=1e-6, aux=1e-4)
# "-1wg": nets made for at most 64 boards run their 3x3 blocks one per launch with several workgroups per board (block_x3_split_kernel,
# round 6); the suffix keeps the one-workgroup-per-board tower kernels, which the four-board fixtures would otherwise never reach
TOL["float16x3-1wg"] = TOL["float32"]
TOL["float16p8-1wg"] = TOL["float16p8"]
TOL["float32-unfused"] = TOL["float32"]
TOL["float16-unfused"] = TOL["float16"]
TOL["float16-perblock"] = TOL["float16"]
TOL["float16-2b"] = TOL["float16"]      # dense residual tower with two boards per workgroup (other nets: same as float16)
TOL["float16-3k"] = TOL["float16"]      # stem / tower / head as three launches; plain float16 runs them as one (forward.hip)


def _run(tmp_path, hip_lib, name, precision):
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, x = nn_cases.make_case(name)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    B = x.shape[0]
    net = HipAPI(0, B, d, precision, keep_logits=True)
    assert net.get_batch_size() == B and net.get_nb_policy_values() == cfg.nb_policy
    assert net.get_nb_input_values_total() == cfg.nb_input_channels * 64
    assert net.get_nb_auxiliary_outputs() == cfg.nb_aux
    assert abs(net.flops_per_position() - ro.flops_per_position(cfg)) < 1.0
    value = np.full(B, 7.0, np.float32)
    probs = np.full(B * cfg.nb_policy, 7.0, np.float32)
    aux = np.full(B * 4, 7.0, np.float32) if cfg.nb_aux else None
    net.predict(np.ascontiguousarray(x.numpy()), value, probs, aux)
    logits = torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy()
    net.close()
    return cfg, sd, x, value, probs.reshape(B, -1), aux, logits


@pytest.mark.parametrize("precision", ["float32", "float16", "float16x3", "float16p8", "float16x3-1wg", "float16p8-1wg", "float16-3k", "float16-perblock",
                                       "float32-unfused", "float16-unfused", "float16x3-perblock", "float16x3-unfused"])
@pytest.mark.parametrize("name", list(nn_cases.CASES))
def test_predict_matches_oracle_and_golden(tmp_path, hip_lib, name, precision):
    cfg, sd, x, value, probs, aux, logits = _run(tmp_path, hip_lib, name, precision)
    tol = TOL[precision]
    o_value, o_logits, o_aux = ro.forward(cfg, sd, x)
    o_probs = torch.softmax(o_logits, dim=1).numpy()
    g = np.load(os.path.join(nn_cases.GOLDEN_DIR, f"nn_{name}.npz"))
    for ref_v, ref_l in ((o_value.numpy().reshape(-1), o_logits.numpy()), (g["value"].reshape(-1), g["logits"])):
        assert np.abs(value - ref_v).max() < tol["value"]
        assert np.abs(logits - ref_l).max() < logit_tol(tol, ref_l)
    assert np.abs(probs - o_probs).max() < tol["prob"]
    assert np.allclose(probs.sum(axis=1), 1.0, atol=1e-4)
    if cfg.nb_aux:
        assert np.abs(aux.reshape(-1, 4) - o_aux.numpy()).max() < tol["aux"]


@pytest.mark.parametrize("variant", ["float16-2b", "float16-1b-8w", "float16-2b-8w"])
@pytest.mark.parametrize("name", ["rise-classical-4", "alphazero-5", "alphazero-3-cv8"])
def test_dense_tower_kernel_shapes(tmp_path, hip_lib, name, variant):
    """restower_kernel<NB, NR>: the same fixtures through two boards per workgroup (odd batch 3 included: the last workgroup's
    second board is empty) and through the 8-thin-waves shape; plain float16 = one board, 4 fat waves."""
    cfg, sd, x, value, probs, aux, logits = _run(tmp_path, hip_lib, name, variant)
    tol = TOL["float16"]
    g = np.load(os.path.join(nn_cases.GOLDEN_DIR, f"nn_{name}.npz"))
    assert np.abs(value - g["value"].reshape(-1)).max() < tol["value"]
    assert np.abs(logits - g["logits"]).max() < logit_tol(tol, g["logits"])


@pytest.mark.parametrize("precision", ["float16", "float16x3", "float16p8", "float16x3-1wg", "float16p8-1wg"])
@pytest.mark.parametrize("case", ["risev2-7", "alphazero-3-cv8"])
@pytest.mark.parametrize("batch", [1, 3, 300])
def test_tower_and_head_kernels_any_batch_size(tmp_path, hip_lib, batch, case, precision):
    """One workgroup per board: batch sizes below / not a multiple of / above the 256 CUs must all be exact per row
    (bottleneck tower and dense residual tower)."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, _ = nn_cases.make_case(case)
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 4242)
    d = nn_cases.export_case(tmp_path, case, cfg, sd)
    net = HipAPI(0, batch, d, precision)
    v, p = np.zeros(batch, np.float32), np.zeros(batch * cfg.nb_policy, np.float32)
    net.predict(np.ascontiguousarray(x.numpy()), v, p)
    net.close()
    o_value, o_logits, _ = ro.forward(cfg, sd, x)
    # dense 3x3 towers contract K = 2304 f16 products per output (9x the 1x1 tower): the oracle's own f16 emulation
    # (sim_dtype=float16) puts the worst of these 300 boards at 0.98e-3 for the value (risev2-7: 0.35e-3) -> 2e-3 here
    value_tol = 2e-3 if cfg.dense_blocks and precision == "float16" else TOL[precision]["value"]
    assert np.abs(v - o_value.numpy().reshape(-1)).max() < value_tol
    assert np.abs(p.reshape(batch, -1) - torch.softmax(o_logits, 1).numpy()).max() < TOL[precision]["prob"]


def test_partial_batch_and_stale_slots(tmp_path, hip_lib):
    """predict always runs the full fixed batch; rows are independent, stale trailing slots must not matter
    (engine/src/searchthread.cpp:407-411)."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, x = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    net = HipAPI(0, 8, d, "float32")
    xin = np.zeros((8, 34, 8, 8), np.float32)
    xin[:4] = x.numpy()
    xin[4:] = 123.0  # garbage in the unused slots
    v, p = np.zeros(8, np.float32), np.zeros(8 * 5184, np.float32)
    net.predict(xin, v, p)
    o_value, o_logits, _ = ro.forward(cfg, sd, x)
    assert np.abs(v[:4] - o_value.numpy().reshape(-1)).max() < 1e-4
    assert np.abs(p.reshape(8, -1)[:4] - torch.softmax(o_logits, 1).numpy()).max() < 1e-6
    net.close()


def test_submit_wait_and_device_resident_paths_agree(tmp_path, hip_lib):
    from crazyara_amd.neuralnetapi import HipAPI, NeuralNetAPIUser
    cfg, sd, x = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    net = HipAPI(0, 4, d, "float16")
    user = NeuralNetAPIUser([net])
    user.input_planes[:] = x.numpy().reshape(-1)
    user.run_inference(2)
    v1, p1 = user.value_outputs.copy(), user.prob_outputs.copy()
    user.value_outputs[:] = 0
    user.prob_outputs[:] = 0
    net.submit(user._p_in, user._p_val, user._p_prob)
    net.wait()
    assert np.array_equal(v1, user.value_outputs) and np.array_equal(p1, user.prob_outputs)
    bufs = net.device_buffers()
    torch.as_tensor(bufs["planes"], device="cuda").copy_(x.cuda())
    torch.cuda.synchronize()
    net.forward_device()
    net.sync()
    assert np.array_equal(torch.as_tensor(bufs["probs"], device="cuda").cpu().numpy().reshape(-1), p1)
    assert np.array_equal(torch.as_tensor(bufs["value"], device="cuda").cpu().numpy(), v1)
    user.close()
    net.close()


@pytest.mark.parametrize("name,precision,batch", [
    ("rise-classical-4", "float16", 70), ("rise-classical-4", "float16-perblock", 9), ("alphazero-5", "float16", 33), ("alphazero-3-cv8", "float32", 9),
    ("rise-classical-3-se", "float16", 40), ("alphazero-3-se", "float16", 40), ("risev2-3-flat", "float16", 64), ("risev33-wdlp", "float16", 64),
    ("risev2-7", "float32", 20), ("risev2-7", "float16-perblock", 20), ("risev2-7", "float16-unfused", 20), ("risev33", "float32-unfused", 8),
    ("risev2-13-lichess", "float16", 64), ("risev2-7", "float16x3", 20), ("risev33-wdlp", "float16x3", 9), ("alphazero-3-se", "float16x3", 9),
    ("risev2-3-flat", "float16x3", 9), ("risev2-7", "float16p8", 20), ("risev33-wdlp", "float16p8", 9), ("risev2-13-lichess", "float16p8", 5)])
def test_every_kernel_family_is_bit_identical_whatever_the_cus_held_before(tmp_path, hip_lib, lds_poison, name, precision, batch):
    """tests/test_fp8.py's call-to-call check (LDS of every CU poisoned with a different pattern before each forward) over the other
    kernel families: dense towers in one launch, per-block and layer-granular kernels, float32, flat and WDLP heads, lichess tables."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, _ = nn_cases.make_case(name)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 78).numpy().reshape(-1)
    net = HipAPI(0, batch, d, precision)
    outs = []
    for pattern in (0x00000000, 0xffffffff, 0x7f7f7f7f, 0x7bff7bff, 0x7f800000, 0x00000000):
        assert lds_poison.poison_lds(pattern, pattern, 0, 0) == 0
        v = np.zeros(batch, np.float32)
        p = np.zeros(batch * cfg.nb_policy, np.float32)
        net.predict(x, v, p)
        outs.append((v, p))
    net.close()
    assert np.isfinite(outs[0][0]).all() and np.isfinite(outs[0][1]).all()
    for v, p in outs[1:]:
        assert np.array_equal(v, outs[0][0]) and np.array_equal(p, outs[0][1])


@pytest.mark.parametrize("name,precision", [("risev2-7", "float16"), ("risev33-wdlp", "float16"), ("risev2-3-flat", "float16"),
                                            ("alphazero-5", "float16"), ("risev2-7", "float32"), ("risev2-7", "float16x3"), ("risev2-7", "float16p8")])
def test_zero_copy_predict_equals_copied_predict(tmp_path, hip_lib, name, precision):
    """predict() with the caller's buffers in pinned memory (NeuralNetAPIUser, neuralnetapiuser.cpp:50-60) issues no copy commands: the
    kernels read the planes and write value / probabilities / aux in place.  Same bits as the copy path (pageable numpy buffers),
    every net family (one-launch forward, separate launches, flat policy head, WDLP aux, float32 layer kernels)."""
    from crazyara_amd.neuralnetapi import HipAPI, NeuralNetAPIUser
    cfg, sd, x = nn_cases.make_case(name)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    B = x.shape[0]
    net = HipAPI(0, B, d, precision)
    v1, p1 = np.full(B, 7.0, np.float32), np.full(B * cfg.nb_policy, 7.0, np.float32)
    a1 = np.full(B * 4, 7.0, np.float32) if cfg.nb_aux else None
    net.predict(np.ascontiguousarray(x.numpy()), v1, p1, a1)                 # pageable: H2D copy, forward, D2H copies
    assert not net.last_submit_zero_copy()
    user = NeuralNetAPIUser([net])
    user.input_planes[:] = x.numpy().reshape(-1)
    user.value_outputs[:] = 7.0
    user.prob_outputs[:] = 7.0
    user.run_inference(2)                                                    # pinned: in place
    assert net.last_submit_zero_copy()
    assert np.array_equal(user.value_outputs, v1) and np.array_equal(user.prob_outputs, p1)
    if cfg.nb_aux:
        assert np.array_equal(user.auxiliary_outputs, a1)
    # the device-side tensors were not the target of that forward: a device-resident replay still works and agrees
    bufs = net.device_buffers()
    torch.as_tensor(bufs["planes"], device="cuda").copy_(x.cuda())
    torch.cuda.synchronize()
    net.forward_device()
    net.sync()
    assert np.array_equal(torch.as_tensor(bufs["probs"], device="cuda").cpu().numpy().reshape(-1), p1)
    user.close()
    net.close()


def test_constructor_errors(tmp_path, hip_lib):
    from crazyara_amd.neuralnetapi import HipAPI
    with pytest.raises(ValueError):
        HipAPI(0, 4, str(tmp_path), "float16")          # no .cranet in directory (neuralnetapi.cpp:65-67)
    cfg, sd, x = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    with pytest.raises(ValueError):
        HipAPI(0, 4, d, "int4")
    net = HipAPI(0, 4, d, "float16")
    assert net.get_version() == 1000000 and net.get_model_name().endswith("-v1.0.cranet")
    net.close()


# ---------------------------------------------------------------------------------------------------------------------
# GPU input-plane builder (csrc/chess/planes_kernel.hip) -- bit-exact against the oracle planes
# ---------------------------------------------------------------------------------------------------------------------
def _random_positions(variant, is960, fen, n, seed):
    import random
    from crazyara_amd import env
    from oracle import chess_oracle as co
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        p = env.Position(fen, is960, variant)
        b = co.Board(fen or None, is960, variant)
        for ply in range(rng.randint(0, 60)):
            moves = b.legal_moves()
            if not moves or b.terminal() != 4:
                break
            mv = rng.choice(moves)
            p.push_uci(b.move_uci(mv))
            b.push(mv)
        out.append((p, b))
    return out


@pytest.mark.parametrize("variant,is960,fen,mode,version", [
    ("crazyhouse", False, "", 0, 1), ("crazyhouse", False, "", 0, 2), ("crazyhouse", False, "", 0, 3),
    ("chess", False, "", 1, 1), ("chess", False, "", 1, 3), ("chess", False, "", 1, "2.7"), ("chess", False, "", 1, "2.8"),
    ("chess", True, "bnnrkbrq/pppppppp/8/8/8/8/PPPPPPPP/BNNRKBRQ w GDgd - 0 1", 1, 3),
    ("3check", False, "", 2, 2), ("kingofthehill", False, "", 2, 3), ("crazyhouse", False, "", 2, 3),
])
def test_gpu_plane_builder_bit_exact(hip_lib, variant, is960, fen, mode, version):
    from crazyara_amd import env, _capi
    from oracle import chess_oracle as co
    lib = _capi.load()
    pos = _random_positions(variant, is960, fen, 24, seed=hash((variant, mode, version)) & 0xFFF)
    layout = env.planes_layout(mode, version)
    C_ = lib.mi_planes_channels(layout)
    descs = b"".join(p.desc(layout) for p, _ in pos)
    for normalize in (True, False):
        out = torch.empty((len(pos), C_, 8, 8), dtype=torch.float32, device="cuda")
        env.planes_from_descs_device(descs, len(pos), layout, normalize, out.data_ptr(), 0)
        got = out.cpu().numpy()
        for i, (p, b) in enumerate(pos):
            exp = co.board_to_planes(b, mode, version, normalize)
            assert np.array_equal(got[i], exp), (b.fen(), normalize)


def test_predict_from_board_descriptors_equals_predict_from_planes(tmp_path, hip_lib):
    """mi_net_submit_boards (192 B/position over PCIe, planes built on the GPU) == predict() on host-built planes."""
    import ctypes as C
    from crazyara_amd import _capi
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, x = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    pos = _random_positions("crazyhouse", False, "", 8, seed=3)
    net = HipAPI(0, 8, d, "float16")
    planes = np.stack([p.planes(0, 1, True) for p, _ in pos]).astype(np.float32)
    v1, p1 = np.zeros(8, np.float32), np.zeros(8 * 5184, np.float32)
    net.predict(planes, v1, p1)
    descs = b"".join(p.desc() for p, _ in pos)
    v2, p2 = np.zeros(8, np.float32), np.zeros(8 * 5184, np.float32)
    lib = _capi.load()
    assert lib.mi_net_submit_boards(net._h, descs, 8, lib.mi_planes_layout(0, 1), v2.ctypes.data, p2.ctypes.data, None) == 0
    net.wait()
    assert np.array_equal(v1, v2) and np.array_equal(p1, p2)
    # wrong layout for this net is rejected loudly
    assert lib.mi_net_submit_boards(net._h, descs, 8, lib.mi_planes_layout(0, 3), v2.ctypes.data, p2.ctypes.data, None) != 0
    net.close()


@pytest.mark.parametrize("flavour", [dict(fold_bn=True, linear="gemm"), dict(fold_bn=False, linear="matmul")])
@pytest.mark.parametrize("name", ["risev2-3", "risev33-wdlp", "rise-classical-4", "alphazero-3-cv8", "risev2-3-flat", "rise-classical-3-se",
                                  "alphazero-3-se"])
def test_onnx_model_directory_loads_and_matches_golden(tmp_path, hip_lib, name, flavour):
    """a model directory holding only the reference's file format ("<prefix>-v<ver>.onnx") goes through
    mi_net_create (ONNX parsed in place, csrc/nn/onnx_import.cpp) and reproduces the reference model's outputs (committed goldens)."""
    import onnx_writer
    from crazyara_amd.neuralnetapi import HipAPI, make_version
    cfg, sd, x = nn_cases.make_case(name)
    d = os.path.join(str(tmp_path), "model")
    os.makedirs(d)
    fname = f"{cfg.name}-v3.0-bsize-{x.shape[0]}.onnx" if flavour["fold_bn"] else f"{cfg.name}-v3.0.onnx"
    with open(os.path.join(d, fname), "wb") as f:
        f.write(onnx_writer.rise_to_onnx(cfg, sd, batch=x.shape[0] if flavour["fold_bn"] else None, **flavour))
    B = x.shape[0]
    tol = TOL["float16"]
    g = np.load(os.path.join(nn_cases.GOLDEN_DIR, f"nn_{name}.npz"))
    for precision in ("float32", "float16x3", "float16"):         # float16 last: it leaves the converted .cranet in the directory
        net = HipAPI(0, B, d, precision, keep_logits=True)
        assert net.get_model_name() == fname and net.get_version() == make_version(3, 0)
        assert net.get_nb_policy_values() == cfg.nb_policy and net.get_nb_auxiliary_outputs() == cfg.nb_aux
        assert abs(net.flops_per_position() - ro.flops_per_position(cfg)) < 1.0
        value, probs = np.zeros(B, np.float32), np.zeros(B * cfg.nb_policy, np.float32)
        aux = np.zeros(B * 4, np.float32) if cfg.nb_aux else None
        net.predict(np.ascontiguousarray(x.numpy()), value, probs, aux)
        logits = torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy()
        net.close()
        tol = TOL[precision]
        assert np.abs(value - g["value"].reshape(-1)).max() < tol["value"]
        assert np.abs(logits - g["logits"]).max() < logit_tol(tol, g["logits"])
        if cfg.nb_aux:
            assert np.abs(aux.reshape(-1, 4) - g["aux"]).max() < tol["aux"]
        if precision == "float16":
            # the offline conversion (mi_onnx_to_cranet) is the same import kept as a file: a directory with the .cranet beside the
            # .onnx picks the container and gives the same numbers bit for bit
            from crazyara_amd import netfile
            converted = netfile.onnx_to_cranet(os.path.join(d, fname))
            net = HipAPI(0, B, d, precision)
            assert net.get_model_name() == os.path.basename(converted)
            v2, p2 = np.zeros(B, np.float32), np.zeros(B * cfg.nb_policy, np.float32)
            net.predict(np.ascontiguousarray(x.numpy()), v2, p2, np.zeros(B * 4, np.float32) if cfg.nb_aux else None)
            net.close()
            assert np.array_equal(v2, value) and np.array_equal(p2, probs)


def _full_size_case(tmp_path, case, B, precision, version, golden_rows=True):
    """One BASELINE configuration at the batch size bench.py runs it with.  Rows 0-3 are the committed reference golden's inputs (they
    must come out as in the 4-board fixture), every row matches the oracle, and the size-independent properties hold: probabilities
    sum to one, values inside the tanh range, a permuted batch gives the permuted outputs bit for bit (one workgroup per board, no
    cross-row arithmetic), and a replay of the same batch gives the same bits."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, xg = nn_cases.make_case(case)
    x = nn_cases.synthetic_planes(B, cfg.nb_input_channels, 777).numpy()
    n_g = xg.shape[0]
    x[:n_g] = xg.numpy()
    d = nn_cases.export_case(tmp_path, case, cfg, sd, version=version)
    net = HipAPI(0, B, d, precision, keep_logits=True)
    v, p = np.zeros(B, np.float32), np.zeros(B * cfg.nb_policy, np.float32)
    aux = np.zeros(B * 4, np.float32) if cfg.nb_aux else None
    net.predict(np.ascontiguousarray(x), v, p, aux)
    logits = torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy().copy()
    p = p.reshape(B, -1)
    tol = TOL[precision]
    if golden_rows:
        g = np.load(os.path.join(nn_cases.GOLDEN_DIR, f"nn_{case}.npz"))
        assert np.abs(v[:n_g] - g["value"].reshape(-1)).max() < tol["value"]
        assert np.abs(logits[:n_g] - g["logits"]).max() < logit_tol(tol, g["logits"])
    o_value, o_logits, _ = ro.forward(cfg, sd, torch.from_numpy(x))
    err_rows = np.abs(logits - o_logits.numpy()).max(axis=1)
    if precision == "float16p8":
        # The 3e-4 of the 4-board fixtures is not the mode's bound over millions of logits: the worst of config 5's 1024 x 5376 measured
        # 5.3e-4 (round 5, still inside north_star's 1e-3).  That this tail belongs to the MODE (oracle.forward_p8, the definition) and
        # not to the kernel is checked on the worst rows: the definition itself is that far from fp32 there, and the kernel sits within
        # the definition's own sensitivity to a 1e-7 perturbation of its weights (the byte images are discontinuous: see
        # test_float16p8_equals_its_emulation) of it.
        worst = np.argsort(err_rows)[-6:]
        xw = torch.from_numpy(x[worst])
        _, e_logits, _ = ro.forward_p8(cfg, sd, xw)
        mode_err = float((e_logits - o_logits[worst]).abs().max())
        sens = 0.0
        for seed in (7, 8):
            g7 = torch.Generator().manual_seed(seed)
            sd2 = {k: (t * (1 + 1e-7 * torch.randn(t.shape, generator=g7)) if t.dtype.is_floating_point and t.dim() > 0 else t) for k, t in sd.items()}
            sens = max(sens, float((ro.forward_p8(cfg, sd2, xw)[1] - e_logits).abs().max()))
        kernel_vs_definition = float(np.abs(logits[worst] - e_logits.numpy()).max())
        numbers = dict(kernel_vs_fp32=float(err_rows.max()), definition_vs_fp32_on_worst_rows=mode_err, kernel_vs_definition=kernel_vs_definition,
                       definition_sensitivity=sens)
        print("float16p8 at full size:", case, B, numbers)
        assert err_rows.max() < 7e-4, numbers
        assert mode_err > 0.4 * float(err_rows.max()), numbers           # the definition itself is that far from fp32 on these rows
        assert kernel_vs_definition < max(1.5e-4, 4.0 * sens), numbers   # ... and the kernel is as close to it as the definition is to itself
    else:
        assert err_rows.max() < logit_tol(tol, o_logits.numpy())                           # float16: measured 3.31e-3 (bound 4.2e-3)
    assert np.abs(v - o_value.numpy().reshape(-1)).max() < tol["value"]
    # (float16p8 over 5.5 million probabilities: 4.8e-6 measured on config 5, the fixtures' 1e-6 is a four-board bound)
    assert np.abs(p - torch.softmax(o_logits, 1).numpy()).max() < (1e-5 if precision == "float16p8" else tol["prob"])
    assert np.allclose(p.sum(axis=1), 1.0, atol=1e-4) and (p >= 0).all() and np.abs(v).max() <= 1.0
    perm = np.random.default_rng(5).permutation(B)
    v2, p2 = np.zeros(B, np.float32), np.zeros(B * cfg.nb_policy, np.float32)
    net.predict(np.ascontiguousarray(x[perm]), v2, p2, aux)
    assert np.array_equal(v2, v[perm]) and np.array_equal(p2.reshape(B, -1), p[perm])
    v3, p3 = np.zeros(B, np.float32), np.zeros(B * cfg.nb_policy, np.float32)
    net.predict(np.ascontiguousarray(x[perm]), v3, p3, aux)               # and replays are deterministic
    assert np.array_equal(v3, v2) and np.array_equal(p3, p2)
    net.close()


@pytest.mark.parametrize("precision", ["float16p8", "float16x3", "float16"])
def test_headline_configuration_at_full_size(tmp_path, hip_lib, precision):
    """BASELINE.json config 2 as bench.py runs it: RISEv2-19, batch 256, in Precision float16p8 (the headline: the fastest mode that
    meets north_star's 1e-3 on the logits -- held to 3e-4 here), float16x3 (1e-4) and float16 (the reference's TensorRT default; its
    logits miss 1e-3, bound as measured)."""
    _full_size_case(tmp_path, "risev2-19", 256, precision, "1.0")


@pytest.mark.parametrize("precision", ["float16p8", "float16x3"])
def test_config3_at_full_size(tmp_path, hip_lib, precision):
    """BASELINE.json config 3 as bench.py searches it: chess RISEv3.3 (3x3 and 5x5 bottleneck runs in tower launches, eca_se gates), batch 512."""
    _full_size_case(tmp_path, "risev33", 512, precision, "3.0")


@pytest.mark.parametrize("precision", ["float16p8", "float16x3"])
def test_config5_at_full_size(tmp_path, hip_lib, precision):
    """BASELINE.json config 5's one-GPU slice as bench.py searches it: RISEv2-13 on the lichess tables (80-channel planes, 5376 policy
    entries), batch 1024 -- four workgroups per compute unit's worth of boards, i.e. the tower kernel's later waves of workgroups."""
    _full_size_case(tmp_path, "risev2-13-lichess", 1024, precision, "3.0")


@pytest.mark.parametrize("precision", ["float32", "float16", "float16x3"])
@pytest.mark.parametrize("channels,family", [(128, "mobile"), (192, "mobile"), (512, "mobile"), (128, "a0"), (320, "classical")])
def test_other_trunk_widths_run_on_the_layer_kernels(tmp_path, hip_lib, channels, family, precision):
    """`channels` is a constructor argument of RiseV3 / AlphaZeroResnet: the fused tower / head kernels are specialised for 256,
    any other multiple of 64 up to 512 goes through the layer-granular kernels and must match the oracle just the same."""
    from crazyara_amd.neuralnetapi import HipAPI
    if family == "mobile":
        cfg = ro.rise_v2_config(3, 34, 81)
        cfg.se_types = [None, "ca_se", "eca_se"]
        cfg.kernels = [3, 5, 3]
        cfg.channels_operating_init, cfg.channel_expansion = 64, 32
    elif family == "a0":
        cfg = ro.alpha_zero_config(2, 34, 81, 4)
    else:
        cfg = ro.rise_classical_config(2, 34, 81)
    cfg.channels = channels
    if cfg.dense_blocks:
        cfg.channels_operating_init = channels
    cfg.name = f"{family}-{channels}"
    sd = ro.make_state_dict(cfg, seed=90 + channels)
    x = nn_cases.synthetic_planes(5, 34, 17)
    d = nn_cases.export_case(tmp_path, cfg.name, cfg, sd)
    net = HipAPI(0, 5, d, precision, keep_logits=True)
    v, p = np.zeros(5, np.float32), np.zeros(5 * cfg.nb_policy, np.float32)
    net.predict(np.ascontiguousarray(x.numpy()), v, p)
    logits = torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy()
    net.close()
    o_value, o_logits, _ = ro.forward(cfg, sd, x)
    tol = TOL[precision]
    # float16 here = f16 storage after every layer (no fused tower): the oracle's own f16 emulation (sim_dtype) moves these values by
    # up to 0.5e-3, the worst board measured 1.2e-3 -> 2e-3; the float32 precision mode holds the 1e-4 of the other tests
    assert np.abs(v - o_value.numpy().reshape(-1)).max() < (2e-3 if precision == "float16" else tol["value"])
    assert np.abs(logits - o_logits.numpy()).max() < logit_tol(tol, o_logits.numpy())


@pytest.mark.parametrize("name,batch", [("risev2-7", 9), ("risev33-wdlp", 5)])
def test_float16x3_two_role_tower_equals_the_symmetric_one_bit_for_bit(tmp_path, hip_lib, name, batch, monkeypatch):
    """tower_x3_roles_kernel (EXPAND / PROJECT waves, the default) and tower_x3_kernel (every wave all three phases) add up every
    output in the same order: identical bits, 3x3 runs between 5x5 blocks and both SE kinds included."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, _ = nn_cases.make_case(name)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 91).numpy().reshape(-1)
    outs = []
    for mode in ("roles", "symmetric"):
        monkeypatch.setenv("CRA_X3_TOWER", mode)
        net = HipAPI(0, batch, d, "float16x3-1wg")
        v, p = np.zeros(batch, np.float32), np.zeros(batch * cfg.nb_policy, np.float32)
        net.predict(x, v, p)
        net.close()
        outs.append((v, p))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("name", ["risev2-3", "risev2-7", "risev2-13", "risev2-19", "risev33", "risev33-wdlp", "risev2-13-lichess"])
def test_float16p8_equals_its_emulation(tmp_path, hip_lib, name):
    """Precision float16p8 is DEFINED by oracle.forward_p8 (f16 main term + two e5m2 cross terms in the two 1x1 contractions of the
    one-launch tower, everything else float16x3): the kernel must sit much closer to that definition than the mode sits to fp32 -- what
    is left is the f32 accumulation order of the matrix unit -- and the mode itself within 3e-4 of fp32 on the logits (north_star: 1e-3)."""
    cfg, sd, x, value, probs, aux, logits = _run(tmp_path, hip_lib, name, "float16p8-1wg")     # (the tower kernels: four boards would run split, below)
    e_value, e_logits, _ = ro.forward_p8(cfg, sd, x)
    o_value, o_logits, _ = ro.forward(cfg, sd, x)
    mode = float((e_logits - o_logits).abs().max())
    kernel_vs_emulation = float(np.abs(logits - e_logits.numpy()).max())
    kernel_vs_fp32 = float(np.abs(logits - o_logits.numpy()).max())
    assert 5e-6 < mode < 3e-4, mode                                    # the mode is not float16x3 (1e-6) and not float16 (1e-3)
    # The 8-bit images are discontinuous functions of the activations (truncation to two mantissa bits): a difference of one f32 ulp between the
    # kernel's and the emulation's accumulation order flips a byte here and there, each flip is worth 2^-13 of a product, and the last flips
    # sit in the policy-map conv itself.  How far such differences carry is measured on the emulation: its weights perturbed by 1e-7 (two draws).
    sens = 0.0
    for seed in (7, 8):
        g = torch.Generator().manual_seed(seed)
        sd2 = {k: (v * (1 + 1e-7 * torch.randn(v.shape, generator=g)) if v.dtype.is_floating_point and v.dim() > 0 else v) for k, v in sd.items()}
        sens = max(sens, float((ro.forward_p8(cfg, sd2, x)[1] - e_logits).abs().max()))
    assert kernel_vs_emulation < max(2e-5, 2.5 * sens), (kernel_vs_emulation, sens, mode)
    assert kernel_vs_fp32 < 3e-4
    assert np.abs(value - e_value.numpy().reshape(-1)).max() < 2.5e-5


@pytest.mark.parametrize("name,towers", [("risev2-19", 1), ("risev33", 5)])
def test_float16p8_launch_structure(tmp_path, hip_lib, name, towers):
    """What the headline mode's forward IS, launch by launch: the stem conv, ONE tower launch per run of blocks of one depthwise size (RISEv2: one
    for the whole tower; RISEv3.3: its 3x3 and 5x5 runs alternate -- no layer-kernel blocks and no SE launches left), the policy head of a
    policy-map net in one launch (both convs and the softmax), the value head."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, x = nn_cases.make_case(name)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    net = HipAPI(0, int(x.shape[0]), d, "float16p8-1wg")
    names = [n for n, _ in net.time_ops(1)]
    net.close()
    assert names == ["conv_gemm_x3_3x3"] + ["tower_p8"] * towers + ["conv_gemm_x3_3x3", "value_head"], names
    # at batch 256 the plain name is that structure; at 64 boards and below the 3x3 runs go one block per launch, several workgroups per board
    net = HipAPI(0, 256, d, "float16p8")
    assert [n for n, _ in net.time_ops(1)] == names
    net.close()
    net = HipAPI(0, 8, d, "float16p8")
    small = [n for n, _ in net.time_ops(1)]
    net.close()
    assert "block_x3_split" in small and "x3_split_finish" in small and ("tower_p8" in small) == (name == "risev33"), small
    assert small[-2:] == ["conv_gemm_x3_3x3", "heads_small"], small       # policy conv 1; policy conv 2 + softmax beside the value head


@pytest.mark.parametrize("precision", ["float16p8", "float16x3"])
@pytest.mark.parametrize("name,batch", [("risev2-7", 8), ("risev2-19", 1), ("risev33-wdlp", 5), ("risev2-13-lichess", 64)])
def test_small_batch_heads_side_by_side_equal_heads_in_sequence(tmp_path, hip_lib, name, batch, precision, monkeypatch):
    """Round 6: a net made for <= 64 boards runs the second policy conv (with the softmax) and the value head as the two roles of ONE launch
    (x3.hip: heads_small_kernel) and spreads the couts of its wide convs (stem, policy conv 1) over four workgroups per board.  Neither changes
    one output's arithmetic: identical bits to the same net with the heads as two launches and two workgroups per wide conv
    (CRA_SMALL_BATCH_HEADS_APART, CRA_SMALL_BATCH_CONV_SPLIT=1), value, probabilities, logits and the WDLP outputs."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, _ = nn_cases.make_case(name)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    xin = np.ascontiguousarray(nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 78).numpy())
    outs = {}
    for apart in (False, True):
        if apart:
            monkeypatch.setenv("CRA_SMALL_BATCH_HEADS_APART", "1")
            monkeypatch.setenv("CRA_SMALL_BATCH_CONV_SPLIT", "1")
        else:
            monkeypatch.delenv("CRA_SMALL_BATCH_HEADS_APART", raising=False)
            monkeypatch.delenv("CRA_SMALL_BATCH_CONV_SPLIT", raising=False)
        net = HipAPI(0, batch, d, precision, keep_logits=True)
        names = [n for n, _ in net.time_ops(1)]
        assert ("heads_small" in names) == (not apart) and ("value_head" in names) == apart, names
        v, p = np.full(batch, 7.0, np.float32), np.full(batch * cfg.nb_policy, 7.0, np.float32)
        aux = np.full(batch * 4, 7.0, np.float32) if cfg.nb_aux else None
        for _ in range(2):
            net.predict(xin, v, p, aux)
        logits = torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy().copy()
        outs[apart] = (v, p, logits, aux)
        net.close()
    for a, b in zip(outs[False], outs[True]):
        assert (a is None and b is None) or np.array_equal(a, b)
    assert np.abs(outs[False][0]).max() <= 1.0 and not np.any(outs[False][1] == 7.0)


@pytest.mark.parametrize("precision", ["float16p8", "float16x3"])
@pytest.mark.parametrize("name,batch", [("risev2-19", 1), ("risev2-19", 3), ("risev2-19", 8), ("risev2-19", 32), ("risev2-7", 8), ("risev33-wdlp", 5),
                                        ("risev2-13-lichess", 16)])
def test_board_split_forward_small_batches(tmp_path, hip_lib, name, batch, precision):
    """Round 6 (VERDICT r05 next #1b): nets made for <= 64 boards run every 3x3 bottleneck block as ONE launch with up to C_op / 128 workgroups
    per board, each owning a share of the block's chunks, partial project sums added as 64-bit fixed point (block_x3_split_kernel).  The
    arithmetic is float16x3's in both modes, so the bound is float16x3's 1e-4 on the logits; integer sums are order-free: two nets, three
    forwards each, identical bits; and the split forward sits within float16x3 round-off of the one-workgroup-per-board tower."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, _ = nn_cases.make_case(name)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 77)
    xin = np.ascontiguousarray(x.numpy())
    o_value, o_logits, o_aux = ro.forward(cfg, sd, x)
    outs = []
    for prec in (precision, precision, "float16x3-1wg"):
        net = HipAPI(0, batch, d, prec, keep_logits=True)
        if prec == precision:
            assert "block_x3_split" in [n for n, _ in net.time_ops(1)]
        for _ in range(3):
            v, p = np.full(batch, 7.0, np.float32), np.full(batch * cfg.nb_policy, 7.0, np.float32)
            aux = np.full(batch * 4, 7.0, np.float32) if cfg.nb_aux else None
            net.predict(xin, v, p, aux)
            logits = torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy()
            outs.append((prec, v, p, logits.copy()))
        net.close()
    tol = TOL[precision]        # (float16p8: the policy head's convs -- and RISEv3.3's 5x5 runs, tower_p8_kernel<5> -- keep the mode's own arithmetic)
    for prec, v, p, logits in outs[:6]:
        assert np.abs(v - o_value.numpy().reshape(-1)).max() < tol["value"]
        assert np.abs(logits - o_logits.numpy()).max() < tol["logit"]
        assert np.abs(p.reshape(batch, -1) - torch.softmax(o_logits, 1).numpy()).max() < tol["prob"]
        assert np.array_equal(v, outs[0][1]) and np.array_equal(p, outs[0][2]) and np.array_equal(logits, outs[0][3])    # run to run, net to net
    if precision == "float16x3":
        assert np.abs(outs[0][3] - outs[6][3]).max() < 2e-5            # against the tower kernel's bits: f32 round-off of another summation order


@pytest.mark.parametrize("name,batch,precision", [("risev2-7", 72, "float16x3"), ("risev2-19", 256, "float16x3"), ("risev2-13-lichess", 16, "float16x3-1wg"),
                                                  ("risev33-wdlp", 8, "float16x3-1wg")])
def test_float16x3_policy_head_in_one_launch_equals_two_launches(tmp_path, hip_lib, name, batch, precision, monkeypatch):
    """Round 6: Precision float16x3 runs the policy head of a policy-map net -- conv 3x3 256 -> 256 + BN + ReLU, conv 3x3 256 -> P, softmax --
    as ONE launch (x3.hip: conv3x3_x3_chain_kernel; the first conv's output goes into the second's operand tiles in LDS instead of through
    HBM).  The arithmetic is the two launches' step for step: identical logits, probabilities and values (CRA_X3_NO_HEAD_CHAIN = the two
    launches)."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, _ = nn_cases.make_case(name)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 80)
    xin = np.ascontiguousarray(x.numpy())
    o_value, o_logits, _ = ro.forward(cfg, sd, x)
    outs = {}
    for chain in (True, False):
        if chain:
            monkeypatch.delenv("CRA_X3_NO_HEAD_CHAIN", raising=False)
        else:
            monkeypatch.setenv("CRA_X3_NO_HEAD_CHAIN", "1")
        net = HipAPI(0, batch, d, precision, keep_logits=True)
        names = [n for n, _ in net.time_ops(1)]
        assert names.count("conv_gemm_x3_3x3") == (2 if chain else 3), names
        v, p = np.full(batch, 7.0, np.float32), np.full(batch * cfg.nb_policy, 7.0, np.float32)
        for _ in range(2):
            net.predict(xin, v, p, np.full(batch * 4, 7.0, np.float32) if cfg.nb_aux else None)
        outs[chain] = (v, p, torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy().copy())
        net.close()
    for a, b in zip(outs[True], outs[False]):
        assert np.array_equal(a, b)
    assert np.abs(outs[True][2] - o_logits.numpy()).max() < TOL["float16x3"]["logit"]
    assert np.abs(outs[True][1].reshape(batch, -1) - torch.softmax(o_logits, 1).numpy()).max() < TOL["float16x3"]["prob"]


@pytest.mark.parametrize("name,batch", [("risev2-19", 1), ("risev2-19", 8), ("risev2-19", 40), ("risev33", 4)])
def test_gate_from_the_images_channel_sums_equals_the_gate_phase(tmp_path, hip_lib, name, batch, monkeypatch):
    """Round 6: a gated block of the split-board forward takes the board's channel means from the channel sums the launch before left per
    image (means are linear) and has its gate before it stages the board, instead of squeezing the staged tiles (x3_se_phase;
    CRA_X3_SPLIT_DEV=8 keeps that form).  Same gate up to f32 round-off of another summation order: logits within 2e-5 of the old form, both
    inside float16x3's bound against the oracle, and run to run identical."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, _ = nn_cases.make_case(name)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 79)
    xin = np.ascontiguousarray(x.numpy())
    o_value, o_logits, _ = ro.forward(cfg, sd, x)
    outs = {}
    for old in (False, True):
        if old:
            monkeypatch.setenv("CRA_X3_SPLIT_DEV", "8")
        else:
            monkeypatch.delenv("CRA_X3_SPLIT_DEV", raising=False)
        net = HipAPI(0, batch, d, "float16x3", keep_logits=True)
        runs = []
        for _ in range(3):
            v, p = np.full(batch, 7.0, np.float32), np.full(batch * cfg.nb_policy, 7.0, np.float32)
            net.predict(xin, v, p, np.full(batch * 4, 7.0, np.float32) if cfg.nb_aux else None)
            runs.append((v, torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy().copy()))
        net.close()
        assert all(np.array_equal(r[0], runs[0][0]) and np.array_equal(r[1], runs[0][1]) for r in runs)
        assert np.abs(runs[0][1] - o_logits.numpy()).max() < TOL["float16x3"]["logit"]
        assert np.abs(runs[0][0] - o_value.numpy().reshape(-1)).max() < TOL["float16x3"]["value"]
        outs[old] = runs[0]
    assert np.abs(outs[False][1] - outs[True][1]).max() < 2e-5
    assert not np.array_equal(outs[False][1], outs[True][1]) or name == "risev33"      # (the switch does switch: other bits in the last place)


@pytest.mark.parametrize("case,B,version,stress", [("risev2-19", 256, "1.0", 3.0), ("risev2-13-lichess", 1024, "3.0", 2.0)])
def test_float16p8_error_grows_with_the_logit_scale_float16x3_holds(tmp_path, hip_lib, case, B, version, stress):
    """float16p8's bound on the seeded random nets (3e-4 on the fixtures, 7e-4 over a million logits; max|logit| 2 ... 7)
    is a property of those weights.  Its cross terms keep two mantissa bits, so its error is RELATIVE to the activations: on the same nets
    with activations 2 - 3 times larger (nn_cases.scale_activations; logits of +-10, what trained nets have) it passes north_star's
    1e-3 -- measured 1.6e-3 at max|logit| 10.4 (RISEv2-19) and 1.04e-3 at 9.8 (RISEv2-13 lichess), profiles/r06/h_precision_vs_logit_scale.txt
    -- while float16x3 (f32-grade products) stays at 1.3e-4 / 8.7e-5 there and inside 1e-3 up to max|logit| ~ 25, where the exact-f32 mode
    itself differs from the CPU oracle by 2e-4.  Bounds held here, for logits up to +-20: float16p8 <= 2.5e-4 x max|logit| (the factor
    itself grows with the scale: 7e-5 at +-2, 1.6e-4 at +-10, 4e-4 at +-30), float16x3 <= 4e-5 x max|logit| and < 1e-3.  Hence bench.py's
    headline mode is float16x3; float16p8 is reported beside it with this bound."""
    from crazyara_amd.neuralnetapi import HipAPI
    cfg, sd, _ = nn_cases.make_case(case)
    x = nn_cases.synthetic_planes(B, cfg.nb_input_channels, 4711)
    xin = np.ascontiguousarray(x.numpy())
    report = {}
    for act in (1.0, stress):
        sds = nn_cases.scale_activations(cfg, sd, act) if act != 1.0 else sd
        d = nn_cases.export_case(tmp_path / f"a{int(act)}", case, cfg, sds, version=version)
        o_value, o_logits, _ = ro.forward(cfg, sds, x)
        mx = float(o_logits.abs().max())
        for precision in ("float16p8", "float16x3"):
            net = HipAPI(0, B, d, precision, keep_logits=True)
            v, p = np.zeros(B, np.float32), np.zeros(B * cfg.nb_policy, np.float32)
            net.predict(xin, v, p, np.zeros(B * 4, np.float32) if cfg.nb_aux else None)
            logits = torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy()
            net.close()
            err = float(np.abs(logits - o_logits.numpy()).max())
            verr = float(np.abs(v - o_value.numpy().reshape(-1)).max())
            report[(act, precision)] = dict(max_logit=round(mx, 2), logit_err=err, per_unit_logit=err / mx, value_err=verr)
    print("precision against the logit scale:", case, B, report)
    for act in (1.0, stress):
        r8, r3 = report[(act, "float16p8")], report[(act, "float16x3")]
        assert r8["logit_err"] < 2.5e-4 * r8["max_logit"], report
        assert r3["logit_err"] < 4e-5 * r3["max_logit"] and r3["logit_err"] < 1e-3, report
        assert r3["value_err"] < 1e-4 and r8["value_err"] < 5e-4, report
    assert 8.0 < report[(stress, "float16p8")]["max_logit"] < 20.0       # the stress case IS at the logit scale of trained nets
    assert report[(1.0, "float16p8")]["logit_err"] < 1e-3                # the seeded nets of the fixtures and of bench.py: inside north_star
    assert report[(stress, "float16p8")]["logit_err"] > 1e-3             # ... and at trained-net scale outside it: why the mode is not the headline
