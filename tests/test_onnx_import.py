"""ONNX model import: csrc/nn/onnx_import.cpp through mi_onnx_to_cranet, host only.

Pinned two ways:
  * tests/golden/onnx/*.onnx are the reference's own PyTorch modules serialised by torch's ONNX exporter (oracle/make_onnx_fixtures.py),
    the bytes a reference user holds; the weights are reproducible from (config, seed), so the import is checked tensor by tensor;
  * tests/onnx_writer.py writes the same networks in the other flavours exporters leave behind (BatchNormalization nodes, MatMul+Add,
    Flatten / Reshape, packed float_data, fp16 weights, pruned plies-to-end branch) at the sizes the GPU tests load.
What has to hold: every conv+BN pair folds (as csrc/nn/rise_net.hip:fold_bn does it) to the same weight and bias as the original
state dict, every Linear / gate tensor is identical, and the meta describes the same architecture.
"""
import os

import numpy as np
import pytest

import nn_cases
import onnx_cases
import onnx_writer
from crazyara_amd import _capi, netfile
from crazyara_amd.rise_config import make_state_dict

ONNX_DIR = os.path.join(nn_cases.GOLDEN_DIR, "onnx")
BN_EPS = 1e-5


@pytest.fixture(scope="module")
def lib(hip_lib):
    return hip_lib


def _np(sd):
    out = {}
    for k, v in sd.items():
        k = "body_spatial." + k[len("body."):] if k.startswith("body.") else k        # AlphaZeroResnet keeps stem + blocks in `body`
        out[k] = v.detach().cpu().numpy().astype(np.float64) if hasattr(v, "detach") else np.asarray(v, np.float64)
    return out


def _fold(t, conv, bn):
    w = np.asarray(t[conv + ".weight"], np.float64)
    if bn is None:
        return w, np.zeros(w.shape[0])
    g, b, m, v = (np.asarray(t[f"{bn}.{s}"], np.float64) for s in ("weight", "bias", "running_mean", "running_var"))
    s = g / np.sqrt(v + BN_EPS)
    return w * s.reshape(-1, 1, 1, 1), b - m * s


def _conv_bn_pairs(cfg):
    pairs = [("body_spatial.0.body.0", "body_spatial.0.body.1")]
    for i in range(len(cfg.kernels)):
        p = f"body_spatial.{i + 1}"
        pairs += [(p + ".body.0", p + ".body.1"), (p + ".body.3", p + ".body.4")]
        if not cfg.dense_blocks:
            pairs.append((p + ".body.6", p + ".body.7"))
    pairs += [("policy_head.body.0", "policy_head.body.1"), ("value_head.body.0", "value_head.body.1")]
    pairs.append(("policy_head.body.3", None if cfg.select_policy_from_plane else "policy_head.body2.0"))
    return pairs


def check_import(cfg, sd, meta, tensors, version, rtol=2e-6, pruned_plys=False):
    sd = _np(sd)
    wdl = cfg.use_wdl and cfg.use_plys_to_end
    expect = dict(arch="rise", source="onnx", input_version=version, nb_input_channels=cfg.nb_input_channels, channels=cfg.channels,
                  channels_operating=",".join(str(c) for c in cfg.channels_operating()), kernels=",".join(str(k) for k in cfg.kernels),
                  se_types=",".join("none" if s is None else "ca_se" if s == "se" else s for s in cfg.se_types),
                  channels_value_head=cfg.channels_value_head, value_fc_size=0 if wdl else cfg.value_fc_size,
                  channels_policy_head=cfg.channels_policy_head, use_wdl=int(wdl), use_plys_to_end=int(wdl), conv_block=cfg.conv_block,
                  select_policy_from_plane=int(cfg.select_policy_from_plane), n_labels=0 if cfg.select_policy_from_plane else cfg.n_labels)
    for k, v in expect.items():
        assert meta[k] == str(v), (k, meta[k], v)
    seen = set()
    for conv, bn in _conv_bn_pairs(cfg):
        w0, b0 = _fold(sd, conv, bn)
        w1, b1 = _fold(tensors, conv, bn)
        scale = np.abs(w0).max()
        assert w1.shape == w0.shape and np.abs(w1 - w0).max() <= rtol * scale + 1e-9, conv
        assert np.abs(b1 - b0).max() <= rtol * max(1.0, np.abs(b0).max()), conv
        seen.add(conv + ".weight")
        if bn:
            seen.update(f"{bn}.{s}" for s in ("weight", "bias", "running_mean", "running_var"))
    exact = []
    for i, se in enumerate(cfg.se_types):
        p = f"body_spatial.{i + 1}"
        if se in ("ca_se", "se"):
            exact += [p + ".se.fc.0.weight", p + ".se.fc.2.weight"]
        elif se == "eca_se":
            exact += [p + ".se.body.0.weight", p + ".se.body.0.bias"]
    if wdl:
        exact += ["value_head.body_wdl.0.weight", "value_head.body_wdl.0.bias"]
        if not pruned_plys:
            exact += ["value_head.body_plys.0.weight", "value_head.body_plys.0.bias"]
        else:
            assert not tensors["value_head.body_plys.0.weight"].any() and not tensors["value_head.body_plys.0.bias"].any()
            seen.update(["value_head.body_plys.0.weight", "value_head.body_plys.0.bias"])
    else:
        exact += [f"value_head.body_final.{i}.{s}" for i in (0, 2) for s in ("weight", "bias")]
    if not cfg.select_policy_from_plane:
        exact += ["policy_head.body3.0.weight", "policy_head.body3.0.bias"]
    for k in exact:
        assert tensors[k].shape == sd[k].shape, k
        assert np.abs(tensors[k] - sd[k]).max() <= rtol * max(1e-30, np.abs(sd[k]).max()), k
        seen.add(k)
    assert seen == set(tensors), set(tensors) ^ seen          # nothing else in the container, nothing missing


def _convert(lib, tmp_path, data, fname):
    src = os.path.join(str(tmp_path), fname)
    with open(src, "wb") as f:
        f.write(data)
    return netfile.read_cranet(netfile.onnx_to_cranet(src))


@pytest.mark.parametrize("name", list(onnx_cases.CASES))
def test_files_of_the_torch_exporter(lib, tmp_path, name):
    cfg, seed, fname, _, stress = onnx_cases.unpack(name)
    with open(os.path.join(ONNX_DIR, fname), "rb") as f:
        data = f.read()
    meta, tensors = _convert(lib, tmp_path, data, fname)
    assert meta["producer"] == "pytorch"
    version = fname.split("-v")[1][:3]
    check_import(cfg, make_state_dict(cfg, seed=seed, stress=stress), meta, tensors, version)      # exporter folds BN in fp32


@pytest.mark.parametrize("flavour", [dict(fold_bn=False, linear="gemm"), dict(fold_bn=True, linear="matmul"),
                                     dict(fold_bn=False, linear="matmul", raw=False), dict(fold_bn=True, linear="gemm", batch=8)])
@pytest.mark.parametrize("name", ["risev2-3", "risev33-wdlp", "rise-classical-4", "alphazero-3-cv8", "risev2-3-flat", "rise-classical-3-se",
                                  "alphazero-3-se"])
def test_other_exporter_flavours_at_full_size(lib, tmp_path, name, flavour):
    cfg, sd, _ = nn_cases.make_case(name)
    data = onnx_writer.rise_to_onnx(cfg, sd, **flavour)
    meta, tensors = _convert(lib, tmp_path, data, f"{cfg.name}-v1.0.onnx")
    check_import(cfg, sd, meta, tensors, "1.0")


def test_fp16_weights_and_pruned_plys_branch(lib, tmp_path):
    cfg, sd, _ = nn_cases.make_case("risev33-wdlp")
    sd16 = {k: (v.half().float() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    meta, tensors = _convert(lib, tmp_path, onnx_writer.rise_to_onnx(cfg, sd16, fold_bn=False, weights_dtype=np.float16), "m-v3.0.onnx")
    check_import(cfg, sd16, meta, tensors, "3.0")
    meta, tensors = _convert(lib, tmp_path, onnx_writer.rise_to_onnx(cfg, sd, prune_plys=True), "p-v3.0.onnx")
    check_import(cfg, sd, meta, tensors, "3.0", pruned_plys=True)


def test_python_reader_sees_the_same_graph(lib):
    """tests/onnx_reader.py (a second, independent decoder of the wire format, test-side only) decodes the exporter's file: same initializers as the C++ importer consumed."""
    from onnx_reader import read_onnx
    cfg, seed, fname, _, _ = onnx_cases.unpack("mobile-se-wdlp")
    g = read_onnx(os.path.join(ONNX_DIR, fname))
    assert g.producer == "pytorch" and [v.name for v in g.inputs] == ["data"] and g.inputs[0].shape == ["batch_size", 12, 8, 8]
    assert [v.name for v in g.outputs] == ["value_out", "policy_out", "auxiliary_out", "wdl_out", "plys_to_end_out"]
    ops = [n.op for n in g.nodes]
    assert ops.count("Conv") == 1 + 3 * 3 + 1 + 3 and ops.count("HardSigmoid") == 2 and "BatchNormalization" not in ops
    sd = make_state_dict(cfg, seed=seed)
    assert np.array_equal(g.initializers["value_head.body_wdl.0.weight"], sd["value_head.body_wdl.0.weight"].numpy())
    data = onnx_writer.rise_to_onnx(cfg, sd, fold_bn=False)
    g2 = read_onnx(data)
    assert [n.op for n in g2.nodes].count("BatchNormalization") == ops.count("Conv") - 2      # gate conv1d and policy-map conv have none
    g3 = read_onnx(os.path.join(ONNX_DIR, onnx_cases.unpack("mobile-shared-constants")[2]))
    assert sum(n.op == "Identity" for n in g3.nodes) >= 5                                       # shared BN constants


# ---- rejection: the importer names what it does not understand ------------------------------------------------------------------
def _fails(lib, tmp_path, data, fname="bad-v1.0.onnx"):
    src = os.path.join(str(tmp_path), fname)
    with open(src, "wb") as f:
        f.write(data)
    with pytest.raises(ValueError) as e:
        netfile.onnx_to_cranet(src)
    return str(e.value)


def test_rejects_what_it_cannot_represent(lib, tmp_path):
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    good = onnx_writer.rise_to_onnx(cfg, sd)
    assert "truncated" in _fails(lib, tmp_path, good[:len(good) // 2])
    assert "ONNX" in _fails(lib, tmp_path, b"CRANET01 definitely not protobuf")
    # a sigmoid gate instead of the reference's hard sigmoid
    W = onnx_writer
    sig = good.replace(W._str(4, "HardSigmoid"), W._str(4, "HardSigmoiX"))
    assert len(sig) == len(good) and "HardSigmoid" in _fails(lib, tmp_path, sig)
    # outputs under other names
    assert "value_out" in _fails(lib, tmp_path, good.replace(b"value_out", b"value_xyz"))
    # stride-2 convolution
    strided = good.replace(W._attr("strides", [1, 1]), W._attr("strides", [2, 1]), 1)
    assert strided != good and "stride" in _fails(lib, tmp_path, strided)
    with pytest.raises(ValueError, match="cannot open"):
        netfile.onnx_to_cranet(os.path.join(str(tmp_path), "missing.onnx"))


def test_model_directory_rules_of_the_reference(lib, tmp_path):
    """get_onnx_model_name (neuralnetapi.cpp:57-73): '-bsize-<B>.onnx' first, else the dynamic file; a file for another fixed batch
    alone is an error.  Discovery runs before any device call, so its errors surface on a CPU-only host too."""
    from crazyara_amd.neuralnetapi import HipAPI
    d = os.path.join(str(tmp_path), "model")
    os.makedirs(d)
    with pytest.raises(ValueError, match="doesn't contain a file ending with .cranet or .onnx"):
        HipAPI(0, 8, d, "float16")
    open(os.path.join(d, "net-v1.0-bsize-4.onnx"), "wb").close()
    with pytest.raises(ValueError, match="should either contain a onnx file supporting the current batch size"):
        HipAPI(0, 8, d, "float16")


def test_damaged_files_raise_and_never_crash(lib, tmp_path):
    """Byte damage anywhere in an exporter file (node table = the first KiBs, weights after it) ends in an error or a valid import."""
    import random
    with open(os.path.join(ONNX_DIR, "mobile-tanh-v1.0-bsize-2.onnx"), "rb") as f:
        exported = f.read()
    cfg, seed, _, _, _ = onnx_cases.unpack("mobile-se-wdlp")            # the other node types: BatchNormalization, MatMul + Add, gates
    written = onnx_writer.rise_to_onnx(cfg, make_state_dict(cfg, seed=seed), fold_bn=False, linear="matmul")
    rng = random.Random(7)
    src, dst = os.path.join(str(tmp_path), "fz-v1.0.onnx"), os.path.join(str(tmp_path), "fz.cranet")
    outcomes = {True: 0, False: 0}
    for it in range(600):
        good = exported if it % 2 else written
        b = bytearray(good)
        if it % 3 == 0:
            b = b[:rng.randrange(len(b))]
        elif it % 3 == 1:
            for _ in range(rng.randint(1, 3)):
                b[rng.randrange(min(len(b), 12000))] = rng.randrange(256)
        else:
            i = rng.randrange(len(b))
            del b[i:i + rng.randint(1, 48)]
        with open(src, "wb") as f:
            f.write(bytes(b))
        try:
            netfile.onnx_to_cranet(src, dst)
            outcomes[True] += 1
        except ValueError:
            outcomes[False] += 1
    assert outcomes[False] > 150
