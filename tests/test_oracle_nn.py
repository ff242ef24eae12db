"""Pins the NN oracle (oracle/rise_oracle.py) against outputs of the reference PyTorch model (tests/golden/nn_*.npz,
made by oracle/make_golden.py from /root/reference) -- CPU only."""
import os

import numpy as np
import pytest
import torch

import nn_cases
from oracle import rise_oracle as ro


@pytest.mark.parametrize("name", list(nn_cases.CASES))
def test_oracle_matches_reference_golden(name):
    cfg, sd, x = nn_cases.make_case(name)
    g = np.load(os.path.join(nn_cases.GOLDEN_DIR, f"nn_{name}.npz"))
    assert np.array_equal(g["x"], x.numpy()), "input generator drifted from the committed fixture"
    torch.set_num_threads(1)
    value, logits, aux = ro.forward(cfg, sd, x)
    # same ops, same summation order family: only thread-partitioning round-off may differ
    assert np.abs(value.numpy() - g["value"]).max() < 2e-6
    assert np.abs(logits.numpy() - g["logits"]).max() < 2e-5
    if cfg.nb_aux:
        assert np.abs(aux.numpy() - g["aux"]).max() < 2e-5
    else:
        assert g["aux"].size == 0


def test_predict_contract_softmax_and_shapes():
    cfg, sd, x = nn_cases.make_case("risev2-3")
    value, probs, aux = ro.predict(cfg, sd, x)
    assert value.shape == (x.shape[0],) and probs.shape == (x.shape[0], 5184) and aux is None
    assert torch.allclose(probs.sum(dim=1), torch.ones(x.shape[0]), atol=1e-5)
    assert float(value.abs().max()) <= 1.0


def test_flops_match_survey_table():
    # 259 / 554 / 1002 MFLOP per position for RISEv2-7/13/19, 523 for RISEv3.3 chess
    for n, mf in ((7, 259), (13, 554), (19, 1002)):
        assert abs(ro.flops_per_position(ro.rise_v2_config(n)) / 1e6 - mf) < 2.0
    assert abs(ro.flops_per_position(ro.rise_v33_config()) / 1e6 - 523) < 2.0


@pytest.mark.reference
def test_oracle_equals_live_reference(has_reference):
    if not has_reference:
        pytest.skip("/root/reference not present (GPU box)")
    from oracle import make_golden
    RiseV3 = make_golden.import_reference()
    cfg, sd, x = nn_cases.make_case("risev33-wdlp")
    model = make_golden.reference_model(RiseV3, cfg)
    model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        out = model(x)
    value, logits, aux = ro.forward(cfg, sd, x)
    assert float((value - out[0]).abs().max()) < 2e-6
    assert float((logits - out[1]).abs().max()) < 2e-5
    assert float((aux - out[2]).abs().max()) < 2e-5


@pytest.mark.parametrize("name", ["risev2-3", "risev2-7", "risev33-wdlp", "alphazero-3-se", "risev2-3-flat", "rise-classical-3-se"])
def test_float16x3_emulation_is_fp32_to_round_off(name):
    """Precision float16x3 (crazyara_amd/csrc/nn/x3.hip) is DEFINED by oracle.forward_x3: every dense contraction on hi/lo-split f16
    operands, three products per term.  Its distance from the pinned fp32 forward is the mode's error: f32 round-off (the GPU tests
    hold the kernels to 1e-4 against fp32; Precision float16 sits at 1e-3 ... 3e-3 on these cases)."""
    cfg, sd, x = nn_cases.make_case(name)
    v, l, a = ro.forward(cfg, sd, x)
    v3, l3, a3 = ro.forward_x3(cfg, sd, x)
    assert (l3 - l).abs().max() < 2e-5 and (v3 - v).abs().max() < 5e-6
    if a is not None:
        assert (a3 - a).abs().max() < 5e-6
    v16, l16, _ = ro.forward(cfg, sd, x, sim_dtype=torch.float16)
    assert (l16 - l).abs().max() > 20 * (l3 - l).abs().max()        # and two orders of magnitude inside the plain f16 emulation


def test_float16x3_split_carries_22_bits():
    """hi + lo of the split reproduces a float to 2^-22 relative (2^-25 absolute where lo is an f16 subnormal)."""
    g = torch.Generator().manual_seed(3)
    for scale in (4.0, 1.0, 0.05, 1e-3):
        t = torch.randn(20000, generator=g) * scale
        hi, lo = ro._split_f16(t)
        err = (hi + lo - t.double()).abs()
        assert (err <= torch.maximum(t.double().abs() * 2.0 ** -22, torch.tensor(2.0 ** -25, dtype=torch.float64))).all()


@pytest.mark.parametrize("name", ["risev2-3", "risev2-7", "risev2-19", "risev33", "risev2-13-lichess"])
def test_float16p8_emulation_is_conformant(name):
    """Precision float16p8 (crazyara_amd/csrc/nn/x3.hip: tower_p8_kernel) is DEFINED by oracle.forward_p8: float16x3 whose tower runs its two 1x1
    contractions as f16 main term + two e5m2 cross terms (truncated activation bytes, compensated weight images).  The definition sits within
    3e-4 of the fp32 oracle on the logits (north_star: 1e-3) and an order of magnitude above float16x3's round-off: a mode of its own, not an alias."""
    cfg, sd, x = nn_cases.make_case(name)
    v32, l32, _ = ro.forward(cfg, sd, x)
    v8, l8, _ = ro.forward_p8(cfg, sd, x)
    v3, l3, _ = ro.forward_x3(cfg, sd, x)
    e8, e3 = float((l8 - l32).abs().max()), float((l3 - l32).abs().max())
    assert e8 < 3e-4 and float((v8 - v32).abs().max()) < 1e-4
    assert e8 > 2 * e3
