"""
ORACLE (test infrastructure only -- never imported by the product path).

CPU fp32 restatement of the reference's RISE network forward pass, written as plain
functional torch ops on a state-dict that uses the reference's own parameter names.

Follows (reference file:line, relative to /root/reference):
  * RiseV3.forward / layer order ........ DeepCrazyhouse/src/domain/neural_net/architectures/pytorch/rise_mobile_v3.py:81-184
  * get_rise_v2_model / get_rise_v33_model  rise_mobile_v3.py:186-241
  * _get_res_blocks (C_op schedule, k=5 rule) rise_mobile_v3.py:36-78
  * _Stem ................................ pytorch/builder_util.py:154-178
  * _BottlekneckResidualBlock ............ builder_util.py:437-475 (residual is added to the SE-scaled x, :473-475)
  * _ChannelAttentionModule (ca_se) ...... builder_util.py:83-114 (bias-free FCs, hard-sigmoid)
  * _EfficientChannelAttentionModule ..... builder_util.py:49-80 (Conv1d over a length-1 sequence == centre-tap linear + bias)
  * _PolicyHead .......................... builder_util.py:206-243 (channel-major flatten = plane*64+sq)
  * _ValueHead ........................... builder_util.py:246-326 (tanh head; WDL variant value=-softmax(wdl)[0]+softmax(wdl)[2])
  * process_value_policy_head ............ builder_util.py:385-398 (aux = cat(wdl, plys))
  * softmax contract of predict() ........ engine/src/nn/tensorrtapi.cpp:378-392, engine/src/nn/neuralnetapi.cpp:241-260
  * ClassicalResidualBlock ............... builder_util.py:401-434 (second activation INSIDE the body, plain add)
  * AlphaZeroResnet / ResidualBlock ...... pytorch/a0_resnet.py:72-183 (activation AFTER the add; state-dict prefix "body.")
  * SE inside the dense blocks ........... ClassicalResidualBlock(se_type): gate on the block INPUT, hard-sigmoid (builder_util.py:416,431-433);
                                           ResidualBlock(use_se): gate on the body OUTPUT, plain sigmoid (a0_resnet.py:94-95,104-106)

Pinning: `oracle/make_golden.py` (run in the build container where /root/reference exists) loads the SAME
state dicts into the imported reference model and stores (input, value, policy logits, aux) under tests/golden/;
tests/test_oracle_nn.py checks this restatement against those fixtures (exact to fp32 round-off).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default (reference never overrides it)


from crazyara_amd.rise_config import (RiseConfig, rise_v2_config, rise_v33_config, rise_classical_config,  # noqa: F401,E402
                                      alpha_zero_config, eca_kernel, make_state_dict)


# --------------------------------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------------------------------
def _bn(sd, name, x):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=False, eps=BN_EPS)


def _q(x, sim_dtype):
    """Optional rounding of an activation tensor to a storage dtype (used to *predict* low-precision error)."""
    return x if sim_dtype is None else x.to(sim_dtype).to(torch.float32)


@torch.no_grad()
def forward(cfg: RiseConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor, sim_dtype=None, taps: dict = None):
    """Returns (value[B,1], policy_logits[B,P*64], aux[B,4] or None).  x: [B,C,8,8] fp32."""
    W = (lambda n: sd[n]) if sim_dtype is None else (lambda n: sd[n].to(sim_dtype).to(torch.float32))
    x = x.to(torch.float32)
    pre = cfg.key_prefix
    h = F.relu(_bn(sd, pre + ".0.body.1", F.conv2d(_q(x, sim_dtype), W(pre + ".0.body.0.weight"), padding=1)))
    h = _q(h, sim_dtype)
    if taps is not None:
        taps["stem"] = h
    for i, (k, se) in enumerate(zip(cfg.kernels, cfg.se_types)):
        p = f"{pre}.{i + 1}"
        if cfg.dense_blocks:
            def gate(z, hard):                           # get_se(...): _ChannelAttentionModule / _EfficientChannelAttentionModule
                y = z.mean(dim=(2, 3))
                if se in ("ca_se", "se"):
                    y = F.linear(F.relu(F.linear(y, sd[p + ".se.fc.0.weight"])), sd[p + ".se.fc.2.weight"])
                else:
                    w = sd[p + ".se.body.0.weight"]
                    y = F.conv1d(y[:, :, None], w, sd[p + ".se.body.0.bias"], padding=w.shape[2] // 2)[:, :, 0]
                y = F.hardsigmoid(y) if hard else torch.sigmoid(y)
                return z * y[:, :, None, None]
            if se is not None and cfg.conv_block == "classical_res_block":
                h = _q(gate(h, True), sim_dtype)         # builder_util.py:431-433: x = se(x) (use_hard_sigmoid=True, :416)
            t = _q(F.relu(_bn(sd, p + ".body.1", F.conv2d(h, W(p + ".body.0.weight"), padding=1))), sim_dtype)
            t = _bn(sd, p + ".body.4", F.conv2d(t, W(p + ".body.3.weight"), padding=1))
            if cfg.conv_block == "classical_res_block":
                h = _q(h + F.relu(t), sim_dtype)         # builder_util.py:413-418,434: act is the body's last module
            else:
                if se is not None:
                    t = gate(_q(t, sim_dtype), False)    # a0_resnet.py:94-95,104-106: out = se(out), use_hard_sigmoid=False
                h = _q(F.relu(h + t), sim_dtype)         # a0_resnet.py:104-107: final_act(x + out)
            if taps is not None:
                taps[f"block{i}"] = h
            continue
        if se in ("ca_se", "se"):
            y = h.mean(dim=(2, 3))
            y = F.relu(F.linear(y, sd[p + ".se.fc.0.weight"]))
            y = F.hardsigmoid(F.linear(y, sd[p + ".se.fc.2.weight"]))
            h = _q(h * y[:, :, None, None], sim_dtype)
        elif se == "eca_se":
            y = h.mean(dim=(2, 3))
            w = sd[p + ".se.body.0.weight"]
            y = F.conv1d(y[:, :, None], w, sd[p + ".se.body.0.bias"], padding=w.shape[2] // 2)[:, :, 0]
            y = F.hardsigmoid(y)
            h = _q(h * y[:, :, None, None], sim_dtype)
        t = _q(F.relu(_bn(sd, p + ".body.1", F.conv2d(h, W(p + ".body.0.weight")))), sim_dtype)
        cop = t.shape[1]
        t = _q(F.relu(_bn(sd, p + ".body.4", F.conv2d(t, W(p + ".body.3.weight"), padding=k // 2, groups=cop))), sim_dtype)
        t = _bn(sd, p + ".body.7", F.conv2d(t, W(p + ".body.6.weight")))
        h = _q(h + t, sim_dtype)
        if taps is not None:
            taps[f"block{i}"] = h
    # policy head
    ph = _q(F.relu(_bn(sd, "policy_head.body.1", F.conv2d(h, W("policy_head.body.0.weight"), padding=1))), sim_dtype)
    pol = F.conv2d(ph, W("policy_head.body.3.weight"), padding=1)
    if cfg.select_policy_from_plane:
        pol = pol.reshape(x.shape[0], -1)                      # channel-major flatten = plane*64 + square
    else:                                                      # builder_util.py:229-232,241-243: BN + act, view, Linear
        pol = _q(F.relu(_bn(sd, "policy_head.body2.0", pol)), sim_dtype).reshape(x.shape[0], -1)
        pol = F.linear(pol, W("policy_head.body3.0.weight"), sd["policy_head.body3.0.bias"])
    # value head
    vh = F.relu(_bn(sd, "value_head.body.1", F.conv2d(h, W("value_head.body.0.weight")))).reshape(x.shape[0], -1)
    aux = None
    if cfg.use_wdl and cfg.use_plys_to_end:
        wdl = F.linear(vh, sd["value_head.body_wdl.0.weight"], sd["value_head.body_wdl.0.bias"])
        plys = torch.sigmoid(F.linear(vh, sd["value_head.body_plys.0.weight"], sd["value_head.body_plys.0.bias"]))
        sm = torch.softmax(wdl, dim=1)
        value = -sm[:, 0:1] + sm[:, 2:3]
        aux = torch.cat((wdl, plys), dim=1)
    else:
        v = F.relu(F.linear(vh, sd["value_head.body_final.0.weight"], sd["value_head.body_final.0.bias"]))
        value = torch.tanh(F.linear(v, sd["value_head.body_final.2.weight"], sd["value_head.body_final.2.bias"]))
    return value, pol, aux


@torch.no_grad()
def predict(cfg, sd, x, sim_dtype=None):
    """NeuralNetAPI::predict contract: value (tanh range), policy AFTER softmax over all nbPolicy entries, aux."""
    value, pol, aux = forward(cfg, sd, x, sim_dtype)
    return value.reshape(-1), torch.softmax(pol, dim=1), aux


def flops_per_position(cfg: RiseConfig) -> float:
    """2*MACs per board position, recomputed from the layer list (SURVEY 8d formula)."""
    C = cfg.channels
    macs = 64 * cfg.nb_input_channels * C * 9
    for k, cop, se in zip(cfg.kernels, cfg.channels_operating(), cfg.se_types):
        if cfg.dense_blocks:
            macs += 2 * 64 * C * C * 9
            if se in ("ca_se", "se"):
                macs += 2 * C * (C // 2)
            elif se == "eca_se":
                macs += C * C
            continue
        macs += 64 * C * cop * 2 + 64 * cop * k * k
        if se in ("ca_se", "se"):
            macs += 2 * C * (C // 2)
        elif se == "eca_se":
            macs += C * C
    macs += 64 * C * C * 9 + 64 * C * cfg.channels_policy_head * 9
    if not cfg.select_policy_from_plane:
        macs += 64 * cfg.channels_policy_head * cfg.n_labels
    macs += 64 * C * cfg.channels_value_head
    if cfg.use_wdl and cfg.use_plys_to_end:
        macs += 4 * 64 * cfg.channels_value_head
    else:
        macs += 64 * cfg.channels_value_head * cfg.value_fc_size + cfg.value_fc_size
    return 2.0 * macs
