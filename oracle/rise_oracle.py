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
    return _heads(cfg, sd, h, sim_dtype)


def _heads(cfg: RiseConfig, sd, h, sim_dtype=None):
    """Policy and value head on the tower output h (the tail of forward)."""
    W = (lambda n: sd[n]) if sim_dtype is None else (lambda n: sd[n].to(sim_dtype).to(torch.float32))
    x = h
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


# --------------------------------------------------------------------------------------------------------------
# Precision fp8 (the product's counterpart of the reference's TensorRT INT8 mode, tensorrtapi.cpp:229-248): emulation of the
# quantisation points of the HIP tower kernel, NOT a restatement of reference code (TensorRT's calibrated INT8 kernels are not in
# /root/reference).  What is pinned is the fp32 forward above; this function says which roundings the fp8 mode adds to it:
#   stem, SE gates, heads ........ as Precision float16 (f16 storage)
#   expand 1x1 ................... A = BN1-folded weights / s1[c] -> e4m3, B = residual stream (f16) -> e4m3, f32 accumulate,
#                                  + b1 / s1, ReLU, f16 (s1[c] = 2^floor(log2 max|w[c]|), folded into the depthwise weights)
#   depthwise .................... f16 weights (w2 * s1), f16 activations, accumulated in f16 by fused multiply-adds in tap order (the
#                                  kernel's v_pk_fma_f16 chain, started at the f16 BN2 bias); output -> e4m3
#   project 1x1 .................. A = BN3-folded weights / s3[c] -> e4m3, B = that e4m3 tile; y = f16(x + s3 * (acc + b3 / s3))
# e4m3 = OCP e4m3fn, round to nearest even, clamped at +-448 (torch.float8_e4m3fn after a clamp).
# --------------------------------------------------------------------------------------------------------------
E4M3_MAX = 448.0


def q_e4m3(x):
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def _fold(sd, conv, bn):
    w = sd[conv + ".weight"].double()
    s = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + BN_EPS)
    return w * s.view(-1, 1, 1, 1), sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * s


def row_scale_pow2(w):
    m = w.abs().flatten(1).max(dim=1).values
    e = torch.floor(torch.log2(m.clamp_min(2.0 ** -24)))
    return torch.where(m > 0, torch.exp2(e), torch.ones_like(m))


def _depthwise_f16_chain(t, w, b, k):
    """depthwise k x k on f16 values with an f16 accumulator: acc = f16(b); acc = f16(x * w + acc) tap by tap, row-major (exactly what a
    chain of fused multiply-adds in f16 computes; taps that fall off the board contribute x = 0)"""
    acc = b.to(torch.float16).double().view(1, -1, 1, 1).expand(t.shape[0], -1, 8, 8)
    tp = F.pad(t, (k // 2, k // 2, k // 2, k // 2)).double()
    wd = w.to(torch.float16).double()
    for dy in range(k):
        for dx in range(k):
            acc = (tp[:, :, dy:dy + 8, dx:dx + 8] * wd[:, 0, dy, dx].view(1, -1, 1, 1) + acc).to(torch.float16).double()
    return acc.float()


def _se_gate(sd, p, se, h, f16_weights=False):
    """hard-sigmoid channel gate of block p on the stream h (builder_util.py:49-114); f16_weights: the gate matrices rounded to f16 as the
    tower kernel holds them (f32 accumulation)"""
    q = (lambda t: t.to(torch.float16).to(torch.float32)) if f16_weights else (lambda t: t)
    y = h.mean(dim=(2, 3))
    if se in ("ca_se", "se"):
        return F.hardsigmoid(F.linear(F.relu(F.linear(y, q(sd[p + ".se.fc.0.weight"]))), q(sd[p + ".se.fc.2.weight"])))
    w = sd[p + ".se.body.0.weight"]
    return F.hardsigmoid(F.conv1d(y[:, :, None], q(w), sd[p + ".se.body.0.bias"], padding=w.shape[2] // 2)[:, :, 0])


@torch.no_grad()
def fp8_block(cfg: RiseConfig, sd, i: int, h: torch.Tensor, se_f16_weights=False):
    """Block i of Precision fp8 from the f16 stream h in front of it (its SE gate included) -> the f16 stream behind it."""
    qh = lambda t: t.to(torch.float16).to(torch.float32)
    k, se = cfg.kernels[i], cfg.se_types[i]
    p = f"{cfg.key_prefix}.{i + 1}"
    if se is not None:
        h = qh(h * _se_gate(sd, p, se, h, se_f16_weights)[:, :, None, None])
    w1, b1 = _fold(sd, p + ".body.0", p + ".body.1")
    w2, b2 = _fold(sd, p + ".body.3", p + ".body.4")
    w3, b3 = _fold(sd, p + ".body.6", p + ".body.7")
    s1, s3 = row_scale_pow2(w1), row_scale_pow2(w3)
    t = F.conv2d(q_e4m3(h), q_e4m3((w1 / s1.view(-1, 1, 1, 1)).float())) + (b1 / s1).float().view(1, -1, 1, 1)
    t = qh(F.relu(t))                                                            # t1 in units of s1
    t = q_e4m3(F.relu(_depthwise_f16_chain(t, (w2 * s1.view(-1, 1, 1, 1)).float(), b2.float(), k)))
    t = F.conv2d(t, q_e4m3((w3 / s3.view(-1, 1, 1, 1)).float())) + (b3 / s3).float().view(1, -1, 1, 1)
    return qh(h + t * s3.float().view(1, -1, 1, 1))


@torch.no_grad()
def fp32_block(cfg: RiseConfig, sd, i: int, h: torch.Tensor):
    """Block i of the pinned fp32 forward from the stream h in front of it (the bottleneck branch of forward(), one block)."""
    k, se = cfg.kernels[i], cfg.se_types[i]
    p = f"{cfg.key_prefix}.{i + 1}"
    if se is not None:
        h = h * _se_gate(sd, p, se, h)[:, :, None, None]
    t = F.relu(_bn(sd, p + ".body.1", F.conv2d(h, sd[p + ".body.0.weight"])))
    t = F.relu(_bn(sd, p + ".body.4", F.conv2d(t, sd[p + ".body.3.weight"], padding=k // 2, groups=t.shape[1])))
    return h + _bn(sd, p + ".body.7", F.conv2d(t, sd[p + ".body.6.weight"]))


@torch.no_grad()
def forward_fp8_tower(cfg: RiseConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor):
    """(value, policy logits, aux) of Precision fp8 as emulated on the CPU (bottleneck-block nets only)."""
    assert not cfg.dense_blocks
    qh = lambda t: t.to(torch.float16).to(torch.float32)
    x = x.to(torch.float32)
    pre = cfg.key_prefix
    # stem as the f16 kernels run it: BN folded into the weights BEFORE they are rounded to f16 (the roundings of this mode amplify a
    # one-ulp difference of the stream into a difference of the size of the mode's own error, so the emulation follows the kernel's
    # order of operations wherever it is cheap to)
    w0, b0 = _fold(sd, pre + ".0.body.0", pre + ".0.body.1")
    h = qh(F.relu(F.conv2d(qh(x), qh(w0.float()), padding=1) + b0.float().view(1, -1, 1, 1)))
    for i in range(len(cfg.kernels)):
        h = fp8_block(cfg, sd, i, h)
    return _heads(cfg, sd, h, torch.float16)


@torch.no_grad()
def predict_fp8_tower(cfg, sd, x):
    value, pol, aux = forward_fp8_tower(cfg, sd, x)
    return value.reshape(-1), torch.softmax(pol, dim=1), aux


# --------------------------------------------------------------------------------------------------------------
# Precision int8 (crazyara_amd/csrc/nn/tower.hip, Q = 2): the calibrated INT8 mode -- the reference's third `Precision` value, TensorRT's
# entropy-calibrated INT8 (tensorrtapi.cpp:334-360, chessbatchstream.cpp:44-94).  Emulation of the mode's rounding points, NOT a restatement
# of reference code (TensorRT's kernels are not in the reference).  Precision fp8's shape with int8 in e4m3's place:
#   * the two 1x1 GEMMs of every bottleneck block on int8 operands (v_mfma_i32_32x32x32_i8), exact int32 accumulation;
#   * activations: ONE step per tensor and block, fixed by calibration (max |.| over the calibration positions -> 127 for the stream in
#     front of the block, -> 255 for the post-ReLU depthwise output, which is stored as u - 128); the quantiser is one f16 FMA
#     x * inv + 1536 (1152) -- a single rounding to the integer grid, half to even -- clamped; inv is an f16 number and the step is 1 / inv;
#   * weights: one step per output row (max |row| / 127), round half to even;
#   * the BN1 / BN3 biases enter the integer accumulators rounded to the accumulator's unit;
#   * t1 = f16(relu(acc) * 2^-7), depthwise in f16 with the dequantisation factors folded into its weights, stream and heads as float16.
# calib = [(max|x_i|, max t2_i)] per block (calibrate_int8 below, or the library's mi_net_calibrate_int8 through the float16 layer kernels).
# --------------------------------------------------------------------------------------------------------------
INT8_ESCALE = 2.0 ** -7


def _f16(t):
    return t.to(torch.float16).to(torch.float32)


def int8_steps(calib):
    """calibration maxima -> [(inv_x, inv_t)] as the kernel holds them: f16 numbers; the steps are their reciprocals"""
    out = []
    for mx, mt in calib:
        out.append((float(_f16(torch.tensor(127.0 / max(float(mx), 1e-6)))), float(_f16(torch.tensor(255.0 / max(float(mt), 1e-6))))))
    return out


@torch.no_grad()
def int8_block(cfg: RiseConfig, sd, i: int, h: torch.Tensor, inv_x: float, inv_t: float, se_f16_weights=False, collect=None):
    """Block i of Precision int8 from the f16 stream h in front of it (its SE gate included) -> the f16 stream behind it."""
    k, se = cfg.kernels[i], cfg.se_types[i]
    p = f"{cfg.key_prefix}.{i + 1}"
    if se is not None:
        h = _f16(h * _se_gate(sd, p, se, h, se_f16_weights)[:, :, None, None])
    w1, b1 = _fold(sd, p + ".body.0", p + ".body.1")
    w2, b2 = _fold(sd, p + ".body.3", p + ".body.4")
    w3, b3 = _fold(sd, p + ".body.6", p + ".body.7")
    s1 = w1.abs().amax(dim=(1, 2, 3)).clamp_min(1e-30) / 127.0
    s3 = w3.abs().amax(dim=(1, 2, 3)).clamp_min(1e-30) / 127.0
    q1 = torch.round(w1 / s1.view(-1, 1, 1, 1)).clamp(-127, 127)
    q3 = torch.round(w3 / s3.view(-1, 1, 1, 1)).clamp(-127, 127)
    if collect is not None:
        collect.append([float(h.abs().max()), 0.0])
    k1 = s1 / inv_x                                                              # value of one unit of the expand accumulator, per row
    k3 = s3 / inv_t
    qx = torch.round(h.double() * inv_x).clamp(-127, 127)                        # f16 x f16 is exact in double; one rounding, half to even
    acc = F.conv2d(qx, q1) + torch.round(b1 / k1).view(1, -1, 1, 1)
    t = _f16((F.relu(acc) * INT8_ESCALE).float())                                # t1
    t = F.relu(_depthwise_f16_chain(t, (w2 * (k1 / INT8_ESCALE).view(-1, 1, 1, 1)).float(), b2.float(), k))
    if collect is not None:
        collect[-1][1] = float(t.max())
    u = torch.round(t.double() * inv_t).clamp(0, 255)
    acc3 = F.conv2d(u, q3) + torch.round(b3 / k3).view(1, -1, 1, 1)              # (the kernel holds u - 128 and starts at + 128 * rowsum: the same integers)
    return _f16((h.double() + acc3 * k3.view(1, -1, 1, 1)).float())


@torch.no_grad()
def forward_int8_tower(cfg: RiseConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor, calib, collect=None):
    """(value, policy logits, aux) of Precision int8 as emulated on the CPU (bottleneck-block nets only)."""
    assert not cfg.dense_blocks
    x = x.to(torch.float32)
    pre = cfg.key_prefix
    w0, b0 = _fold(sd, pre + ".0.body.0", pre + ".0.body.1")
    h = _f16(F.relu(F.conv2d(_f16(x), _f16(w0.float()), padding=1) + b0.float().view(1, -1, 1, 1)))
    steps = int8_steps(calib)
    for i in range(len(cfg.kernels)):
        h = int8_block(cfg, sd, i, h, steps[i][0], steps[i][1], collect=collect)
    return _heads(cfg, sd, h, torch.float16)


@torch.no_grad()
def calibrate_int8(cfg: RiseConfig, sd, x_calib: torch.Tensor):
    """[(max |stream in front of block i| (gated), max depthwise output of block i)] of the float16 forward over the calibration positions"""
    pre = cfg.key_prefix
    w0, b0 = _fold(sd, pre + ".0.body.0", pre + ".0.body.1")
    h = _f16(F.relu(F.conv2d(_f16(x_calib.float()), _f16(w0.float()), padding=1) + b0.float().view(1, -1, 1, 1)))
    out = []
    for i, (k, se) in enumerate(zip(cfg.kernels, cfg.se_types)):
        p = f"{pre}.{i + 1}"
        if se is not None:
            h = _f16(h * _se_gate(sd, p, se, h)[:, :, None, None])
        w1, b1 = _fold(sd, p + ".body.0", p + ".body.1")
        w2, b2 = _fold(sd, p + ".body.3", p + ".body.4")
        w3, b3 = _fold(sd, p + ".body.6", p + ".body.7")
        t = _f16(F.relu(F.conv2d(h, _f16(w1.float())) + b1.float().view(1, -1, 1, 1)))
        t = _f16(F.relu(F.conv2d(t, _f16(w2.float()), padding=k // 2, groups=t.shape[1]) + b2.float().view(1, -1, 1, 1)))
        out.append((float(h.abs().max()), float(t.max())))
        h = _f16(h + F.conv2d(t, _f16(w3.float())) + b3.float().view(1, -1, 1, 1))
    return out


# --------------------------------------------------------------------------------------------------------------
# Precision float16x3 (crazyara_amd/csrc/nn/x3.hip): emulation of the mode's rounding points, NOT a restatement of reference code.
# What is pinned is the fp32 forward above; this function says what the split-operand mode adds to it:
#   every dense contraction (stem, expand / project 1x1, dense 3x3, policy convs, value conv, value FC1, flat-policy Linear):
#       BN folded into the weights in double; weights and input activations split a = hi + lo (hi = rne_f16(a), lo = rne_f16(a - hi));
#       product = hi*hi + hi*lo + lo*hi (the lo*lo term is dropped), exact accumulation here (f32 in the kernel)
#   depthwise, SE gates, last value FC / WDLP, softmax: fp32 as in Precision float32
# --------------------------------------------------------------------------------------------------------------
def _split_f16(t):
    hi = t.float().to(torch.float16).double()
    lo = (t.double() - hi).float().to(torch.float16).double()
    return hi, lo


def _x3_conv(x, w, padding=0):
    xh, xl = _split_f16(x)
    wh, wl = _split_f16(w)
    return (F.conv2d(xh, wh, padding=padding) + F.conv2d(xh, wl, padding=padding) + F.conv2d(xl, wh, padding=padding)).float()


def _x3_layer(sd, x, conv, bn, padding=0):
    if bn:
        w, b = _fold(sd, conv, bn)
        return _x3_conv(x, w, padding) + b.float().view(1, -1, 1, 1)
    return _x3_conv(x, sd[conv + ".weight"].double(), padding)


# Precision float16p8 = float16x3 whose tower launches (3x3 and 5x5 bottleneck blocks at 256 channels) run BOTH of their 1x1 contractions as
#   f16 main term   hi(a) * hi(W')                          W' = w * 2^p, p = 11 - floor(log2(max |w|)) over the layer (BN folded, double)
# + e5m2 cross term t8(hi(a)) * r8((W' - hi(W')) * c)       hi(.) = rne_f16; a = the f32 operand (residual stream / depthwise output)
# + e5m2 cross term t8(lo(a)) * r8(hi(W') * c)              lo(a) = rne_f16(a - hi(a)) (the difference is exact in f32)
# t8 = the HIGH BYTE of the f16 value (e5m2 has f16's exponent field: truncation to two mantissa bits, one byte permute in the kernel),
# r8 = e5m2 round to nearest even (host side), c = 1 / (1 - ln 2 / 8): the mean loss of the truncation taken back on the weight images
# (scripts/studies/p8_format_study.py).  All three products carry 2^p; exact accumulation here (f32 in the kernel), the sum times 2^-p in
# front of the BN bias.  The same contraction runs the policy head's dense 3x3 convs whose input channels are a multiple of 128 (conv 1 of every
# head, the policy-map conv of the select_policy_from_plane heads).  Everything else is forward_x3.
P8_TRUNC_COMPENSATION = 1.0 / (1.0 - 0.125 * 0.6931471805599453)


def _e5m2_high_byte(x16):
    b = x16.contiguous().view(torch.int16).to(torch.int32) & 0xFF00
    return torch.where(b >= 0x8000, b - 0x10000, b).to(torch.int16).view(torch.float16).double()


def _e5m2_rne(x):
    return x.float().to(torch.float8_e5m2).double()


def _p8_conv(x, w, padding=0):
    m = w.abs().max()
    e = torch.floor(torch.log2(m)) if float(m) > 0 else torch.tensor(0.0, dtype=torch.float64)
    p = 11.0 - float(e)
    W = w * (2.0 ** p)
    wh = W.float().to(torch.float16).double()
    xh16 = x.float().to(torch.float16)
    xh = xh16.double()
    xl16 = (x.double() - xh).float().to(torch.float16)      # the difference is exact in f32
    main = F.conv2d(xh, wh, padding=padding)
    c1 = F.conv2d(_e5m2_high_byte(xh16), _e5m2_rne((W - wh) * P8_TRUNC_COMPENSATION), padding=padding)
    c2 = F.conv2d(_e5m2_high_byte(xl16), _e5m2_rne(wh * P8_TRUNC_COMPENSATION), padding=padding)
    return ((main + c1 + c2) * (2.0 ** -p)).float()


def _p8_layer(sd, x, conv, bn, padding=0):
    """_x3_layer with the contraction of Precision float16p8 (the policy head's dense 3x3 convs: x3.hip conv3x3_p8_kernel)"""
    if bn:
        w, b = _fold(sd, conv, bn)
        return _p8_conv(x, w, padding) + b.float().view(1, -1, 1, 1)
    return _p8_conv(x, sd[conv + ".weight"].double(), padding)


@torch.no_grad()
def forward_p8(cfg: RiseConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor):
    """(value, policy logits, aux) of Precision float16p8 as emulated on the CPU."""
    return forward_x3(cfg, sd, x, p8=True)


@torch.no_grad()
def forward_x3(cfg: RiseConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor, p8: bool = False):
    """(value, policy logits, aux) of Precision float16x3 as emulated on the CPU (every net family).  p8: see forward_p8."""
    x = x.to(torch.float32)
    pre = cfg.key_prefix
    h = F.relu(_x3_layer(sd, x, pre + ".0.body.0", pre + ".0.body.1", 1))
    for i, (k, se) in enumerate(zip(cfg.kernels, cfg.se_types)):
        p = f"{pre}.{i + 1}"

        def gate(z, hard):
            y = z.mean(dim=(2, 3))
            if se in ("ca_se", "se"):
                y = F.linear(F.relu(F.linear(y, sd[p + ".se.fc.0.weight"])), sd[p + ".se.fc.2.weight"])
            else:
                w = sd[p + ".se.body.0.weight"]
                y = F.conv1d(y[:, :, None], w, sd[p + ".se.body.0.bias"], padding=w.shape[2] // 2)[:, :, 0]
            y = F.hardsigmoid(y) if hard else torch.sigmoid(y)
            return z * y[:, :, None, None]
        if cfg.dense_blocks:
            if se is not None and cfg.conv_block == "classical_res_block":
                h = gate(h, True)
            t = F.relu(_x3_layer(sd, h, p + ".body.0", p + ".body.1", 1))
            t = _x3_layer(sd, t, p + ".body.3", p + ".body.4", 1)
            if cfg.conv_block == "classical_res_block":
                h = h + F.relu(t)
            else:
                if se is not None:
                    t = gate(t, False)
                h = F.relu(h + t)
            continue
        if se is not None:
            h = gate(h, True)
        in_p8_tower = p8 and k in (3, 5) and h.shape[1] == 256      # a block of the tower launches (x3.hip: tower_p8_kernel<3 | 5>)
        if in_p8_tower:
            w1, b1 = _fold(sd, p + ".body.0", p + ".body.1")
            t = F.relu(_p8_conv(h, w1) + b1.float().view(1, -1, 1, 1))
        else:
            t = F.relu(_x3_layer(sd, h, p + ".body.0", p + ".body.1"))
        cop = t.shape[1]
        t = F.relu(_bn(sd, p + ".body.4", F.conv2d(t, sd[p + ".body.3.weight"], padding=k // 2, groups=cop)))
        if in_p8_tower:
            w3, b3 = _fold(sd, p + ".body.6", p + ".body.7")
            h = h + (_p8_conv(t, w3) + b3.float().view(1, -1, 1, 1))
        else:
            h = h + _x3_layer(sd, t, p + ".body.6", p + ".body.7")
    B = x.shape[0]
    head_layer = _p8_layer if p8 and h.shape[1] % 128 == 0 else _x3_layer
    ph = F.relu(head_layer(sd, h, "policy_head.body.0", "policy_head.body.1", 1))
    if cfg.select_policy_from_plane:
        pol = head_layer(sd, ph, "policy_head.body.3", "", 1).reshape(B, -1)
    else:
        pol = F.relu(_x3_layer(sd, ph, "policy_head.body.3", "policy_head.body2.0", 1)).reshape(B, -1)
        pol = _x3_conv(pol[:, :, None, None], sd["policy_head.body3.0.weight"].double()[:, :, None, None]).reshape(B, -1) \
            + sd["policy_head.body3.0.bias"]
    vh = F.relu(_x3_layer(sd, h, "value_head.body.0", "value_head.body.1")).reshape(B, -1)
    aux = None
    if cfg.use_wdl and cfg.use_plys_to_end:
        wdl = F.linear(vh, sd["value_head.body_wdl.0.weight"], sd["value_head.body_wdl.0.bias"])
        plys = torch.sigmoid(F.linear(vh, sd["value_head.body_plys.0.weight"], sd["value_head.body_plys.0.bias"]))
        sm = torch.softmax(wdl, dim=1)
        value = -sm[:, 0:1] + sm[:, 2:3]
        aux = torch.cat((wdl, plys), dim=1)
    else:
        v = _x3_conv(vh[:, :, None, None], sd["value_head.body_final.0.weight"].double()[:, :, None, None]).reshape(B, -1)
        v = F.relu(v + sd["value_head.body_final.0.bias"])
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
