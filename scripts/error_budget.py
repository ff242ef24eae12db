"""Per-site rounding-error budget of Precision float16 for a RISE bottleneck net (CPU, float64 emulation).

Every place where the HIP path rounds to f16 is a named *site*; the script runs the network in float64 with one site (or a set)
switched on and reports the maximum error of the policy logits / value against the all-off run.  BN is folded into the conv
weights before rounding, as the loader does (csrc/nn/rise_net.hip).  Used to decide which rounding points are worth moving
(DESIGN 4.2); it is a design tool, not a test.

  python scripts/error_budget.py [--net risev2-19] [--boards 64] [--seed 14]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from crazyara_amd import rise_config  # noqa: E402

SITES = ["w_stem", "w_expand", "w_dw", "w_project", "w_se", "w_policy1", "w_policy2", "w_value",
         "a_planes", "a_stem", "a_stream", "a_se_scaled", "a_b_operand", "a_t1", "a_t2", "dw_acc_f16", "a_policy_mid", "a_head_in"]

D = torch.float64


def q16(t):
    return t.to(torch.float16).to(D)


def fold(sd, conv, bn, eps=1e-5):
    w = sd[conv + ".weight"].to(D)
    g, b = sd[bn + ".weight"].to(D), sd[bn + ".bias"].to(D)
    m, v = sd[bn + ".running_mean"].to(D), sd[bn + ".running_var"].to(D)
    s = g / torch.sqrt(v + eps)
    return w * s[:, None, None, None], b - m * s


def dw_f16_accumulate(t, w, bias, k):
    """depthwise conv accumulating tap by tap in f16 (v_pk_fma_f16): acc = fma(x, w, acc) rounded to f16 each step"""
    B, C, H, Wd = t.shape
    pad = k // 2
    tp = F.pad(t, (pad, pad, pad, pad))
    acc = q16(bias)[None, :, None, None].expand(B, C, H, Wd).clone()
    for dy in range(k):
        for dx in range(k):
            acc = q16(acc + tp[:, :, dy:dy + H, dx:dx + Wd] * w[None, :, 0, dy, dx, None, None])
    return acc


@torch.no_grad()
def emu(cfg, sd, x, on):
    on = set(on)
    R = lambda site, t: q16(t) if site in on else t  # noqa: E731
    pre = cfg.key_prefix
    x = R("a_planes", x.to(D))
    w, b = fold(sd, pre + ".0.body.0", pre + ".0.body.1")
    h = F.relu(F.conv2d(x, R("w_stem", w), b, padding=1))
    h = R("a_stem", h)
    for i, (k, se) in enumerate(zip(cfg.kernels, cfg.se_types)):
        p = f"{pre}.{i + 1}"
        if se in ("ca_se", "se"):
            y = h.mean(dim=(2, 3))
            y = F.relu(F.linear(y, R("w_se", sd[p + ".se.fc.0.weight"].to(D))))
            y = F.hardsigmoid(F.linear(y, R("w_se", sd[p + ".se.fc.2.weight"].to(D))))
            h = R("a_se_scaled", h * y[:, :, None, None])
        elif se == "eca_se":
            y = h.mean(dim=(2, 3))
            wse = sd[p + ".se.body.0.weight"].to(D)
            y = F.conv1d(y[:, :, None], R("w_se", wse), sd[p + ".se.body.0.bias"].to(D), padding=wse.shape[2] // 2)[:, :, 0]
            h = R("a_se_scaled", h * F.hardsigmoid(y)[:, :, None, None])
        w1, b1 = fold(sd, p + ".body.0", p + ".body.1")
        t = F.relu(F.conv2d(R("a_b_operand", h), R("w_expand", w1), b1))
        t = R("a_t1", t)
        w2, b2 = fold(sd, p + ".body.3", p + ".body.4")
        cop = t.shape[1]
        if "dw_acc_f16" in on:
            t = F.relu(dw_f16_accumulate(t, R("w_dw", w2), b2, k))
        else:
            t = F.relu(F.conv2d(t, R("w_dw", w2), b2, padding=k // 2, groups=cop))
        t = R("a_t2", t)
        w3, b3 = fold(sd, p + ".body.6", p + ".body.7")
        t = F.conv2d(t, R("w_project", w3), b3)
        h = R("a_stream", h + t)
    hh = R("a_head_in", h)
    wp, bp = fold(sd, "policy_head.body.0", "policy_head.body.1")
    ph = R("a_policy_mid", F.relu(F.conv2d(hh, R("w_policy1", wp), bp, padding=1)))
    pol = F.conv2d(ph, R("w_policy2", sd["policy_head.body.3.weight"].to(D)), padding=1).reshape(x.shape[0], -1)
    wv, bv = fold(sd, "value_head.body.0", "value_head.body.1")
    vh = F.relu(F.conv2d(hh, R("w_value", wv), bv)).reshape(x.shape[0], -1)
    v = F.relu(F.linear(vh, sd["value_head.body_final.0.weight"].to(D), sd["value_head.body_final.0.bias"].to(D)))
    value = torch.tanh(F.linear(v, sd["value_head.body_final.2.weight"].to(D), sd["value_head.body_final.2.bias"].to(D)))
    return value.reshape(-1), pol, h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="risev2-19")
    ap.add_argument("--boards", type=int, default=64)
    ap.add_argument("--seed", type=int, default=14)
    args = ap.parse_args()
    import nn_cases
    n = int(args.net.split("-")[1])
    cfg = rise_config.rise_v2_config(n, 34, 81)
    sd = rise_config.make_state_dict(cfg, seed=args.seed, stress=True)
    x = nn_cases.synthetic_planes(args.boards, 34, args.seed + 1000)
    v0, p0, h0 = emu(cfg, sd, x, ())
    print(f"{args.net}, {args.boards} boards: max|logit| {p0.abs().max():.3f}  rms logit {p0.pow(2).mean().sqrt():.3f}  "
          f"rms stream {h0.pow(2).mean().sqrt():.3f}")

    def report(label, on):
        v, p, h = emu(cfg, sd, x, on)
        print(f"  {label:<58s} logit max {float((p - p0).abs().max()):.2e} rms {float((p - p0).pow(2).mean().sqrt()):.2e}   "
              f"value max {float((v - v0).abs().max()):.2e}   stream rms {float((h - h0).pow(2).mean().sqrt()):.2e}")

    for s in SITES:
        if s == "a_b_operand":
            continue
        report(s, [s])
    product = [s for s in SITES if s != "a_b_operand"]
    report("ALL (the product path today)", product)
    report("all weights only", [s for s in SITES if s.startswith("w_")])
    report("all activations only", [s for s in product if not s.startswith("w_")])
    report("today minus dw_acc_f16", [s for s in product if s != "dw_acc_f16"])
    f32stream = [s for s in product if s not in ("a_stream", "a_se_scaled", "a_stem")] + ["a_b_operand"]
    report("f32 residual stream (operand rounding kept)", f32stream)
    report("f32 stream, f32 dw accumulate", [s for s in f32stream if s != "dw_acc_f16"])
    report("f32 stream, f32 dw acc, un-rounded policy mid", [s for s in f32stream if s not in ("dw_acc_f16", "a_policy_mid")])
    report("tower exact, heads f16", ["w_policy1", "w_policy2", "w_value", "a_policy_mid", "a_head_in"])
    report("heads exact, tower f16", [s for s in product if s not in ("w_policy1", "w_policy2", "w_value", "a_policy_mid", "a_head_in")])


if __name__ == "__main__":
    main()
