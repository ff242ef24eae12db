"""CPU study (no GPU): which 8-bit format can carry the cross terms of Precision float16p8, and in which of the tower's two GEMMs?

The e4m3 form (oracle.forward_p8) needs v_cvt_scalef32_pk_fp8_* conversions, which are slow on gfx950 (profiles/NOTES.md, round 4).  e5m2
("bf8") has f16's exponent: the bf8 image of an f16 value is its HIGH BYTE -- one v_perm_b32 per four values, no conversion instruction
(truncation), or one v_pk_add_u16 in front of it (round half up).  This script measures the max logit / value error against the fp32 oracle
for: e4m3 in the expand GEMM (the shipped definition), bf8 in the expand GEMM, bf8 in both GEMMs; activations truncated / rounded half up /
rounded to nearest even / truncated with the mean loss taken back on the weights (weights are converted on the host: always nearest even).

usage: python scripts/studies/p8_format_study.py [net ...]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nn_cases  # noqa: E402
from oracle import rise_oracle as ro  # noqa: E402


def bf8_of_f16(x16, mode):
    """the e5m2 image of f16 values (returned as f64): high byte of the f16, after truncation / + half an ulp / round to nearest even"""
    b = x16.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    if mode == "trunc":
        b = b & 0xFF00
    elif mode == "half_up":
        b = (b + 0x80) & 0xFF00
    else:
        b = (b + 0x7F + ((b >> 8) & 1)) & 0xFF00
    b = torch.where(b >= 0x8000, b - 0x10000, b).to(torch.int16)
    return b.view(torch.float16).double()


def scale_of(w):
    m = w.abs().max()
    e = torch.floor(torch.log2(m)) if float(m) > 0 else torch.tensor(0.0, dtype=torch.float64)
    return 11.0 - float(e)


TRUNC_COMPENSATION = 1.0 / (1.0 - 0.125 * 0.6931471805599453)      # truncation to two mantissa bits loses 2^e / 8 on average; E[1.f] = 1 / ln 2


def conv_bf8(x, w, mode):
    comp = 1.0
    if mode == "trunc_comp":                                          # the bias of truncated activations taken back on the (host-made) weight images
        mode, comp = "trunc", TRUNC_COMPENSATION
    p = scale_of(w)
    W = w * (2.0 ** p)
    wh16 = W.float().to(torch.float16)
    wh = wh16.double()
    wl16 = (W - wh).float().to(torch.float16)
    xh16 = x.float().to(torch.float16)
    xh = xh16.double()
    xl16 = (x.double() - xh).float().to(torch.float16)
    main = F.conv2d(xh, wh)
    c1 = F.conv2d(bf8_of_f16(xh16, mode), bf8_of_f16(((W - wh) * comp).float().to(torch.float16), "rne"))
    c2 = F.conv2d(bf8_of_f16(xl16, mode), bf8_of_f16((wh * comp).float().to(torch.float16), "rne"))
    return ((main + c1 + c2) * (2.0 ** -p)).float()


@torch.no_grad()
def forward_variant(cfg, sd, x, expand, project):
    """forward_x3 with the tower's expand / project contraction replaced: each of None (float16x3), 'e4m3', ('bf8', mode)"""
    x = x.to(torch.float32)
    pre = cfg.key_prefix
    h = F.relu(ro._x3_layer(sd, x, pre + ".0.body.0", pre + ".0.body.1", 1))

    def contraction(kind, a, w):
        if kind is None:
            return ro._x3_conv(a, w)
        if kind == "e4m3":
            return ro._p8_conv(a, w)
        return conv_bf8(a, w, kind[1])
    for i, (k, se) in enumerate(zip(cfg.kernels, cfg.se_types)):
        p = f"{pre}.{i + 1}"
        if se is not None:
            y = h.mean(dim=(2, 3))
            if se in ("ca_se", "se"):
                y = F.linear(F.relu(F.linear(y, sd[p + ".se.fc.0.weight"])), sd[p + ".se.fc.2.weight"])
            else:
                w = sd[p + ".se.body.0.weight"]
                y = F.conv1d(y[:, :, None], w, sd[p + ".se.body.0.bias"], padding=w.shape[2] // 2)[:, :, 0]
            h = h * F.hardsigmoid(y)[:, :, None, None]
        in_tower = k == 3 and h.shape[1] == 256
        w1, b1 = ro._fold(sd, p + ".body.0", p + ".body.1")
        t = F.relu(contraction(expand if in_tower else None, h, w1) + b1.float().view(1, -1, 1, 1))
        cop = t.shape[1]
        t = F.relu(ro._bn(sd, p + ".body.4", F.conv2d(t, sd[p + ".body.3.weight"], padding=k // 2, groups=cop)))
        w3, b3 = ro._fold(sd, p + ".body.6", p + ".body.7")
        h = h + contraction(project if in_tower else None, t, w3) + b3.float().view(1, -1, 1, 1)
    B = x.shape[0]
    ph = F.relu(ro._x3_layer(sd, h, "policy_head.body.0", "policy_head.body.1", 1))
    if cfg.select_policy_from_plane:
        pol = ro._x3_layer(sd, ph, "policy_head.body.3", "", 1).reshape(B, -1)
    else:
        pol = F.relu(ro._x3_layer(sd, ph, "policy_head.body.3", "policy_head.body2.0", 1)).reshape(B, -1)
        pol = ro._x3_conv(pol[:, :, None, None], sd["policy_head.body3.0.weight"].double()[:, :, None, None]).reshape(B, -1) \
            + sd["policy_head.body3.0.bias"]
    vh = F.relu(ro._x3_layer(sd, h, "value_head.body.0", "value_head.body.1")).reshape(B, -1)
    if cfg.use_wdl and cfg.use_plys_to_end:
        wdl = F.linear(vh, sd["value_head.body_wdl.0.weight"], sd["value_head.body_wdl.0.bias"])
        sm = torch.softmax(wdl, dim=1)
        value = -sm[:, 0:1] + sm[:, 2:3]
    else:
        v = ro._x3_conv(vh[:, :, None, None], sd["value_head.body_final.0.weight"].double()[:, :, None, None]).reshape(B, -1)
        v = F.relu(v + sd["value_head.body_final.0.bias"])
        value = torch.tanh(F.linear(v, sd["value_head.body_final.2.weight"], sd["value_head.body_final.2.bias"]))
    return value, pol


def main():
    names = sys.argv[1:] or ["risev2-3", "risev2-7", "risev2-19", "risev2-13-lichess"]
    variants = [("float16x3", None, None), ("e4m3 expand (shipped float16p8)", "e4m3", None), ("e4m3 both", "e4m3", "e4m3")]
    for mode in ("rne", "half_up", "trunc", "trunc_comp"):
        variants.append((f"bf8 {mode} expand", ("bf8", mode), None))
        variants.append((f"bf8 {mode} both", ("bf8", mode), ("bf8", mode)))
    print(f"{'variant':38s}" + "".join(f"{n:>26s}" for n in names) + "      (max |logit err|, max |value err| vs the fp32 oracle)")
    rows = {v[0]: [] for v in variants}
    for n in names:
        cfg, sd, x = nn_cases.make_case(n)
        if any(k != 3 for k in cfg.kernels) or cfg.dense_blocks:
            print(f"# {n}: not a 3x3 bottleneck tower, skipped")
        v32, l32, _ = ro.forward(cfg, sd, x)
        for name, e, p in variants:
            v, l = forward_variant(cfg, sd, x, e, p)
            rows[name].append((float((l - l32).abs().max()), float((v - v32).abs().max())))
    for name, _, _ in variants:
        print(f"{name:38s}" + "".join(f"{a:14.2e}{b:12.2e}" for a, b in rows[name]))


if __name__ == "__main__":
    main()
