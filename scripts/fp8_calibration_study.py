"""Would the reference's INT8-style calibration (tensorrtapi.cpp:334-360: per-tensor activation ranges from the 232 calibration plies of
chessbatchstream.cpp:44-94) help Precision fp8?  CPU study on the oracle's emulation of the mode (oracle/rise_oracle.fp8_block): per block a
power-of-two activation scale sa for the e4m3 copy of the stream (and sb for the depthwise output), chosen from the calibration
positions so that the largest calibration value lands at 2^7 (of e4m3's 448), applied to OTHER positions.  Prints the error against
fp32 with and without the scales.  Test infrastructure only (imports oracle/)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from crazyara_amd import env, openings  # noqa: E402
from oracle import rise_oracle as ro  # noqa: E402


def planes_of(fens):
    return torch.from_numpy(np.stack([env.Position(f, False, "crazyhouse").planes(0, 1, True) for f in fens]).astype(np.float32))


@torch.no_grad()
def fp8_forward(cfg, sd, x, sa=None, sb=None, collect=None):
    """forward_fp8_tower with optional per-block power-of-two activation scales; collect: dict that receives max|stream|, max|t2| per block"""
    qh = lambda t: t.to(torch.float16).to(torch.float32)
    pre = cfg.key_prefix
    w0, b0 = ro._fold(sd, pre + ".0.body.0", pre + ".0.body.1")
    h = qh(F.relu(F.conv2d(qh(x), qh(w0.float()), padding=1) + b0.float().view(1, -1, 1, 1)))
    for i, (k, se) in enumerate(zip(cfg.kernels, cfg.se_types)):
        p = f"{pre}.{i + 1}"
        if se is not None:
            h = qh(h * ro._se_gate(sd, p, se, h)[:, :, None, None])
        w1, b1 = ro._fold(sd, p + ".body.0", p + ".body.1")
        w2, b2 = ro._fold(sd, p + ".body.3", p + ".body.4")
        w3, b3 = ro._fold(sd, p + ".body.6", p + ".body.7")
        s1, s3 = ro.row_scale_pow2(w1), ro.row_scale_pow2(w3)
        a = 1.0 if sa is None else sa[i]
        bb = 1.0 if sb is None else sb[i]
        t = F.conv2d(ro.q_e4m3(h * a), ro.q_e4m3((w1 / s1.view(-1, 1, 1, 1)).float())) / a + (b1 / s1).float().view(1, -1, 1, 1)
        t = qh(F.relu(t))
        t = F.relu(ro._depthwise_f16_chain(t, (w2 * s1.view(-1, 1, 1, 1)).float(), b2.float(), k))
        if collect is not None:
            collect.setdefault("x", []).append(float(h.abs().max()))
            collect.setdefault("t2", []).append(float(t.abs().max()))
        t = ro.q_e4m3(t * bb) / bb
        t = F.conv2d(t, ro.q_e4m3((w3 / s3.view(-1, 1, 1, 1)).float())) + (b3 / s3).float().view(1, -1, 1, 1)
        h = qh(h + t * s3.float().view(1, -1, 1, 1))
    return ro._heads(cfg, sd, h, torch.float16)


def main():
    fens = openings.position_fens("crazyhouse")
    calib, test = fens[0::2], fens[1::2][:64]
    for nblocks, seed in ((19, 14), (7, 12)):
        cfg = ro.rise_v2_config(nblocks, 34, 81)
        sd = ro.make_state_dict(cfg, seed=seed, stress=True)
        col = {}
        fp8_forward(cfg, sd, planes_of(calib), collect=col)
        pow2 = lambda m: 2.0 ** np.floor(np.log2(128.0 / max(m, 1e-9)))
        sa = [pow2(m) for m in col["x"]]
        sb = [pow2(m) for m in col["t2"]]
        xt = planes_of(test)
        v32, l32, _ = ro.forward(cfg, sd, xt)
        rows = []
        for name, a, b in (("scale 1 (the product)", None, None), ("stream scale per block", sa, None), ("stream + depthwise-output scales", sa, sb)):
            v, l, _ = fp8_forward(cfg, sd, xt, a, b)
            rows.append((name, float((v - v32).abs().max()), float((l - l32).abs().max()),
                         float((torch.softmax(l, 1) - torch.softmax(l32, 1)).abs().max())))
        print(f"RISEv2-{nblocks}: {len(calib)} calibration positions, {len(test)} test positions; max|stream| per block {min(col['x']):.2f} .. {max(col['x']):.2f}, "
              f"max|t2| {min(col['t2']):.2f} .. {max(col['t2']):.2f}; scales sa 2^{int(np.log2(min(sa)))} .. 2^{int(np.log2(max(sa)))}")
        for name, ev, el, ep in rows:
            print(f"   {name:36s} |value| {ev:.3e}   |logit| {el:.3e}   |prob| {ep:.3e}")


if __name__ == "__main__":
    main()
