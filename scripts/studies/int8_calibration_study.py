"""Round 6: would a CALIBRATED INT8 mode -- the reference's third `Precision` value, TensorRT's
entropy-calibrated INT8 (tensorrtapi.cpp:334-360, calibration batches from the plies of chessbatchstream.cpp:44-94) -- be worth building on
gfx950's v_mfma_i32_32x32x32_i8?  Round 3's study that rejected calibration was made on the e4m3 emulation (three mantissa bits wherever a
scale puts the values); this one emulates int8 itself.

The mode emulated (the shape of Precision fp8, with int8 in its place): the two 1x1 GEMMs of every bottleneck block on int8 operands --
activations with ONE scale per tensor and block fixed from the calibration positions (`max`: the largest calibration magnitude -> 127;
`p9999`: the 99.99th percentile, the usual stand-in for TensorRT's entropy calibrator, values beyond it saturate; `dynamic`: the scale of
the batch itself, an upper bound no calibrated engine reaches), weights with one scale per output row (max|row| -> 127), exact int32
accumulation, dequantised to f16; the post-ReLU depthwise output optionally as UNSIGNED 8 bit (zero point folded into the bias: one more
bit); stem, depthwise, SE gates, heads in float16 as in Precision float16 / fp8.  Calibration = every second position of the opening set and
the two calibration games (crazyara_amd/openings.py), test = 64 of the others.  Prints |value|, |logit|, |prob| error against the fp32
oracle next to Precision float16's and Precision fp8's own emulations on the same positions.
Test infrastructure only (imports oracle/)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from crazyara_amd import env, openings  # noqa: E402
from oracle import rise_oracle as ro  # noqa: E402


def planes_of(fens):
    return torch.from_numpy(np.stack([env.Position(f, False, "crazyhouse").planes(0, 1, True) for f in fens]).astype(np.float32))


def qh(t):
    return t.to(torch.float16).to(torch.float32)


def quant(t, scale, unsigned=False):
    """int8 image of t at `scale` (value of one step) as exact doubles; unsigned: 0 ... 255"""
    q = torch.round(t.double() / scale)
    return q.clamp(0, 255) if unsigned else q.clamp(-127, 127)


@torch.no_grad()
def int8_forward(cfg, sd, x, sx=None, st=None, unsigned_t2=False, collect=None):
    """sx[i] / st[i]: activation steps of block i's stream / depthwise output (None: the batch's own maximum = dynamic)"""
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
        s1 = w1.abs().amax(dim=(1, 2, 3)).clamp_min(1e-12) / 127.0          # per expand row
        s3 = w3.abs().amax(dim=(1, 2, 3)).clamp_min(1e-12) / 127.0          # per cout
        q1 = torch.round(w1.double() / s1.view(-1, 1, 1, 1)).clamp(-127, 127)
        q3 = torch.round(w3.double() / s3.view(-1, 1, 1, 1)).clamp(-127, 127)
        a = float(h.abs().max()) / 127.0 if sx is None else sx[i]
        if collect is not None:
            collect.setdefault("x", []).append(h.abs().flatten())
        t = F.conv2d(quant(h, a), q1) * (a * s1.double()).view(1, -1, 1, 1) + b1.double().view(1, -1, 1, 1)
        t = qh(F.relu(t.float()))
        t = F.relu(ro._depthwise_f16_chain(t, w2.float(), b2.float(), k))
        if collect is not None:
            collect.setdefault("t2", []).append(t.abs().flatten())
        top = 255.0 if unsigned_t2 else 127.0
        bb = float(t.abs().max()) / top if st is None else st[i] * (127.0 / top)
        t = F.conv2d(quant(t, bb, unsigned_t2), q3) * (bb * s3.double()).view(1, -1, 1, 1) + b3.double().view(1, -1, 1, 1)
        h = qh(h + t.float())
    return ro._heads(cfg, sd, h, torch.float16)


def main():
    fens = openings.position_fens("crazyhouse")
    calib, test = fens[0::2], fens[1::2][:64]
    for nblocks, seed in ((19, 14), (7, 12), (13, 13)):
        cfg = ro.rise_v2_config(nblocks, 34, 81)
        sd = ro.make_state_dict(cfg, seed=seed, stress=True)
        col = {}
        int8_forward(cfg, sd, planes_of(calib), collect=col)
        steps = {}
        for name, fn in (("max", lambda v: float(v.max())), ("p9999", lambda v: float(torch.quantile(v[torch.randperm(v.numel())[:2_000_000]].double(), 0.9999)))):
            steps[name] = ([max(fn(v), 1e-9) / 127.0 for v in col["x"]], [max(fn(v), 1e-9) / 127.0 for v in col["t2"]])
        xt = planes_of(test)
        v32, l32, _ = ro.forward(cfg, sd, xt)
        p32 = torch.softmax(l32, 1)

        def row(name, out):
            v, l, _ = out
            return (name, float((v - v32).abs().max()), float((v - v32).abs().mean()), float((l - l32).abs().max()), float((torch.softmax(l, 1) - p32).abs().max()),
                    float((torch.softmax(l, 1).argmax(1) == p32.argmax(1)).float().mean()))
        rows = [row("Precision float16 (emulated)", ro.forward(cfg, sd, xt, sim_dtype=torch.float16)),
                row("Precision fp8 = e4m3 (emulated)", ro.forward_fp8_tower(cfg, sd, xt))]
        for cal in ("max", "p9999"):
            rows.append(row(f"int8, {cal} calibration, signed", int8_forward(cfg, sd, xt, *steps[cal])))
            rows.append(row(f"int8, {cal} calibration, unsigned t2", int8_forward(cfg, sd, xt, *steps[cal], unsigned_t2=True)))
        rows.append(row("int8, dynamic per-batch scales, unsigned t2", int8_forward(cfg, sd, xt, unsigned_t2=True)))
        print(f"RISEv2-{nblocks} (stress-scaled random init): {len(calib)} calibration positions, {len(test)} test positions")
        print(f"   {'mode':44s} |value| max   mean      |logit| max  |prob| max   same best move")
        for name, ev, em, el, ep, top in rows:
            print(f"   {name:44s} {ev:.3e}  {em:.3e}  {el:.3e}    {ep:.3e}    {top:.3f}")


if __name__ == "__main__":
    main()
