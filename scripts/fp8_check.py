"""Development (GPU): Precision fp8 against its CPU emulation (oracle.forward_fp8_tower) and against fp32; timing beside float16."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nn_cases
from crazyara_amd import rise_config as rc
from crazyara_amd.neuralnetapi import HipAPI
from crazyara_amd import build
from oracle import rise_oracle as ro
build.build()
cases = [tuple(a.split("x")) for a in sys.argv[1:]] or [("3", "8"), ("7", "37"), ("19", "256")]
for case in cases:
    nblk, B = int(case[0]), int(case[1])
    cfg = rc.rise_v2_config(nblk)
    sd = rc.make_state_dict(cfg, seed=1, stress=True)
    if len(case) > 2 and case[2] == "id":
        # diagnostic: depthwise = identity (centre tap 1, BN2 = identity), so that the only arithmetic between the two e4m3 roundings of
        # a block is exact: kernel and emulation then differ by f32 summation order alone
        for i, k in enumerate(cfg.kernels):
            p = f"{cfg.key_prefix}.{i + 1}"
            w = torch.zeros_like(sd[p + ".body.3.weight"]); w[:, 0, k // 2, k // 2] = 1.0
            sd[p + ".body.3.weight"] = w
            sd[p + ".body.4.weight"] = torch.ones_like(sd[p + ".body.4.weight"]) * float(np.sqrt(1.0 + ro.BN_EPS))
            sd[p + ".body.4.bias"] = torch.zeros_like(sd[p + ".body.4.bias"])
            sd[p + ".body.4.running_mean"] = torch.zeros_like(sd[p + ".body.4.running_mean"])
            sd[p + ".body.4.running_var"] = torch.ones_like(sd[p + ".body.4.running_var"])
    tmp = tempfile.mkdtemp()
    d = nn_cases.export_case(tmp, "b", cfg, sd)
    x = nn_cases.synthetic_planes(B, 34, 5)
    nref = min(B, 16)
    v32, p32, _ = ro.predict(cfg, sd, x[:nref])
    v8, p8, _ = ro.predict_fp8_tower(cfg, sd, x[:nref])
    for prec in ("float16", "fp8-3k", "fp8"):
        net = HipAPI(0, B, d, prec)
        value = np.zeros(B, np.float32); probs = np.zeros(B * cfg.nb_policy, np.float32)
        net.predict(x.numpy().reshape(-1), value, probs)
        pr = probs.reshape(B, -1)[:nref]
        torch.as_tensor(net.device_buffers()["planes"], device="cuda").copy_(x.cuda()); torch.cuda.synchronize()
        net.time_forward(5)
        ms = net.time_forward(30) / 30
        ops = net.time_ops(5)
        agg = {}
        for n, t in ops: agg[n] = agg.get(n, 0) + t
        print(f"RISEv2-{nblk} B={B} {prec}: {ms:.4f} ms/forward {B / ms * 1e3:.0f} evals/s per-op {({k: round(v, 4) for k, v in agg.items()})}\n"
              f"      vs fp32: |dvalue| {np.abs(value[:nref] - v32.numpy()).max():.2e} |dprob| {np.abs(pr - p32.numpy()).max():.2e}   "
              f"vs fp8 emulation: |dvalue| {np.abs(value[:nref] - v8.numpy()).max():.2e} |dprob| {np.abs(pr - p8.numpy()).max():.2e}  "
              f"(emulation vs fp32: {(v8 - v32).abs().max():.2e} / {(p8 - p32).abs().max():.2e})  nan {np.isnan(probs).any()}", flush=True)
        net.close()
