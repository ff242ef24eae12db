"""Development: the merged-roles tower (towerm.hip, Precision "float16-m3k") against the 8-wave tower ("float16-3k"): outputs and time."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nn_cases
from crazyara_amd import rise_config as ro
from crazyara_amd.neuralnetapi import HipAPI
from crazyara_amd import build
build.build()
for nblk, B in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or ((3, 8), (7, 37), (19, 256), (19, 512)):
    cfg = ro.rise_v2_config(nblk)
    sd = ro.make_state_dict(cfg, seed=1, stress=True)
    tmp = tempfile.mkdtemp()
    d = nn_cases.export_case(tmp, "b", cfg, sd)
    x = nn_cases.synthetic_planes(B, 34, 5)
    res = {}
    for prec in ("float16-3k", "float16-m3k", "float16"):
        net = HipAPI(0, B, d, prec)
        value = np.zeros(B, np.float32); probs = np.zeros(B * cfg.nb_policy, np.float32)
        net.predict(x.numpy().reshape(-1), value, probs)
        torch.as_tensor(net.device_buffers()["planes"], device="cuda").copy_(x.cuda()); torch.cuda.synchronize()
        net.time_forward(5)
        ms = net.time_forward(30) / 30
        ops = net.time_ops(5)
        agg = {}
        for n, t in ops: agg[n] = agg.get(n, 0) + t
        res[prec] = (value.copy(), probs.copy())
        print(f"RISEv2-{nblk} B={B} {prec}: {ms:.4f} ms/forward  per-op {({k: round(v, 4) for k, v in agg.items()})}", flush=True)
        net.close()
    a, b = res["float16-3k"], res["float16-m3k"]
    print(f"   merged vs 8-wave: |dvalue| {np.abs(a[0]-b[0]).max():.3e}  |dprobs| {np.abs(a[1]-b[1]).max():.3e}  nan {np.isnan(b[1]).any()}", flush=True)
