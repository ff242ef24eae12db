"""Scratch: time the forward of a RISEv2-N net at a batch size (device-resident)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nn_cases
from crazyara_amd import rise_config as ro
from crazyara_amd.neuralnetapi import HipAPI
from crazyara_amd import build
build.build()
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 19
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = ro.rise_v2_config(nblk)
sd = ro.make_state_dict(cfg, seed=1)
tmp = tempfile.mkdtemp()
d = nn_cases.export_case(tmp, "b", cfg, sd)
for prec in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("float16", "float16-perblock", "float16-unfused", "float32")):
    net = HipAPI(0, B, d, prec)
    x = nn_cases.synthetic_planes(B, 34, 5)
    torch.as_tensor(net.device_buffers()["planes"], device="cuda").copy_(x.cuda()); torch.cuda.synchronize()
    net.time_forward(5)
    ms = net.time_forward(20) / 20
    fl = net.flops_per_position() * B
    print(f"{prec} B={B} blocks={nblk}: {ms:.3f} ms/forward  {B/ms*1e3:.0f} evals/s  {fl/ms/1e9:.1f} TFLOP/s")
    ops = net.time_ops(3)
    agg = {}
    for n, t in ops: agg[n] = agg.get(n, 0) + t
    print("   per-op ms:", {k: round(v, 3) for k, v in agg.items()}, "sum", round(sum(agg.values()), 3))
    net.close()
