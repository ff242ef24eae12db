"""Scratch: time the forward of a dense-residual-tower net (classical / alphazero) at a batch size."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nn_cases
from crazyara_amd import rise_config as ro
from crazyara_amd.neuralnetapi import HipAPI
kind = sys.argv[1] if len(sys.argv) > 1 else "alphazero"
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 19
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cfg = ro.alpha_zero_config(nblk, 34, 81, 8) if kind == "alphazero" else ro.rise_classical_config(nblk, 34, 81)
sd = ro.make_state_dict(cfg, seed=1)
d = nn_cases.export_case(tempfile.mkdtemp(), "b", cfg, sd)
for prec in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("float16-1b-8w", "float16-1b", "float16-2b-8w", "float16-2b")):
    net = HipAPI(0, B, d, prec)
    x = nn_cases.synthetic_planes(B, 34, 5)
    torch.as_tensor(net.device_buffers()["planes"], device="cuda").copy_(x.cuda()); torch.cuda.synchronize()
    net.time_forward(5)
    ms = net.time_forward(20) / 20
    fl = net.flops_per_position() * B
    print(f"{cfg.name} {prec} B={B}: {ms:.3f} ms/forward  {B/ms*1e3:.0f} evals/s  {fl/ms/1e9:.1f} TFLOP/s")
    ops = net.time_ops(3)
    agg = {}
    for n, t in ops: agg[n] = agg.get(n, 0) + t
    print("   per-op ms:", {k: round(v, 3) for k, v in agg.items()})
    net.close()
