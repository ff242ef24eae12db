"""Runs a few device-resident forwards of the headline net (for rocprofv3 --pmc / --kernel-trace passes)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crazyara_amd import build, netfile, rise_config
from crazyara_amd.neuralnetapi import HipAPI
build.build()
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 19
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prec = sys.argv[3] if len(sys.argv) > 3 else "float16"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
cfg = rise_config.rise_v2_config(nblk)
sd = rise_config.make_state_dict(cfg, seed=2024)
tmp = tempfile.mkdtemp()
netfile.export_rise(os.path.join(tmp, f"{cfg.name}-v1.0.cranet"), cfg, sd)
net = HipAPI(0, B, tmp, prec)
x = (torch.rand(B, 34, 8, 8) < 0.1).float()
torch.as_tensor(net.device_buffers()["planes"], device="cuda").copy_(x.cuda()); torch.cuda.synchronize()
for _ in range(iters):
    net.forward_device()
net.sync()
print("done")
