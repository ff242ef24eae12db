"""NN-only timing of the BASELINE.json configurations (device-resident forwards, graph replay): evals/s per config."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nn_cases
from crazyara_amd import build, rise_config
from crazyara_amd.neuralnetapi import HipAPI
build.build()
CONFIGS = [
    ("config 1: crazyhouse RISEv2-7, batch 8", lambda: rise_config.rise_v2_config(7, 34, 81), 8, "1.0"),
    ("config 2: crazyhouse RISEv2-19, batch 256", lambda: rise_config.rise_v2_config(19, 34, 81), 256, "1.0"),
    ("          crazyhouse RISEv2-19, batch 512", lambda: rise_config.rise_v2_config(19, 34, 81), 512, "1.0"),
    ("          crazyhouse RISEv2-19, batch 1024", lambda: rise_config.rise_v2_config(19, 34, 81), 1024, "1.0"),
    ("          crazyhouse RISEv2-13 (reference's named net), batch 256", lambda: rise_config.rise_v2_config(13, 34, 81), 256, "1.0"),
    ("config 3: chess RISEv3.3, batch 512", lambda: rise_config.rise_v33_config(52, 76, False), 512, "3.0"),
    ("          chess RISEv3.3 WDLP, batch 512", lambda: rise_config.rise_v33_config(52, 76, True), 512, "3.0"),
    ("config 5: lichess (3check/KOTH tables) RISEv2-13 80ch, batch 1024", lambda: rise_config.rise_v2_config(13, 80, 84), 1024, "3.0"),
    ("N9:       AlphaZero-19 (dense 3x3 tower, value head 4 ch), batch 256", lambda: rise_config.alpha_zero_config(19, 34, 81, 4), 256, "1.0"),
    ("N9:       rise-classical-19 (dense 3x3 tower), batch 256", lambda: rise_config.rise_classical_config(19, 34, 81), 256, "1.0"),
    ("N9:       rise-classical-19, batch 512", lambda: rise_config.rise_classical_config(19, 34, 81), 512, "1.0"),
]
prec = sys.argv[1] if len(sys.argv) > 1 else "float16"
for name, mk, B, ver in CONFIGS:
    cfg = mk()
    sd = rise_config.make_state_dict(cfg, seed=1)
    d = nn_cases.export_case(tempfile.mkdtemp(), "c", cfg, sd, version=ver)
    net = HipAPI(0, B, d, prec)
    x = nn_cases.synthetic_planes(B, cfg.nb_input_channels, 5)
    torch.as_tensor(net.device_buffers()["planes"], device="cuda").copy_(x.cuda()); torch.cuda.synchronize()
    net.time_forward(10)
    it = 100
    ms = net.time_forward(it) / it
    ops = {}
    for n, t in net.time_ops(3): ops[n] = ops.get(n, 0) + t
    print(f"{name}: {ms:.3f} ms/forward  {B/ms*1e3:,.0f} evals/s  {net.flops_per_position()*B/ms/1e9:.0f} TFLOP/s  ops {{{', '.join(f'{k}: {v:.3f}' for k, v in ops.items())}}}", flush=True)
    net.close()
