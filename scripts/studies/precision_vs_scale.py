"""Round 6: the precision modes' logit error against the size of the activations / logits.  The seeded random nets of the
fixtures sit at max|logit| ~ 2 (RISEv2-19) ... 6.6 (RISEv2-13 lichess); trained nets reach +-10 and more.  nn_cases.scale_activations makes
the same nets with activations `act` times larger; every mode runs the same boards on the GPU and is compared with the fp32 oracle.
usage (GPU box): python scripts/studies/precision_vs_scale.py [out file]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import nn_cases  # noqa: E402
from crazyara_amd.neuralnetapi import HipAPI  # noqa: E402
from oracle import rise_oracle as ro  # noqa: E402

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
for case, B, version in (("risev2-19", 256, "1.0"), ("risev2-13-lichess", 256, "3.0"), ("risev2-7", 64, "1.0")):
    cfg, sd, _ = nn_cases.make_case(case)
    x = nn_cases.synthetic_planes(B, cfg.nb_input_channels, 4711)
    xin = np.ascontiguousarray(x.numpy())
    print(f"{case}, {B} boards: max |logit error| against the fp32 oracle (and |value error|)", file=out)
    print(f"  {'act':>4s} {'max|logit|':>10s} {'max|stream|':>11s}  " + "  ".join(f"{m:>22s}" for m in ("float16p8", "float16x3", "float32", "float16")), file=out)
    for act in (1.0, 1.5, 2.0, 3.0, 4.0, 8.0):
        sds = nn_cases.scale_activations(cfg, sd, act) if act != 1.0 else sd
        d = nn_cases.export_case(tempfile.mkdtemp(), case, cfg, sds, version=version)
        taps = {}
        o_value, o_logits, _ = ro.forward(cfg, sds, x, taps=taps)
        stream = max(float(v.abs().max()) for k, v in taps.items()) if taps else float("nan")
        cells = []
        for precision in ("float16p8", "float16x3", "float32", "float16"):
            net = HipAPI(0, B, d, precision, keep_logits=True)
            v, p = np.zeros(B, np.float32), np.zeros(B * cfg.nb_policy, np.float32)
            net.predict(xin, v, p, np.zeros(B * 4, np.float32) if cfg.nb_aux else None)
            logits = torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy()
            net.close()
            cells.append(f"{float(np.abs(logits - o_logits.numpy()).max()):.2e} ({float(np.abs(v - o_value.numpy().reshape(-1)).max()):.1e})")
        print(f"  {act:4.1f} {float(o_logits.abs().max()):10.2f} {stream:11.1f}  " + "  ".join(f"{c:>22s}" for c in cells), file=out)
    out.flush()
