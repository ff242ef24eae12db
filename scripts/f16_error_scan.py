"""Measured maxima of |logit|, |value|, |prob| error of a precision mode against the fp32 oracle, per parity case and for the
headline configuration at full size (RISEv2-19, 256 boards).  The tolerances of tests/test_nn_parity_gpu.py are set from this
table (<= 1.25 x the measured maximum); DESIGN 4.2 quotes it.

  python scripts/f16_error_scan.py [float16] [float16-3k ...]
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import nn_cases  # noqa: E402
from crazyara_amd import build  # noqa: E402
from crazyara_amd.neuralnetapi import HipAPI  # noqa: E402
from oracle import rise_oracle as ro  # noqa: E402

build.build()
precisions = sys.argv[1:] or ["float16"]


def run(cfg, sd, x, precision, version):
    d = nn_cases.export_case(tempfile.mkdtemp(), cfg.name, cfg, sd, version=version)
    B = x.shape[0]
    net = HipAPI(0, B, d, precision, keep_logits=True)
    v, p = np.zeros(B, np.float32), np.zeros(B * cfg.nb_policy, np.float32)
    aux = np.zeros(B * 4, np.float32) if cfg.nb_aux else None
    net.predict(np.ascontiguousarray(x.numpy()), v, p, aux)
    logits = torch.as_tensor(net.device_buffers()["logits"], device="cuda").cpu().numpy().copy()
    net.close()
    ov, ol, oa = ro.forward(cfg, sd, x)
    op = torch.softmax(ol, 1).numpy()
    return dict(logit=float(np.abs(logits - ol.numpy()).max()), value=float(np.abs(v - ov.numpy().reshape(-1)).max()),
                prob=float(np.abs(p.reshape(B, -1) - op).max()),
                aux=float(np.abs(aux.reshape(-1, 4) - oa.numpy()).max()) if cfg.nb_aux else 0.0,
                max_logit=float(np.abs(ol.numpy()).max()))


for precision in precisions:
    print(f"== {precision}")
    for name in nn_cases.CASES:
        cfg, sd, x = nn_cases.make_case(name)
        r = run(cfg, sd, x, precision, "3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
        print(f"  {name:<26s} B={x.shape[0]:<4d} logit {r['logit']:.3e} (max |logit| {r['max_logit']:.2f})  value {r['value']:.3e}  "
              f"prob {r['prob']:.3e}  aux {r['aux']:.3e}", flush=True)
    cfg, sd, xg = nn_cases.make_case("risev2-19")
    x = nn_cases.synthetic_planes(256, cfg.nb_input_channels, 777)
    x[:4] = xg
    r = run(cfg, sd, x, precision, "1.0")
    print(f"  {'risev2-19 headline':<26s} B=256  logit {r['logit']:.3e} (max |logit| {r['max_logit']:.2f})  value {r['value']:.3e}  "
          f"prob {r['prob']:.3e}", flush=True)
