"""Per-op event times of one net (development): python scripts/time_ops_net.py <family> <batch> <precision>
family: risev2-19 | risev33 | risev2-13-lichess"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crazyara_amd import netfile, rise_config  # noqa: E402
from crazyara_amd.neuralnetapi import HipAPI  # noqa: E402


def main():
    fam, batch, prec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    if fam == "risev33":
        cfg, version = rise_config.rise_v33_config(52, 76, False), "3.0"
    elif fam == "risev2-13-lichess":
        cfg, version = rise_config.rise_v2_config(13, 80, 84), "3.0"
    else:
        cfg, version = rise_config.rise_v2_config(int(fam.split("-")[1]), 34, 81), "1.0"
    sd = rise_config.make_state_dict(cfg, seed=31, stress=True)
    d = tempfile.mkdtemp(prefix="cra_ops_")
    netfile.export_rise(os.path.join(d, f"{cfg.name}-v{version}.cranet"), cfg, sd, input_version=version)
    net = HipAPI(0, batch, d, prec)
    rng = np.random.default_rng(0)
    x = (rng.random((batch, cfg.nb_input_channels, 8, 8)) < 0.1).astype(np.float32)
    v = np.zeros(batch, np.float32)
    p = np.zeros(batch * cfg.nb_policy, np.float32)
    net.predict(x, v, p)
    net.time_ops(3)
    ops = net.time_ops(10)
    total = sum(ms for _, ms in ops)
    print(f"{fam} batch {batch} {prec}: {len(ops)} launches, {total:.4f} ms per forward by op events, {net.time_forward(50) / 50:.4f} ms per forward")
    agg = {}
    for i, (name, ms) in enumerate(ops):
        print(f"  {i:3d} {name:22s} {ms:.4f}")
        agg[name] = agg.get(name, 0.0) + ms
    for k, ms in sorted(agg.items(), key=lambda kv: -kv[1]):
        print(f"  sum {k:22s} {ms:.4f}  ({100 * ms / total:.1f} %)")
    net.close()


if __name__ == "__main__":
    main()
