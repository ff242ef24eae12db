"""Shared NN parity cases: configs, deterministic weights, deterministic inputs (no reference needed at run time)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import rise_oracle as ro  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def small_risev2():
    """3-block RISEv2 parametrisation (C_op 128/192/256), ca_se on the last two blocks, crazyhouse v1 planes."""
    cfg = ro.rise_v2_config(3, 34, 81)
    cfg.se_types = [None, "ca_se", "ca_se"]
    cfg.name = "risev2-3"
    return cfg


def small_risev2_flat():
    cfg = small_risev2()
    cfg.select_policy_from_plane = False
    cfg.n_labels = 2272
    cfg.name = "risev2-3-flat"
    return cfg


CASES = {
    # name: (config factory, seed, stress-init, batch)
    "risev2-3": (small_risev2, 11, True, 4),
    "risev2-7": (lambda: ro.rise_v2_config(7, 34, 81), 12, True, 8),            # BASELINE config 1
    "risev2-13": (lambda: ro.rise_v2_config(13, 34, 81), 13, True, 4),          # the reference's named RISEv2
    "risev2-19": (lambda: ro.rise_v2_config(19, 34, 81), 14, True, 4),          # BASELINE config 2 (headline)
    "risev2-13-default-init": (lambda: ro.rise_v2_config(13, 34, 81), 15, False, 4),
    "risev33": (lambda: ro.rise_v33_config(52, 76, False), 16, True, 4),        # BASELINE config 3
    "risev33-wdlp": (lambda: ro.rise_v33_config(52, 76, True), 17, True, 4),    # released ClassicAra head
    "risev2-13-lichess": (lambda: ro.rise_v2_config(13, 80, 84), 18, True, 2),  # MultiAra tables (config 5)
    # dense 3x3 siblings of the same zoo
    "rise-classical-4": (lambda: ro.rise_classical_config(4, 34, 81), 19, True, 4),
    "alphazero-5": (lambda: ro.alpha_zero_config(5, 52, 76, 4), 20, True, 4),
    "alphazero-3-cv8": (lambda: ro.alpha_zero_config(3, 34, 81, 8), 21, True, 3),
    # SE gates inside the dense blocks: on the block input with hard-sigmoid (ClassicalResidualBlock(se_type)), on the body output with
    # plain sigmoid (AlphaZero ResidualBlock(use_se))
    "rise-classical-3-se": (lambda: ro.rise_classical_config(3, 34, 81, se_types=[None, "ca_se", "eca_se"]), 23, True, 4),
    "alphazero-3-se": (lambda: ro.alpha_zero_config(3, 34, 81, 4, use_se=True), 24, True, 4),
    # flat-label policy head (select_policy_from_plane=False): Linear(P*64 -> 2272 crazyhouse labels)
    "risev2-3-flat": (small_risev2_flat, 22, True, 5),
}


def make_case(name):
    factory, seed, stress, batch = CASES[name]
    cfg = factory()
    sd = ro.make_state_dict(cfg, seed=seed, stress=stress)
    x = synthetic_planes(batch, cfg.nb_input_channels, seed + 1000)
    return cfg, sd, x


def scale_activations(cfg, sd, act=8.0):
    """A copy of `sd` whose activations -- stem output, every tensor of the bottleneck blocks, the heads' hidden layers -- and policy
    logits are `act` times larger (trained nets have logits of +-10 and more; the seeded random nets sit at +-2): the stem's BatchNorm scales
    its output, every later BatchNorm takes running_mean * act and bias * act (BN(act u) with those = act BN(u)), the value head's last
    Linear is divided by `act` so that the value does not saturate.  SE gates see larger means (not homogeneous): a different net, not a
    rescaling of the same function -- a stress case for the precision modes' ABSOLUTE error bounds."""
    out = {k: v.clone() for k, v in sd.items()}
    pre = cfg.key_prefix
    for k in (pre + ".0.body.1.weight", pre + ".0.body.1.bias"):
        out[k] = out[k] * act
    for k in list(out):
        if k.startswith(pre + ".0.body.1."):
            continue
        if k.endswith(".running_mean") or (k.endswith(".bias") and k[:-5] + ".running_mean" in out):
            out[k] = out[k] * act
    out["value_head.body_final.0.bias"] = out["value_head.body_final.0.bias"] * act
    out["value_head.body_final.2.weight"] = out["value_head.body_final.2.weight"] / act
    for k in ("value_head.body_wdl.0.weight", "value_head.body_plys.0.weight"):
        if k in out:
            out[k] = out[k] / act
    return out


def synthetic_planes(batch, channels, seed):
    """Board-like planes: ~88 % zeros, sparse ones, a few constant planes with fractional values."""
    rng = np.random.default_rng(seed)
    x = (rng.random((batch, channels, 8, 8)) < 0.10).astype(np.float32)
    for b in range(batch):
        for c in rng.choice(channels, size=max(2, channels // 8), replace=False):
            x[b, c] = rng.choice([0.0, 1.0, 0.25, 1.0 / 32, 3.0 / 8, 0.002 * (b + 1)])
    return torch.from_numpy(x)


def export_case(tmpdir, name, cfg, sd, version="1.0"):
    from crazyara_amd import netfile
    d = os.path.join(str(tmpdir), name)
    os.makedirs(d, exist_ok=True)
    netfile.export_rise(os.path.join(d, f"{cfg.name}-v{version}.cranet"), cfg, sd, input_version=version)
    return d
