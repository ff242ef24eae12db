"""Generates tests/golden/nn_*.npz by running the REFERENCE PyTorch model (imported from /root/reference).

Run in the build container only:  python oracle/make_golden.py
Each fixture stores the input planes and the reference's outputs (value, policy logits, aux) for weights that are
reproducible from (config, seed) via oracle.rise_oracle.make_state_dict -- the weights themselves are not stored.
The reference imports timm.models.layers.DropPath (rise_mobile_v3.py:27); timm is absent, so a 4-line identity stub
package is put on sys.path (DropPath(p=0) is the identity in eval mode anyway).
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def import_reference():
    stub = tempfile.mkdtemp(prefix="timm_stub_")
    os.makedirs(os.path.join(stub, "timm", "models"))
    open(os.path.join(stub, "timm", "__init__.py"), "w").close()
    open(os.path.join(stub, "timm", "models", "__init__.py"), "w").close()
    with open(os.path.join(stub, "timm", "models", "layers.py"), "w") as f:
        f.write("import torch\nclass DropPath(torch.nn.Module):\n    def __init__(self, p=0.0):\n        super().__init__()\n"
                "    def forward(self, x):\n        return x\n")
    sys.path.insert(0, stub)
    sys.path.insert(0, "/root/reference")
    from DeepCrazyhouse.src.domain.neural_net.architectures.pytorch.rise_mobile_v3 import RiseV3
    return RiseV3


def reference_model(RiseV3, cfg):
    n = len(cfg.kernels)
    if cfg.conv_block == "a0_res_block":
        from DeepCrazyhouse.src.domain.neural_net.architectures.pytorch.a0_resnet import AlphaZeroResnet
        return AlphaZeroResnet(n_labels=2272, channels=cfg.channels, nb_input_channels=cfg.nb_input_channels,
                               channels_value_head=cfg.channels_value_head, channels_policy_head=cfg.channels_policy_head,
                               num_res_blocks=n, value_fc_size=cfg.value_fc_size, act_type="relu",
                               select_policy_from_plane=cfg.select_policy_from_plane,
                               use_wdl=cfg.use_wdl, use_plys_to_end=cfg.use_plys_to_end, use_mlp_wdl_ply=False,
                               use_se=any(t is not None for t in cfg.se_types)).eval()
    return RiseV3(nb_input_channels=cfg.nb_input_channels, board_height=8, board_width=8, channels=cfg.channels,
                  channels_operating_init=cfg.channels_operating_init, channel_expansion=cfg.channel_expansion,
                  act_types=["relu"] * n, channels_value_head=cfg.channels_value_head, value_fc_size=cfg.value_fc_size,
                  channels_policy_head=cfg.channels_policy_head, dropout_rate=0, select_policy_from_plane=cfg.select_policy_from_plane,
                  kernels=cfg.kernels, se_types=cfg.se_types, use_avg_features=False, n_labels=cfg.n_labels,
                  use_wdl=cfg.use_wdl, use_plys_to_end=cfg.use_plys_to_end, use_mlp_wdl_ply=False, conv_block=cfg.conv_block).eval()


def main():
    import nn_cases
    RiseV3 = import_reference()
    torch.set_num_threads(1)   # fixed summation order for the stored fp32 goldens
    os.makedirs(nn_cases.GOLDEN_DIR, exist_ok=True)
    only = sys.argv[1:]
    for name in nn_cases.CASES:
        if only and name not in only:
            continue
        cfg, sd, x = nn_cases.make_case(name)
        model = reference_model(RiseV3, cfg)
        model.load_state_dict(sd, strict=True)
        with torch.no_grad():
            out = model(x)
        value, logits = out[0], out[1]
        aux = out[2] if len(out) > 2 else torch.zeros(0)
        path = os.path.join(nn_cases.GOLDEN_DIR, f"nn_{name}.npz")
        np.savez_compressed(path, x=x.numpy(), value=value.numpy(), logits=logits.numpy(), aux=aux.numpy())
        print(f"{name}: value {value.flatten()[:3].tolist()} |logit|max {float(logits.abs().max()):.3f} -> {path} "
              f"({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
