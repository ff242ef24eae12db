"""Tiny RISE-family configurations for the ONNX import fixtures (tests/golden/onnx, written by oracle/make_onnx_fixtures.py).

The importer is host code and takes any channel count, so the files torch's exporter writes from the reference's modules can stay a
few dozen KiB; the weights are reproducible from (config, seed) through crazyara_amd.rise_config.make_state_dict.
"""
from crazyara_amd.rise_config import RiseConfig


def _cfg(**kw):
    base = dict(nb_input_channels=12, channels=16, channels_operating_init=8, channel_expansion=4, kernels=[3, 3], se_types=[None, None],
                channels_value_head=2, value_fc_size=8, channels_policy_head=5, n_labels=40)
    base.update(kw)
    return RiseConfig(**base)


# name: (config, seed, file name, exported batch or None for the dynamic axis[, stress-init (default True)])
CASES = {
    # RISEv3.3-shaped: 5x5 depthwise, both gate types, WDL + plies-to-end head, all five outputs, dynamic batch
    "mobile-se-wdlp": (_cfg(kernels=[3, 5, 3], se_types=[None, "eca_se", "ca_se"], use_wdl=True, use_plys_to_end=True, name="mobile-se-wdlp"),
                       31, "mobile-se-wdlp-v3.0.onnx", None),
    # RISEv2-shaped: tanh value head, policy map, fixed batch ("-bsize-2")
    "mobile-tanh": (_cfg(se_types=[None, "ca_se"], name="mobile-tanh"), 32, "mobile-tanh-v1.0-bsize-2.onnx", 2),
    "mobile-flat": (_cfg(select_policy_from_plane=False, name="mobile-flat"), 33, "mobile-flat-v2.8.onnx", None),
    "classical": (_cfg(channels_operating_init=16, channel_expansion=0, conv_block="classical_res_block", name="classical"),
                  34, "classical-v1.0.onnx", None),
    "alphazero": (_cfg(channels_operating_init=16, channel_expansion=0, conv_block="a0_res_block", channels_value_head=1, name="alphazero"),
                  35, "alphazero-v3.0.onnx", None),
    # channel gates INSIDE the dense blocks: ClassicalResidualBlock(se_type) gates the block input (hard-sigmoid, builder_util.py:416,431-433),
    # AlphaZero's ResidualBlock(use_se) gates the body output with a plain sigmoid before the shortcut (a0_resnet.py:94-107)
    "classical-se": (_cfg(channels_operating_init=16, channel_expansion=0, conv_block="classical_res_block", se_types=["ca_se", "eca_se"],
                          name="classical-se"), 37, "classical-se-v1.0.onnx", None),
    "alphazero-se": (_cfg(channels_operating_init=16, channel_expansion=0, conv_block="a0_res_block", channels_value_head=1,
                          se_types=["se", "se"], name="alphazero-se"), 38, "alphazero-se-v3.0.onnx", None),
    # torch-default initialisation: every BatchNorm has the same statistics, the exporter shares them through Identity nodes
    "mobile-shared-constants": (_cfg(kernels=[3, 5, 3], se_types=[None, "eca_se", "ca_se"], use_wdl=True, use_plys_to_end=True,
                                     name="mobile-shared-constants"), 36, "mobile-shared-constants-v3.0.onnx", None, False),
}


def unpack(name):
    c = CASES[name]
    return c[0], c[1], c[2], c[3], (c[4] if len(c) > 4 else True)
