"""Writer/reader for the CRANET01 model container consumed by the HIP backend.

Takes a state dict keyed like the reference's PyTorch RiseV3 module
(DeepCrazyhouse/src/domain/neural_net/architectures/pytorch/rise_mobile_v3.py:81-184), so a checkpoint saved by the
reference trainer (`save_torch_state`, trainer_agent_pytorch.py) can be exported unchanged.  The file name carries
the input-representation version as the reference's ONNX export does ("-v<maj>.<min>", trainer_agent_pytorch.py:588-633,
parsed by engine/src/nn/neuralnetapi.cpp:194-227).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Mapping

import numpy as np

MAGIC = b"CRANET01"


def write_cranet(path: str, meta: Mapping[str, object], tensors: Mapping[str, "np.ndarray"]) -> str:
    lines = []
    for k, v in meta.items():
        if isinstance(v, (list, tuple)):
            v = ",".join("none" if e is None else str(e) for e in v)
        elif isinstance(v, bool):
            v = int(v)
        lines.append(f"{k} {v}")
    blob = bytearray()
    for name, t in tensors.items():
        a = np.ascontiguousarray(np.asarray(t, dtype=np.float32))
        off = len(blob)
        blob += a.tobytes()
        blob += b"\0" * ((-len(blob)) % 16)
        dims = " ".join(str(d) for d in a.shape)
        lines.append(f"tensor {name} {a.ndim} {dims} {off}".replace("  ", " "))
    header = ("\n".join(lines) + "\n").encode()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(header)))
        f.write(header)
        f.write(bytes(blob))
    return path


def read_cranet(path: str):
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != MAGIC:
        raise ValueError(f"{path}: not a CRANET01 file")
    (hlen,) = struct.unpack("<Q", raw[8:16])
    header = raw[16:16 + hlen].decode()
    blob = raw[16 + hlen:]
    meta: Dict[str, str] = {}
    tensors: Dict[str, np.ndarray] = {}
    for line in header.splitlines():
        if not line:
            continue
        parts = line.split(" ")
        if parts[0] == "tensor":
            name, nd = parts[1], int(parts[2])
            shape = [int(x) for x in parts[3:3 + nd]]
            off = int(parts[3 + nd])
            n = int(np.prod(shape)) if shape else 1
            tensors[name] = np.frombuffer(blob, dtype=np.float32, count=n, offset=off).reshape(shape)
        else:
            meta[parts[0]] = " ".join(parts[1:])
    return meta, tensors


def export_rise(path: str, cfg, state_dict, input_version: str = "1.0", variant: str = "crazyhouse") -> str:
    """cfg: any object with the RiseV3 constructor fields (crazyara_amd/rise_config.py:RiseConfig)."""
    meta = dict(
        arch="rise", variant=variant, input_version=input_version,
        nb_input_channels=cfg.nb_input_channels, channels=cfg.channels,
        channels_operating_init=cfg.channels_operating_init, channel_expansion=cfg.channel_expansion,
        kernels=list(cfg.kernels), se_types=list(cfg.se_types),
        channels_value_head=cfg.channels_value_head, value_fc_size=cfg.value_fc_size,
        channels_policy_head=cfg.channels_policy_head, use_wdl=int(cfg.use_wdl), use_plys_to_end=int(cfg.use_plys_to_end),
        conv_block=getattr(cfg, "conv_block", "mobile_bottlekneck_res_block"),
        select_policy_from_plane=int(getattr(cfg, "select_policy_from_plane", True)), n_labels=getattr(cfg, "n_labels", 2272),
    )
    tensors = {}
    for k, v in state_dict.items():
        if k.endswith("num_batches_tracked"):
            continue
        if k.startswith("body."):      # AlphaZeroResnet keeps stem + blocks in `body` (a0_resnet.py:142-144): one naming in the file
            k = "body_spatial." + k[len("body."):]
        tensors[k] = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
    return write_cranet(path, meta, tensors)


def onnx_to_cranet(onnx_path: str, cranet_path: str = "") -> str:
    """Convert one of the reference's ONNX model files (trainer_agent_pytorch.py:588-633) to the container, through the library's
    own importer (csrc/nn/onnx_import.cpp; `mi_net_create` reads .onnx files with the same code).  Host only."""
    from . import _capi
    if not cranet_path:
        cranet_path = os.path.splitext(onnx_path)[0] + ".cranet"
    if _capi.load().mi_onnx_to_cranet(onnx_path.encode(), cranet_path.encode()):
        raise ValueError(_capi.last_error())
    return cranet_path
