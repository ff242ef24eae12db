"""Test infrastructure: serialises a RISE-family model (config + state dict) as an ONNX file, by hand-written protobuf encoding.

The `onnx` package is not part of this image.  The importer (crazyara_amd/csrc/nn/onnx_import.cpp) is pinned against bytes written by
torch's own exporter from the reference's modules (tests/golden/onnx, oracle/make_onnx_fixtures.py); this writer adds what those
tiny fixtures cannot: full-size graphs the GPU can load, and the other flavours exporters leave behind --
  fold_bn=False  separate BatchNormalization nodes (unsimplified / MXNet-era exports), fold_bn=True: conv weight+bias
  linear="gemm"  Gemm(transB=1) | "matmul": MatMul with a transposed weight + Add(bias)
The graph follows the forward pass of rise_mobile_v3.py / a0_resnet.py / builder_util.py as restated in oracle/rise_oracle.py.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Sequence

import numpy as np


# ---- protobuf wire format -------------------------------------------------------------------------------------------------
def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _int(num: int, v: int) -> bytes:
    return _varint(num << 3) + _varint(v)


def _ld(num: int, payload: bytes) -> bytes:
    return _varint(num << 3 | 2) + _varint(len(payload)) + payload


def _str(num: int, s: str) -> bytes:
    return _ld(num, s.encode())


def _f32(num: int, v: float) -> bytes:
    return _varint(num << 3 | 5) + struct.pack("<f", v)


def tensor(name: str, arr: np.ndarray, as_raw: bool = True) -> bytes:
    arr = np.asarray(arr)
    out = b"".join(_int(1, int(d)) for d in arr.shape)
    if arr.dtype == np.int64:
        out += _int(2, 7)
        out += _ld(9, arr.astype("<i8").tobytes()) if as_raw else _ld(7, b"".join(_varint(int(v)) for v in arr.reshape(-1)))
    elif arr.dtype == np.float16:
        out += _int(2, 10) + _ld(9, arr.astype("<f2").tobytes())
    else:
        out += _int(2, 1)
        a = arr.astype("<f4")
        out += _ld(9, a.tobytes()) if as_raw else _ld(4, a.tobytes())        # float_data is a packed repeated field
    return out + _str(8, name)


def _attr(name: str, value) -> bytes:
    out = _str(1, name)
    if isinstance(value, float):
        return out + _f32(2, value) + _int(20, 1)
    if isinstance(value, int):
        return out + _int(3, value) + _int(20, 2)
    if isinstance(value, str):
        return out + _ld(4, value.encode()) + _int(20, 3)
    if isinstance(value, np.ndarray):
        return out + _ld(5, tensor("", value)) + _int(20, 4)
    return out + b"".join(_int(8, int(v)) for v in value) + _int(20, 7)     # ints


def node(op: str, inputs: Sequence[str], outputs: Sequence[str], **attrs) -> bytes:
    out = b"".join(_str(1, i) for i in inputs) + b"".join(_str(2, o) for o in outputs)
    out += _str(3, outputs[0] + "_node") + _str(4, op)
    return out + b"".join(_ld(5, _attr(k, v)) for k, v in attrs.items())


def value_info(name: str, shape: Sequence) -> bytes:
    dims = b"".join(_ld(1, _str(2, d) if isinstance(d, str) else _int(1, d)) for d in shape)
    return _str(1, name) + _ld(2, _ld(1, _int(1, 1) + _ld(2, dims)))


def model(nodes: List[bytes], inits: List[bytes], inputs: List[bytes], outputs: List[bytes], producer="crazyara_amd-tests", opset=17) -> bytes:
    g = b"".join(_ld(1, n) for n in nodes) + _str(2, "main") + b"".join(_ld(5, t) for t in inits)
    g += b"".join(_ld(11, i) for i in inputs) + b"".join(_ld(12, o) for o in outputs)
    return _int(1, 8) + _str(2, producer) + _ld(7, g) + _ld(8, _str(1, "") + _int(2, opset))


# ---- RISE graph -------------------------------------------------------------------------------------------------------------
class _Builder:
    def __init__(self, sd: Dict[str, "np.ndarray"], fold_bn: bool, linear: str, weights_dtype=np.float32, raw: bool = True):
        self.sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in sd.items()}
        self.fold_bn, self.linear_kind, self.wdt, self.raw = fold_bn, linear, weights_dtype, raw
        self.nodes: List[bytes] = []
        self.inits: List[bytes] = []
        self.n = 0

    def name(self, stem: str) -> str:
        self.n += 1
        return f"{stem}_{self.n}"

    def init(self, stem: str, arr: np.ndarray) -> str:
        nm = self.name(stem)
        if arr.dtype != np.int64:
            arr = arr.astype(self.wdt)
        self.inits.append(tensor(nm, arr, self.raw))
        return nm

    def op(self, op: str, inputs: Sequence[str], out: Optional[str] = None, **attrs) -> str:
        out = out or self.name(op.lower())
        self.nodes.append(node(op, inputs, [out], **attrs))
        return out

    def conv_bn(self, x: str, conv: str, bn: Optional[str], relu: bool, groups: int = 1) -> str:
        w = self.sd[conv + ".weight"].astype(np.float64)
        k = w.shape[2]
        attrs = dict(dilations=[1, 1], group=groups, kernel_shape=[k, k], pads=[k // 2] * 4, strides=[1, 1])
        if bn is None:
            y = self.op("Conv", [x, self.init("w", w)], **attrs)
        elif self.fold_bn:
            g, b, m, v = (self.sd[f"{bn}.{s}"].astype(np.float64) for s in ("weight", "bias", "running_mean", "running_var"))
            s = g / np.sqrt(v + 1e-5)
            y = self.op("Conv", [x, self.init("w", w * s[:, None, None, None]), self.init("b", b - m * s)], **attrs)
        else:
            y = self.op("Conv", [x, self.init("w", w)], **attrs)
            y = self.op("BatchNormalization", [y] + [self.init(s, self.sd[f"{bn}.{t}"]) for s, t in
                                                      (("gamma", "weight"), ("beta", "bias"), ("mean", "running_mean"), ("var", "running_var"))],
                        epsilon=1e-5, momentum=0.9)
        return self.op("Relu", [y]) if relu else y

    def fc(self, x: str, name: str, bias: bool, out: Optional[str] = None) -> str:
        w = self.sd[name + ".weight"]
        if self.linear_kind == "gemm":
            ins = [x, self.init("fcw", w)] + ([self.init("fcb", self.sd[name + ".bias"])] if bias else [])
            return self.op("Gemm", ins, out, alpha=1.0, beta=1.0, transB=1)
        y = self.op("MatMul", [x, self.init("fcw", np.ascontiguousarray(w.T))], None if bias else out)
        return self.op("Add", [y, self.init("fcb", self.sd[name + ".bias"])], out) if bias else y

    def flatten(self, x: str, width: int, out: Optional[str] = None) -> str:
        if self.linear_kind == "gemm":
            return self.op("Flatten", [x], out, axis=1)
        return self.op("Reshape", [x, self.init("shape", np.array([-1, width], np.int64))], out)


def rise_to_onnx(cfg, sd, batch=None, fold_bn: bool = True, linear: str = "gemm", weights_dtype=np.float32, raw: bool = True,
                 prune_plys: bool = False) -> bytes:
    """batch=None: dynamic batch axis ('batch_size'), else the fixed size of a "-bsize-<B>" file."""
    b = _Builder(sd, fold_bn, linear, weights_dtype, raw)
    C = cfg.channels
    pre = "body_spatial" if f"body_spatial.0.body.0.weight" in b.sd else "body"
    x = b.conv_bn("data", f"{pre}.0.body.0", f"{pre}.0.body.1", True)
    for i, (k, cop, se) in enumerate(zip(cfg.kernels, cfg.channels_operating(), cfg.se_types)):
        p = f"{pre}.{i + 1}"
        if cfg.conv_block == "a0_res_block":             # ResidualBlock(use_se): the gate sits on the branch output, plain sigmoid
            y = b.conv_bn(x, p + ".body.0", p + ".body.1", True)
            y = b.conv_bn(y, p + ".body.3", p + ".body.4", False)
            if se is not None:
                g = b.flatten(b.op("GlobalAveragePool", [y]), C)
                g = b.op("Relu", [b.fc(g, p + ".se.fc.0", False)])
                g = b.op("Sigmoid", [b.fc(g, p + ".se.fc.2", False)])
                g = b.op("Reshape", [g, b.init("shape", np.array([-1, C, 1, 1], np.int64))])
                y = b.op("Mul", [y, g])
            x = b.op("Relu", [b.op("Add", [x, y])])
            continue
        if se in ("ca_se", "se"):
            y = b.flatten(b.op("GlobalAveragePool", [x]), C)
            y = b.op("Relu", [b.fc(y, p + ".se.fc.0", False)])
            y = b.op("HardSigmoid", [b.fc(y, p + ".se.fc.2", False)], alpha=1.0 / 6.0, beta=0.5)
            y = b.op("Reshape", [y, b.init("shape", np.array([-1, C, 1, 1], np.int64))])
            x = b.op("Mul", [x, y])
        elif se == "eca_se":
            w = b.sd[p + ".se.body.0.weight"]
            y = b.op("Reshape", [b.op("GlobalAveragePool", [x]), b.init("shape", np.array([-1, C, 1], np.int64))])
            y = b.op("Conv", [y, b.init("w", w), b.init("b", b.sd[p + ".se.body.0.bias"])], dilations=[1], group=1, kernel_shape=[w.shape[2]],
                     pads=[w.shape[2] // 2] * 2, strides=[1])
            y = b.op("HardSigmoid", [y], alpha=1.0 / 6.0, beta=0.5)
            y = b.op("Reshape", [y, b.init("shape", np.array([-1, C, 1, 1], np.int64))])
            x = b.op("Mul", [x, y])
        if cfg.conv_block == "mobile_bottlekneck_res_block":
            y = b.conv_bn(x, p + ".body.0", p + ".body.1", True)
            y = b.conv_bn(y, p + ".body.3", p + ".body.4", True, groups=cop)
            y = b.conv_bn(y, p + ".body.6", p + ".body.7", False)
            x = b.op("Add", [x, y])
        elif cfg.conv_block == "classical_res_block":
            y = b.conv_bn(x, p + ".body.0", p + ".body.1", True)
            y = b.conv_bn(y, p + ".body.3", p + ".body.4", True)
            x = b.op("Add", [y, x])
        else:
            y = b.conv_bn(x, p + ".body.0", p + ".body.1", True)
            y = b.conv_bn(y, p + ".body.3", p + ".body.4", False)
            x = b.op("Relu", [b.op("Add", [x, y])])
    # value head
    outputs = ["value_out", "policy_out"]
    v = b.flatten(b.conv_bn(x, "value_head.body.0", "value_head.body.1", True), 64 * cfg.channels_value_head)
    if cfg.use_wdl and cfg.use_plys_to_end:
        wdl = b.fc(v, "value_head.body_wdl.0", True, "wdl_out")
        if not prune_plys:
            b.op("Sigmoid", [b.fc(v, "value_head.body_plys.0", True)], "plys_to_end_out")
            b.op("Concat", ["wdl_out", "plys_to_end_out"], "auxiliary_out", axis=1)
            outputs += ["auxiliary_out", "wdl_out", "plys_to_end_out"]
        sm = b.op("Softmax", [wdl], axis=1)
        parts = [b.name("split") for _ in range(3)]
        b.nodes.append(node("Split", [sm, b.init("split", np.array([1, 1, 1], np.int64))], parts, axis=1))
        b.op("Add", [b.op("Neg", [parts[0]]), parts[2]], "value_out")
    else:
        y = b.op("Relu", [b.fc(v, "value_head.body_final.0", True)])
        b.op("Tanh", [b.fc(y, "value_head.body_final.2", True)], "value_out")
    # policy head
    y = b.conv_bn(x, "policy_head.body.0", "policy_head.body.1", True)
    if cfg.select_policy_from_plane:
        y = b.conv_bn(y, "policy_head.body.3", None, False)
        b.flatten(y, 64 * cfg.channels_policy_head, "policy_out")
    else:
        y = b.conv_bn(y, "policy_head.body.3", "policy_head.body2.0", True)
        y = b.flatten(y, 64 * cfg.channels_policy_head)
        b.fc(y, "policy_head.body3.0", True, "policy_out")
    bdim = "batch_size" if batch is None else int(batch)
    widths = {"value_out": 1, "policy_out": cfg.nb_policy, "auxiliary_out": 4, "wdl_out": 3, "plys_to_end_out": 1}
    return model(b.nodes, b.inits, [value_info("data", [bdim, cfg.nb_input_channels, 8, 8])],
                 [value_info(o, [bdim, widths[o]]) for o in outputs])
