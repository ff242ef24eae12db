"""Dependency-free reader for ONNX model files ("ONNX weight import").

The reference's on-disk model format of record is ONNX (trainer_agent_pytorch.py:588-633 writes it, tensorrtapi.cpp:239-295 parses
it through TensorRT's parser).  Neither `onnx` nor a generated protobuf module is part of this image, so the handful of messages the
importer needs are decoded straight from the protobuf wire format (field numbers from the published onnx.proto3):

    ModelProto      ir_version=1 producer_name=2 graph=7 opset_import=8
    GraphProto      node=1 name=2 initializer=5 input=11 output=12
    NodeProto       input=1 output=2 name=3 op_type=4 attribute=5
    AttributeProto  name=1 f=2 i=3 s=4 t=5 floats=7 ints=8 type=20
    TensorProto     dims=1 data_type=2 float_data=4 int32_data=5 int64_data=7 name=8 raw_data=9 double_data=10
    ValueInfoProto  name=1 type=2;  TypeProto.tensor_type=1 {elem_type=1 shape=2};  TensorShapeProto.dim=1 {dim_value=1 dim_param=2}
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np

_VARINT, _FIXED64, _BYTES, _FIXED32 = 0, 1, 2, 5


class OnnxFormatError(ValueError):
    pass


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise OnnxFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise OnnxFormatError("varint too long")


def _fields(buf: bytes) -> Iterator[Tuple[int, int, Union[int, bytes]]]:
    """Yields (field number, wire type, value) of one message; length-delimited values are memory slices."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == _VARINT:
            val, pos = _varint(buf, pos)
        elif wt == _FIXED64:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == _FIXED32:
            val, pos = buf[pos:pos + 4], pos + 4
        elif wt == _BYTES:
            n, pos = _varint(buf, pos)
            if pos + n > end:
                raise OnnxFormatError("length-delimited field runs past its message")
            val, pos = buf[pos:pos + n], pos + n
        else:
            raise OnnxFormatError(f"unsupported wire type {wt}")
        yield num, wt, val
    if pos != end:
        raise OnnxFormatError("message ends inside a field")


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(val, wt) -> List[int]:
    if wt == _VARINT:
        return [_signed(val)]
    out, pos = [], 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(_signed(v))
    return out


def _packed_floats(val, wt) -> List[float]:
    if wt == _FIXED32:
        return [struct.unpack("<f", val)[0]]
    return list(np.frombuffer(val, "<f4"))


# TensorProto.DataType -> numpy
_DTYPES = {1: "<f4", 2: "u1", 3: "i1", 4: "<u2", 5: "<i2", 6: "<i4", 7: "<i8", 9: "?", 10: "<f2", 11: "<f8", 12: "<u4", 13: "<u8"}


def _tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype = 0
    name = ""
    raw: Optional[bytes] = None
    floats: List[float] = []
    ints: List[int] = []
    doubles: List[float] = []
    for num, wt, val in _fields(buf):
        if num == 1:
            dims += _packed_varints(val, wt)
        elif num == 2:
            dtype = val
        elif num == 4:
            floats += _packed_floats(val, wt)
        elif num in (5, 7):
            ints += _packed_varints(val, wt)
        elif num == 8:
            name = bytes(val).decode()
        elif num == 9:
            raw = bytes(val)
        elif num == 10:
            doubles += list(np.frombuffer(val, "<f8")) if wt == _BYTES else [struct.unpack("<d", val)[0]]
        elif num == 13 or num == 14:
            raise OnnxFormatError(f"tensor {name!r}: external data is not supported")
    if dtype not in _DTYPES:
        raise OnnxFormatError(f"tensor {name!r}: unsupported data type {dtype}")
    dt = np.dtype(_DTYPES[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, dt)
    elif dtype == 1:
        arr = np.asarray(floats, dt)
    elif dtype == 11:
        arr = np.asarray(doubles, dt)
    elif dtype == 10:                                       # fp16 travels as uint16 bit patterns in int32_data
        arr = np.asarray(ints, "<u2").view("<f2")
    else:
        arr = np.asarray(ints, dt)
    n = int(np.prod(dims)) if dims else 1
    if arr.size != n:
        raise OnnxFormatError(f"tensor {name!r}: {arr.size} elements for dims {dims}")
    return name, arr.reshape(dims).copy()


@dataclass
class Node:
    op: str
    inputs: List[str]
    outputs: List[str]
    name: str = ""
    attrs: Dict[str, object] = field(default_factory=dict)


@dataclass
class ValueInfo:
    name: str
    elem_type: int = 0
    shape: List[Union[int, str, None]] = field(default_factory=list)     # int, symbolic name or None per axis


@dataclass
class Graph:
    nodes: List[Node]
    initializers: Dict[str, np.ndarray]
    inputs: List[ValueInfo]                                                # graph inputs that are not initializers
    outputs: List[ValueInfo]
    producer: str = ""
    opset: int = 0


def _attribute(buf: bytes) -> Tuple[str, object]:
    name = ""
    f = i = s = t = None
    floats: List[float] = []
    ints: List[int] = []
    atype = 0
    for num, wt, val in _fields(buf):
        if num == 1:
            name = bytes(val).decode()
        elif num == 2:
            f = struct.unpack("<f", val)[0]
        elif num == 3:
            i = _signed(val)
        elif num == 4:
            s = bytes(val)
        elif num == 5:
            t = _tensor(val)[1]
        elif num == 7:
            floats += _packed_floats(val, wt)
        elif num == 8:
            ints += _packed_varints(val, wt)
        elif num == 20:
            atype = val
    # AttributeType: FLOAT=1 INT=2 STRING=3 TENSOR=4 FLOATS=6 INTS=7
    if atype == 1 or (atype == 0 and f is not None):
        return name, f
    if atype == 2 or (atype == 0 and i is not None):
        return name, i
    if atype == 3 or (atype == 0 and s is not None):
        return name, s.decode(errors="replace") if s is not None else ""
    if atype == 4 or (atype == 0 and t is not None):
        return name, t
    if atype == 6:
        return name, floats
    if atype == 7 or ints:
        return name, ints
    return name, floats if floats else None


def _node(buf: bytes) -> Node:
    n = Node("", [], [])
    for num, wt, val in _fields(buf):
        if num == 1:
            n.inputs.append(bytes(val).decode())
        elif num == 2:
            n.outputs.append(bytes(val).decode())
        elif num == 3:
            n.name = bytes(val).decode()
        elif num == 4:
            n.op = bytes(val).decode()
        elif num == 5:
            k, v = _attribute(val)
            n.attrs[k] = v
    return n


def _value_info(buf: bytes) -> ValueInfo:
    vi = ValueInfo("")
    for num, wt, val in _fields(buf):
        if num == 1:
            vi.name = bytes(val).decode()
        elif num == 2:
            for n2, _, v2 in _fields(val):
                if n2 != 1:                                                # only tensor types
                    continue
                for n3, _, v3 in _fields(v2):
                    if n3 == 1:
                        vi.elem_type = v3
                    elif n3 == 2:
                        for n4, _, v4 in _fields(v3):
                            if n4 != 1:
                                continue
                            dim: Union[int, str, None] = None
                            for n5, w5, v5 in _fields(v4):
                                if n5 == 1:
                                    dim = _signed(v5)
                                elif n5 == 2:
                                    dim = bytes(v5).decode()
                            vi.shape.append(dim)
    return vi


def read_onnx(path_or_bytes: Union[str, bytes]) -> Graph:
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        raw = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            raw = f.read()
    graph_buf = None
    producer = ""
    opset = 0
    try:
        for num, wt, val in _fields(raw):
            if num == 7 and wt == _BYTES:
                graph_buf = val
            elif num == 2 and wt == _BYTES:
                producer = bytes(val).decode(errors="replace")
            elif num == 8 and wt == _BYTES:
                domain, version = "", 0
                for n2, _, v2 in _fields(val):
                    if n2 == 1:
                        domain = bytes(v2).decode()
                    elif n2 == 2:
                        version = v2
                if domain in ("", "ai.onnx"):
                    opset = max(opset, version)
        if graph_buf is None:
            raise OnnxFormatError("no graph in the model file")
        nodes: List[Node] = []
        inits: Dict[str, np.ndarray] = {}
        inputs: List[ValueInfo] = []
        outputs: List[ValueInfo] = []
        for num, wt, val in _fields(graph_buf):
            if num == 1:
                nodes.append(_node(val))
            elif num == 5:
                name, arr = _tensor(val)
                inits[name] = arr
            elif num == 11:
                inputs.append(_value_info(val))
            elif num == 12:
                outputs.append(_value_info(val))
    except (IndexError, struct.error, UnicodeDecodeError) as e:
        raise OnnxFormatError(f"not a readable ONNX file: {e}") from e
    inputs = [v for v in inputs if v.name not in inits]                    # IR < 4 lists the initializers as inputs too
    return Graph(nodes, inits, inputs, outputs, producer, opset)
