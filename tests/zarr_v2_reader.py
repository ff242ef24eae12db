"""An independent reader of zarr format-2 arrays, written from the published storage specification (zarr.readthedocs.io,
"Zarr storage specification version 2") -- test-side only.  It is NOT derived from the writer under test: it handles everything
the spec allows for the pieces involved here (any chunk grid incl. partial edge chunks, missing chunks = fill_value, C / F chunk
order, both dimension separators, little / big endian numeric dtypes, compressor null / zlib / gzip / bz2 by codec id), so that a
writer that merely agrees with itself would still fail on a wrong key, name, byte order or chunk shape.

    .zgroup   {"zarr_format": 2}
    <array>/.zarray  {"zarr_format": 2, "shape", "chunks", "dtype", "compressor", "fill_value", "order", "filters"[, "dimension_separator"]}
    <array>/<i>.<j>...   one file per chunk, ALWAYS the full chunk shape (edge chunks are padded), encoded by filters then compressor
"""
import bz2
import gzip
import itertools
import json
import os
import zlib

import numpy as np


def open_group(root):
    with open(os.path.join(root, ".zgroup")) as f:
        meta = json.load(f)
    if meta.get("zarr_format") != 2:
        raise ValueError("not a zarr v2 group")
    return sorted(d for d in os.listdir(root) if os.path.isfile(os.path.join(root, d, ".zarray")))


def _decompress(buf, compressor):
    if compressor is None:
        return buf
    cid = compressor.get("id")
    if cid == "zlib":
        return zlib.decompress(buf)
    if cid == "gzip":
        return gzip.decompress(buf)
    if cid == "bz2":
        return bz2.decompress(buf)
    raise NotImplementedError(f"compressor {cid!r} (no codec library in this image)")


def read_array(root, name):
    d = os.path.join(root, name)
    with open(os.path.join(d, ".zarray")) as f:
        meta = json.load(f)
    for key in ("zarr_format", "shape", "chunks", "dtype", "compressor", "fill_value", "order", "filters"):
        if key not in meta:
            raise ValueError(f"{name}/.zarray lacks the mandatory key {key!r}")
    if meta["zarr_format"] != 2:
        raise ValueError("zarr_format must be 2")
    if meta["filters"] not in (None, []):
        raise NotImplementedError("filters")
    if not isinstance(meta["dtype"], str):
        raise NotImplementedError("structured dtypes")
    dtype = np.dtype(meta["dtype"])                      # "<i2", ">f4", "|u1" ... (numpy typestr, as the spec defines it)
    shape, chunks = tuple(meta["shape"]), tuple(meta["chunks"])
    if len(shape) != len(chunks) or any(c <= 0 for c in chunks):
        raise ValueError("shape / chunks mismatch")
    order = meta["order"]
    if order not in ("C", "F"):
        raise ValueError("order must be C or F")
    sep = meta.get("dimension_separator", ".")
    fill = meta["fill_value"]
    if fill is None:
        fill = 0
    elif isinstance(fill, str):
        fill = {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}[fill]
    out = np.full(shape, fill, dtype)
    grid = [range((s + c - 1) // c) for s, c in zip(shape, chunks)]
    nbytes = int(np.prod(chunks)) * dtype.itemsize
    for idx in itertools.product(*grid):
        path = os.path.join(d, sep.join(str(i) for i in idx)) if sep == "." else os.path.join(d, *[str(i) for i in idx])
        if not os.path.exists(path):
            continue                                     # an absent chunk holds the fill value
        with open(path, "rb") as f:
            raw = _decompress(f.read(), meta["compressor"])
        if len(raw) != nbytes:
            raise ValueError(f"chunk {path}: {len(raw)} bytes, a full chunk has {nbytes}")
        chunk = np.frombuffer(raw, dtype).reshape(chunks, order=order)
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sel] = chunk[tuple(slice(0, sl.stop - sl.start) for sl in sel)]
    return out
