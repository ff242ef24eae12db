"""Training-sample exporter of the self-play loop: the wire format the reference's Python trainers read.

Follows engine/src/rl/traindataexporter.cpp: one zarr (v2) group with the arrays
    x              int16   [N][C][8][8]   un-normalised input planes (get_state_planes(normalize=false), :175-196)
    y_value        int16   [N]            +1 / -1 / 0: the game result seen from the side to move (:66-78, 287-296)
    y_policy       float32 [N][NB_LABELS] the MCTS policy scattered to the flat label index, mirrored for Black (:198-224)
    y_best_move_q  float32 [N]            Q of the selected move (:49-63)
    plys_to_end    int16   [N]            plies from the sample to the end of its game (:80-93, 298-302)
    phase_vector   int16   [N]            game phase of the sample (:95-108)
    start_indices  int32   [N]            sample index at which game g starts (:226-233)
chunked [chunk_size] along N (:262-283).  The reference writes through z5; this writer emits the same directory layout with
plain little-endian chunks ("compressor": null), which every zarr reader opens.  `read_array` is the minimal reader the tests use.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Sequence

import numpy as np

from . import env

WHITE_WIN, DRAWN, BLACK_WIN = 1, 0, -1


class _ZarrArray:
    def __init__(self, root: str, name: str, dtype: str, shape: Sequence[int], chunks: Sequence[int]):
        self.dir = os.path.join(root, name)
        self.dtype, self.shape, self.chunks = np.dtype(dtype), tuple(shape), tuple(chunks)
        os.makedirs(self.dir, exist_ok=True)
        meta = {"zarr_format": 2, "shape": list(self.shape), "chunks": list(self.chunks), "dtype": self.dtype.str,
                "compressor": None, "fill_value": 0, "order": "C", "filters": None}
        with open(os.path.join(self.dir, ".zarray"), "w") as f:
            json.dump(meta, f, indent=1)

    def write_rows(self, start: int, data: np.ndarray) -> None:
        """data: [n, ...] rows start .. start+n of the first axis (the only chunked axis)."""
        data = np.ascontiguousarray(data, self.dtype)
        cs = self.chunks[0]
        pos = 0
        while pos < data.shape[0]:
            row = start + pos
            ci, off = row // cs, row % cs
            n = min(cs - off, data.shape[0] - pos)
            path = os.path.join(self.dir, ".".join([str(ci)] + ["0"] * (len(self.shape) - 1)))
            if os.path.exists(path):
                chunk = np.fromfile(path, self.dtype).reshape((cs,) + self.shape[1:])
            else:
                chunk = np.zeros((cs,) + self.shape[1:], self.dtype)
            chunk[off:off + n] = data[pos:pos + n]
            chunk.tofile(path)
            pos += n


def read_array(root: str, name: str) -> np.ndarray:
    """Minimal zarr-v2 reader for uncompressed, first-axis-chunked arrays (tests / inspection)."""
    d = os.path.join(root, name)
    with open(os.path.join(d, ".zarray")) as f:
        meta = json.load(f)
    assert meta["zarr_format"] == 2 and meta["compressor"] is None and meta["order"] == "C"
    shape, chunks, dtype = tuple(meta["shape"]), tuple(meta["chunks"]), np.dtype(meta["dtype"])
    out = np.full(shape, meta["fill_value"], dtype)
    for ci in range((shape[0] + chunks[0] - 1) // chunks[0]):
        path = os.path.join(d, ".".join([str(ci)] + ["0"] * (len(shape) - 1)))
        if os.path.exists(path):
            chunk = np.fromfile(path, dtype).reshape((chunks[0],) + shape[1:])
            n = min(chunks[0], shape[0] - ci * chunks[0])
            out[ci * chunks[0]:ci * chunks[0] + n] = chunk[:n]
    return out


class TrainDataExporter:
    """TrainDataExporter(fileName, numPhases=1, ..., numberChunks, chunkSize) (traindataexporter.cpp:136-156)."""

    def __init__(self, path: str, mode: int, version_major: int, nb_labels: int, number_chunks: int = 200, chunk_size: int = 128):
        self.path, self.mode, self.version = path, mode, version_major
        self.nb_labels, self.chunk_size = nb_labels, chunk_size
        self.number_samples = number_chunks * chunk_size
        self.channels = env._capi.load().mi_planes_channels(env.planes_layout(mode, version_major))
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, ".zgroup"), "w") as f:
            json.dump({"zarr_format": 2}, f)
        n, c = self.number_samples, chunk_size
        self.arr: Dict[str, _ZarrArray] = {
            "start_indices": _ZarrArray(path, "start_indices", "<i4", (n,), (c,)),
            "x": _ZarrArray(path, "x", "<i2", (n, self.channels, 8, 8), (c, self.channels, 8, 8)),
            "y_value": _ZarrArray(path, "y_value", "<i2", (n,), (c,)),
            "y_policy": _ZarrArray(path, "y_policy", "<f4", (n, nb_labels), (c, nb_labels)),
            "y_best_move_q": _ZarrArray(path, "y_best_move_q", "<f4", (n,), (c,)),
            "plys_to_end": _ZarrArray(path, "plys_to_end", "<i2", (n,), (c,)),
            "phase_vector": _ZarrArray(path, "phase_vector", "<i2", (n,), (c,)),
        }
        self.game_idx = 0
        self.start_idx = 0
        self._save_start_idx()
        self.new_game()

    # ---- per game ------------------------------------------------------------------------------------------------------
    def new_game(self) -> None:
        self._x: List[np.ndarray] = []
        self._policy: List[np.ndarray] = []
        self._q: List[float] = []
        self._stm: List[int] = []

    def is_file_full(self) -> bool:
        return self.start_idx >= self.number_samples

    def save_sample(self, pos: env.Position, moves: Sequence[int], policy: Sequence[float], best_move_q: float) -> None:
        """`moves` / `policy`: EvalInfo::legalMoves / policyProbSmall (entries beyond len(policy) count as 0)."""
        if self.start_idx + len(self._x) >= self.number_samples:
            return                                                # "Extended number of maximum samples"
        self._x.append(pos.planes(self.mode, self.version, False).astype(np.int16).reshape(self.channels, 8, 8))
        pol = np.zeros(self.nb_labels, np.float32)
        for m, p in zip(moves, policy):
            pol[pos.policy_index(m, self.mode, False)] = p        # action_to_index<classic, (not)Mirrored>
        self._policy.append(pol)
        self._q.append(float(best_move_q))
        self._stm.append(1 if pos.side_to_move() == 0 else -1)    # -(col * 2 - 1)

    def export_game_samples(self, result: int) -> int:
        """result: WHITE_WIN / DRAWN / BLACK_WIN.  Returns the number of samples written."""
        n = len(self._x)
        if n == 0 or self.start_idx >= self.number_samples:
            return 0
        value = np.array(self._stm, np.int16) * np.int16(-1 if result == BLACK_WIN else 0 if result == DRAWN else 1)
        plys = (n - np.arange(n)).astype(np.int16)                # (idx - n) * -1
        s = self.start_idx
        self.arr["x"].write_rows(s, np.stack(self._x))
        self.arr["y_value"].write_rows(s, value)
        self.arr["y_best_move_q"].write_rows(s, np.array(self._q, np.float32))
        self.arr["y_policy"].write_rows(s, np.stack(self._policy))
        self.arr["plys_to_end"].write_rows(s, plys)
        self.arr["phase_vector"].write_rows(s, np.zeros(n, np.int16))     # single-phase nets: get_phase == 0
        self.start_idx += n
        self.game_idx += 1
        self._save_start_idx()
        self.new_game()
        return n

    def _save_start_idx(self) -> None:
        if self.game_idx < self.number_samples:
            self.arr["start_indices"].write_rows(self.game_idx, np.array([self.start_idx], np.int32))
