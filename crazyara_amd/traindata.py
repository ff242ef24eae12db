"""Training-sample exporter of the self-play loop: Python handle on the C++ `TrainDataExporter`
(crazyara_amd/csrc/rl/traindata.cpp through `mi_traindata_*`), which restates engine/src/rl/traindataexporter.cpp: one zarr (v2)
group with the arrays x / y_value / y_policy / y_best_move_q / plys_to_end / phase_vector / start_indices, chunked along the sample
axis, raw little-endian chunks.  Nothing is written from Python; tests read the files back with an independent reader
(tests/zarr_v2_reader.py)."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _capi, env

WHITE_WIN, DRAWN, BLACK_WIN = 1, 0, -1                      # the results of crazyara_amd.selfplay
_RESULT_CODE = {DRAWN: 0, WHITE_WIN: 1, BLACK_WIN: 2}       # enum Result of the reference (engine/src/state.h)


class TrainDataExporter:
    """TrainDataExporter(fileName, numPhases=1, ..., numberChunks, chunkSize) (traindataexporter.cpp:136-156)."""

    def __init__(self, path: str, mode: int, version_major, nb_labels: int = 0, number_chunks: int = 200, chunk_size: int = 128):
        self._lib = _capi.load()
        major, minor = env.split_version(version_major)
        self._h = self._lib.mi_traindata_create(path.encode(), mode, major, minor, number_chunks, chunk_size)
        if not self._h:
            raise RuntimeError(_capi.last_error())
        self.path, self.mode = path, mode
        i = self.info()
        self.nb_labels, self.channels, self.number_samples = i["nb_labels"], i["channels"], i["number_samples"]
        if nb_labels and nb_labels != self.nb_labels:
            raise ValueError(f"mode {mode} has {self.nb_labels} labels, not {nb_labels}")

    def info(self) -> dict:
        ns, si, gi = C.c_uint(), C.c_uint(), C.c_uint()
        nl, ch, full = C.c_int(), C.c_int(), C.c_int()
        self._lib.mi_traindata_info(self._h, C.byref(ns), C.byref(si), C.byref(gi), C.byref(nl), C.byref(ch), C.byref(full))
        return dict(number_samples=ns.value, start_index=si.value, game_index=gi.value, nb_labels=nl.value, channels=ch.value,
                    is_full=bool(full.value))

    @property
    def start_idx(self) -> int:
        return self.info()["start_index"]

    @property
    def game_idx(self) -> int:
        return self.info()["game_index"]

    def set_phases(self, num_phases: int, game_phase_definition: int = 0) -> None:
        """numPhases / gamePhaseDefinition of the reference exporter's constructor (0 = lichess, 1 = movecount): save_sample writes
        pos->get_phase(numPhases, gamePhaseDefinition) into phase_vector (traindataexporter.cpp:91-103)."""
        if self._lib.mi_traindata_set_phases(self._h, int(num_phases), int(game_phase_definition)):
            raise RuntimeError(_capi.last_error())

    def new_game(self) -> None:
        if self._lib.mi_traindata_new_game(self._h):
            raise RuntimeError(_capi.last_error())

    def is_file_full(self) -> bool:
        return self.info()["is_full"]

    def save_sample(self, pos: env.Position, moves: Sequence[int], policy: Sequence[float], best_move_q: float) -> None:
        """`moves` / `policy`: EvalInfo::legalMoves / policyProbSmall (entries beyond len(policy) count as 0)."""
        mv = (C.c_uint32 * len(moves))(*moves)
        pol = np.ascontiguousarray(policy, np.float64)
        if self._lib.mi_traindata_save_sample(self._h, pos._h, mv, len(moves), pol.ctypes.data_as(C.POINTER(C.c_double)), len(pol),
                                              float(best_move_q)):
            raise RuntimeError(_capi.last_error())

    def save_search_sample(self, pool, tree: int) -> None:
        """The sample of a searched tree of a pool, taken inside the library (root position, moves, MCTS policy, bestMoveQ)."""
        if self._lib.mi_search_save_sample(pool._h, tree, self._h):
            raise RuntimeError(_capi.last_error())

    def export_game_samples(self, result: int) -> int:
        """result: WHITE_WIN / DRAWN / BLACK_WIN.  Returns the number of samples written."""
        n = C.c_uint()
        if self._lib.mi_traindata_export_game_samples(self._h, _RESULT_CODE[result], C.byref(n)):
            raise RuntimeError(_capi.last_error())
        return n.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.mi_traindata_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
