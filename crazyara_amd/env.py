"""Host-side environment wrapper over the C ABI: the `State` operations the search needs (engine/src/state.h:287-509 as
implemented by BoardState, engine/src/environments/chess_related/boardstate.cpp:42-277), plus input planes and policy
indices.  Thin ctypes glue -- all logic lives in libcrazyara_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np

from . import _capi

MODE_CRAZYHOUSE, MODE_CHESS, MODE_LICHESS = 0, 1, 2
TERMINAL_LOSS, TERMINAL_DRAW, TERMINAL_WIN, TERMINAL_CUSTOM, TERMINAL_NONE = 0, 1, 2, 3, 4


def split_version(version):
    """3 -> (3, 0); "2.8" / 2.8 -> (2, 8)."""
    if isinstance(version, (int, np.integer)):
        return int(version), 0
    major, _, minor = str(version).partition(".")
    return int(major), int(minor or 0)


def planes_layout(mode: int, version) -> int:
    major, minor = split_version(version)
    return _capi.load().mi_planes_layout_minor(mode, major, minor)


class Position:
    def __init__(self, fen: str = "", is960: bool = False, variant: str = "chess", _handle=None):
        self._lib = _capi.load()
        self._h = _handle or self._lib.mi_pos_create((fen or "").encode(), int(is960), variant.encode())
        if not self._h:
            raise ValueError(_capi.last_error())

    def clone(self) -> "Position":
        return Position(_handle=self._lib.mi_pos_clone(self._h))

    def fen(self) -> str:
        buf = C.create_string_buffer(256)
        self._lib.mi_pos_fen(self._h, buf, 256)
        return buf.value.decode()

    def side_to_move(self) -> int:
        return self._lib.mi_pos_side_to_move(self._h)

    def legal_moves(self) -> List[int]:
        buf = (C.c_uint32 * 512)()
        n = self._lib.mi_pos_legal_moves(self._h, buf, 512)
        return list(buf[:n])

    def move_uci(self, move: int) -> str:
        buf = C.create_string_buffer(16)
        self._lib.mi_pos_move_to_uci(self._h, move, buf, 16)
        return buf.value.decode()

    def move_san(self, move: int) -> str:
        buf = C.create_string_buffer(16)
        if self._lib.mi_pos_move_to_san(self._h, move, buf, 16) < 0:
            raise ValueError(_capi.last_error())
        return buf.value.decode()

    def legal_uci(self) -> List[str]:
        return sorted(self.move_uci(m) for m in self.legal_moves())

    def uci_to_move(self, uci: str) -> int:
        return self._lib.mi_pos_uci_to_move(self._h, uci.encode())

    def push(self, move: int) -> None:
        if self._lib.mi_pos_do_move(self._h, move):
            raise ValueError(_capi.last_error())

    def push_uci(self, uci: str) -> bool:
        m = self.uci_to_move(uci)
        if not m:
            return False
        self.push(m)
        return True

    def san_to_move(self, san: str) -> int:
        """The legal move whose SAN (Board::pgn_move form, mi_pos_move_to_san) is `san`; check / mate marks and "e.p." are ignored, 0 if
        none matches.  The inverse a PGN opening book needs (zh-50_startpos.pgn)."""
        want = san.rstrip("+#!?").replace("e.p.", "").strip()
        for m in self.legal_moves():
            if self.move_san(m).rstrip("+#") == want:
                return m
        return 0

    def push_san(self, san: str) -> bool:
        m = self.san_to_move(san)
        if not m:
            return False
        self.push(m)
        return True

    def terminal(self) -> int:
        return self._lib.mi_pos_terminal(self._h)

    def game_phase(self, num_phases: int = 3, definition: int = 0) -> int:
        """Board::get_phase (board.cpp:540-587): definition 0 lichess (0 / 1 / 2), 1 movecount."""
        return self._lib.mi_pos_game_phase(self._h, int(num_phases), int(definition))

    def in_check(self) -> bool:
        return bool(self._lib.mi_pos_in_check(self._h))

    def insufficient_material(self) -> bool:
        return bool(self._lib.mi_pos_insufficient_material(self._h))

    def steps_from_null(self) -> int:
        return self._lib.mi_pos_plies_from_null(self._h)

    def number_repetitions(self) -> int:
        return self._lib.mi_pos_number_repetitions(self._h)

    def perft(self, depth: int) -> int:
        return int(self._lib.mi_pos_perft(self._h, depth))

    def planes(self, mode: int, version_major, normalize: bool = True, repetitions: int = -1) -> np.ndarray:
        """version_major: the major number, or "maj.min" (chess "2.7" / "2.8" are distinct layouts)."""
        layout = planes_layout(mode, version_major)
        c = self._lib.mi_planes_channels(layout)
        out = np.empty((c, 8, 8), np.float32)
        if self._lib.mi_pos_planes(self._h, layout, int(normalize), repetitions, out.ctypes.data):
            raise RuntimeError(_capi.last_error())
        return out

    def desc(self, layout: int = -1) -> bytes:
        """192-byte board descriptor; pass the layout when it reads the legal-move features (chess 2.7 / 2.8)."""
        buf = C.create_string_buffer(192)
        if (self._lib.mi_pos_desc(self._h, buf) if layout < 0 else self._lib.mi_pos_desc_for(self._h, layout, buf)):
            raise RuntimeError(_capi.last_error())
        return buf.raw

    def policy_index(self, move, mode: int, is_policy_map: bool = True) -> int:
        if isinstance(move, str):
            move = self.uci_to_move(move)
        return self._lib.mi_pos_policy_index(self._h, move, mode, int(is_policy_map))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi_pos_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def policy_tables(mode: int):
    lib = _capi.load()
    n = lib.mi_policy_nb_labels(mode)
    labels = [lib.mi_policy_label(mode, i, 0).decode() for i in range(n)]
    mirrored = [lib.mi_policy_label(mode, i, 1).decode() for i in range(n)]
    flat = [lib.mi_policy_flat_plane_idx(mode, i) for i in range(n)]
    return labels, mirrored, flat


def planes_from_descs_device(descs: bytes, n: int, layout: int, normalize: bool, d_planes_ptr: int, device_id: int = 0):
    lib = _capi.load()
    if lib.mi_planes_from_descs_device(descs, n, layout, int(normalize), d_planes_ptr, device_id):
        raise RuntimeError(_capi.last_error())


def planes_from_descs_host(descs: bytes, n: int, layout: int, normalize: bool = True) -> np.ndarray:
    """n 192-byte descriptors -> float planes [n][C][8][8] on the host (the input of a CPU net behind the same leaf collector)."""
    lib = _capi.load()
    c = lib.mi_planes_channels(layout)
    out = np.empty((n, c, 8, 8), np.float32)
    if lib.mi_planes_from_descs_host(descs, n, layout, int(normalize), out.ctypes.data_as(_capi.c_float_p)):
        raise RuntimeError(_capi.last_error())
    return out
