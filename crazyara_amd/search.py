"""Python handle on the native MCTS search pool (C ABI mi_search_*): many trees feeding shared GPU batches."""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional

import numpy as np

from . import _capi


class SearchSettingsC(C.Structure):
    _fields_ = [("batch_size", C.c_int), ("cpuct_init", C.c_float), ("cpuct_base", C.c_float),
                ("node_policy_temperature", C.c_float), ("virtual_style", C.c_int), ("virtual_mix_threshold", C.c_uint),
                ("virtual_offset_strength", C.c_double), ("q_value_weight", C.c_float), ("q_veto_delta", C.c_float),
                ("mode", C.c_int), ("version_major", C.c_int), ("is_policy_map", C.c_int), ("clone_keeps_last_moves", C.c_int),
                ("epsilon_greedy_counter", C.c_int), ("epsilon_checks_counter", C.c_int), ("seed", C.c_uint), ("mcts_solver", C.c_int),
                ("dirichlet_epsilon", C.c_float), ("dirichlet_alpha", C.c_float), ("version_minor", C.c_int)]


class GoLimitsC(C.Structure):
    """mi_go_limits: the SearchLimits fields TimeManager::get_time_for_move reads."""
    _fields_ = [("movetime", C.c_longlong), ("nodes", C.c_ulonglong), ("simulations", C.c_ulonglong), ("movestogo", C.c_int),
                ("depth", C.c_int), ("time", C.c_int * 2), ("inc", C.c_int * 2), ("move_overhead", C.c_int), ("infinite", C.c_int)]


def time_for_move(side: int, move_number: int, *, movetime=0, nodes=0, simulations=0, movestogo=0, depth=0, wtime=0, btime=0,
                  winc=0, binc=0, move_overhead=20, infinite=False) -> int:
    """Milliseconds a `go` with these clock arguments searches (TimeManager::get_time_for_move); 0 = no time limit."""
    lim = GoLimitsC(movetime, nodes, simulations, movestogo, depth, (C.c_int * 2)(wtime, btime), (C.c_int * 2)(winc, binc),
                    move_overhead, int(infinite))
    r = _capi.load().mi_time_for_move(C.byref(lim), side, move_number)
    if r < 0:
        raise ValueError(_capi.last_error())
    return r


class SearchStatsC(C.Structure):
    _fields_ = [("nodes", C.c_ulonglong), ("nn_evals", C.c_ulonglong), ("batches", C.c_ulonglong), ("simulations", C.c_ulonglong),
                ("seconds", C.c_double), ("depth_avg", C.c_double), ("depth_max", C.c_uint)]


EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))


def default_settings(**overrides) -> SearchSettingsC:
    s = SearchSettingsC()
    _capi.load().mi_search_default_settings(C.byref(s))
    for k, v in overrides.items():
        setattr(s, k, v)
    return s


class SearchPool:
    def __init__(self, settings: SearchSettingsC, net_a=None, net_b=None,
                 eval_fn: Optional[Callable] = None, fn_batch: int = 0, fn_nb_policy: int = 0):
        self._lib = _capi.load()
        self._cb = None
        if eval_fn is not None:
            def _tramp(user, descs, n, value, probs):
                try:
                    raw = C.string_at(descs, n * 192)
                    v, p = eval_fn([raw[i * 192:(i + 1) * 192] for i in range(n)])
                    np.ctypeslib.as_array(value, shape=(n,))[:] = np.asarray(v, np.float32)
                    np.ctypeslib.as_array(probs, shape=(n, fn_nb_policy))[:] = np.asarray(p, np.float32)
                    return 0
                except Exception as e:  # noqa: BLE001
                    print("evaluator callback raised:", repr(e))
                    return 1
            self._cb = EVAL_FN(_tramp)
        self._nets = [net_a, net_b]
        self._h = self._lib.mi_search_create(C.byref(settings), net_a._h if net_a else None, net_b._h if net_b else None,
                                             self._cb if self._cb else C.cast(None, EVAL_FN), None, fn_batch, fn_nb_policy)
        if not self._h:
            raise RuntimeError(_capi.last_error())

    def add_position(self, fen: str = "", is960: bool = False, variant: str = "crazyhouse") -> int:
        t = self._lib.mi_search_add_position(self._h, (fen or "").encode(), int(is960), variant.encode())
        if t < 0:
            raise ValueError(_capi.last_error())
        return t

    def run(self, simulations: int = 0, nodes: int = 0, threads: int = 1, movetime_ms: int = 0) -> SearchStatsC:
        """Searches every active tree to its limits (absolute on the root's counters, SearchThread::nodes_limits_ok); movetime_ms > 0
        also ends the searches that long after the start of the call (SearchLimits::movetime).  ctypes releases the GIL for the
        call, so `stop()` may come from another Python thread."""
        st = SearchStatsC()
        if self._lib.mi_search_run_timed(self._h, simulations, nodes, movetime_ms, threads, C.byref(st)):
            raise RuntimeError(_capi.last_error())
        return st

    def announce_go(self) -> None:
        """The commanding thread's half of `go`: from now on a `stop` belongs to the coming `run`, also when it arrives before the
        search thread has entered it (mi_search_announce_go)."""
        self._lib.mi_search_announce_go(self._h)

    def cancel_go(self) -> bool:
        """Withdraws the oldest announced search that no `run` has taken (the commanding thread dropped it); True if one was waiting
        (mi_search_cancel_go)."""
        return self._lib.mi_search_cancel_go(self._h) == 1

    def stop(self) -> None:
        """Ends the `run` that is executing in another thread or has been announced (SearchThread::stop); no effect otherwise."""
        self._lib.mi_search_stop(self._h)

    def root_children(self, tree: int):
        cap = 512
        moves = (C.c_uint32 * cap)()
        visits = (C.c_uint32 * cap)()
        q = (C.c_float * cap)()
        pri = (C.c_float * cap)()
        n = self._lib.mi_search_root_children(self._h, tree, cap, moves, visits, q, pri)
        return list(moves[:n]), list(visits[:n]), np.array(q[:n], np.float32), np.array(pri[:n], np.float32)

    def tree_info(self, tree: int):
        rv, nc, alloc, val = C.c_uint(), C.c_uint(), C.c_uint(), C.c_float()
        self._lib.mi_search_tree_info(self._h, tree, C.byref(rv), C.byref(nc), C.byref(alloc), C.byref(val))
        return dict(root_visits=rv.value, node_count=nc.value, allocated=alloc.value, root_value=val.value)

    def apply_move(self, tree: int, uci: str) -> bool:
        """A move was played on the tree's board: keep the searched subtree below it (True) or restart from the new position."""
        kept = C.c_int()
        if self._lib.mi_search_apply_move(self._h, tree, uci.encode(), C.byref(kept)):
            raise ValueError(_capi.last_error())
        return bool(kept.value)

    def fen(self, tree: int) -> str:
        buf = C.create_string_buffer(256)
        if self._lib.mi_search_tree_fen(self._h, tree, buf, 256):
            raise RuntimeError(_capi.last_error())
        return buf.value.decode()

    def root_policy(self, tree: int):
        """(policy over the expanded root children as Node::get_mcts_policy returns it, Q of the best move)."""
        buf = (C.c_double * 512)()
        q = C.c_float()
        n = self._lib.mi_search_root_policy(self._h, tree, 512, buf, C.byref(q))
        if n < 0:
            raise RuntimeError(_capi.last_error())
        return np.array(buf[:n], np.float64), float(q.value)

    def pv(self, tree: int) -> dict:
        """EvalInfo pv[0] / centipawns[0] / movesToMate[0] of the tree (what a UCI front end prints)."""
        buf = C.create_string_buffer(4096)
        cp, mate = C.c_int(), C.c_int()
        n = self._lib.mi_search_pv(self._h, tree, buf, 4096, C.byref(cp), C.byref(mate))
        if n < 0:
            raise RuntimeError(_capi.last_error())
        return dict(pv=buf.value.decode().split(), centipawns=cp.value, moves_to_mate=mate.value)

    def pv_multi(self, tree: int, multipv: int) -> list:
        """The lines of a Multi_PV output (EvalInfo pv / bestMoveQ / centipawns / movesToMate per line)."""
        out = []
        for idx in range(multipv):
            buf = C.create_string_buffer(4096)
            cp, mate, q = C.c_int(), C.c_int(), C.c_float()
            n = self._lib.mi_search_pv_multi(self._h, tree, idx, multipv, buf, 4096, C.byref(cp), C.byref(mate), C.byref(q))
            if n < 0:
                raise RuntimeError(_capi.last_error())
            if n == 0:
                break
            out.append(dict(pv=buf.value.decode().split(), centipawns=cp.value, moves_to_mate=mate.value, best_move_q=q.value))
        return out

    def tree_dump(self, tree: int) -> np.ndarray:
        """The whole tree as the flat word list of mi_search_tree_dump (depth-first records of every selected node)."""
        cap = 1 << 22
        buf = (C.c_uint32 * cap)()
        n = self._lib.mi_search_tree_dump(self._h, tree, buf, cap)
        if n < 0:
            raise RuntimeError(_capi.last_error())
        return np.ctypeslib.as_array(buf)[:n].copy()

    def debug_replay(self):
        """(differing words, report) of mi_search_debug_replay: the lanes' recorded batches evaluated again, alone on the device, and
        compared bit for bit with what the searches consumed.  Needs CRA_LANE_RECORD=1 in the environment when the pool is made."""
        cap = 1 << 17
        buf = C.create_string_buffer(cap)
        n = self._lib.mi_search_debug_replay(self._h, buf, cap)
        if n < 0:
            raise RuntimeError(_capi.last_error())
        return int(n), buf.value.decode()

    def set_shared_collectors(self, k: int) -> None:
        """k >= 1 collectors per tree in every lane (SearchThreads sharing one tree, crazyara.cpp:555-561): the single-`go` mode.
        0 = many-trees mode (one collector per tree, each tree in one lane)."""
        if self._lib.mi_search_set_shared_collectors(self._h, int(k)):
            raise ValueError(_capi.last_error())

    def set_state_budget(self, budget: int) -> None:
        """Stored leaf states per tree (the reference's MCTS_STORE_STATES): 0 = every simulation replays its path from the root."""
        if self._lib.mi_search_set_state_budget(self._h, int(budget)):
            raise RuntimeError(_capi.last_error())

    def set_adaptive_quota(self, cap: int) -> None:
        """cap > 0: the trees of a lane that are still searching share the whole batch (throughput setting for self-play); 0 = fixed quota"""
        if self._lib.mi_search_set_adaptive_quota(self._h, int(cap)):
            raise RuntimeError(_capi.last_error())

    def set_active(self, tree: int, active: bool) -> None:
        if self._lib.mi_search_set_active(self._h, tree, int(active)):
            raise ValueError(_capi.last_error())

    def reset_position(self, tree: int, fen: str = "", is960: bool = False, variant: str = "crazyhouse") -> None:
        if self._lib.mi_search_reset_position(self._h, tree, (fen or "").encode(), int(is960), variant.encode()):
            raise ValueError(_capi.last_error())

    def add_lane(self, net) -> None:
        """One more evaluator lane (call before add_position): one more batch in flight."""
        if self._lib.mi_search_add_lane(self._h, net._h):
            raise RuntimeError(_capi.last_error())
        self._nets.append(net)

    def root_solved(self, tree: int) -> dict:
        """Solver verdict on the root: node_type 0 WIN / 1 DRAW / 2 LOSS / 6 UNSOLVED, plies to the end, mating child index."""
        nt, ply, mate = C.c_int(), C.c_int(), C.c_int()
        if self._lib.mi_search_root_solved(self._h, tree, C.byref(nt), C.byref(ply), C.byref(mate)):
            raise RuntimeError(_capi.last_error())
        return dict(node_type=nt.value, end_in_ply=ply.value, checkmate_idx=mate.value)

    def best_move(self, tree: int) -> str:
        buf = C.create_string_buffer(16)
        if self._lib.mi_search_best_move(self._h, tree, buf, 16):
            raise RuntimeError(_capi.last_error())
        return buf.value.decode()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi_search_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
