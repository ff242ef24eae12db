"""
ORACLE (test infrastructure only -- never imported by the product path).

ctypes handle on oracle/_ref/libcrazyara_ref.so: the REFERENCE'S OWN MCTS code (engine/src/agents/mctsagent.cpp,
searchthread.cpp, node.{h,cpp}, nodedata.cpp, evalinfo.cpp, util/blazeutil.h ... compiled unmodified from /root/reference by
oracle/ref/build_ref.py) behind a few C entry points (oracle/ref/ref_driver.cpp).  It pins SURVEY 8a rows M1-M10: the tests run
one search through this library and through the product's `mi_search_*` with the same evaluator callback and compare the trees.

What is NOT the reference's in that library (see oracle/ref/shim/): the chess environment behind the State interface (this
repository's Position -- the reference's needs the absent Stockfish fork; it is pinned separately by the reference's own
known-answer tests) and the stand-in for the absent blaze headers (plain element-wise loops in the natural C++ promotion order).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable

import numpy as np

from crazyara_amd.search import EVAL_FN, SearchSettingsC      # the settings struct is shared by both sides of the comparison

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libcrazyara_ref.so")
_lib = None


def available() -> bool:
    """True when the library exists (prebuilt) or can be built (reference present)."""
    from oracle.ref import build_ref
    return os.path.exists(LIB_PATH) or build_ref.reference_present()


def load():
    global _lib
    if _lib is not None:
        return _lib
    from oracle.ref import build_ref
    path = build_ref.build()
    if path is None:
        raise RuntimeError("oracle/_ref is not built and /root/reference is absent")
    lib = C.CDLL(path)
    _declare(lib)
    _lib = lib
    return lib


def _declare(lib):
    lib.ref_last_error.restype = C.c_char_p
    lib.ref_agent_create.restype = C.c_void_p
    lib.ref_agent_create.argtypes = [C.POINTER(SearchSettingsC), EVAL_FN, C.c_void_p, C.c_int]
    lib.ref_agent_destroy.argtypes = [C.c_void_p]
    lib.ref_set_use_mcgs.argtypes = [C.c_int]
    lib.ref_set_use_mcgs.restype = None
    lib.ref_agent_set_position.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p]
    lib.ref_agent_go.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    lib.ref_agent_apply_move.argtypes = [C.c_void_p, C.c_char_p]
    lib.ref_agent_fen.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.ref_agent_root_children.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                            C.POINTER(C.c_float)]
    lib.ref_agent_root_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint)] * 2 + [C.POINTER(C.c_float)] + [C.POINTER(C.c_int)] * 3 + \
        [C.POINTER(C.c_uint), C.POINTER(C.c_int)]
    lib.ref_time_for_move.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.ref_time_for_move.restype = C.c_int
    lib.ref_agent_set_multipv.argtypes = [C.c_void_p, C.c_int]
    lib.ref_agent_set_multipv.restype = None
    lib.ref_agent_pv_at.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    lib.ref_agent_pv_at.restype = C.c_int
    lib.ref_agent_pv.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ref_agent_pv.restype = C.c_int
    lib.ref_agent_eval.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_int, C.POINTER(C.c_float),
                                   C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    lib.ref_agent_tree_dump.restype = C.c_long
    lib.ref_agent_tree_dump.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_long]
    lib.ref_agent_net_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    lib.ref_get_current_cput.restype = C.c_float
    lib.ref_get_current_cput.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.ref_first_and_second_max.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ref_get_quantile.restype = C.c_float
    lib.ref_get_quantile.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_float]
    lib.ref_apply_quantile_clipping.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_float]
    lib.ref_sharpen_distribution.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_float]
    lib.ref_value_to_centipawn.restype = C.c_int
    lib.ref_value_to_centipawn.argtypes = [C.c_float]


LIB_HIP_PATH = os.path.join(HERE, "_ref", "libcrazyara_ref_hip.so")


def hip_available() -> bool:
    from oracle.ref import build_ref
    return os.path.exists(LIB_HIP_PATH) or build_ref.reference_present()


LIB_HIP_RELEASE_PATH = os.path.join(HERE, "_ref", "libcrazyara_ref_hip_release.so")
# the reference's SearchThread with integration/searchthread_hip.patch applied (descriptor-fed batches, priors gathered on the GPU)
LIB_HIP_PATCHED_PATH = os.path.join(HERE, "_ref", "libcrazyara_ref_hip_patched.so")
LIB_HIP_PATCHED_RELEASE_PATH = os.path.join(HERE, "_ref", "libcrazyara_ref_hip_patched_release.so")
_libs_hip = {}


def _declare_hip(lib):
    lib.ref_agent_create_hip.restype = C.c_void_p
    lib.ref_agent_create_hip.argtypes = [C.POINTER(SearchSettingsC), C.c_char_p, C.c_int, C.c_char_p]
    lib.ref_agent_create_hip_threads.restype = C.c_void_p
    lib.ref_agent_create_hip_threads.argtypes = [C.POINTER(SearchSettingsC), C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    lib.ref_hipapi_create.restype = C.c_void_p
    lib.ref_hipapi_create.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.c_char_p, C.c_int]
    lib.ref_hipapi_destroy.argtypes = [C.c_void_p]
    lib.ref_hipapi_info.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
    lib.ref_hipapi_model_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.ref_hipapi_validate.argtypes = [C.c_void_p]
    lib.ref_hipapi_run_inference.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.POINTER(C.c_float), C.POINTER(C.c_double)]


def hip_patched_available() -> bool:
    from oracle.ref import build_ref
    return os.path.exists(LIB_HIP_PATCHED_PATH) or build_ref.reference_present()


def load_hip(release: bool = False, patched: bool = False):
    """The same library with integration/hipapi.h compiled in (the reference-side binding of the product library).
    release = True: the -O3 -DNDEBUG build of the same sources (the reference's Release configuration), for throughput measurements.
    patched = True: the build whose SearchThread carries integration/searchthread_hip.patch (every library has its own copy of the
    reference's globals, so agents of the plain and of the patched build live side by side in one process)."""
    key = (bool(release), bool(patched))
    if key in _libs_hip:
        return _libs_hip[key]
    path = {(False, False): LIB_HIP_PATH, (True, False): LIB_HIP_RELEASE_PATH,
            (False, True): LIB_HIP_PATCHED_PATH, (True, True): LIB_HIP_PATCHED_RELEASE_PATH}[key]
    from oracle.ref import build_ref
    build_ref.build()
    if not os.path.exists(path):
        raise RuntimeError(f"oracle/_ref/{os.path.basename(path)} is not built")
    from crazyara_amd import _capi
    _capi.load()                                  # the product library first (RTLD_GLOBAL), then the binding that links against it
    lib = C.CDLL(path)
    _declare(lib)
    _declare_hip(lib)
    _libs_hip[key] = lib
    return lib


def _err(lib=None):
    return ((lib or load()).ref_last_error() or b"").decode()


class RefHipAPI:
    """integration/hipapi.h's HipAPI constructed and used through the reference's NeuralNetAPI base class / NeuralNetAPIUser."""

    def __init__(self, model_dir: str, device_id: int, batch: int, precision: str, mode: int):
        self._lib = load_hip()
        self._h = self._lib.ref_hipapi_create(model_dir.encode(), device_id, batch, precision.encode(), mode)
        if not self._h:
            raise RuntimeError(_err(self._lib))

    def info(self) -> dict:
        out = (C.c_long * 8)()
        self._lib.ref_hipapi_info(self._h, out)
        keys = ("version", "is_policy_map", "nb_input_values_total", "nb_policy_values", "batch_size", "nb_auxiliary_outputs",
                "has_auxiliary_outputs", "game_phase")
        return dict(zip(keys, [int(v) for v in out]))

    def model_name(self) -> str:
        buf = C.create_string_buffer(512)
        self._lib.ref_hipapi_model_name(self._h, buf, 512)
        return buf.value.decode()

    def validate(self) -> int:
        return self._lib.ref_hipapi_validate(self._h)

    def run_inference(self, planes: np.ndarray, iterations: int = 1):
        i = self.info()
        B = i["batch_size"]
        planes = np.ascontiguousarray(planes, np.float32)
        assert planes.size == B * i["nb_input_values_total"]
        value, probs = np.zeros(B, np.float32), np.zeros(B * i["nb_policy_values"], np.float32)
        aux = np.zeros(max(1, B * i["nb_auxiliary_outputs"]), np.float32)
        sec = C.c_double()
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        if self._lib.ref_hipapi_run_inference(self._h, iterations, fp(planes), fp(value), fp(probs), fp(aux), C.byref(sec)):
            raise RuntimeError(_err(self._lib))
        return value, probs.reshape(B, -1), aux, sec.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ref_hipapi_destroy(self._h)
            self._h = None


def set_use_mcgs(on: bool) -> None:
    """SearchSettings::useMCGS of the agents created from now on (UCI Search_Type mcgs | mcts, crazyara.cpp:736); default False."""
    load().ref_set_use_mcgs(1 if on else 0)


class RefAgent:
    """One MCTSAgent of the reference (one SearchThread, one tree) with a callback evaluator.

    eval_fn(list of 192-byte board descriptors) -> (values, probs[n][nb_policy]) -- the signature the product's callback lane uses."""

    def __init__(self, settings: SearchSettingsC, eval_fn: Callable = None, nb_policy: int = 0, hip_model_dir: str = None,
                 device_id: int = 0, precision: str = "float16", threads: int = 1, release: bool = False, patched: bool = False):
        self.nb_policy = nb_policy
        if hip_model_dir is not None:                       # the agent's nets are HipAPI objects: `go` evaluates on the GPU
            self._lib = load_hip(release, patched)
            self._cb = None
            # threads = the UCI option `Threads`: that many SearchThreads (own batch net each) on the one tree, crazyara.cpp:548-563
            self._h = self._lib.ref_agent_create_hip_threads(C.byref(settings), hip_model_dir.encode(), device_id, precision.encode(), int(threads))
            if not self._h:
                raise RuntimeError(_err(self._lib))
            return
        self._lib = load()

        def _tramp(user, descs, n, value, probs):
            try:
                raw = C.string_at(descs, n * 192)
                v, p = eval_fn([raw[i * 192:(i + 1) * 192] for i in range(n)])
                np.ctypeslib.as_array(value, shape=(n,))[:] = np.asarray(v, np.float32)
                np.ctypeslib.as_array(probs, shape=(n, nb_policy))[:] = np.asarray(p, np.float32)
                return 0
            except Exception as e:  # noqa: BLE001
                print("evaluator callback raised:", repr(e))
                return 1
        self._cb = EVAL_FN(_tramp)
        self._h = self._lib.ref_agent_create(C.byref(settings), self._cb, None, nb_policy)
        if not self._h:
            raise RuntimeError(_err(self._lib))

    def set_position(self, fen: str = "", is960: bool = False, variant: str = "crazyhouse"):
        if self._lib.ref_agent_set_position(self._h, (fen or "").encode(), int(is960), variant.encode()):
            raise ValueError(_err(self._lib))

    def go(self, simulations: int = 0, nodes: int = 0):
        if self._lib.ref_agent_go(self._h, simulations, nodes):
            raise RuntimeError(_err(self._lib))

    def apply_move(self, uci: str):
        if self._lib.ref_agent_apply_move(self._h, uci.encode()):
            raise ValueError(_err(self._lib))

    def fen(self) -> str:
        buf = C.create_string_buffer(256)
        self._lib.ref_agent_fen(self._h, buf, 256)
        return buf.value.decode()

    def root_children(self):
        cap = 512
        moves, visits = (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
        q, pri = (C.c_float * cap)(), (C.c_float * cap)()
        n = self._lib.ref_agent_root_children(self._h, cap, moves, visits, q, pri)
        return list(moves[:n]), list(visits[:n]), np.array(q[:n], np.float32), np.array(pri[:n], np.float32)

    def root_info(self) -> dict:
        rv, nc, fv = C.c_uint(), C.c_uint(), C.c_uint()
        val = C.c_float()
        nt, ply, mate, nl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if self._lib.ref_agent_root_info(self._h, C.byref(rv), C.byref(nc), C.byref(val), C.byref(nt), C.byref(ply), C.byref(mate),
                                         C.byref(fv), C.byref(nl)):
            raise RuntimeError("no searched root")
        return dict(root_visits=rv.value, node_count=nc.value, root_value=val.value, node_type=nt.value, end_in_ply=ply.value,
                    checkmate_idx=mate.value, free_visits=fv.value, n_legal=nl.value)

    def eval_info(self) -> dict:
        pol = (C.c_double * 512)()
        uci = C.create_string_buffer(16)
        q = C.c_float()
        nodes, sel = C.c_uint(), C.c_uint()
        n = self._lib.ref_agent_eval(self._h, 512, pol, uci, 16, C.byref(q), C.byref(nodes), C.byref(sel))
        if n < 0:
            raise RuntimeError("eval info")
        return dict(policy=np.array(pol[:n], np.float64), best_move=uci.value.decode(), best_q=float(q.value), nodes=nodes.value,
                    sel_depth=sel.value)

    def set_multipv(self, k: int) -> None:
        self._lib.ref_agent_set_multipv(self._h, k)

    def pv_multi(self, multipv: int) -> list:
        out = []
        for idx in range(multipv):
            buf = C.create_string_buffer(4096)
            cp, mate, q = C.c_int(), C.c_int(), C.c_float()
            n = self._lib.ref_agent_pv_at(self._h, idx, buf, 4096, C.byref(cp), C.byref(mate), C.byref(q))
            if n <= 0:
                break
            out.append(dict(pv=buf.value.decode().split(), centipawns=cp.value, moves_to_mate=mate.value, best_move_q=q.value))
        return out

    def pv(self) -> dict:
        buf = C.create_string_buffer(4096)
        cp, mate = C.c_int(), C.c_int()
        n = self._lib.ref_agent_pv(self._h, buf, 4096, C.byref(cp), C.byref(mate))
        if n < 0:
            raise RuntimeError("pv")
        return dict(pv=buf.value.decode().split(), centipawns=cp.value, moves_to_mate=mate.value)

    def tree_dump(self) -> np.ndarray:
        cap = 1 << 22
        buf = (C.c_uint32 * cap)()
        n = self._lib.ref_agent_tree_dump(self._h, buf, cap)
        if n < 0:
            raise RuntimeError("tree dump buffer too small")
        return np.ctypeslib.as_array(buf)[:n].copy()

    def net_counters(self) -> dict:
        c = (C.c_ulonglong * 4)()
        self._lib.ref_agent_net_counters(self._h, c)
        return dict(root_calls=c[0], root_evals=c[1], batch_calls=c[2], batch_evals=c[3])

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ref_agent_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def time_for_move(limits, side: int, move_number: int) -> int:
    """TimeManager::get_time_for_move of the reference build on a crazyara_amd.search.GoLimitsC."""
    return load().ref_time_for_move(C.byref(limits), side, move_number)
