"""ctypes binding of include/crazyara_hip.h (the C ABI a cgo/JNI/C++ shim would bind)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcrazyara_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)

# name -> (restype, argtypes); kept in one table so tests can check that every symbol of the header is exported
SIGNATURES = {
    "mi_last_error": (C.c_char_p, []),
    "mi_version": (C.c_char_p, []),
    "mi_device_count": (C.c_int, []),
    "mi_host_alloc": (C.c_void_p, [C.c_size_t]),
    "mi_host_free": (None, [C.c_void_p]),
    "mi_net_create": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.c_char_p]),
    "mi_net_destroy": (None, [C.c_void_p]),
    "mi_net_design": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "mi_net_model_name": (C.c_char_p, [C.c_void_p]),
    "mi_net_flops_per_position": (C.c_double, [C.c_void_p]),
    "mi_net_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_net_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_net_wait": (C.c_int, [C.c_void_p]),
    "mi_net_device_buffers": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_void_p)] * 5),
    "mi_net_forward_device": (C.c_int, [C.c_void_p]),
    "mi_net_sync": (C.c_int, [C.c_void_p]),
    "mi_net_stream": (C.c_void_p, [C.c_void_p]),
    "mi_net_time_forward": (C.c_int, [C.c_void_p, C.c_int, c_float_p]),
    "mi_net_op_count": (C.c_int, [C.c_void_p]),
    "mi_net_time_ops": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), c_float_p]),
}

_lib = None


def load():
    """Loads the HIP library; raises (never falls back) when it is absent or lacks a declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m crazyara_amd.build` (hipcc, gfx950). "
            "crazyara_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().mi_last_error().decode()
