"""Host-side mirror of the reference's NN plugin surface on top of the C ABI.

`HipAPI` has the public interface of `NeuralNetAPI` (engine/src/nn/neuralnetapi.h:148-311) as implemented by
`TensorrtAPI` (engine/src/nn/tensorrtapi.cpp:43-237): same constructor arguments, same getters, same `predict`
contract (whole fixed batch, fp32 NCHW planes in, value[B] (tanh range) / policy[B*nbPolicy] (softmaxed) / aux out,
blocking).  `NeuralNetAPIUser` mirrors the buffer-owning base class (engine/src/nn/neuralnetapiuser.cpp:34-109).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional

import numpy as np

from . import _capi

VERSION_SEP_MAJOR = 1000000  # engine/src/version.h:37-38
VERSION_SEP_MINOR = 1000


def make_version(major: int, minor: int, patch: int = 0) -> int:
    return major * VERSION_SEP_MAJOR + minor * VERSION_SEP_MINOR + patch


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def calibrate_int8(model_directory: str, device_id: int = 0, planes=None) -> None:
    """mi_net_calibrate_int8: writes <model file>.int8calib beside the model.  planes: float32 [n][C][8][8] calibration boards, or None
    for the plies of the reference's calibration games (chessbatchstream.cpp:44-94)."""
    lib = _capi.load()
    if planes is None:
        rc = lib.mi_net_calibrate_int8(model_directory.encode(), int(device_id), None, 0)
    else:
        import numpy as np
        arr = np.ascontiguousarray(planes, dtype=np.float32)
        rc = lib.mi_net_calibrate_int8(model_directory.encode(), int(device_id), arr.ctypes.data_as(C.c_void_p), int(arr.shape[0]))
    if rc:
        raise RuntimeError(_capi.last_error())


class HipAPI:
    def __init__(self, device_id: int, batch_size: int, model_directory: str, precision: str = "float16", keep_logits: bool = False):
        self._lib = _capi.load()
        self.precision_requested = precision
        if precision == "int8" and self._lib.mi_net_has_int8_calibration(model_directory.encode()) == 0:
            # the option layer (integration/hipapi.h does the same; TensorRT runs its calibrator when the engine cache is missing,
            # tensorrtapi.cpp:297-360): Precision int8 on a model without a calibration file calibrates it first, on the plies of the
            # reference's calibration games
            print("info string HipAPI: run INT8 quantization calibration", file=sys.stderr)
            if self._lib.mi_net_calibrate_int8(model_directory.encode(), int(device_id), None, 0):
                raise RuntimeError(_capi.last_error())
        self._h = self._lib.mi_net_create(model_directory.encode(), int(device_id), int(batch_size), precision.encode())
        if not self._h:
            msg = _capi.last_error()
            # the reference throws invalid_argument / runtime_error from the constructor (neuralnetapi.cpp:65-70,173)
            raise (ValueError if "directory" in msg or "precision" in msg or "batch" in msg else RuntimeError)(msg)
        if keep_logits:      # tests / analysis: forwards also leave policy_out (pre-softmax) in device_buffers()["logits"]
            self._lib.mi_net_keep_logits(self._h, 1)
        shape = (C.c_int * 4)()
        npol, naux, ver, phase = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._lib.mi_net_design(self._h, shape, C.byref(npol), C.byref(naux), C.byref(ver), C.byref(phase))
        self.input_shape = tuple(shape)
        self._nb_policy, self._nb_aux, self._version, self._phase = npol.value, naux.value, ver.value, phase.value
        self.device_id = device_id
        self.precision = precision

    # ---- NeuralNetAPI getters (neuralnetapi.h:207-299) ----
    def get_batch_size(self) -> int:
        return self.input_shape[0]

    def get_nb_input_values_total(self) -> int:
        return self.input_shape[1] * self.input_shape[2] * self.input_shape[3]

    def get_nb_policy_values(self) -> int:
        return self._nb_policy

    def get_policy_output_length(self) -> int:
        return self._nb_policy * self.get_batch_size()

    def get_nb_auxiliary_outputs(self) -> int:
        return self._nb_aux

    def has_auxiliary_outputs(self) -> bool:
        return self._nb_aux > 0

    def get_version(self) -> int:
        return self._version

    def get_game_phase(self) -> int:
        return self._phase

    def get_model_name(self) -> str:
        return self._lib.mi_net_model_name(self._h).decode()

    def get_device_name(self) -> str:
        return f"gpu_{self.device_id}"

    def is_policy_map(self, nb_labels: int) -> bool:
        """isPolicyMap := policyLen != NB_LABELS (tensorrtapi.cpp:157)."""
        return self._nb_policy != nb_labels

    def flops_per_position(self) -> float:
        return float(self._lib.mi_net_flops_per_position(self._h))

    # ---- predict (neuralnetapi.h:230-237) ----
    @staticmethod
    def _ptr(a) -> int:
        if a is None:
            return 0
        if isinstance(a, np.ndarray):
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
            return a.ctypes.data
        return int(a)  # raw address (e.g. from mi_host_alloc)

    def predict(self, input_planes, value_output, prob_outputs, auxiliary_outputs=None) -> None:
        if self._lib.mi_net_predict(self._h, self._ptr(input_planes), self._ptr(value_output), self._ptr(prob_outputs),
                                    self._ptr(auxiliary_outputs)):
            raise RuntimeError(_capi.last_error())

    def submit(self, input_planes, value_output, prob_outputs, auxiliary_outputs=None) -> None:
        if self._lib.mi_net_submit(self._h, self._ptr(input_planes), self._ptr(value_output), self._ptr(prob_outputs),
                                   self._ptr(auxiliary_outputs)):
            raise RuntimeError(_capi.last_error())

    def wait(self) -> None:
        if self._lib.mi_net_wait(self._h):
            raise RuntimeError(_capi.last_error())

    def last_submit_zero_copy(self) -> bool:
        """True if the last predict / submit found all its buffers pinned and issued no copy commands (mi_net_last_submit_zero_copy)."""
        return bool(self._lib.mi_net_last_submit_zero_copy(self._h))

    # ---- device-resident path ----
    def device_buffers(self):
        """dict of zero-copy device views usable with torch.as_tensor(view, device='cuda')."""
        p = [C.c_void_p() for _ in range(5)]
        self._lib.mi_net_device_buffers(self._h, *[C.byref(x) for x in p])
        B, Cc = self.input_shape[0], self.input_shape[1]
        out = {
            "planes": _DevArray(p[0].value, (B, Cc, 8, 8)),
            "value": _DevArray(p[1].value, (B,)),
            "probs": _DevArray(p[2].value, (B, self._nb_policy)),
            "logits": _DevArray(p[3].value, (B, self._nb_policy)),
        }
        if p[4].value:
            out["aux"] = _DevArray(p[4].value, (B, self._nb_aux))
        return out

    def block_dump(self):
        """Test hook (mi_net_block_dump): device view [blocks + 1][B][64][256] float16 of the residual stream in front of the tower's first
        block and behind every block, filled by every forward from now on."""
        n = C.c_int()
        p = self._lib.mi_net_block_dump(self._h, C.byref(n))
        if not p:
            raise RuntimeError(_capi.last_error())
        return _DevArray(p, (n.value, self.input_shape[0], 64, 256), "<f2")

    def forward_device(self) -> None:
        if self._lib.mi_net_forward_device(self._h):
            raise RuntimeError(_capi.last_error())

    def sync(self) -> None:
        if self._lib.mi_net_sync(self._h):
            raise RuntimeError(_capi.last_error())

    def time_forward(self, iters: int) -> float:
        ms = C.c_float()
        if self._lib.mi_net_time_forward(self._h, int(iters), C.byref(ms)):
            raise RuntimeError(_capi.last_error())
        return ms.value

    def time_ops(self, iters: int):
        n = self._lib.mi_net_op_count(self._h)
        names = (C.c_char_p * n)()
        ms = (C.c_float * n)()
        if self._lib.mi_net_time_ops(self._h, int(iters), names, ms):
            raise RuntimeError(_capi.last_error())
        return [(names[i].decode(), ms[i] / iters) for i in range(n)]

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.mi_net_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NeuralNetAPIUser:
    """Owns the pinned host I/O buffers sized from the first net (neuralnetapiuser.cpp:34-75) and runs the
    `inference` benchmark loop (neuralnetapiuser.cpp:104-109)."""

    def __init__(self, nets):
        self.nets = list(nets)
        net = self.nets[0]
        lib = _capi.load()
        self._lib = lib
        B = net.get_batch_size()

        def pinned(n):
            p = lib.mi_host_alloc(n * 4)
            if not p:
                raise RuntimeError(_capi.last_error())
            arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n,))
            return p, arr

        self._p_in, self.input_planes = pinned(B * net.get_nb_input_values_total())
        self._p_val, self.value_outputs = pinned(B)
        self._p_prob, self.prob_outputs = pinned(B * net.get_nb_policy_values())
        self._p_aux, self.auxiliary_outputs = (pinned(B * net.get_nb_auxiliary_outputs())
                                                if net.has_auxiliary_outputs() else (None, None))

    def run_inference(self, iterations: int) -> None:
        for _ in range(iterations):
            self.nets[0].predict(self.input_planes, self.value_outputs, self.prob_outputs, self.auxiliary_outputs)

    def close(self):
        for p in (self._p_in, self._p_val, self._p_prob, self._p_aux):
            if p:
                self._lib.mi_host_free(p)
        self._p_in = self._p_val = self._p_prob = self._p_aux = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
