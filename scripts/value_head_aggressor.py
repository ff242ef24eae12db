"""Which kernel has to share the compute unit for value_head_kernel's FC1 partial sums to come out wrong?  (profiles/NOTES.md, round 4)

Net A (float16x3, one-launch value head, stage checksums on) runs ONE full forward, then only its value head launch again and again on
its own stream, each launch checked against the first; net B loops ONE of its ops on another stream.  Prints, per op of B, how many of
A's launches differed.   python scripts/value_head_aggressor.py [launches]
"""
import ctypes as C
import json
import os
import sys
import tempfile
import threading

os.environ["CRA_X3_VALUE_HEAD"] = "one"
os.environ["CRA_VALUE_HEAD_DEBUG"] = "1"
os.environ.setdefault("CRA_VALUE_HEAD_LDS_PAD", "-1")      # the round-3 form that shares compute units (0 = the shipped, exclusive form)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import nn_cases
from crazyara_amd import _capi
from crazyara_amd.neuralnetapi import HipAPI, NeuralNetAPIUser, _DevArray

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
lib = _capi.load()
lib.mi_dev_value_head_debug.restype = C.c_void_p
lib.mi_dev_value_head_debug.argtypes = [C.c_void_p]
lib.mi_dev_launch_op.argtypes = [C.c_void_p, C.c_int, C.c_int]
tmp = tempfile.mkdtemp(prefix="cra_aggr_")
cfg, sd, _ = nn_cases.make_case("risev2-3")
d = nn_cases.export_case(tmp, "risev2-3", cfg, sd)
nets = [HipAPI(0, 64, d, "float16x3") for _ in range(2)]
users = [NeuralNetAPIUser([n]) for n in nets]
rng = np.random.default_rng(5)
for n, u in zip(nets, users):
    u.input_planes[:] = (rng.random(u.input_planes.shape) < 0.1).astype(np.float32)
    n.predict(u.input_planes, u.value_outputs, u.prob_outputs)          # every buffer of the forward holds this batch from now on
names = [nm for nm, _ in nets[0].time_ops(1)]
vh = names.index("value_head")
A, B = nets
view = torch.as_tensor(_DevArray(lib.mi_dev_value_head_debug(A._h), (64 * (8 + 1024),)), device="cuda")
lib.mi_dev_launch_op(A._h, vh, 1)
A.sync()
ref = view[:64 * 8].reshape(64, 8)[:, :6].clone()
results = {}
for k, name in list(enumerate(names)) + [(-1, "nothing")]:
    stop = threading.Event()

    def aggressor():
        while not stop.is_set():
            if k >= 0:
                lib.mi_dev_launch_op(B._h, k, 16)
                B.sync()
    th = threading.Thread(target=aggressor)
    th.start()
    bad = 0
    for _ in range(N):
        lib.mi_dev_launch_op(A._h, vh, 1)
        A.sync()
        if not torch.equal(view[:64 * 8].reshape(64, 8)[:, :6], ref):
            bad += 1
    stop.set()
    th.join()
    results[f"{k}:{name}"] = bad
    print(f"aggressor op {k:2d} {name:22s}: {bad} of {N} value head launches differ", flush=True)
print("RESULT " + json.dumps(results))
