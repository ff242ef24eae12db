"""Development: latency of one evaluator lane (descriptors in -> results out) and rate of two lanes alternating, without any host-side
search work in between; compare with the device-resident forward.  usage: python scripts/lane_latency.py [blocks=19] [batch=256] [precision=float16]"""
import ctypes as C
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from crazyara_amd import _capi, env, netfile, openings, rise_config
from crazyara_amd.neuralnetapi import HipAPI

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 19
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
PREC = sys.argv[3] if len(sys.argv) > 3 else "float16"
cfg = rise_config.rise_v2_config(blocks, 34, 81)
sd = rise_config.make_state_dict(cfg, seed=1)
tmp = tempfile.mkdtemp()
netfile.export_rise(os.path.join(tmp, "m-v1.0.cranet"), cfg, sd)
lib = _capi.load()
nets = [HipAPI(0, B, tmp, PREC) for _ in range(2)]
fens = openings.position_fens("crazyhouse")
descs = b"".join(env.Position(fens[i % len(fens)], False, "crazyhouse").desc() for i in range(B))
layout = lib.mi_planes_layout(0, 1)
bufs = []
for n in nets:
    d = lib.mi_host_alloc(len(descs)); C.memmove(d, descs, len(descs))
    v = lib.mi_host_alloc(4 * B); p = lib.mi_host_alloc(4 * B * cfg.nb_policy)
    bufs.append((d, v, p))

def submit(i):
    d, v, p = bufs[i]
    assert lib.mi_net_submit_boards(nets[i]._h, d, B, layout, v, p, None) == 0

for i in (0, 1):
    submit(i); nets[i].wait()
dev = nets[0].time_forward(50) / 50
t0 = time.perf_counter()
for _ in range(100):
    submit(0); nets[0].wait()
one = (time.perf_counter() - t0) / 100
submit(0)
t0 = time.perf_counter()
for _ in range(100):
    submit(1); nets[0].wait(); submit(0); nets[1].wait()
two = (time.perf_counter() - t0) / 200
nets[0].wait()
# the search pool's path: priors gathered on the GPU, no copy commands (kernels read / write the pinned buffers in place)
stride = 160
gb = []
rng = np.random.default_rng(0)
for n in nets:
    idx = lib.mi_host_alloc(2 * B * stride); cnt = lib.mi_host_alloc(4 * B); out = lib.mi_host_alloc(4 * B * stride)
    np.ctypeslib.as_array(C.cast(idx, C.POINTER(C.c_uint16)), (B * stride,))[:] = rng.integers(0, cfg.nb_policy, B * stride)
    np.ctypeslib.as_array(C.cast(cnt, C.POINTER(C.c_uint32)), (B,))[:] = 60
    gb.append((idx, cnt, out))

def submit_g(i):
    d, v, p = bufs[i]
    idx, cnt, out = gb[i]
    assert lib.mi_net_submit_boards_gathered(nets[i]._h, d, B, layout, idx, cnt, stride, v, out, None) == 0

for i in (0, 1):
    submit_g(i); nets[i].wait()
t0 = time.perf_counter()
for _ in range(100):
    submit_g(0); nets[0].wait()
one_g = (time.perf_counter() - t0) / 100
submit_g(0)
t0 = time.perf_counter()
for _ in range(100):
    submit_g(1); nets[0].wait(); submit_g(0); nets[1].wait()
two_g = (time.perf_counter() - t0) / 200
nets[0].wait()
def spin(seconds):
    t = time.perf_counter() + seconds
    while time.perf_counter() < t:
        pass

# the same with the host busy for 0.17 ms between a lane's results and its next submit (what apply + collect cost in the pool)
submit_g(0)
t0 = time.perf_counter()
for _ in range(100):
    spin(170e-6); submit_g(1); nets[0].wait(); spin(170e-6); submit_g(0); nets[1].wait()
two_gd = (time.perf_counter() - t0) / 200
nets[0].wait()
print(f"gathered, zero-copy, 0.17 ms of host work per batch: two lanes alternating {two_gd * 1e3:.3f} ms per batch")
print(f"gathered, zero-copy: one lane {one_g * 1e3:.3f} ms per batch | two lanes alternating {two_g * 1e3:.3f} ms per batch")
print(f"device-resident forward {dev:.3f} ms | one lane, full probability vectors back: {one * 1e3:.3f} ms per batch | "
      f"two lanes alternating: {two * 1e3:.3f} ms per batch")
