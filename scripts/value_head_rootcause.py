"""Register, load or LDS?  The known co-residency fault of value_head_kernel's FC1 (profiles/NOTES.md rounds 4 and 5), taken apart.

Net A (float16x3, one-launch value head WITHOUT an LDS fence, the PROBE instantiation: CRA_VALUE_HEAD_VARIANT=16; ROOTCAUSE_EXTRA_VARIANT=32
selects the packed-FMA form of FC1, which only a CRA_BUILD_PACKED_FP32=1 library still contains) relaunches only its
value head; net B loops its policy-map conv (conv_gemm_x3_kernel<3, 1, 8, 4>) on another stream.  Every launch of A leaves, per board:
  region 0  the FC1 partial sums as the END of the kernel reads them from LDS           (round 4's dump)
  region 1  the same sums read from LDS right behind the barrier
  region 2  the sums stored to global memory straight from the accumulator registers (never through LDS)
  region 3  per lane and component, the sum of the bit patterns of the 128 weight words the lane loaded
  header    HW_ID of the board's four waves, XCC_ID
For every differing launch the script says which regions differ and where (wave, lane, component), i.e.
  3 differs                      -> the LOADED words were wrong in the registers (load return path or register file), before any arithmetic
  2 differs, 3 equal             -> the accumulator went wrong in the FMA chain
  0 / 1 differ, 2 equal          -> the registers were right, the way through LDS (ds_write_b128 / ds_read) was not
usage: python scripts/value_head_rootcause.py [launches] [batch] [net: risev2-3 | risev2-19]
"""
import ctypes as C
import json
import os
import sys
import tempfile
import threading
from collections import Counter

os.environ["CRA_X3_VALUE_HEAD"] = "one"
os.environ["CRA_VALUE_HEAD_DEBUG"] = "1"
os.environ.setdefault("CRA_VALUE_HEAD_LDS_PAD", "-1")
os.environ["CRA_VALUE_HEAD_VARIANT"] = str(16 | int(os.environ.get("ROOTCAUSE_EXTRA_VARIANT", "0")))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import nn_cases  # noqa: E402
from crazyara_amd import _capi  # noqa: E402
from crazyara_amd.neuralnetapi import HipAPI, NeuralNetAPIUser, _DevArray  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 64
CASE = sys.argv[3] if len(sys.argv) > 3 else "risev2-3"
lib = _capi.load()
lib.mi_dev_value_head_debug.restype = C.c_void_p
lib.mi_dev_value_head_debug.argtypes = [C.c_void_p]
lib.mi_dev_launch_op.argtypes = [C.c_void_p, C.c_int, C.c_int]
tmp = tempfile.mkdtemp(prefix="cra_root_")
cfg, sd, _ = nn_cases.make_case(CASE)
d = nn_cases.export_case(tmp, CASE, cfg, sd)
nets = [HipAPI(0, BATCH, d, "float16x3") for _ in range(2)]
users = [NeuralNetAPIUser([n]) for n in nets]
rng = np.random.default_rng(5)
for n, u in zip(nets, users):
    u.input_planes[:] = (rng.random(u.input_planes.shape) < 0.1).astype(np.float32)
    n.predict(u.input_planes, u.value_outputs, u.prob_outputs)
names = [nm for nm, _ in nets[0].time_ops(1)]
vh = names.index("value_head")
conv = max(i for i, nm in enumerate(names) if nm.startswith("conv_gemm"))          # the policy-map conv (the last conv of the forward)
A, B = nets
words = BATCH * (8 + 1024) + BATCH * (16 + 3 * 1024)
view = torch.as_tensor(_DevArray(lib.mi_dev_value_head_debug(A._h), (words,)), device="cuda").view(torch.int32)
hdr0 = BATCH * (8 + 1024)


def regions(t):
    r0 = t[BATCH * 8:hdr0].reshape(BATCH, 1024)
    pr = t[hdr0:].reshape(BATCH, 16 + 3 * 1024)
    return r0, pr[:, 16:16 + 1024], pr[:, 16 + 1024:16 + 2048], pr[:, 16 + 2048:16 + 3072], pr[:, :16]


lib.mi_dev_launch_op(A._h, vh, 1)
A.sync()
ref = [r.clone() for r in regions(view)]
stop = threading.Event()


def aggressor():
    while not stop.is_set():
        lib.mi_dev_launch_op(B._h, conv, 16)
        B.sync()


th = threading.Thread(target=aggressor)
th.start()
events, patterns, sites, simds = [], Counter(), Counter(), Counter()
for it in range(N):
    lib.mi_dev_launch_op(A._h, vh, 1)
    A.sync()
    cur = regions(view)
    diff = [not torch.equal(cur[i], ref[i]) for i in range(4)]
    if any(diff):
        key = "".join(str(i) for i in range(4) if diff[i])
        patterns[key] += 1
        ev = {"launch": it, "regions_differing": key}
        for i in range(4):
            if diff[i]:
                idx = (cur[i] != ref[i]).nonzero()
                b = int(idx[0, 0])
                cols = idx[idx[:, 0] == b][:, 1].tolist()
                # s_part layout: [quarter kq = wave][fc]: word = kq * fc + 4 * lane + component
                fc = 1024 // 4
                where = sorted({(c // fc, (c % fc) // 4, c % 4) for c in cols})
                ev[f"region{i}"] = {"boards": sorted(set(idx[:, 0].tolist()))[:4], "first_board_words": len(cols),
                                    "wave_lane_component": where[:20]}
                for w_, l_, c_ in where:
                    sites[(i, w_, l_ // 16, c_)] += 1
                hw = int(cur[4][b, where[0][0]])
                ev["hw_id_of_the_wave"] = hex(hw & 0xffffffff)
                simds[(hw >> 4) & 3] += 1
        if len(events) < 30:
            events.append(ev)
stop.set()
th.join()
print(f"net {CASE} batch {BATCH}: {sum(patterns.values())} of {N} value head launches differ beside op {conv} ({names[conv]})")
print("which regions differ together (0 = LDS read at the end, 1 = LDS read behind the barrier, 2 = registers -> global, 3 = loaded-word checksums):")
for k, v in patterns.most_common():
    print(f"  regions {k}: {v}")
print("sites (region, wave, quarter-wave, component): count")
for k, v in sorted(sites.items(), key=lambda kv: -kv[1])[:24]:
    print(f"  {k}: {v}")
print("SIMD of the failing wave (HW_ID bits 5:4):", dict(simds))
for ev in events[:12]:
    print(json.dumps(ev))
print("RESULT " + json.dumps({"differing": sum(patterns.values()), "launches": N, "patterns": dict(patterns)}))
