"""Development: which LDS bytes does a forward read without having written them?

Fills every CU's LDS with zeros except a byte range that gets NaN bytes (tests/support/lds_poison.hip), runs the forward, and bisects the
range while the output differs from the all-zero run.  Found the 5 x 5 depthwise tap that read one row past the t1 tile.
usage (GPU box): python scripts/lds_read_before_write.py [precision ...]      default: fp8-3k fp8 float16-3k float16"""
import ctypes
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import nn_cases
from crazyara_amd import build
from crazyara_amd.neuralnetapi import HipAPI


def poison_lib():
    src = os.path.join(ROOT, "tests", "support", "lds_poison.hip")
    out = os.path.join(ROOT, "tests", "support", "_build", "liblds_poison.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-O2", *build.device_flags(), "-shared", "-fPIC", src, "-o", out], check=True, cwd="/tmp")
    lib = ctypes.CDLL(out)
    lib.poison_lds.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    return lib


def bisect(lib, name, batch, precision, pattern=0xffffffff, floor=64):
    cfg, sd, _ = nn_cases.make_case(name)
    d = nn_cases.export_case(tempfile.mkdtemp(), name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 77).numpy().reshape(-1)
    net = HipAPI(0, batch, d, precision)

    def run(lo, hi):
        assert lib.poison_lds(pattern, 0, lo, hi) == 0
        v = np.zeros(batch, np.float32)
        p = np.zeros(batch * cfg.nb_policy, np.float32)
        net.predict(x, v, p)
        return np.concatenate([v, p])

    ref = run(0, 0)
    if not np.array_equal(ref, run(0, 0)):
        print(f"{precision} {name}: differs from call to call even behind a zeroed LDS -- a race, not a stale read", flush=True)
        net.close()
        return
    hits, todo = [], [(0, 160 * 1024)]
    while todo:
        lo, hi = todo.pop()
        o = run(lo, hi)
        if np.array_equal(o, ref, equal_nan=True):
            continue
        if hi - lo <= floor:
            hits.append((lo, hi, int(np.isnan(o).sum()), float(np.nanmax(np.abs(o - ref)))))
            continue
        mid = (lo + hi) // 2
        todo += [(mid, hi), (lo, mid)]
    net.close()
    hits.sort()
    merged = []
    for h in hits:
        if merged and merged[-1][1] == h[0]:
            merged[-1] = (merged[-1][0], h[1], merged[-1][2] + h[2], max(merged[-1][3], h[3]))
        else:
            merged.append(h)
    print(f"{precision} {name}: {len(merged)} LDS ranges read before written (absolute LDS addresses: static arrays first)", flush=True)
    for lo, hi, nn, md in merged[:40]:
        print(f"   [{lo}, {hi})  {hi - lo} B   NaNs in the output {nn}   max |difference| {md:.3g}", flush=True)


if __name__ == "__main__":
    lib = poison_lib()
    for precision in (sys.argv[1:] or ["fp8-3k", "fp8", "float16-3k", "float16"]):
        for name, batch in (("risev33", 8), ("risev2-3", 8)):
            bisect(lib, name, batch, precision)
