"""Builds libcrazyara_hip.so (hipcc, gfx950 only) in-tree under crazyara_amd/lib/."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcrazyara_hip.so")


def sources():
    out = []
    for root, _, files in os.walk(CSRC):
        for f in sorted(files):
            if f.endswith((".hip", ".cpp")):
                out.append(os.path.join(root, f))
    return sorted(out)


def _stamp():
    h = hashlib.sha256()
    for root, _, files in os.walk(CSRC):
        for f in sorted(files):
            p = os.path.join(root, f)
            h.update(os.path.relpath(p, CSRC).encode())      # relative: the checkout may live elsewhere on the GPU box
            with open(p, "rb") as fh:
                h.update(fh.read())
    with open(os.path.join(os.path.dirname(HERE), "include", "crazyara_hip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(device_flags()).encode())
    return h.hexdigest()


# No packed f32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any kernel of the library.  Round 5 found that the results of
# v_pk_fma_f32 -- which the compiler forms on its own from scalar source -- come out wrong in lanes 48-63 when a wave of ANOTHER workgroup on
# the same SIMD issues MFMAs (profiles/NOTES.md round 5, scripts/ubench/neighbour_mfma.hip: 3.0 M wrong sums in 3000 launches beside an
# MFMA-only neighbour, 0 with v_fmac_f32; in the product: value_head_kernel beside the float16x3 policy conv, 84 % of its launches).  Which
# kernels of a library meet on a SIMD depends on the other lanes / nets / processes of the GPU, so the instruction class is switched off
# for every device compile (same bits: the packed forms are two IEEE FMAs); tests/test_isa_hazards.py checks the listings.
# CRA_BUILD_PACKED_FP32=1 builds with them (A/B timing only).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def device_flags():
    return [] if os.environ.get("CRA_BUILD_PACKED_FP32") else list(NO_PACKED_FP32)


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_file = LIB + ".stamp"
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.relpath(src, CSRC).replace(os.sep, "_") + ".o")
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", *device_flags(),
               "-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
