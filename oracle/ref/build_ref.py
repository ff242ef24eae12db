"""oracle/_ref: the REFERENCE'S OWN search code, compiled from the sources where they lie under /root/reference.

Test infrastructure only (never imported by the product).  Recipe, not sources: nothing from the reference is copied into this
repository; the outputs go to oracle/_ref/ (git-ignored, shipped to the GPU box with the other built files).

What is compiled (engine/src, unmodified):
    node.cpp (through node_tu.cpp), nodedata.cpp, searchthread.cpp, evalinfo.cpp, state.cpp, stateobj.cpp,
    util/{blazeutil,communication,randomgen}.cpp, agents/{agent,mctsagent}.cpp, agents/util/gcthread.cpp,
    agents/config/{searchsettings,searchlimits,playsettings}.cpp, manager/{threadmanager,timemanager,treemanager}.cpp,
    nn/{neuralnetapi,neuralnetapiuser,neuralnetdesign}.cpp
against
    shim/blaze/Math.h       stand-in for the absent blaze submodule (natural-order element-wise semantics, see its header)
    shim/pommermanstate.h   the reference's MODE_POMMERMAN hook (stateobj.h:39-40,53-55) filled with a State over this repository's
                            chess Position (the reference's own BoardState needs the absent Stockfish fork)
    ref_driver.cpp          extern "C" entry points for the tests

The reference's build system is not run.  `python oracle/ref/build_ref.py` rebuilds; build() returns the library path, or None when
/root/reference is absent and no prebuilt library exists (the GPU box uses the prebuilt file).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CRA_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF, "engine", "src")
OUT = os.path.join(os.path.dirname(HERE), "_ref")
LIB = os.path.join(OUT, "libcrazyara_ref.so")
LIB_HIP = os.path.join(OUT, "libcrazyara_ref_hip.so")     # + integration/hipapi.h, linked against the product library
LIB_HIP_RELEASE = os.path.join(OUT, "libcrazyara_ref_hip_release.so")
# the reference's SearchThread with integration/searchthread_hip.patch applied (descriptor-fed batches, gathered priors): the patch is
# applied to a COPY of searchthread.cpp made here at build time (oracle/_ref/patched/, git-ignored) -- nothing of the reference is committed
LIB_HIP_PATCHED = os.path.join(OUT, "libcrazyara_ref_hip_patched.so")                  # asserting build: the parity test
LIB_HIP_PATCHED_RELEASE = os.path.join(OUT, "libcrazyara_ref_hip_patched_release.so")  # -O3 -DNDEBUG: the drop-in throughput leg
PATCH = os.path.join(ROOT, "integration", "searchthread_hip.patch")

REFERENCE_SOURCES = [
    "nodedata.cpp", "searchthread.cpp", "evalinfo.cpp", "state.cpp", "stateobj.cpp",
    "util/blazeutil.cpp", "util/communication.cpp", "util/randomgen.cpp",
    "agents/agent.cpp", "agents/mctsagent.cpp", "agents/util/gcthread.cpp",
    "agents/config/searchsettings.cpp", "agents/config/searchlimits.cpp", "agents/config/playsettings.cpp",
    "manager/threadmanager.cpp", "manager/timemanager.cpp", "manager/treemanager.cpp",
    "nn/neuralnetapi.cpp", "nn/neuralnetapiuser.cpp", "nn/neuralnetdesign.cpp",
]
SHIM_SOURCES = ["node_tu.cpp", "ref_driver.cpp"]                      # node_tu.cpp = #include "node.cpp" + a seeding hook
PRODUCT_ENV_SOURCES = ["chess/position.cpp", "chess/policy.cpp", "chess/planes_host.cpp"]   # the environment behind the State adapter

# DYNAMIC_NN_ARCH: the reference's default (engine/CMakeLists.txt:14,99-100): buffer sizes come from the loaded net, not from constants
FLAGS = ["-std=c++17", "-O2", "-fPIC", "-DMODE_POMMERMAN", "-DDISABLE_UCI_INFO", "-DDYNAMIC_NN_ARCH", "-w"]


def reference_present() -> bool:
    return os.path.isfile(os.path.join(SRC, "node.cpp"))


def _inputs():
    files = [os.path.join(SRC, s) for s in REFERENCE_SOURCES] + [os.path.join(SRC, "node.cpp"), os.path.join(SRC, "node.h")]
    files += [os.path.join(HERE, s) for s in SHIM_SOURCES] + [os.path.join(HERE, "shim", "pommermanstate.h"),
                                                              os.path.join(HERE, "shim", "blaze", "Math.h"), __file__]
    files += [os.path.join(ROOT, "crazyara_amd", "csrc", s) for s in PRODUCT_ENV_SOURCES]
    files += [os.path.join(ROOT, "crazyara_amd", "csrc", "chess", h) for h in ("position.h", "policy.h", "planes.h", "planes_host.h")]
    files.append(os.path.join(ROOT, "include", "crazyara_hip.h"))
    files.append(os.path.join(ROOT, "integration", "hipapi.h"))
    files.append(PATCH)
    return files


def _stamp() -> str:
    h = hashlib.sha256()
    for f in _inputs():
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False):
    if not reference_present():
        return LIB if os.path.exists(LIB) else None
    os.makedirs(OUT, exist_ok=True)
    stamp_file = LIB + ".stamp"
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    gxx = shutil.which("g++")
    if gxx is None:
        raise RuntimeError("g++ not found")
    inc = ["-I", os.path.join(HERE, "shim"), "-I", SRC, "-I", os.path.join(SRC, "nn"), "-I", os.path.join(SRC, "agents"),
           "-I", os.path.join(ROOT, "crazyara_amd", "csrc")]
    objdir = os.path.join(OUT, "obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for base, names in ((SRC, REFERENCE_SOURCES), (HERE, SHIM_SOURCES), (os.path.join(ROOT, "crazyara_amd", "csrc"), PRODUCT_ENV_SOURCES)):
        for n in names:
            obj = os.path.join(objdir, n.replace("/", "_") + ".o")
            cmd = [gxx] + FLAGS + inc + ["-c", os.path.join(base, n), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((n, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for n, obj, p in jobs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"g++ failed on {n}:\n{out}")
        objs.append(obj)
    # -Bsymbolic: the rand() / srand() of ref_driver.cpp bind the reference's calls inside this library (seeded exploration)
    r = subprocess.run([gxx, "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIB] + objs + ["-lpthread"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    # the same objects + the driver compiled with the HipAPI section, linked against the product library (found through an rpath
    # relative to this file's location, so the pair travels to the GPU box)
    product_lib_dir = os.path.join(ROOT, "crazyara_amd", "lib")
    if os.path.exists(os.path.join(product_lib_dir, "libcrazyara_hip.so")):
        drv = os.path.join(objdir, "ref_driver_hip.o")
        r = subprocess.run([gxx] + FLAGS + inc + ["-DREF_WITH_HIPAPI", "-I", os.path.join(ROOT, "include"), "-c",
                            os.path.join(HERE, "ref_driver.cpp"), "-o", drv], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed on ref_driver.cpp (HipAPI build):\n" + r.stdout)
        others = [o for o in objs if not o.endswith("ref_driver.cpp.o")]
        r = subprocess.run([gxx, "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIB_HIP] + others + [drv, "-L", product_lib_dir, "-lcrazyara_hip",
                            "-Wl,-rpath,$ORIGIN/../../crazyara_amd/lib", "-lpthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed (HipAPI build):\n" + r.stdout)
        # The same once more as a RELEASE build (-O3 -DNDEBUG: what the reference's CMake Release configuration compiles) for the throughput
        # leg of bench.py: with Threads > 1 the reference's own assert on the virtual-loss counter (node.h:506) fires under its threads'
        # races, which a release engine does not contain; the parity tests keep the asserting build above.
        rel_dir = os.path.join(OUT, "obj_release")
        os.makedirs(rel_dir, exist_ok=True)
        rel_flags = [f for f in FLAGS if f != "-O2"] + ["-O3", "-DNDEBUG"]
        jobs = []
        for base, names in ((SRC, REFERENCE_SOURCES), (HERE, SHIM_SOURCES), (os.path.join(ROOT, "crazyara_amd", "csrc"), PRODUCT_ENV_SOURCES)):
            for n in names:
                obj = os.path.join(rel_dir, n.replace("/", "_") + ".o")
                extra = ["-DREF_WITH_HIPAPI", "-I", os.path.join(ROOT, "include")] if n == "ref_driver.cpp" else []
                jobs.append((n, obj, subprocess.Popen([gxx] + rel_flags + inc + extra + ["-c", os.path.join(base, n), "-o", obj],
                                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        rel_objs = []
        for n, obj, p in jobs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError(f"g++ failed on {n} (release build):\n{out}")
            rel_objs.append(obj)
        r = subprocess.run([gxx, "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIB_HIP_RELEASE] + rel_objs + ["-L", product_lib_dir, "-lcrazyara_hip",
                            "-Wl,-rpath,$ORIGIN/../../crazyara_amd/lib", "-lpthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed (HipAPI release build):\n" + r.stdout)
        # ---- the patched SearchThread (HIP_BACKEND): same objects, searchthread.cpp replaced by the patched copy ----
        pdir = os.path.join(OUT, "patched")
        os.makedirs(pdir, exist_ok=True)
        patched_src = os.path.join(pdir, "searchthread.cpp")
        shutil.copyfile(os.path.join(SRC, "searchthread.cpp"), patched_src)
        r = subprocess.run(["patch", "-p3", "--no-backup-if-mismatch", patched_src, PATCH], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("integration/searchthread_hip.patch does not apply to the reference's searchthread.cpp:\n" + r.stdout)
        hip_defs = ["-DHIP_BACKEND", "-DHIP_ENGINE_MODE=refshim::config().mode", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "integration"),
                    "-I", os.path.join(pdir, "inc")]
        # the patch includes "nn/hipapi.h" (where a maintainer puts the file): give the include path that shape
        os.makedirs(os.path.join(pdir, "inc", "nn"), exist_ok=True)
        shutil.copyfile(os.path.join(ROOT, "integration", "hipapi.h"), os.path.join(pdir, "inc", "nn", "hipapi.h"))
        for flags, base_objs, drv_obj, lib in ((FLAGS, others, drv, LIB_HIP_PATCHED),
                                               (rel_flags, [o for o in rel_objs if not o.endswith("searchthread.cpp.o")], None, LIB_HIP_PATCHED_RELEASE)):
            obj = os.path.join(pdir, ("rel_" if lib == LIB_HIP_PATCHED_RELEASE else "") + "searchthread_patched.o")
            r = subprocess.run([gxx] + flags + inc + hip_defs + ["-c", patched_src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError("g++ failed on the patched searchthread.cpp:\n" + r.stdout)
            link_objs = [o for o in base_objs if not o.endswith("searchthread.cpp.o")] + [obj] + ([drv_obj] if drv_obj else [])
            r = subprocess.run([gxx, "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", lib] + link_objs + ["-L", product_lib_dir, "-lcrazyara_hip",
                                "-Wl,-rpath,$ORIGIN/../../crazyara_amd/lib", "-lpthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError("link failed (patched SearchThread build):\n" + r.stdout)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
