"""The compiled gfx950 kernels keep every reader of an MFMA result far enough behind the MFMA.

The compiler pads its own instructions; the kernels' asm epilogues (v_fma_mix / v_cvt_pk straight from the accumulators) are the
kernel author's business (csrc/nn/device_utils.h: mfma_retire).  A schedule that put the stem's last MFMA two issue slots in front of
its asm pack made the one-launch fp8 forward differ from call to call; this test compiles the kernels to assembly (hipcc -S, no GPU
needed) and runs scripts/isa_mfma_hazards.py over every kernel."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["forward", "tower", "restower", "head", "stem", "block", "policy_value"]


def test_no_reader_in_the_shadow_of_an_mfma(tmp_path):
    from crazyara_amd import build
    nn = os.path.join(ROOT, "crazyara_amd", "csrc", "nn")
    files = [f for f in FILES if os.path.exists(os.path.join(nn, f + ".hip"))]
    assert {"forward", "tower", "restower", "head", "stem"} <= set(files)
    procs = []
    for f in files:
        out = tmp_path / (f + ".s")
        cmd = [build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S",
               os.path.join(nn, f + ".hip"), "-o", str(out)]
        procs.append((f, out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=str(tmp_path))))
    kernels = 0
    for f, out, p in procs:
        log, _ = p.communicate()
        assert p.returncode == 0, log
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_mfma_hazards.py"), str(out)], stdout=subprocess.PIPE, text=True)
        assert r.returncode == 0
        lines = r.stdout.strip().split("\n")
        assert lines[-1] == "total 0", f + ":\n" + r.stdout[-3000:]
        kernels += sum(1 for l in lines if l.endswith("0 short distances"))
    assert kernels >= 8 + 2 + 4 + 1 + 4          # forward x 8, tower x 2, restower x 4, head, stem x 4


def test_the_scanner_sees_a_short_distance(tmp_path):
    listing = tmp_path / "k.s"
    listing.write_text("_Zk:\n\tv_mfma_f32_32x32x16_f16 v[2:17], v[38:41], v[34:37], v[2:17]\n\tv_add_u32 v60, v1, v80\n\t;;#ASMSTART\n"
                       "\tv_fma_mixlo_f16 v58, v3, 1.0, v42\n\t;;#ASMEND\n\ts_nop 15\n\tv_mov_b32 v70, v4\n.Lfunc_end0:\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_mfma_hazards.py"), str(listing)], stdout=subprocess.PIPE, text=True)
    assert "total 1" in r.stdout and "ASM v_fma_mixlo_f16 touches v3" in r.stdout, r.stdout
