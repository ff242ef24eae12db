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
FILES = ["forward", "tower", "restower", "head", "stem", "x3", "kernels", "rise_net", "block", "policy_value"]


def test_no_reader_in_the_shadow_of_an_mfma(tmp_path):
    from crazyara_amd import build
    nn = os.path.join(ROOT, "crazyara_amd", "csrc", "nn")
    files = [f for f in FILES if os.path.exists(os.path.join(nn, f + ".hip"))] + ["../chess/planes_kernel"]    # (every .hip of the library)
    assert {"forward", "tower", "restower", "head", "stem", "x3", "kernels"} <= set(files)      # x3 + kernels: the headline forward
    procs = []
    for f in files:
        out = tmp_path / (os.path.basename(f) + ".s")
        cmd = [build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", *build.device_flags(), "-x", "hip", "--cuda-device-only", "-S",
               os.path.join(nn, f + ".hip"), "-o", str(out)]
        procs.append((f, out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=str(tmp_path))))
    kernels = 0
    packed = []
    for f, out, p in procs:
        log, _ = p.communicate()
        assert p.returncode == 0, log
        # no packed f32 arithmetic anywhere: v_pk_fma_f32 comes out wrong beside an MFMA-issuing wave of another workgroup on the SIMD
        # (round 5, scripts/ubench/neighbour_mfma.hip; crazyara_amd/build.py NO_PACKED_FP32)
        packed += [f + ": " + l.strip() for l in open(out) if l.strip().startswith(("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"))]
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_mfma_hazards.py"), str(out)], stdout=subprocess.PIPE, text=True)
        assert r.returncode == 0
        lines = r.stdout.strip().split("\n")
        assert lines[-1] == "total 0", f + ":\n" + r.stdout[-3000:]
        kernels += sum(1 for l in lines if l.endswith("0 short distances"))
    assert not packed, packed[:5]
    assert kernels >= 8 + 2 + 4 + 1 + 4 + 20     # forward x 8, tower x 2, restower x 4, head, stem x 4, x3's towers / convs + kernels.hip


def test_the_scanner_stops_at_an_unconditional_branch(tmp_path):
    """What the listing prints behind `s_branch` is another path's code (round 4's 18 false positives in x3.s were all of this kind);
    and the 16x16x128 8-bit MFMA is 8 passes: 11 wait states, not the 18 of the 32x32x64 form."""
    listing = tmp_path / "k.s"
    listing.write_text("_Zk:\n\tv_mfma_f32_16x16x32_f16 v[2:5], v[18:21], v[34:37], v[2:5]\n\ts_branch .LBB0_4\n.LBB0_31:\n\tv_mov_b32_e32 v5, 0\n"
                       "\tv_mfma_scale_f32_16x16x128_f8f6f4 v[6:9], v[18:25], v[34:41], v[6:9], v1, v1 op_sel_hi:[0,0,0] cbsz:1 blgp:1\n\ts_nop 10\n"
                       "\tv_mov_b32 v70, v6\n.Lfunc_end0:\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_mfma_hazards.py"), str(listing)], stdout=subprocess.PIPE, text=True)
    assert r.stdout.strip().endswith("total 0"), r.stdout
    listing.write_text(listing.read_text().replace("s_nop 10", "s_nop 8"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_mfma_hazards.py"), str(listing)], stdout=subprocess.PIPE, text=True)
    assert "total 1" in r.stdout, r.stdout


def test_the_scanner_sees_a_short_distance(tmp_path):
    listing = tmp_path / "k.s"
    listing.write_text("_Zk:\n\tv_mfma_f32_32x32x16_f16 v[2:17], v[38:41], v[34:37], v[2:17]\n\tv_add_u32 v60, v1, v80\n\t;;#ASMSTART\n"
                       "\tv_fma_mixlo_f16 v58, v3, 1.0, v42\n\t;;#ASMEND\n\ts_nop 15\n\tv_mov_b32 v70, v4\n.Lfunc_end0:\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_mfma_hazards.py"), str(listing)], stdout=subprocess.PIPE, text=True)
    assert "total 1" in r.stdout and "ASM v_fma_mixlo_f16 touches v3" in r.stdout, r.stdout
