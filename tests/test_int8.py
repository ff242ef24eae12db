"""Precision int8: the calibrated INT8 mode (the reference: TensorRT INT8 with Int8EntropyCalibrator2 over ChessBatchStream,
engine/src/nn/tensorrtapi.cpp:334-360, chessbatchstream.cpp:44-94).  Defined by oracle/rise_oracle.int8_block / forward_int8_tower:
int8 operands in the two GEMMs of every bottleneck block (v_mfma_i32_32x32x32_i8), one calibrated step per activation tensor and block,
one step per weight row, exact int32 sums.  Integer sums do not depend on their order, so the kernel is pinned to the emulation far more
tightly than Precision fp8 can be: block by block, teacher-forced, almost every element identical."""
import os

import numpy as np
import pytest
import torch

import nn_cases
from oracle import rise_oracle as ro


def test_emulation_error_envelope_against_fp32_and_fp8():
    """the study's numbers (scripts/studies/int8_calibration_study.py, profiles/r06/i_*): value within 1e-2 of fp32 and closer than e4m3"""
    for name in ("risev2-3", "risev2-7"):
        cfg, sd, x = nn_cases.make_case(name)
        xc = nn_cases.synthetic_planes(48, cfg.nb_input_channels, 99)
        calib = ro.calibrate_int8(cfg, sd, xc)
        assert len(calib) == len(cfg.kernels) and all(a > 0 and b > 0 for a, b in calib)
        v32, l32, _ = ro.forward(cfg, sd, x)
        v8, l8, _ = ro.forward_int8_tower(cfg, sd, x, calib)
        vf, lf, _ = ro.forward_fp8_tower(cfg, sd, x)
        e8, ef = float((v8 - v32).abs().max()), float((vf - v32).abs().max())
        assert 1e-4 < e8 < 1.5e-2, (name, e8)
        assert float((l8 - l32).abs().max()) < float((lf - l32).abs().max()), name
        assert torch.equal(ro.forward_int8_tower(cfg, sd, x, calib)[1], l8)


def _export(tmp_path, name):
    factory, seed, stress, _ = nn_cases.CASES[name]
    cfg = factory()
    sd = ro.make_state_dict(cfg, seed=seed, stress=stress)
    d = nn_cases.export_case(tmp_path, name, cfg, sd, version="3.0" if cfg.nb_input_channels in (52, 64, 80) else "1.0")
    return cfg, sd, d


def _calib_file(d):
    f = [os.path.join(d, n) for n in os.listdir(d) if n.endswith(".int8calib")]
    assert len(f) == 1, f
    lines = open(f[0]).read().split("\n")
    assert lines[0] == "crazyara-int8-calibration 1"
    n = int(lines[2].split()[1])
    return [tuple(float(v) for v in lines[3 + i].split()) for i in range(n)]


@pytest.mark.gpu
def test_int8_needs_a_calibration_and_makes_one_from_the_reference_games(tmp_path, hip_lib, capfd):
    """The C ABI refuses Precision int8 on a model without its calibration file and names the call that makes it; the option layer
    (integration/hipapi.h and its Python mirror) runs that call first, like TensorRT runs its calibrator when no engine cache exists:
    the default positions are the plies of the reference's calibration games, encoded by the library itself -- the same maxima as a
    calibration on planes the Python environment made of the same plies."""
    from crazyara_amd import _capi, env, openings
    from crazyara_amd.neuralnetapi import HipAPI, calibrate_int8
    cfg, sd, d = _export(tmp_path, "risev2-3")
    lib = _capi.load()
    assert lib.mi_net_has_int8_calibration(d.encode()) == 0
    assert not lib.mi_net_create(d.encode(), 0, 4, b"int8")
    msg = _capi.last_error()
    assert "int8" in msg and "mi_net_calibrate_int8" in msg
    net = HipAPI(0, 4, d, "int8")                                      # the option layer calibrates
    assert "run INT8 quantization calibration" in capfd.readouterr().err
    assert lib.mi_net_has_int8_calibration(d.encode()) == 1
    x = nn_cases.make_case("risev2-3")[2]
    v, p = np.zeros(4, np.float32), np.zeros(4 * cfg.nb_policy, np.float32)
    net.predict(np.ascontiguousarray(x.numpy()), v, p)
    net.close()
    v32, p32, _ = ro.predict(cfg, sd, x)
    assert np.abs(v - v32.numpy()).max() < 3e-2 and np.abs(p.reshape(4, -1) - p32.numpy()).max() < 1e-3
    default = _calib_file(d)
    planes = []
    for g in openings.games("crazyhouse"):
        pos = env.Position("", False, "crazyhouse")
        for mv in [None] + g:
            if mv is not None:
                assert pos.push_uci(mv)
            planes.append(pos.planes(0, 1, True))
    assert len(planes) == 234                                           # 232 plies + the two start positions
    os.remove([os.path.join(d, n) for n in os.listdir(d) if n.endswith(".int8calib")][0])
    calibrate_int8(d, 0, np.stack(planes))
    assert _calib_file(d) == default
    # ... and they are the float16 forward's maxima: the oracle's float16 emulation of the same pass
    emu = ro.calibrate_int8(cfg, sd, torch.from_numpy(np.stack(planes)))
    for (a, b), (ea, eb) in zip(default, emu):
        assert abs(a - ea) <= 0.02 * ea and abs(b - eb) <= 0.02 * eb, (default, emu)


def _f16_ulp(t):
    e = torch.floor(torch.log2(t.abs().clamp_min(2.0 ** -14)))
    return torch.exp2(e - 10)


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("risev2-3", 4), ("risev2-19", 6), ("risev33", 5), ("risev2-13-lichess", 3), ("risev33-wdlp", 3)])
def test_int8_tower_block_by_block_teacher_forced(tmp_path, hip_lib, name, batch):
    """Block i of the kernel (its f16 stream in front of and behind every block: mi_net_block_dump) against oracle.int8_block run from the
    kernel's OWN input of block i with the calibration file's steps.  The integer GEMMs are exact on both sides; what can differ is an
    f16 rounding of the depthwise chain (one unit of a t2 byte) or of the SE gate: almost every element must be IDENTICAL and the rest
    within a small fraction of the block's own mode error."""
    from crazyara_amd.neuralnetapi import HipAPI, calibrate_int8
    cfg, sd, d = _export(tmp_path, name)
    xc = nn_cases.synthetic_planes(40, cfg.nb_input_channels, 555)
    calibrate_int8(d, 0, xc.numpy())
    steps = ro.int8_steps(_calib_file(d))
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 4000)
    net = HipAPI(0, batch, d, "int8")
    dump = net.block_dump()
    v, p = np.zeros(batch, np.float32), np.zeros(batch * cfg.nb_policy, np.float32)
    net.predict(np.ascontiguousarray(x.numpy()), v, p, np.zeros(batch * 4, np.float32) if cfg.nb_aux else None)
    tiles = torch.as_tensor(dump, device="cuda").cpu().float()
    net.close()
    assert tiles.shape[0] == len(cfg.kernels) + 1 and torch.isfinite(tiles).all()
    nchw = lambda t: t.permute(0, 2, 1).reshape(batch, 256, 8, 8).contiguous()
    rows = []
    for i in range(len(cfg.kernels)):
        h_in, h_gpu = nchw(tiles[i]), nchw(tiles[i + 1])
        h_emu = ro.int8_block(cfg, sd, i, h_in, steps[i][0], steps[i][1], se_f16_weights=True)
        err_mode = h_emu - ro.fp32_block(cfg, sd, i, h_in)
        mode_max, mode_rms = float(err_mode.abs().max()), float(err_mode.pow(2).mean().sqrt())
        diff = h_gpu - h_emu
        rows.append((i, float((diff.abs() - _f16_ulp(h_emu)).max()) / mode_max, float(diff.pow(2).mean().sqrt()) / mode_rms,
                     float((diff != 0).float().mean()), mode_max))
    report = "; ".join(f"b{i}: max {a:.2f} rms {b:.3f} differ {c:.3f}" for i, a, b, c, _ in rows)
    print(f"{name}: (|gpu - emulation| - ulp) / block mode error: {report}")
    if os.path.isdir("gpurun_out"):
        with open(os.path.join("gpurun_out", "int8_block_by_block.txt"), "a") as f:
            f.write(f"{name} batch {batch}: {report}\n")
    for i, mx, rms, differ, mode_max in rows:
        assert mode_max > 1e-4, (i, mode_max)
        assert mx <= 0.5 and rms <= 0.1 and differ < 0.25, (name, i, mx, rms, differ)


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("risev2-7", 19), ("risev2-19", 37), ("risev33", 5)])
def test_int8_end_to_end_envelope(tmp_path, hip_lib, name, batch):
    """End to end against fp32 and against the other modes on the same boards: the value within 2e-2 of fp32 (the study: 6 - 8e-3 on
    the calibration games' neighbours), closer than Precision fp8's; one launch and three launches give the same bits; float16 on the same
    directory is untouched by the calibration file."""
    from crazyara_amd.neuralnetapi import HipAPI, calibrate_int8
    cfg, sd, d = _export(tmp_path, name)
    calibrate_int8(d, 0, nn_cases.synthetic_planes(64, cfg.nb_input_channels, 556).numpy())
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 4001)
    xin = np.ascontiguousarray(x.numpy())
    v32, p32, _ = ro.predict(cfg, sd, x)
    outs = {}
    for prec in ("int8", "int8-3k", "fp8", "float16"):
        net = HipAPI(0, batch, d, prec)
        v, p = np.zeros(batch, np.float32), np.zeros(batch * cfg.nb_policy, np.float32)
        net.predict(xin, v, p, np.zeros(batch * 4, np.float32) if cfg.nb_aux else None)
        net.close()
        outs[prec] = (v, p.reshape(batch, -1))
    assert np.array_equal(outs["int8"][0], outs["int8-3k"][0]) and np.array_equal(outs["int8"][1], outs["int8-3k"][1])
    err = {k: (float(np.abs(v - v32.numpy()).max()), float(np.abs(p - p32.numpy()).max())) for k, (v, p) in outs.items()}
    print("int8 end to end:", name, err)
    assert err["int8"][0] < 2e-2 and err["int8"][1] < 5e-4, err
    assert err["int8"][0] < err["fp8"][0] * 1.2 + 1e-3, err
    assert err["float16"][0] < 1e-3, err
    assert np.abs(outs["int8"][1].sum(axis=1) - 1.0).max() < 1e-4


@pytest.mark.gpu
def test_int8_forward_is_bit_identical_whatever_the_cus_held_before(tmp_path, hip_lib, lds_poison):
    from crazyara_amd.neuralnetapi import HipAPI, calibrate_int8
    cfg, sd, d = _export(tmp_path, "risev33")
    calibrate_int8(d, 0, nn_cases.synthetic_planes(32, cfg.nb_input_channels, 557).numpy())
    batch = 64
    x = nn_cases.synthetic_planes(batch, cfg.nb_input_channels, 77).numpy().reshape(-1)
    net = HipAPI(0, batch, d, "int8")
    outs = []
    for pattern in (0x00000000, 0xffffffff, 0x7f7f7f7f, 0x7bff7bff, 0x80808080, 0x00000000):
        assert lds_poison.poison_lds(pattern, pattern, 0, 0) == 0
        v, p = np.zeros(batch, np.float32), np.zeros(batch * cfg.nb_policy, np.float32)
        net.predict(x, v, p)
        outs.append((v, p))
    net.close()
    assert np.isfinite(outs[0][0]).all() and np.isfinite(outs[0][1]).all()
    for v, p in outs[1:]:
        assert np.array_equal(v, outs[0][0]) and np.array_equal(p, outs[0][1])


@pytest.mark.gpu
def test_int8_is_refused_where_the_tower_kernel_does_not_run(tmp_path, hip_lib):
    from crazyara_amd.neuralnetapi import calibrate_int8
    cfg, sd, d = _export(tmp_path, "alphazero-3-cv8")                 # dense 3 x 3 blocks: no bottleneck tower
    with pytest.raises(RuntimeError, match="bottleneck"):
        calibrate_int8(d, 0, nn_cases.synthetic_planes(8, cfg.nb_input_channels, 1).numpy())
