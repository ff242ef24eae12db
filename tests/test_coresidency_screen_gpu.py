"""No kernel of a conformant forward comes out different because a kernel of another net runs beside it.

Round 4 found ONE such pair by accident (the float16x3 value head beside the policy-map conv: one FC1 accumulator wrong in 20-80 % of the
launches) and round 5 its cause: v_pk_fma_f32 goes wrong beside a wave of another workgroup that issues MFMAs on the same SIMD
(profiles/NOTES.md; scripts/ubench/neighbour_mfma.hip).  The library holds no packed f32 arithmetic since (tests/test_isa_hazards.py); this
is the behavioural half of that guarantee, the small form of scripts/coresidency_screen.py: net A runs its forward op by op, every op's
output buffers are recorded, then every op is relaunched ALONE, every launch compared word for word on the device, while net B loops one of
ITS ops on a second stream -- every (victim, aggressor) pair, round 4's harness geometry (RISEv2-3, batch 64: the one where the known pair
failed in 1072 of 2000 launches with the packed value head)."""
import ctypes as C
import threading
import time

import numpy as np
import pytest
import torch

import nn_cases
from crazyara_amd import _capi
from crazyara_amd.neuralnetapi import HipAPI, NeuralNetAPIUser

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["float16x3", "float16p8"])
def test_every_op_reproduces_beside_every_op_of_a_second_net(tmp_path, hip_lib, monkeypatch, precision):
    monkeypatch.setenv("CRA_X3_VALUE_HEAD", "one")              # the one-launch value head (the default), unfenced since round 5
    lib = _capi.load()
    lib.mi_dev_launch_op.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.mi_dev_screen_prepare.argtypes = [C.c_void_p]
    lib.mi_dev_screen_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_long)]
    lib.mi_dev_screen_run.restype = C.c_long
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp_path, "risev2-3", cfg, sd)
    batch = 64
    # "-1wg": the one-workgroup-per-board tower kernels this harness was built around (every op writes buffers of its own, so an op can be
    # relaunched alone on the state the forward left); a batch of 64 would otherwise run split-board (round 6), whose launches hand
    # their partial sums on through two alternating buffers -- that forward is screened whole, below
    A, B = HipAPI(0, batch, d, precision + "-1wg"), HipAPI(0, batch, d, precision + "-1wg")
    users = [NeuralNetAPIUser([n]) for n in (A, B)]
    rng = np.random.default_rng(1)
    for n, u in zip((A, B), users):
        u.input_planes[:] = (rng.random(u.input_planes.shape) < 0.1).astype(np.float32)
        n.predict(u.input_planes, u.value_outputs, u.prob_outputs)
    other = torch.from_numpy((rng.random((batch, cfg.nb_input_channels, 8, 8)) < 0.1).astype(np.float32)).cuda()
    torch.as_tensor(A.device_buffers()["planes"], device="cuda").copy_(other)       # the screen's planes differ from the forward before
    torch.cuda.synchronize()
    n_ops = lib.mi_dev_screen_prepare(A._h)
    assert n_ops >= 4, _capi.last_error()
    names = [nm for nm, _ in A.time_ops(1)]
    assert "value_head" in names and len(names) == n_ops
    red = {}
    for j in range(n_ops):
        stop = threading.Event()

        def aggressor(j=j):
            while not stop.is_set():
                lib.mi_dev_launch_op(B._h, j, 16)
                B.sync()
        th = threading.Thread(target=aggressor)
        th.start()
        time.sleep(0.01)
        for k in range(n_ops):
            words = C.c_long(0)
            bad = lib.mi_dev_screen_run(A._h, k, 400, C.byref(words))
            assert bad >= 0, _capi.last_error()
            if bad:
                red[(f"victim {k}:{names[k]}", f"aggressor {j}:{names[j]}")] = (int(bad), int(words.value))
        stop.set()
        th.join()
    for u in users:
        u.close()
    A.close()
    B.close()
    assert not red, red


@pytest.mark.parametrize("precision", ["float16x3", "float16p8"])
@pytest.mark.parametrize("batch", [8, 64])
def test_split_board_forward_reproduces_beside_a_second_forward(tmp_path, hip_lib, precision, batch):
    """The split-board forward of small batches (block_x3_split_kernel: one launch per bottleneck block, partial sums handed on through
    two alternating buffers) screened as a WHOLE: net A's forward is repeated 300 times on the device and every result compared bit
    for bit with the first, while net B (same model, other planes; then a one-workgroup-per-board net of batch 256) replays its own
    forward on a second stream without pause."""
    cfg, sd, _ = nn_cases.make_case("risev2-7")
    d = nn_cases.export_case(tmp_path, "risev2-7", cfg, sd)
    rng = np.random.default_rng(3)
    A = HipAPI(0, batch, d, precision, keep_logits=True)
    assert "block_x3_split" in [nm for nm, _ in A.time_ops(1)]
    for b_batch, b_prec in ((batch, precision), (256, precision)):
        B = HipAPI(0, b_batch, d, b_prec)
        for n, bb in ((A, batch), (B, b_batch)):
            x = torch.from_numpy((rng.random((bb, cfg.nb_input_channels, 8, 8)) < 0.1).astype(np.float32)).cuda()
            torch.as_tensor(n.device_buffers()["planes"], device="cuda").copy_(x)
        torch.cuda.synchronize()
        stop = threading.Event()

        def aggressor():
            while not stop.is_set():
                for _ in range(8):
                    B.forward_device()
                B.sync()
        th = threading.Thread(target=aggressor)
        th.start()
        time.sleep(0.01)
        bufs = A.device_buffers()
        first = None
        bad = 0
        for it in range(300):
            A.forward_device()
            A.sync()
            out = (torch.as_tensor(bufs["logits"], device="cuda").clone(), torch.as_tensor(bufs["value"], device="cuda").clone())
            if first is None:
                first = out
            elif not (torch.equal(out[0], first[0]) and torch.equal(out[1], first[1])):
                bad += 1
        stop.set()
        th.join()
        B.close()
        assert bad == 0, (bad, b_batch)
    A.close()
