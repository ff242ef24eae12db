"""Co-residency screen (VERDICT r04 #1c): does ANY kernel of a conformant forward come out different when ANY kernel of a second net
runs beside it?

Round 4 found one such pair by accident (value_head_kernel beside conv_gemm_x3_kernel<3, 1, 8, 4>: one FC1 accumulator wrong in 20-30 % of
the launches) and fenced it with 144 KB of LDS; this screens every (victim, aggressor) pair of ops of the conformant forwards.  Round 5's
first run of it (library with packed f32 arithmetic, fence off; profiles/r05/a_screen_*) showed exactly that pair red in every net and
nothing else; with the cause removed (v_pk_fma_f32, profiles/NOTES.md round 5) the shipped library has no fence and every cell is 0
(profiles/r05/e_screen_*).

  victim    : net A has run a forward of OTHER planes, then the planes of the screen; RiseNet::dev_screen_prepare runs the forward op by op
              and records every op's output buffers.  Then op k ALONE, `launches` times on A's stream, every launch compared on the device
              with the recorded bits (an op that is not idempotent gets its buffers put back before every launch).
  aggressor : net B (same model and mode, own weights / buffers / stream) loops ONE of its ops on another host thread.
  control   : a library built with CRA_BUILD_PACKED_FP32=1 and CRA_VALUE_HEAD_VARIANT=32 (FC1 as the compiler writes it) shows the known
              pair red; the shipped library must show 0 everywhere.  --fence adds round 4's 144 KB LDS fence (A/B).

  --cross   : the aggressor is a net of ANOTHER model / mode (victim:aggressor pairs, e.g. p8-v2:f16-v2 -- the float16 one-launch forward --
              or x3-v2:p8-v33), every op of it in turn
  --search  : one more aggressor column, "search lanes": a two-lane SearchPool of the aggressor's model searching crazyhouse / chess
              positions on its own host threads -- the plane builder, the forward and the gather kernel of real lanes, as a second engine
              process or another pool of the same process would run them

usage: python scripts/coresidency_screen.py [--launches 1000] [--batch 256] [--configs p8-v2,x3-v2,p8-v33,x3-v33] [--cross a:b,c:d] [--search]
                                            [--fence] [--out file.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import threading
import time

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=1000)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--configs", default="p8-v2,x3-v2,p8-v33,x3-v33")
ap.add_argument("--fence", action="store_true", help="round 4's 144 KB LDS fence around the value head (A/B; the shipped kernel has none)")
ap.add_argument("--cross", default="", help="victim:aggressor config pairs, comma separated (instead of --configs)")
ap.add_argument("--search", action="store_true", help="add a live two-lane search of the aggressor's model as one more aggressor")
ap.add_argument("--out", default=None)
args = ap.parse_args()
os.environ["CRA_X3_VALUE_HEAD"] = "one"
if args.fence:
    os.environ["CRA_VALUE_HEAD_LDS_PAD"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from crazyara_amd import _capi, netfile, openings, rise_config, search  # noqa: E402
from crazyara_amd.neuralnetapi import HipAPI, NeuralNetAPIUser  # noqa: E402

lib = _capi.load()
lib.mi_dev_launch_op.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.mi_dev_screen_prepare.argtypes = [C.c_void_p]
lib.mi_dev_screen_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_long)]
lib.mi_dev_screen_run.restype = C.c_long
lib.mi_dev_screen_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]

CONFIGS = {
    "p8-v2": (lambda: rise_config.rise_v2_config(19, 34, 81), "1.0", "float16p8"),
    "x3-v2": (lambda: rise_config.rise_v2_config(19, 34, 81), "1.0", "float16x3"),
    "p8-v33": (lambda: rise_config.rise_v33_config(52, 76, False), "3.0", "float16p8"),
    "x3-v33": (lambda: rise_config.rise_v33_config(52, 76, False), "3.0", "float16x3"),
    "x3-v2-3": (lambda: rise_config.rise_v2_config(3, 34, 81), "1.0", "float16x3"),          # round 4's harness net (batch 64)
    "f16-v2": (lambda: rise_config.rise_v2_config(19, 34, 81), "1.0", "float16"),            # aggressors of other modes: the one-launch forward
    "fp8-v2": (lambda: rise_config.rise_v2_config(19, 34, 81), "1.0", "fp8"),
    "f32-v2-7": (lambda: rise_config.rise_v2_config(7, 34, 81), "1.0", "float32"),
}


def planes(batch, channels, seed):
    rng = np.random.default_rng(seed)
    return (rng.random((batch, channels, 8, 8)) < 0.1).astype(np.float32)


report = {"launches": args.launches, "batch": args.batch, "fence": bool(args.fence), "configs": {}}
pairs = [tuple(p.split(":")) for p in args.cross.split(",") if p] if args.cross else [(c, c) for c in args.configs.split(",")]


def make_net_dir(config):
    make, version, precision = CONFIGS[config]
    cfg = make()
    sd = rise_config.make_state_dict(cfg, seed=77, stress=True)
    d = tempfile.mkdtemp(prefix="cra_screen_")
    netfile.export_rise(os.path.join(d, f"{cfg.name}-v{version}.cranet"), cfg, sd, input_version=version)
    return cfg, d, precision, version


for vic_name, agg_name in pairs:
    name = vic_name if vic_name == agg_name else f"{vic_name}:{agg_name}"
    cfg, d, precision, version = make_net_dir(vic_name)
    cfg_b, d_b, precision_b, version_b = (cfg, d, precision, version) if agg_name == vic_name else make_net_dir(agg_name)
    if args.batch <= 64:
        # nets made for <= 64 boards run the split-board forward, whose launches pass partial images through two alternating buffers and end
        # in the policy conv that overwrites the first block's input: an op replayed on its own does not see the input it had inside the
        # forward (every cell of such an op reads as different, beside any neighbour and beside none).  The op-by-op screen therefore takes the
        # one-workgroup-per-board form; the split forward is screened whole (tests/test_coresidency_screen_gpu.py).
        precision = precision + "-1wg" if precision in ("float16x3", "float16p8") else precision
        precision_b = precision_b + "-1wg" if precision_b in ("float16x3", "float16p8") else precision_b
    A, B = HipAPI(0, args.batch, d, precision), HipAPI(0, args.batch, d_b, precision_b)
    users = [NeuralNetAPIUser([n]) for n in (A, B)]
    for n, u, seed, c_ in ((A, users[0], 1, cfg), (B, users[1], 2, cfg_b)):
        u.input_planes[:] = planes(args.batch, c_.nb_input_channels, seed).reshape(-1)
        n.predict(u.input_planes, u.value_outputs, u.prob_outputs, u.auxiliary_outputs if n.has_auxiliary_outputs() else None)
    # the planes of the screen in A's device-side input (the op-by-op forward reads them there)
    torch.as_tensor(A.device_buffers()["planes"], device="cuda").copy_(torch.from_numpy(planes(args.batch, cfg.nb_input_channels, 3)).cuda())
    torch.as_tensor(B.device_buffers()["planes"], device="cuda").copy_(torch.from_numpy(planes(args.batch, cfg_b.nb_input_channels, 4)).cuda())
    torch.cuda.synchronize()
    B.forward_device()
    B.sync()
    n_ops = lib.mi_dev_screen_prepare(A._h)
    assert n_ops > 0, lib.mi_last_error()
    names = [nm for nm, _ in A.time_ops(1)]      # (runs the forward once more on the same planes: same bits, the record stands)
    names_b = names if agg_name == vic_name else [nm for nm, _ in B.time_ops(1)]
    pool = None
    if args.search:                              # real lanes of the aggressor's model: two more nets, a pool, its own host threads
        chess_like = cfg_b.nb_input_channels in (52, 39)
        st = search.default_settings(mode=1 if chess_like else 0, version_major=int(version_b.split(".")[0]), batch_size=16)
        lane_nets = [HipAPI(0, args.batch, d_b, precision_b) for _ in range(2)]
        pool = search.SearchPool(st, net_a=lane_nets[0], net_b=lane_nets[1])
        variant_b = "chess" if chess_like else "crazyhouse"
        fens = openings.position_fens(variant_b)
        n_trees = 2 * args.batch // 16
        for i in range(n_trees):
            pool.add_position(fens[(i * 7) % len(fens)], False, variant_b)
    infos = []
    for k in range(n_ops):
        buf = C.create_string_buffer(256)
        lib.mi_dev_screen_info(A._h, k, buf, 256)
        infos.append(buf.value.decode())
        print(f"[{name}] op {k:2d}: {infos[-1]}", flush=True)
    matrix = {}
    t0 = time.perf_counter()
    for j, agg in [(-1, "nothing")] + list(enumerate(names_b)) + ([(-2, "search lanes")] if pool is not None else []):
        stop = threading.Event()

        def aggressor():
            sims = 0
            while not stop.is_set():
                if j >= 0:
                    lib.mi_dev_launch_op(B._h, j, 16)
                    B.sync()
                elif j == -2:
                    for i_ in range(n_trees):                       # fresh trees every round: the search runs for the whole column
                        pool.reset_position(i_, fens[(i_ * 7 + sims) % len(fens)], False, variant_b)
                    sims += 1
                    pool.run(simulations=400, threads=4)
                else:
                    time.sleep(0.01)
        th = threading.Thread(target=aggressor)
        th.start()
        row = {}
        for k, vic in enumerate(names):
            words = C.c_long(0)
            bad = lib.mi_dev_screen_run(A._h, k, args.launches, C.byref(words))
            assert bad >= 0, lib.mi_last_error()
            row[f"{k}:{vic}"] = [int(bad), int(words.value)]
        stop.set()
        th.join()
        matrix[f"{j}:{agg}"] = row
        red = {v: b for v, b in row.items() if b[0]}
        print(f"[{name}] aggressor {j:2d} {agg:16s}: " + (f"RED {red}" if red else "all victims clean"), flush=True)
    if pool is not None:
        pool.close()
        for n in lane_nets:
            n.close()
    report["configs"][name] = {"precision": precision, "model": cfg.name, "aggressor_precision": precision_b, "aggressor_model": cfg_b.name,
                               "ops": infos, "seconds": round(time.perf_counter() - t0, 1),
                               "bad_launches_and_pieces_by_aggressor_then_victim": matrix}
    for u in users:
        u.close()
    A.close()
    B.close()
line = json.dumps(report)
if args.out:
    with open(args.out, "w") as f:
        f.write(line + "\n")
# compact table: victims down, aggressors across, cells = differing launches
for name, r in report["configs"].items():
    m = r["bad_launches_and_pieces_by_aggressor_then_victim"]
    aggs = list(m)
    print(f"\n== {name} ({r['precision']}, {r['model']}, batch {args.batch}, {args.launches} launches per cell; rows = victim op, columns = aggressor op) ==")
    print(" " * 22 + " ".join(f"{a.split(':')[0]:>5s}" for a in aggs))
    for v in m[aggs[0]]:
        print(f"{v:22s}" + " ".join(f"{m[a][v][0]:5d}" for a in aggs))
red_total = sum(b[0] for r in report["configs"].values() for row in r["bad_launches_and_pieces_by_aggressor_then_victim"].values() for b in row.values())
print(f"\nRESULT red cells' launches in all: {red_total}")
