"""Two-lane search reproducibility: bisecting harness (profiles/NOTES.md round 4).

The scenario of tests/test_search_gpu.py::test_priors_gathered_on_the_gpu_equal_whole_probability_vectors: 8 crazyhouse trees, two
evaluator lanes (two nets, two streams), 4 host threads, batch 64, 240 simulations.  A tree lives in ONE lane, so its result is a pure
function of the network outputs: two runs that differ got different numbers from the GPU.  Every run is recorded (CRA_LANE_RECORD) and
replayed single-stream afterwards (mi_search_debug_replay): a differing word names batch, slot and output.

    python scripts/lane_divergence.py                  # the table of configurations, one child process each
    python scripts/lane_divergence.py --child NAME     # one configuration in this process (environment already set)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# name -> (environment, options)
# CRA_VALUE_HEAD_LDS_PAD=-1: the one-launch value head as it was in round 3 (44 KB of LDS, sharing compute units with other kernels) -- the
# form that fails; without it the kernel takes a compute unit's LDS for itself (the shipped form)
CONFIGS = {
    "three_roles": ({"CRA_X3_VALUE_HEAD": "three"}, {}),
    "one_roles": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1"}, {}),
    "one_symmetric": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_X3_TOWER": "symmetric"}, {}),
    "one_roles_hwq1": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "GPU_MAX_HW_QUEUES": "1"}, {}),
    "one_roles_serialize": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "AMD_SERIALIZE_KERNEL": "3"}, {}),
    "one_roles_lanesync": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_LANE_SYNC": "1"}, {}),
    "one_roles_nograph": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_LANE_NO_GRAPH": "1"}, {}),
    "one_roles_turns": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_FORCE_FORWARD_TURNS": "1"}, {}),
    "one_roles_onelane": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1"}, {"lanes": 1}),
    "one_roles_whole_vectors": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_GATHER_PER_SLOT": "0"}, {}),
    "three_roles_nograph": ({"CRA_X3_VALUE_HEAD": "three", "CRA_LANE_NO_GRAPH": "1"}, {}),
    "float16": ({}, {"precision": "float16"}),
    # the value head's stage checksums under the storm of concurrent predicts: which stage differs first?
    "dbg_one": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_VALUE_HEAD_DEBUG": "1"}, {"predicts": 20000, "runs": 0}),
    "dbg_one_alone_on_cu": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_DEBUG": "1", "CRA_VALUE_HEAD_LDS_PAD": "0"}, {"predicts": 20000, "runs": 0}),
    "dbg_one_hwq1": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_VALUE_HEAD_DEBUG": "1", "GPU_MAX_HW_QUEUES": "1"}, {"predicts": 20000, "runs": 0}),
    "storm_three": ({"CRA_X3_VALUE_HEAD": "three"}, {"predicts": 20000, "runs": 0}),
    # the shipped kernel (exclusive compute unit) against the round-3 form that shares compute units (CRA_VALUE_HEAD_LDS_PAD=-1)
    "shipped_one": ({"CRA_X3_VALUE_HEAD": "one"}, {"predicts": 40000, "runs": 100}),
    "sharing_one": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1"}, {"predicts": 40000, "runs": 100}),
    "dbg_one_own_lds": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_VALUE_HEAD_DEBUG": "1", "CRA_VALUE_HEAD_VARIANT": "1"}, {"predicts": 20000, "runs": 0}),
    "dbg_one_no_pk": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_VALUE_HEAD_DEBUG": "1", "CRA_VALUE_HEAD_VARIANT": "2"}, {"predicts": 20000, "runs": 0}),
    "one_no_dbg_storm": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1"}, {"predicts": 20000, "runs": 0}),
    "big_default": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_VALUE_HEAD_DEBUG": "1"}, {"predicts": 80000, "runs": 0}),
    "big_own_lds": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_VALUE_HEAD_DEBUG": "1", "CRA_VALUE_HEAD_VARIANT": "1"}, {"predicts": 80000, "runs": 0}),
    "big_vmcnt0": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_VALUE_HEAD_DEBUG": "1", "CRA_VALUE_HEAD_VARIANT": "4"}, {"predicts": 80000, "runs": 0}),
    "big_nt_loads": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_VALUE_HEAD_DEBUG": "1", "CRA_VALUE_HEAD_VARIANT": "8"}, {"predicts": 80000, "runs": 0}),
    "big_alone_on_cu": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_DEBUG": "1", "CRA_VALUE_HEAD_LDS_PAD": "0"}, {"predicts": 80000, "runs": 0}),
    "big_default_again": ({"CRA_X3_VALUE_HEAD": "one", "CRA_VALUE_HEAD_LDS_PAD": "-1", "CRA_VALUE_HEAD_DEBUG": "1"}, {"predicts": 80000, "runs": 0}),
}


def child(name, runs, predicts):
    import numpy as np
    import nn_cases
    from crazyara_amd import _capi, openings, search
    from crazyara_amd.neuralnetapi import HipAPI, NeuralNetAPIUser

    env, opt = CONFIGS[name]
    precision = opt.get("precision", "float16x3")
    lanes = opt.get("lanes", 2)
    tmp = tempfile.mkdtemp(prefix="cra_lane_")
    cfg, sd, _ = nn_cases.make_case("risev2-3")
    d = nn_cases.export_case(tmp, "risev2-3", cfg, sd)
    fens = openings.position_fens("crazyhouse")[20:28]
    out = {"name": name, "env": env, "precision": precision, "lanes": lanes}

    # ---- (1) forwards of two nets from two host threads, fixed inputs, zero-copy predict: every output against the net alone ----
    nets = [HipAPI(0, 64, d, precision) for _ in range(2)]
    users = [NeuralNetAPIUser([n]) for n in nets]
    rng = np.random.default_rng(5)
    for u in users:
        u.input_planes[:] = (rng.random(u.input_planes.shape) < 0.1).astype(np.float32)
    ref = []
    for n, u in zip(nets, users):
        n.predict(u.input_planes, u.value_outputs, u.prob_outputs)
        ref.append((u.value_outputs.copy(), u.prob_outputs.copy()))
    bad = [[0, 0, 0], [0, 0, 0]]      # predicts that differ, of those: value differs, probabilities differ
    worst = [0.0, 0.0]
    predicts = opt.get("predicts", predicts)
    runs = opt.get("runs", runs)
    lib = _capi.load()
    dbg_view, dbg_ref, dbg_events = [None, None], [None, None], []
    if "CRA_VALUE_HEAD_DEBUG" in env:
        import ctypes as C
        import torch
        from crazyara_amd.neuralnetapi import _DevArray
        lib.mi_dev_value_head_debug.restype = C.c_void_p
        lib.mi_dev_value_head_debug.argtypes = [C.c_void_p]
        for i, n in enumerate(nets):
            ptr = lib.mi_dev_value_head_debug(n._h)
            assert ptr, "no debug buffer: CRA_VALUE_HEAD_DEBUG must be set before the net is built"
            dbg_view[i] = torch.as_tensor(_DevArray(ptr, (64 * (8 + 1024),)), device="cuda")
            n.predict(users[i].input_planes, users[i].value_outputs, users[i].prob_outputs)
            dbg_ref[i] = dbg_view[i].cpu().numpy().copy()
    stage_names = ["board", "conv_w", "conv_out", "fc1_parts", "fc2_sum", "value", "hw_id", "xcc_id"]

    def loop(i):
        n, u = nets[i], users[i]
        for _ in range(predicts):
            u.value_outputs[:] = np.nan
            n.predict(u.input_planes, u.value_outputs, u.prob_outputs)
            dv = not np.array_equal(u.value_outputs.view(np.uint32), ref[i][0].view(np.uint32))
            dp = not np.array_equal(u.prob_outputs.view(np.uint32), ref[i][1].view(np.uint32))
            if dv or dp:
                bad[i][0] += 1
                bad[i][1] += int(dv)
                bad[i][2] += int(dp)
                worst[i] = max(worst[i], float(np.nanmax(np.abs(u.value_outputs - ref[i][0]))) if not np.isnan(u.value_outputs).any() else float("inf"))
                if dv and dbg_view[i] is not None and len(dbg_events) < 40:
                    raw = dbg_view[i].cpu().numpy()
                    got = raw[:64 * 8].reshape(64, 8)
                    parts, parts_ref = raw[64 * 8:].reshape(64, 1024), dbg_ref[i][64 * 8:].reshape(64, 1024)
                    dbg_ref8 = dbg_ref[i][:64 * 8].reshape(64, 8)
                    for b in np.nonzero(u.value_outputs.view(np.uint32) != ref[i][0].view(np.uint32))[0]:
                        wrong = np.nonzero(parts[b].view(np.uint32) != parts_ref[b].view(np.uint32))[0]
                        diff = [stage_names[c] for c in range(6) if got[b, c].view(np.uint32) != dbg_ref8[b, c].view(np.uint32)]
                        dbg_events.append({"net": i, "board": int(b), "stages_that_differ": diff,
                                           "got": [float(v) for v in got[b, :6]], "ref": [float(v) for v in dbg_ref8[b, :6]],
                                           "wrong_partial_sums": [int(w) for w in wrong[:64]], "n_wrong": int(len(wrong)),
                                           "wrong_delta": [float(parts[b, w] - parts_ref[b, w]) for w in wrong[:16]],
                                           "hw_id": hex(int(got[b, 6].view(np.uint32))), "xcc_id": hex(int(got[b, 7].view(np.uint32))),
                                           "ref_hw_id": hex(int(dbg_ref8[b, 6].view(np.uint32))), "ref_xcc_id": hex(int(dbg_ref8[b, 7].view(np.uint32))),
                                           "value": float(u.value_outputs[b]), "ref_value": float(ref[i][0][b])})
    th = [threading.Thread(target=loop, args=(i,)) for i in range(2)]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    out["concurrent_predicts"] = {"per_net": predicts, "differing": bad, "worst_value_delta": worst, "seconds": round(time.time() - t0, 2)}
    if dbg_events:
        out["value_head_stage_events"] = dbg_events
    for u in users:
        u.close()

    # ---- (2) the searches, recorded and replayed ----
    os.environ["CRA_LANE_RECORD"] = "1"
    first = None
    equal = 0
    replay_bad = 0
    reports = []
    t0 = time.time()
    for r in range(runs):
        st = search.default_settings(mode=0, version_major=1, batch_size=16, seed=3)
        pool = search.SearchPool(st, net_a=nets[0], net_b=nets[1] if lanes == 2 else None)
        for f in fens:
            pool.add_position(f, False, "crazyhouse")
        pool.run(simulations=240, threads=4)
        dumps = [pool.tree_dump(i).tobytes() for i in range(len(fens))]
        n_bad, text = pool.debug_replay()
        replay_bad += n_bad
        if n_bad and len(reports) < 6:
            reports.append(f"run {r}: " + text[-1500:])
        if first is None:
            first = dumps
            equal += 1
        else:
            same = [a == b for a, b in zip(first, dumps)]
            equal += int(all(same))
            if not all(same) and len(reports) < 6:
                reports.append(f"run {r}: trees that differ from run 0: {[i for i, s in enumerate(same) if not s]} (replay: {n_bad} differing words)")
        pool.close()
    out["searches"] = {"runs": runs, "equal_to_first": equal, "replay_differing_words": replay_bad, "seconds": round(time.time() - t0, 2)}
    out["reports"] = reports
    for n in nets:
        n.close()
    print("RESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child")
    ap.add_argument("--runs", type=int, default=60)
    ap.add_argument("--predicts", type=int, default=300)
    ap.add_argument("--configs", default=",".join(CONFIGS))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.child:
        child(a.child, a.runs, a.predicts)
        return
    lines = []
    for name in a.configs.split(","):
        env = dict(os.environ)
        env.update(CONFIGS[name][0])
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, "--runs", str(a.runs), "--predicts", str(a.predicts)],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            line = res[-1][7:] if res else json.dumps({"name": name, "error": r.stdout[-1500:], "rc": r.returncode})
        except subprocess.TimeoutExpired:
            line = json.dumps({"name": name, "error": "timeout"})
        print(f"[{time.time() - t0:6.1f} s] {line}", flush=True)
        lines.append(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
