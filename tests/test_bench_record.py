"""bench.py's stdout line stays parseable: round 4's line grew to 21.7 KB and the driver's record lost `roofline` / `cpu_baseline`
(VERDICT r04).  compact_record() is pure, so the whole of round 4's committed result (profiles/r04/aj_bench.json) is fed through it
here: the line must stay under 6 KB, carry the contract's keys, the roofline and the CPU baseline, and name the detail file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _round4_full():
    old = json.load(open(os.path.join(ROOT, "profiles", "r04", "aj_bench.json")))
    full = {k: old[k] for k in ("metric", "value", "value_precision", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "config", "timed_region", "roofline", "cpu_baseline", "pcie_inclusive",
                                "dropin_reference_search", "mcts_configs", "game_configs", "mcts", "mcts_other_mode")}
    full["modes"] = {m: old[m] for m in ("float16", "float16x3", "float32", "fp8")}
    full["bench_seconds"] = old["summary"]["bench_seconds"]
    return full


def test_line_is_small_and_complete():
    import bench
    full = _round4_full()
    line = bench.compact_record(full, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 6144, len(text)
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in back, k
    assert back["roofline"]["frac"] == full["roofline"]["frac"] and back["roofline"]["bound"] == "mfma"
    assert back["roofline"]["achieved"] and back["roofline"]["peak"] and back["roofline"]["unit"] == "TFLOP/s"
    assert "traffic" in back["roofline"] and "per_op_ms" in back["roofline"]
    assert back["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and back["cpu_baseline"]["cores"] and back["cpu_baseline"]["kind"]
    assert back["detail"] == "bench_detail.json"
    assert list(back)[-1] == "summary"                                       # a truncating log keeps the tail
    # no companion rates inside roofline, nothing repeated at the top level
    assert not [k for k in back["roofline"] if "nodes_per_sec" in k or "evals_per_sec" in k]
    assert not [k for k in back if k.startswith(("config2_", "value_float16"))]
    # SURVEY 8(d) Metric 1 (the reference's `inference` loop, host copies included) sits at the top level beside the device-resident `value`
    assert back["value_device_resident"] == back["value"]
    assert back["value_pcie_inclusive_one_user"] == full["pcie_inclusive"]["one_net_evals_per_sec"]
    assert back["value_pcie_inclusive_two_users"] == full["pcie_inclusive"]["two_nets_in_flight_evals_per_sec"]
    # the other half of the metric and the reference-default mode are in the summary
    s = back["summary"]
    assert s["config2_mcts_nodes_per_sec_float16p8"] == full["mcts"]["mcts_nodes_per_sec"]
    assert s["nn_evals_per_sec_float16"] == full["modes"]["float16"]["evals_per_sec"]
    assert s["config3_nodes_per_sec"] == full["mcts_configs"]["config3"]["mcts_nodes_per_sec"]


def test_line_shrinks_when_the_result_grows():
    import bench
    full = _round4_full()
    for i in range(200):                                                      # a run with many more legs than today's
        full["mcts_configs"][f"config1_extra_leg_{i}"] = dict(full["mcts_configs"]["config1"])
    line = bench.compact_record(full, "d.json")
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert line["roofline"]["frac"] and line["cpu_baseline"]["value"]


def test_round6_reporting_keys():
    """NN evals/s at batch 8 / 512 / 1024 and config 3 with both value heads, the copy-path rates, the dominant kernel
    against the RAW dense f16 peak, the whole forward on the timed region's clock -- and, from a multi-rank run, configs 4 and 5 as sharded
    game legs with per-rank arrays."""
    import bench
    full = _round4_full()
    full["nn_by_batch"] = {k: {"evals_per_sec": 1000.0 + i, "ms_per_step": 0.5, "frac": 0.1} for i, k in
                           enumerate(("batch8", "batch512", "batch1024", "config3_tanh", "config3_wdlp"))}
    full["pcie_inclusive"].update({"copy_path_one_net_evals_per_sec": 1.0, "copy_path_two_nets_evals_per_sec": 2.0,
                                   "zero_copy_forced_two_nets_evals_per_sec": 3.0})
    full["roofline"]["frac_of_dense_f16_peak"] = 0.17
    full["roofline"]["whole_forward"] = {"ms_per_step": 0.6, "event_ms_per_step": 0.66, "achieved": 400.0, "frac": 0.33}
    full["game_configs"] = {"config4_selfplay": {"games_per_min": 900.0, "games": 24, "moves": 1000, "seconds": 1.6, "mcts_nodes_per_sec": 1.5e6,
                                                 "per_rank_games_per_min": [300.0, 310.0, 290.0], "per_rank_games": [8, 8, 8]},
                            "config5_arena": {"games_per_min": 2000.0, "games": 96, "moves": 5000, "seconds": 2.9, "mcts_nodes_per_sec": 1.8e6,
                                              "per_rank_games_per_min": [700.0, 650.0, 660.0], "per_rank_games": [32, 32, 32]}}
    line = bench.compact_record(full, "d.json")
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    s = line["summary"]
    for k in ("batch8", "batch512", "batch1024", "config3_tanh", "config3_wdlp"):
        assert s[f"nn_evals_per_sec_{k}"] == full["nn_by_batch"][k]["evals_per_sec"]
    assert s["copy_path_one_user"] == 1.0 and s["copy_path_two_users"] == 2.0 and s["zero_copy_forced_two_users"] == 3.0
    assert line["roofline"]["frac_of_dense_f16_peak"] == 0.17
    assert line["roofline"]["whole_forward_ms"] == 0.6 and line["roofline"]["whole_forward_frac"] == 0.33
    assert s["config4_selfplay_games_per_min"] == 900.0 and s["config4_selfplay_per_rank_games_per_min"] == [300.0, 310.0, 290.0]
    assert s["config5_arena_games_per_min"] == 2000.0 and len(s["config5_arena_per_rank_games_per_min"]) == 3


def test_line_limit_holds_for_an_oversized_result_and_for_legs_without_rates():
    """a large roofline.pmc block, many modes and game legs must not push the line past the limit, and a leg that recorded a skip
    or an error (no rate field) must not raise after the whole bench has run."""
    import bench
    full = _round4_full()
    full["roofline"]["pmc"] = {f"COUNTER_{i}": 123456789.0 + i for i in range(150)}
    for i in range(60):
        full["modes"][f"mode_{i}"] = dict(full["modes"]["float16"])
        full["game_configs"][f"config_extra_{i}"] = {"games_per_min": 1.0, "mcts_nodes_per_sec": 2.0, "games_per_min_float16": 3.0}
    full["modes"]["broken"] = {"error": "hipErrorOutOfMemory"}
    full["mcts_configs"]["config_skipped"] = {"skipped": "no reference build"}
    full["game_configs"]["config_failed"] = {"error": "x"}
    line = bench.compact_record(full, "d.json")
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert line["summary"]["nn_evals_per_sec"] == full["value"]
