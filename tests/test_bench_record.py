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
    assert not [k for k in back if k.startswith(("config2_", "value_float16", "value_pcie"))]
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
