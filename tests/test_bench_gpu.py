"""The measurement contract of bench.py as the driver parses it: ONE JSON line, the contract's keys, and a `roofline` object that STARTS with
scalars (the driver's record keeps its first ~23 keys and drops unknown top-level keys: VERDICT r05 weak #4)."""
import json
import os
import subprocess
import sys

import pytest

gpu = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline"]
ROOFLINE_HEAD = ["bound", "achieved", "peak", "unit", "frac", "traffic"]
ROOFLINE_SCALARS = ["frac_algorithmic", "dominant_ms", "dominant_frac_algorithmic", "dominant_frac_executed", "stage1_hbm_frac", "stage1_kernel_ms",
                    "exact_value", "exact_direct_value", "latency_ms_per_image", "train_step_ms", "kernel_ms_per_step", "share_of_step",
                    "launches_per_step", "value_with_dead_layer1_computed", "useful_tflops", "step_direct_conv_equiv_tflops"]


@gpu
def test_bench_line_contract_on_a_small_workload():
    """`python bench.py --workload full_b8_n42_vits --steps 2 --warmup 1 --no-latency-leg` (a child process): the line a driver reads."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "full_b8_n42_vits", "--steps", "2", "--warmup", "1",
                        "--no-latency-leg"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert d["unit"] == "crops/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["config"]["workload"].startswith("full_b8_n42_vits") and d["config"]["input_batches"] == 2 and d["config"]["prefetch_query"] is True
    roof = d["roofline"]
    keys = list(roof)
    assert keys[:6] == ROOFLINE_HEAD, keys[:6]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    head = keys[:23]
    for k in ROOFLINE_SCALARS:
        assert k in head, (k, head)
    for k in head[1:]:
        assert k in ("unit",) or roof[k] is None or isinstance(roof[k], (int, float)), (k, roof[k])       # scalars first, strings after
    for k in keys[23:]:
        if isinstance(roof[k], str):
            assert len(roof[k]) <= 100, (k, len(roof[k]))
    assert roof["exact_value"] == d["exact_value"] and roof["stage1_hbm_frac"] == d["roofline_stage1"]["frac"]
    assert roof["value_with_dead_layer1_computed"] == d["dead_layer1_computed"]["value"] and d["config"]["dpt_layer1_branch"].startswith("not computed")
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "crops/s" and isinstance(cb["sample"], str)
    x = d["exact_mode"]["f16x3_vs_exact"]
    assert x["pairs_with_same_template"] > 0 and x["flow_max_abs_px"] < 1e-2 and x["keypoint_slots_equal"] > 0.999
