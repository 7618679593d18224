"""The driver's contract with bench.py (one JSON line on stdout, the metric of BASELINE.json, roofline and
cpu_baseline objects) checked on a short run.  GPU only: the render path has no CPU fallback."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_one_json_line_with_roofline_and_cpu_baseline():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2",
                          "--voices", "131072"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["steps"] == 10 and d["warmup"] == 2 and d["n_gpus"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] in str(base.get("unit", d["unit"])) or d["metric"].startswith("voice-samples")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["value"] > 100 * c["value"] / c["cores"]          # sanity: a GPU, not a fallback
    assert r["launches_timed"] >= 5                            # (10 steps: every launch bracketed)
    # the driver-run line also carries the other BASELINE workloads and the harder regimes of the main one
    for key in ("c1", "c2", "c4", "low_latency", "long_block"):
        o = d[key]
        assert o["value"] > 0 and o["ms_per_step"] > 0 and o["kernel"].startswith("sk_render") and o["kernel_ms_mean"] > 0
        assert abs(o["frac"] - o["achieved"] / o["peak"]) < 1e-9
        assert o["launches_timed"] >= 3
    assert d["c1"]["voices"] == 4096 and d["c2"]["voices"] == 65536 and d["c4"]["voices"] == 262144
    for key in ("envelopes_in_motion", "live_control", "fixed_point"):
        assert d[key]["value"] > 0 and d[key]["ms_per_step"] > 0
    assert d["envelopes_in_motion"]["ms_per_step"] >= d["ms_per_step"] * 0.9      # ramps cost, they never speed a block up


@pytest.mark.gpu
def test_n_gt_1_code_path_through_a_one_rank_rccl_group():
    """bench.py --rehearse-dist: the shard's RCCL reduce with one rank (what the driver runs with N ranks)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29591", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--voices", "131072",
                          "--rehearse-dist", "--no-cpu"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-1500:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "strong" and d["output_finite"]
    assert "ncclReduce" in d["config"]["parallelism"]
