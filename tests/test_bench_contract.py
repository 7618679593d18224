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
