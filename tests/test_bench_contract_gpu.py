"""The driver's contract for bench.py: one JSON line on stdout with the agreed keys (short run, GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_training_line_has_the_contract_keys():
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["unit"] == "tiles/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 100
    assert abs(d["value"] - 16 * 1000.0 / d["ms_per_step"]) < 0.01 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_sampling_line():
    d = _run("--mode", "sample", "--euler-steps", "3", "--no-cpu-baseline")
    assert d["unit"] == "tiles/s" and d["config"]["finite"] is True and d["roofline"]["achieved"] > 100


def test_pix2pix_line():
    """Separate line for the pix2pix G + D step (row a13 is not in the reference)."""
    d = _run("--mode", "pix2pix")
    assert d["unit"] == "tiles/s" and "pix2pix" in d["metric"] and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["achieved"] > 10
    assert d["config"]["loss_d"] == d["config"]["loss_d"] and d["config"]["loss_g"] == d["config"]["loss_g"]   # finite
