"""bench.py and __graft_entry__.smoke() run end to end on a reduced problem: the JSON line carries the contract's
fields (metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling / dtype / data / config.workload) and
the roofline object, and nothing in the solver's interfaces has drifted away from what the benchmark wraps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_small_run_emits_contract_line():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "30000", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert "workload" in d["config"] and d["data"] == "synthetic"
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert abs(d["value"] - 30000 * 8192 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]


def test_graft_entry_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
