"""``python bench.py --gpus N`` starts its own ranks (no torch.distributed.run wrapper needed) and rank 0 reports how
many ranks the backend's all-reduce summed over.  Driven here with 2 ranks over gloo on CPU (XGPR_DIST_BACKEND);
on a GPU node the same launcher starts one RCCL rank per GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(extra_env, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    return res


def test_self_launch_two_gloo_ranks():
    res = _run({"XGPR_DIST_BACKEND": "gloo"}, "--gpus", "2", "--dist-check", "--rows", "1000001")
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout          # rank 0 prints the one JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2
    assert out["n_ranks_seen"] == 2             # the all-reduce of ones really summed over two ranks
    assert out["backend"] == "gloo"
    assert out["shard_rows_per_rank"] == [500001.0, 500000.0]
    assert out["allreduce_w_us_back_to_back"] > 0 and out["allreduce_w_bytes"] == 8 * 8192


def test_child_failure_is_propagated():
    # the ranks cannot initialise (unknown backend): torch.distributed.run exits non-zero, and so must the launcher
    res = _run({"XGPR_DIST_BACKEND": "no_such_backend"}, "--gpus", "2", "--dist-check")
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
