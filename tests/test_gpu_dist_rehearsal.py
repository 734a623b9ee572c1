"""The collective path on the one GPU of the test box: the driver's launch line with ONE rank whose collectives go
through the backend (XGPR_DIST_FORCE=1) -- RCCL communicator set up on the device, all-reduce and barrier issued for
real.  What N ranks add to this (the exchange over xGMI) only the driver's multi-GPU run can show."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_one_rank_through_rccl():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"XGPR_DIST_FORCE": "1", "XGPR_RCCL_DIRECT": "1",        # the direct path is opt-in
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-check", "--rows", "1000"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["backend"] == "nccl" and out["world_size"] == 1 and out["n_ranks_seen"] == 1
    assert out["allreduce_w_us_back_to_back"] > 0 and out["shard_rows_per_rank"] == [1000.0]
    # the float64 sums go through the C ABI's on-stream all-reduce (a communicator of its own), and agree with torch's
    assert out["allreduce_path"].startswith("xgpr_allreduce_sum_f64") and out["direct_equals_torch_allreduce"] is True


def test_one_rank_torch_distributed_fallback():
    """The default (XGPR_RCCL_DIRECT unset) keeps every collective on torch.distributed."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "XGPR_RCCL_DIRECT")}
    env.update({"XGPR_DIST_FORCE": "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29548", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-check", "--rows", "1000"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["allreduce_path"].startswith("torch.distributed") and out["n_ranks_seen"] == 1


def test_whole_bench_through_the_backend_with_one_rank():
    """The driver's launch line, every collective of the run issued through RCCL (one rank): data generation, the
    preconditioner's all-reduces, the timed iterations, the solve to tolerance and the reporting all complete and the
    line carries the distributed fields."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "XGPR_RCCL_DIRECT")}
    env.update({"XGPR_DIST_FORCE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rows", "40000", "--steps", "4",
           "--warmup", "1", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    dist = out["distributed"]
    assert dist["backend"] == "nccl" and dist["n_ranks_seen"] == 1 and dist["allreduce_path"].startswith("torch.distributed")
    assert len(dist["per_rank"]) == 1 and dist["per_rank"][0]["rows"] == 40000 and dist["per_rank"][0]["allreduce_ms_per_iter"] > 0
    assert out["steps"] == 4 and out["value"] > 0 and out["fit_to_tol"]["converged"] is True
    assert out["final_loss_check"]["ok"] is None          # no stored value for this size


def _two_gloo_ranks(extra_env, tmp_path):
    """`python bench.py --gpus 2` (it starts its own ranks) with both ranks on device 0 over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "XGPR_RCCL_DIRECT", "XGPR_BENCH_CHILD")}
    env.update({"XGPR_DIST_BACKEND": "gloo", "XGPR_LOCAL_DEVICE": "0", "XGPR_BENCH_CHILD_FILE": str(tmp_path / "child.json")})
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "31250", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-configs"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, lines


def test_multi_rank_bench_starts_the_direct_path_rehearsal_after_its_line(tmp_path):
    """N > 1: after rank 0 has printed the job's line, a FRESH job with XGPR_RCCL_DIRECT=1 runs under a watchdog
    (bench.direct_rccl_child) and reports to stderr and a side file; stdout stays the one line, the exit code the job's
    own.  (Over gloo the child keeps the torch path -- the direct one needs RCCL -- which is what this rehearsal can show.)"""
    res, lines = _two_gloo_ranks({}, tmp_path)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["distributed"]["n_ranks_seen"] == 2 and len(out["build_id"]) == 64
    assert "direct-rccl child:" in res.stderr
    rec = json.load(open(tmp_path / "child.json"))
    assert rec["status"] == "exited 0", rec
    assert rec["line"]["n_gpus"] == 2 and rec["line"]["n_ranks_seen"] == 2 and rec["line"]["build_id"] == out["build_id"]
    assert rec["line"]["allreduce_path"].startswith("torch.distributed")
    assert len(rec["line"]["allreduce_ms_per_iter"]) == 2


def test_a_direct_path_rehearsal_that_does_not_finish_changes_nothing(tmp_path):
    """The watchdog at 1 s: the child (which needs far longer to start its ranks) is killed with its process group; the
    job's line and exit code are what they are without it."""
    res, lines = _two_gloo_ranks({"XGPR_BENCH_CHILD_TIMEOUT": "1"}, tmp_path)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    rec = json.load(open(tmp_path / "child.json"))
    assert rec["status"].startswith("timed out after 1 s") and rec["line"] is None
