"""bench.py's multi-rank path without a GPU: (1) the synthetic dataset is ONE dataset whatever the number of ranks
(the shards of 1, 2, 4, 8 ranks concatenate to identical (x, y)); (2) a solve of it gives the same final loss on one
rank and on two gloo ranks, within the tolerance ``final_loss_check`` applies when sums are exchanged, and that check is
reached and passes on every rank; (3) the opt-in direct-RCCL set-up, with ``xgpr_rccl_comm_init`` failing on ONE rank
only, is dropped by both ranks together without a hang and the sums keep working through torch.distributed.

The product's hot operators only exist on the GPU, so -- in the test only -- the kernel object is served from the CPU
oracle (tests/test_dist_cpu.py's double); dataset, CG driver, communicator and bench.py's own functions are the product's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

ROWS, DIM, RFFS, STEPS = 2400, 16, 64, 12


def _bounds(n, world, rank):
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@pytest.mark.parametrize("n", [40000, 15625 * 3 + 17, 1000])
def test_shards_of_any_world_size_concatenate_to_one_dataset(n):
    import bench
    dev = torch.device("cpu")
    x1, y1 = bench.make_shard(0, n, 8, dev)
    assert x1.shape == (n, 8) and y1.shape == (n,) and x1.dtype == torch.float32 and y1.dtype == torch.float64
    for world in (2, 4, 8):
        parts = [bench.make_shard(*_bounds(n, world, r), 8, dev) for r in range(world)]
        assert torch.equal(torch.cat([p[0] for p in parts]), x1), world
        assert torch.equal(torch.cat([p[1] for p in parts]), y1), world
    xe, ye = bench.make_shard(7, 7, 8, dev)
    assert xe.shape == (0, 8) and ye.shape == (0,)


def test_default_bench_shards_align_with_the_seeded_blocks():
    import bench
    for world in (1, 2, 4, 8):
        for r in range(world):
            lo, hi = _bounds(1_000_000, world, r)
            assert lo % bench.DATA_BLOCK_ROWS == 0 and hi % bench.DATA_BLOCK_ROWS == 0


def test_final_loss_check_logic():
    import bench
    key = (10, 2, 4, 1, 5)
    table = {key: 0.0125}
    assert bench.final_loss_check((11, 2, 4, 1, 5), 0.3, 1, table) == {"expected": None, "rtol": None, "ok": None}
    assert bench.final_loss_check(key, 0.0125 * (1 + 5e-7), 1, table)["ok"] is True
    assert bench.final_loss_check(key, 0.0125 * (1 + 5e-6), 1, table)["ok"] is False      # one rank: 1e-6
    assert bench.final_loss_check(key, 0.0125 * (1 + 5e-6), 8, table)["ok"] is True       # exchanged sums: 1e-5
    assert bench.final_loss_check(key, 0.0125 * (1 + 5e-5), 8, table)["ok"] is False
    assert bench.final_loss_check(key, float("nan"), 2, table)["ok"] is False
    # the stored table's keys are (rows, dim, rffs, rank, steps) -- no world size: one dataset, one value for every N
    assert all(len(k) == 5 for k in bench.EXPECTED_FINAL_LOSS)


def _solve(comm):
    """bench.py's steps on a small problem: this rank's rows of the dataset, the product's dataset + CG driver."""
    import bench
    from test_dist_cpu import OracleBackedKernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import ConjugateGrad, calc_zty
    lo, hi = comm.shard_bounds(ROWS)
    x, y = bench.make_shard(lo, hi, DIM, torch.device("cpu"))
    ds = build_regression_dataset(x, y, chunk_size=500, device="cpu", comm=comm, already_sharded=True)
    assert ds.get_ndatapoints() == ROWS and ds.get_local_ndatapoints() == hi - lo
    kern = OracleBackedKernel(RFFS, DIM, np.array([0.1, 1.0]))
    zty, _ = calc_zty(ds, kern)
    resid = torch.zeros((RFFS, 2, 1), dtype=torch.float64)
    resid[:, 0, 0] = zty / ROWS
    _, _, niter, losses = ConjugateGrad(comm).fit(ds, kern, None, resid, maxiter=STEPS, tol=0.0, verbose=False)
    assert niter == STEPS
    return losses[-1]


def _worker(rank, world, port, outdir, expected):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import bench
    from xgpr_amd import dist as xd
    comm = xd.init_from_env(device_type="cpu")
    loss = _solve(comm)
    key = (ROWS, DIM, RFFS, 0, STEPS)
    chk = bench.final_loss_check(key, loss, comm.world_size, {key: expected})
    # what follows the timed region on rank 0 only (probes, CPU baseline, printing) must hold no collective: rank 1
    # tears down here while rank 0 still "works"
    if rank == 0:
        import time
        time.sleep(0.5)
    np.savez(os.path.join(outdir, f"loss_rank{rank}.npz"), loss=loss, ok=chk["ok"], rtol=chk["rtol"])
    torch.distributed.destroy_process_group()


def test_two_gloo_ranks_reach_the_loss_check_and_pass_it(tmp_path):
    from xgpr_amd import dist as xd
    expected = _solve(xd.Comm())                      # one rank, this process
    assert 0 < expected < 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), expected), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "loss_rank0.npz"), np.load(tmp_path / "loss_rank1.npz")
    assert float(r0["loss"]) == float(r1["loss"])     # replicated CG state
    assert bool(r0["ok"]) and bool(r1["ok"]) and float(r0["rtol"]) == 1e-5
    # same dataset, sums in another order: rounding differences of 1e-16, amplified by the (un-preconditioned) recurrence
    assert abs(float(r0["loss"]) / expected - 1.0) < 1e-6


def test_eight_gloo_ranks_solve_the_same_problem(tmp_path):
    """The world size the driver's scaling run ends at: eight ranks (300 rows each, one thread each) reach and pass the loss
    check; every rank holds the same replicated CG state; the loss is the one-rank value to the exchanged-sums tolerance."""
    from xgpr_amd import dist as xd
    expected = _solve(xd.Comm())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(8, port, str(tmp_path), expected), nprocs=8, join=True)
    res = [np.load(tmp_path / f"loss_rank{r}.npz") for r in range(8)]
    assert len({float(r["loss"]) for r in res}) == 1
    assert all(bool(r["ok"]) and float(r["rtol"]) == 1e-5 for r in res)
    # (eight partial sums per dot product instead of one: 1.9e-6 after 12 un-preconditioned iterations; two ranks: < 1e-6)
    assert abs(float(res[0]["loss"]) / expected - 1.0) < 1e-5


class _FakeRcclLib:
    """Stands in for libxgpr_hip.so's xgpr_rccl_* entry points: set-up succeeds everywhere except
    xgpr_rccl_comm_init on ``fail_rank``."""

    def __init__(self, rank, fail_rank, log):
        self.rank, self.fail_rank, self.log = rank, fail_rank, log

    def xgpr_rccl_unique_id(self, buf):
        return 0

    def xgpr_rccl_comm_init(self, handle_ref, nranks, ident, rank):
        self.log.append("init")
        return -5 if rank == self.fail_rank else 0

    def xgpr_allreduce_sum_f64(self, *a):
        self.log.append("direct_sum")          # would block for ever on a communicator only one rank holds
        return 0

    def xgpr_rccl_comm_destroy(self, handle):
        self.log.append("destroy")
        return 0

    def xgpr_last_error(self):
        return b"injected"


def _worker_rccl(rank, world, port, outdir, fail_rank):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), XGPR_RCCL_DIRECT="1",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from xgpr_amd import dist as xd
    comm = xd.init_from_env(device_type="cpu")          # gloo: the direct path is not even considered
    assert comm.direct_rccl is False
    log = []
    comm._direct_requested = lambda: True               # as on an RCCL job with XGPR_RCCL_DIRECT=1
    comm._load_rccl_entry_points = lambda: _FakeRcclLib(rank, fail_rank, log)
    used = comm.enable_direct_rccl(torch.device("cpu"))
    v = torch.full((4,), float(rank + 1), dtype=torch.float64)
    comm.all_reduce_(v)                                  # the sums still work, through torch.distributed
    np.savez(os.path.join(outdir, f"rccl_rank{rank}.npz"), used=used, direct=comm.direct_rccl, v=v.numpy(),
             log=np.array(log))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [0, 1])
def test_one_sided_rccl_init_failure_drops_the_direct_path_on_both_ranks(tmp_path, fail_rank):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_rccl, args=(2, port, str(tmp_path), fail_rank), nprocs=2, join=True)
    for rank in (0, 1):
        r = np.load(tmp_path / f"rccl_rank{rank}.npz")
        assert not bool(r["used"]) and not bool(r["direct"])
        assert np.array_equal(r["v"], np.full(4, 3.0))
        log = list(r["log"])
        assert "direct_sum" not in log                   # nobody issued a collective on the half-made communicator
        assert log == (["init"] if rank == fail_rank else ["init", "destroy"])


# ---- the follow-up job that rehearses the opt-in direct RCCL path (bench.direct_rccl_child): started after the main line,
# watched, and never able to change the parent's stdout / exit code
class _Args:
    gpus, warmup, rows, dim, rffs, rank_precond = 2, 1, 2400, 16, 64, 8


def _child_line():
    return {"n_gpus": 2, "ms_per_step": 1.25, "value": 3.0e9, "final_loss": 0.5, "build_id": "ab" * 32,
            "final_loss_check": {"expected": 0.5, "rtol": 1e-5, "ok": True},
            "distributed": {"n_ranks_seen": 2, "allreduce_path": "xgpr_allreduce_sum_f64", "direct_equals_torch_allreduce": True,
                            "allreduce_w_us_back_to_back": 21.0,
                            "per_rank": [{"rank": 0, "allreduce_ms_per_iter": 0.02}, {"rank": 1, "allreduce_ms_per_iter": 0.021}]}}


def test_direct_child_is_started_with_the_opt_in_and_its_line_is_recorded(tmp_path, capsys, monkeypatch):
    import json
    import bench
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("MASTER_PORT", "1")
    prog = ("import os, json, sys\n"
            "assert os.environ['XGPR_RCCL_DIRECT'] == '1' and os.environ['XGPR_BENCH_CHILD'] == '1'\n"
            "assert not any(k in os.environ for k in ('RANK', 'WORLD_SIZE', 'MASTER_PORT', 'LOCAL_RANK'))   # a fresh job: it starts its own ranks\n"
            "print('noise before the line'); print(json.dumps(%r)); sys.stderr.write('child stderr')\n" % (_child_line(),))
    side = tmp_path / "child.json"
    rec = bench.direct_rccl_child(_Args, limit_s=30, cmd=[sys.executable, "-c", prog.replace("True", "True")], side_file=str(side))
    assert rec["status"] == "exited 0"
    assert rec["line"]["direct_equals_torch_allreduce"] is True and rec["line"]["final_loss_check"]["ok"] is True
    assert rec["line"]["allreduce_ms_per_iter"] == [0.02, 0.021] and rec["line"]["n_ranks_seen"] == 2
    assert json.load(open(side))["line"]["ms_per_step"] == 1.25
    cap = capsys.readouterr()
    assert cap.out == "" and "direct-rccl child:" in cap.err           # stdout belongs to the ONE JSON line of the main job


def test_direct_child_that_hangs_is_killed_with_its_whole_process_group(tmp_path, capsys):
    import time
    import bench
    pidfile = tmp_path / "grandchild.pid"
    # the child starts a grandchild (as torchrun starts the ranks) and both sleep for ever, ignoring SIGTERM
    prog = ("import os, signal, subprocess, sys, time\n"
            "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
            "p = subprocess.Popen([sys.executable, '-c', 'import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(1000)'])\n"
            "open(%r, 'w').write(str(p.pid))\n"
            "time.sleep(1000)\n" % str(pidfile))
    t0 = time.perf_counter()
    rec = bench.direct_rccl_child(_Args, limit_s=3, cmd=[sys.executable, "-c", prog], side_file=str(tmp_path / "c.json"))
    assert time.perf_counter() - t0 < 40
    assert rec["status"].startswith("timed out after 3 s") and rec["line"] is None
    gpid = int(open(pidfile).read())
    for _ in range(50):
        try:
            os.kill(gpid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        pytest.fail("the hung child's rank process survived the watchdog")
    assert capsys.readouterr().out == ""


def test_direct_child_failures_never_escape(tmp_path, capsys):
    import bench
    rec = bench.direct_rccl_child(_Args, limit_s=5, cmd=["/nonexistent/python"], side_file=str(tmp_path / "a.json"))
    assert rec["status"].startswith("not started")
    rec = bench.direct_rccl_child(_Args, limit_s=20, cmd=[sys.executable, "-c", "import sys; print('no json here'); sys.exit(3)"],
                                  side_file=str(tmp_path / "b.json"))
    assert rec["status"] == "exited 3" and rec["line"] is None
    rec = bench.direct_rccl_child(_Args, limit_s="not a number", cmd=[sys.executable, "-c", "pass"], side_file="/proc/nope/x.json")
    assert rec["status"].startswith("error in the parent")
    assert capsys.readouterr().out == ""
