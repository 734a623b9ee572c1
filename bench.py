#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps 20 --warmup 3          (N > 1: starts its own N ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
Matern-5/2 kernel, N = 1e6 datapoints, d = 1024, 8192 random features, rows sharded
contiguously over the ranks (STRONG scaling: N is the whole job), rank-512 randomized-Nystrom
preconditioner.  Synthetic data (X ~ N(0,1)/sqrt(d) from the device RNG, seeded per GLOBAL row block -- the same
dataset at every N; y = sin(X a) + 0.1 eps), resident in HBM before the timed region.

A "step" is one preconditioned-CG iteration (reference fitting_toolkit/cg_tools.py:255-287):
one pass of the fused feature-generation + Z^T(Z p) kernel over all N rows (every iteration
regenerates all N x M random features; Z is never written), the RCCL all-reduce of w, the
preconditioner apply and the float64 vector updates, including the per-iteration convergence
check.  value = N * M * K / t  random features per second (whole job); cg_iters_per_sec = K / t.

Besides the contract fields the JSON line carries
  roofline      -- the dominant kernel (ztz3_kernel, the fused matvec): algorithmic HBM bytes (4*d per row)
                   over its measured duration (HIP events on the launch stream)
  featgen_op    -- the stand-alone cudaRBFFeatureGen-equivalent operator (Z materialised as
                   float64), with its own HBM roofline (4*d + 8*M bytes per row)
  precond_build -- the randomized-Nystrom build that precedes the timed steps: its dense work 2*N*rank*M
                   float64 flop over its wall time, against the FP64 matrix peak
  distributed   -- ranks the all-reduce summed over, per-rank step / kernel / all-reduce times
  conv_featgen  -- the convolution feature operator at BASELINE configs[3]'s shape (8192 sequences per call)
  cpu_baseline  -- the CPU oracle (OpenMP port of the reference CPU algorithm) timed on this
                   box's host cores on a bounded row sample (rank 0, --gpus 1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_MFMA_PEAK_TFLOPS = 78.6   # dense FP64 matrix peak: 256 CUs x 4 SIMDs x 2048 flop / 64 cycles x 2.4 GHz


PROFILE_ROUND = "r6"          # the committed rocprofv3 summaries this file quotes: profiles/r6_*


def stored_profile(name, build_id=None):
    """(object, note) of profiles/<PROFILE_ROUND>_<name>.  With ``build_id``: a stored MEASUREMENT of a kernel is quoted only
    while the library that produced it is the one loaded now (the summaries record xgpr_build_id() of the run they came from);
    on a mismatch the object is None and the note says why -- a stale figure is not carried silently across kernel changes."""
    path = os.path.join(ROOT, "profiles", "%s_%s" % (PROFILE_ROUND, name))
    try:
        obj = json.load(open(path))
    except (OSError, ValueError):
        return None, "profiles/%s_%s is missing" % (PROFILE_ROUND, name)
    if build_id is not None and obj.get("build_id") != build_id:
        return None, ("profiles/%s_%s was measured on build %s; the loaded library is %s: not quoted (re-run tools/collect_profiles.sh)"
                      % (PROFILE_ROUND, name, str(obj.get("build_id"))[:12], build_id[:12]))
    return obj, "profiles/%s_%s" % (PROFILE_ROUND, name)


def mfma_busy_of(kernels):
    """SQ_VALU_MFMA_BUSY_CYCLES over SIMD-cycles for the named kernels, from the short-launch counter pass
    (tools/r6_mfma_busy.sh -> profiles/r6_mfma_clock.json; stored)."""
    obj, src = stored_profile("mfma_clock.json")
    if obj is None:
        return None
    out = {}
    for k in kernels:
        ent = (obj.get(k) or [None])[0]
        if ent:
            out[k] = {"mfma_busy_frac_of_simd_cycles": ent["mfma_busy_frac_of_simd_cycles"], "clock_GHz": ent["clock_GHz"], "rows": ent["rows"]}
    out["source"] = src + " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE on launches short enough not to stop the counter; stored)"
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000, help="N, total over all ranks")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--rffs", type=int, default=8192)
    ap.add_argument("--rank-precond", type=int, default=512)
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="CPU-baseline budget: whole 8192-row chunks are processed until this much time is spent")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` object (build + fit of cfg2 / cfg4 / cfg5 at their per-GPU shares, ~15 s)")
    ap.add_argument("--no-direct-child", action="store_true",
                    help="N > 1: do not start the follow-up job that rehearses the opt-in direct RCCL path (direct_rccl_child)")
    ap.add_argument("--dist-check", action="store_true",
                    help="rendezvous only: start / join the ranks, count them with an all-reduce, time the per-iteration "
                         "all-reduce, print the JSON line and exit (no HIP kernels; also runs on CPU over gloo)")
    return ap.parse_args()


def self_launch(args):
    """``python bench.py --gpus N`` with N > 1 and no torch.distributed.run environment: start the N ranks here.
    Runs BEFORE anything touches the GPU (this process never does): a child ``python -m torch.distributed.run``
    (one process per GPU, rendezvous on 127.0.0.1 at a free port) is started with the same arguments, waited
    for, and its exit code becomes ours.  Never an exec of a process that has initialised the GPU."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


TORCHRUN_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RESTART_COUNT",
                "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE",
                "TORCH_NCCL_ASYNC_ERROR_HANDLING", "NCCL_ASYNC_ERROR_HANDLING")


def direct_rccl_child(args, limit_s=None, cmd=None, side_file=None):
    """First contact of the opt-in on-stream all-reduce (XGPR_RCCL_DIRECT=1: ``xgpr_allreduce_sum_f64`` on a communicator
    created through the C ABI, xgpr_amd/dist.py) with more than one GPU, without putting the driver's number at risk:
    AFTER rank 0 has printed the job's JSON line and the process group is gone, rank 0 starts a FRESH job -- a child
    ``python bench.py --gpus N --steps 20 --no-configs --no-cpu-baseline`` in a session of its own, which starts its own N
    ranks (self_launch; never an exec of this process, which has touched the GPU) -- under a wall-clock watchdog.  What the
    child's line says about the direct path (``direct_equals_torch_allreduce``, ``final_loss_check``, the all-reduce time
    per iteration, ms per step) goes to stderr and to a side file.  Nothing here can change this process's exit code or
    its stdout: every failure (no start, non-zero exit, no JSON, time-out -> the child's whole process group is killed)
    is a status string in that record.  Returns the record."""
    import signal
    rec = {"what": "bench.py --gpus %d with XGPR_RCCL_DIRECT=1, started after the main line was printed" % args.gpus,
           "status": None}
    try:
        limit_s = float(os.environ.get("XGPR_BENCH_CHILD_TIMEOUT", "180")) if limit_s is None else float(limit_s)
        if cmd is None:
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", "20", "--warmup", str(args.warmup),
                   "--rows", str(args.rows), "--dim", str(args.dim), "--rffs", str(args.rffs), "--rank-precond",
                   str(args.rank_precond), "--no-configs", "--no-cpu-baseline", "--no-direct-child"]
        env = {k: v for k, v in os.environ.items() if k not in TORCHRUN_ENV}
        env["XGPR_RCCL_DIRECT"] = "1"
        env["XGPR_BENCH_CHILD"] = "1"
        rec["limit_s"] = limit_s
        t0 = time.perf_counter()
        try:
            proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        except OSError as exc:
            rec["status"] = "not started: %s" % exc
            proc = None
        if proc is not None:
            try:
                out, err = proc.communicate(timeout=limit_s)
                rec["status"] = "exited %d" % proc.returncode
            except subprocess.TimeoutExpired:
                for sig, wait_s in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
                    try:
                        os.killpg(proc.pid, sig)          # the child's own session: its torchrun and every rank
                    except ProcessLookupError:
                        break
                    try:
                        proc.wait(timeout=wait_s)
                        break
                    except subprocess.TimeoutExpired:
                        continue
                try:
                    out, err = proc.communicate(timeout=5.0)
                except (subprocess.TimeoutExpired, ValueError, OSError):
                    out, err = b"", b""
                rec["status"] = "timed out after %.0f s: process group killed" % limit_s
            rec["seconds"] = time.perf_counter() - t0
            line = None
            for cand in reversed(out.decode(errors="replace").strip().splitlines()):
                if cand.startswith("{"):
                    try:
                        line = json.loads(cand)
                        break
                    except ValueError:
                        continue
            if line is None:
                rec["line"] = None
                rec["stderr_tail"] = err.decode(errors="replace")[-1500:]
            else:
                dist_o = line.get("distributed", {})
                rec["line"] = {"n_gpus": line.get("n_gpus"), "ms_per_step": line.get("ms_per_step"), "value": line.get("value"),
                               "allreduce_path": dist_o.get("allreduce_path"), "n_ranks_seen": dist_o.get("n_ranks_seen"),
                               "direct_equals_torch_allreduce": dist_o.get("direct_equals_torch_allreduce"),
                               "allreduce_w_us_back_to_back": dist_o.get("allreduce_w_us_back_to_back"),
                               "allreduce_ms_per_iter": [r.get("allreduce_ms_per_iter") for r in dist_o.get("per_rank", [])],
                               "final_loss": line.get("final_loss"), "final_loss_check": line.get("final_loss_check"),
                               "build_id": line.get("build_id")}
    except Exception as exc:        # noqa: BLE001 -- by contract nothing escapes
        rec["status"] = "error in the parent: %r" % (exc,)
    try:
        side_file = side_file or os.environ.get("XGPR_BENCH_CHILD_FILE") or os.path.join(ROOT, "gpurun_out", "bench_direct_rccl_child.json")
        os.makedirs(os.path.dirname(side_file), exist_ok=True)
        with open(side_file, "w") as f:
            json.dump(rec, f, indent=1)
        rec["side_file"] = side_file
    except OSError:
        pass
    try:
        print("direct-rccl child: " + json.dumps(rec), file=sys.stderr, flush=True)
    except Exception:               # noqa: BLE001
        pass
    return rec


def dist_summary(comm, device, m):
    """What rank 0 reports about the job's communicator: the number of ranks the backend's all-reduce actually
    summed over, the backend, and the time of the per-iteration exchange (the all-reduce of w: m float64) measured
    on this device with the other ranks taking part."""
    import torch
    import torch.distributed as dist
    ones = torch.ones(1, dtype=torch.float64, device=device)
    comm.all_reduce_(ones)
    out = {"n_ranks_seen": int(round(float(ones.item()))), "world_size": comm.world_size,
           "backend": dist.get_backend() if comm.through_backend else "none (single rank)",
           "allreduce_path": ("xgpr_allreduce_sum_f64: ncclAllReduce enqueued on the compute stream (communicator created through "
                              "the C ABI)" if getattr(comm, "direct_rccl", False) else
                              "torch.distributed.all_reduce (ProcessGroup's stream, chained to the compute stream with events)")
                             if comm.through_backend else "none"}
    if comm.through_backend and getattr(comm, "direct_rccl", False):
        # cross-check of the direct path against torch.distributed's own all-reduce on a non-trivial vector
        import torch.distributed as dist2
        chk = torch.arange(1, 1025, dtype=torch.float64, device=device) * (comm.rank + 1)
        ref = chk.clone()
        comm.all_reduce_(chk)
        dist2.all_reduce(ref, op=dist2.ReduceOp.SUM)
        out["direct_equals_torch_allreduce"] = bool(torch.equal(chk, ref))
    if comm.through_backend:
        w = torch.zeros(m, dtype=torch.float64, device=device)
        for _ in range(5):
            comm.all_reduce_(w)
        comm.barrier()
        reps = 50
        if device.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                comm.all_reduce_(w)
            e1.record()
            torch.cuda.synchronize()
            out["allreduce_w_us_back_to_back"] = 1e3 * e0.elapsed_time(e1) / reps
        else:
            t0 = time.perf_counter()
            for _ in range(reps):
                comm.all_reduce_(w)
            out["allreduce_w_us_back_to_back"] = 1e6 * (time.perf_counter() - t0) / reps
        out["allreduce_w_bytes"] = 8 * m
    return out


def gather_per_rank(comm, device, values):
    """values: list of floats of this rank -> list (per rank) of lists, on every rank."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor(values, dtype=torch.float64, device=device)
    if not comm.through_backend:
        return [mine.tolist()]
    bufs = [torch.zeros_like(mine) for _ in range(comm.world_size)]
    dist.all_gather(bufs, mine)
    return [b.tolist() for b in bufs]


DATA_BLOCK_ROWS = 15625       # rows per seeded block of the synthetic dataset (1e6 rows = 64 blocks; 125 000 = 8 blocks)


def make_shard(lo, hi, d, device):
    """Rows [lo, hi) of THE synthetic dataset -- the same dataset whatever the number of ranks.  Row block b (global rows
    b*DATA_BLOCK_ROWS ...) has its own generator seeded with b, is always drawn (and its y computed) at the full block
    shape, then sliced to the shard, so that the concatenation of the shards of 1, 2, 4 or 8 ranks is one and the same
    (x, y) bit for bit: X ~ N(0,1)/sqrt(d) float32, y = sin(X a) + 0.1 eps float64."""
    import numpy as np
    import torch
    ga = torch.Generator(device=device)
    ga.manual_seed(7)
    a = torch.randn(d, generator=ga, device=device, dtype=torch.float32) * 3.0
    x = torch.empty((hi - lo, d), dtype=torch.float32, device=device)
    y = torch.empty(hi - lo, dtype=torch.float64, device=device)
    B = DATA_BLOCK_ROWS
    for b in range(lo // B, (hi + B - 1) // B if hi > lo else lo // B):
        gen = torch.Generator(device=device)
        gen.manual_seed(123_000_000 + b)
        xb = torch.randn((B, d), generator=gen, device=device, dtype=torch.float32) / np.sqrt(d)
        yb = torch.sin(xb @ a).double() + 0.1 * torch.randn(B, generator=gen, device=device, dtype=torch.float64)
        s0, s1 = max(lo, b * B), min(hi, (b + 1) * B)          # global rows of this block held by the shard
        x[s0 - lo:s1 - lo] = xb[s0 - b * B:s1 - b * B]
        y[s0 - lo:s1 - lo] = yb[s0 - b * B:s1 - b * B]
    return x, y


# Loss (relative residual) of the preconditioned CG solve after `steps` iterations, measured on MI355X with ONE rank:
# (rows, dim, rffs, rank, steps) -> value.  The dataset does not depend on the number of ranks (make_shard), the
# preconditioner and the iterates only through the summation order of the all-reduced sums (and the eigensolver's last
# digits), so the same value holds at every N: rtol 1e-6 on one rank, 1e-5 when sums are exchanged.
EXPECTED_FINAL_LOSS = {(1_000_000, 1024, 8192, 512, 20): 0.0135079259394}      # 2 / 4 gloo ranks on one device: ...9398, ...9397


def final_loss_check(key, loss, world_size, table=None):
    """The timed iterations are a real solve: compare its final loss with the stored one.  Returns the object the JSON
    line carries ({"expected", "rtol", "ok"}; ok is None when no value is stored for this configuration)."""
    table = EXPECTED_FINAL_LOSS if table is None else table
    if key not in table:
        return {"expected": None, "rtol": None, "ok": None}
    rtol = 1e-6 if world_size == 1 else 1e-5
    ok = bool(abs(loss / table[key] - 1.0) <= rtol)          # False for NaN
    return {"expected": table[key], "rtol": rtol, "ok": ok}


def cpu_baseline(args, budget_s):
    """The reference CPU path on a bounded sample of the same workload: per 8192-row chunk, feature generation
    followed by ``Z.T @ (Z @ p)`` in numpy (the host BLAS), as fitting_toolkit/cg_tools.py:189-191 does.  Feature
    generation runs through the REFERENCE's own compiled arithmetic core when oracle/_ref/libxgpr_ref.so is
    present (kind "reference": hadamard_transforms.cpp / shared_rfgen_ops.cpp under the row loop and OpenMP team
    of rbf_ops.cpp:73-100), otherwise through the oracle's C restatement (kind "port"); both are bit-identical."""
    import numpy as np
    from oracle import oracle as orc
    kind = "reference" if orc.RefCore.available() else "port"
    orc.build(ref=False)
    threads = orc.Oracle().num_threads()               # the OpenMP default team size both libraries run with
    ops = orc.RefCore() if kind == "reference" else orc.Oracle()
    rng = np.random.default_rng(5)
    d, m = args.dim, args.rffs
    radem, chi = orc.draw_sorf_params(m, d, 123)
    orc.matern_rescale(chi, 2.5, 123)
    p = rng.standard_normal(m)
    chunk = 8192
    x = (rng.standard_normal((chunk, d)) / np.sqrt(d)).astype(np.float32)
    z = np.zeros((chunk, m))
    ops.cpuRBFFeatureGen(x, z, radem, chi, True)       # warm-up (page in, thread pool)
    t_feat = t_mv = 0.0
    done = 0
    while (t_feat + t_mv) < budget_s and done < args.rows:
        z[:] = 0
        t0 = time.perf_counter()
        ops.cpuRBFFeatureGen(x, z, radem, chi, True)
        z[:, 0] = 1.0
        t1 = time.perf_counter()
        w = z.T @ (z @ p)
        t2 = time.perf_counter()
        t_feat += t1 - t0
        t_mv += t2 - t1
        done += chunk
    return {
        "value": done * m / (t_feat + t_mv), "unit": "random-features/s",
        "cores": threads, "kind": kind,
        "sample": f"{done} rows of the same workload (d={d}, M={m}) in {chunk}-row chunks: "
                  f"feature-gen {t_feat:.2f} s + numpy Z^T(Zp) {t_mv:.2f} s; scaled linearly in rows",
        "featgen_only_features_per_s": done * m / t_feat,
        "cg_iters_per_sec_extrapolated": 1.0 / ((t_feat + t_mv) * args.rows / done),
        "host_cpus": os.cpu_count(),
    }


def conv_featgen_probe(device, nseq=8192):
    """The convolution feature operator (cudaConv1dFGen's drop-in) at BASELINE configs[3]'s shape: one-hot protein-like
    sequences, L <= 512, 21 channels, conv_width 9, 16384 RFFs, 'sqrt' averaging; 8192 sequences per call (the window the
    feature cache is built in).  The kernel is vector-pipe bound: `priced` is count x measured issue cost of its k-mer
    loop (profiles/r6_conv_inst_table.json, tools/count_loop_insts.py conv 8) over the measured time."""
    import numpy as np
    import torch
    from xgpr_amd.kernels import make_kernel
    L, C, m, w = 512, 21, 16384, 9
    g = torch.Generator(device=device).manual_seed(3)
    x = torch.nn.functional.one_hot(torch.randint(0, C, (nseq, L), device=device, generator=g), C).to(torch.float32)
    sl = torch.randint(64, L + 1, (nseq,), generator=torch.Generator().manual_seed(5)).numpy().astype(np.int32)
    kern = make_kernel("Conv1dRBF", (nseq, L, C), m, 123, device, {"conv_width": w, "averaging": "sqrt"})
    kern.set_hyperparams(np.array([1.0, 0.8]), logspace=False)
    kern.transform_x(x, sl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        z = kern.transform_x(x, sl)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del z
    kmers = int((sl.astype(np.int64) - w + 1).sum())
    tiles = kmers * (m // 2 // 1024)                     # wave tiles (1024 frequencies of one k-mer)
    out = {"workload": "BASELINE configs[3] shape: Conv1dRBF, %d sequences (L 64..512, 21 channels, conv_width 9), %d RFFs" % (nseq, m),
           "ms": ms, "sequences_per_s": nseq / (ms * 1e-3), "kmers": kmers, "tile_transforms_per_s": tiles / (ms * 1e-3),
           "note": "whole operator call (transform_x: scaling of the input copy, ordering kernel, feature kernel, float64 output)"}
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_conv_inst_table.json")))
        pipe_ms = tab["priced_vector_ns_per_tile_per_simd"] * tiles / 1024 * 1e-6
        out["vector_pipe"] = {"valu_insts_per_kmer_tile": tab["valu_instructions"],
                              "priced": {"pipe_ms": pipe_ms, "frac": pipe_ms / ms,
                                         "source": "profiles/%s_conv_inst_table.json x profiles/r3_valu_cost.json (stored)" % PROFILE_ROUND}}
    except (OSError, KeyError, ValueError):
        pass
    return out


def config_share(which, device, rows=None):
    """One of the BASELINE configs that are not the headline, at the per-GPU share of its 8-GPU size (cfg2: whole), on
    this GPU: randomized-Nystrom preconditioner build + CG fit to 1e-6 through the product's own entry points
    (RandNysPreconditioner, cg_fit_lib_internal -- the calls xGPRegression.fit(mode="cg") makes).  Synthetic inputs of
    the shape SURVEY 8(d) names; y ~ N(0,1).  Returns the timing object of the JSON line's ``configs`` entry."""
    import numpy as np
    import torch
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import cg_fit_lib_internal
    g = torch.Generator(device=device).manual_seed(123)
    sl = None
    if which == "cfg2":      # RBF, N=1e5 d=256, 4096 RFFs, single GPU
        n = rows or 100_000
        d, m, rank, method, chunk = 256, 4096, 512, "srht", 8192
        x = torch.randn(n, d, device=device, generator=g) / d ** 0.5
        kern = make_kernel("RBF", (n, d), m, 123, device, {})
        what = "RBF, N=%d (whole), d=256, 4096 RFFs" % n
    elif which == "cfg4":    # Conv1d RBF, L<=512, 21 channels one-hot, 16384 RFFs; N=5e5 over 8 GPUs -> 62500 per GPU
        n = rows or 62_500
        L, C, m, rank, method, chunk = 512, 21, 16384, 512, "srht", 1024
        x = torch.nn.functional.one_hot(torch.randint(0, C, (n, L), device=device, generator=g), C).to(torch.float32)
        sl = torch.randint(64, L + 1, (n,), generator=torch.Generator().manual_seed(5)).numpy().astype(np.int32)
        kern = make_kernel("Conv1dRBF", (n, L, C), m, 123, device, {"conv_width": 9, "averaging": "sqrt"})
        what = "Conv1dRBF, %d sequences (1/8 of 5e5; L 64..512, 21 channels, conv_width 9), 16384 RFFs" % n
    elif which == "cfg5":    # RBF, d=512, 32768 RFFs, rank-2048 srht_2; N=2e6 over 8 GPUs -> 250000 per GPU
        n = rows or 250_000
        d, m, rank, method, chunk = 512, 32768, 2048, "srht_2", 8192
        x = torch.randn(n, d, device=device, generator=g) / d ** 0.5
        kern = make_kernel("RBF", (n, d), m, 123, device, {})
        what = "RBF, %d rows (1/8 of 2e6), d=512, 32768 RFFs" % n
    else:
        raise ValueError(which)
    y = torch.randn(n, dtype=torch.float64, device=device, generator=g)
    ds = build_regression_dataset(x, y, sl, chunk_size=chunk, device=device)
    kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0
    # small untimed build + fits first: library initialisation and first launches are not part of a build
    nw = min(n, 4096)
    ds_w = build_regression_dataset(x[:nw], y[:nw], None if sl is None else sl[:nw], chunk_size=chunk, device=device)
    pre_w = RandNysPreconditioner(kern, ds_w, min(rank, 256), False, 123, method)
    modes = (True,) if which == "cfg4" else (False, True)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for cache in modes:
            cg_fit_lib_internal(kern, ds_w, 1e-6, 5, pre_w, False, cache_features=cache)
    del ds_w, pre_w
    pre, t_pre = timed(lambda: RandNysPreconditioner(kern, ds, rank, False, 123, method))
    out = {"workload": what, "rank": rank, "method": method, "precond_build_s": t_pre,
           "achieved_ratio": float(pre.achieved_ratio), "fits": []}
    for cache in modes:
        t_cache = 0.0
        if cache:
            torch.cuda.empty_cache()          # the build's scratch goes back to the driver before the big allocation
            _, t_cache = timed(lambda: ds.feature_cache(kern))
        (w, niter, losses), t_fit = timed(lambda: cg_fit_lib_internal(kern, ds, 1e-6, 200, pre, False, cache_features=cache))
        out["fits"].append({"features": "resident float32 cache" if cache else "regenerated every iteration",
                            "iterations": int(niter), "seconds": t_fit, "ms_per_iteration": 1e3 * t_fit / niter,
                            "feature_cache_s": t_cache if cache else None, "final_err": losses[-1]})
    ds._zcache = None
    del ds, pre, x, y, kern
    torch.cuda.empty_cache()
    return out


def nmll_probe(device, rows=262144):
    """The approximate NMLL (one preconditioned CG solve with k = 26 right-hand sides + stochastic Lanczos
    quadrature, xgp_regression.py:264-367) at cfg3's shape on a 262 144-row shard with resident float32 features: what an
    iteration of the batched solve costs beside its block matvec."""
    import numpy as np
    import torch
    from xgpr_amd.kernels import make_kernel, block_workspace_bytes
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.nmll import approximate_nmll
    d, m, k = 1024, 8192, 26
    g = torch.Generator(device=device).manual_seed(123)
    x = torch.randn(rows, d, device=device, generator=g) / d ** 0.5
    a = torch.randn(d, device=device, generator=g)
    y = (torch.sin(x @ a) + 0.1 * torch.randn(rows, device=device, generator=g)).double()
    ds = build_regression_dataset(x, y, chunk_size=16384, device=device)
    kern = make_kernel("Matern", (rows, d), m, 123, device, {"matern_nu": 2.5})
    kern.set_hyperparams(np.array([0.3, 1.0]), logspace=False)
    pre = RandNysPreconditioner(kern, ds, 512, False, 123, "srht")
    zc = ds.feature_cache(kern)
    det = {}
    approximate_nmll(kern, ds, pre, None, 123, True, det)                  # first launches, untimed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    val = approximate_nmll(kern, ds, pre, None, 123, True, det)
    torch.cuda.synchronize()
    t_nmll = time.perf_counter() - t0
    vb = torch.randn((m, k), dtype=torch.float64, device=device, generator=g)
    wb = torch.empty_like(vb)
    bws = torch.empty(block_workspace_bytes(rows, m, k), dtype=torch.uint8, device=device)
    kern.ztz_block_cached(zc, vb, wb, bws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        kern.ztz_block_cached(zc, vb, wb, bws)
    e1.record()
    e1.synchronize()
    mv_ms = e0.elapsed_time(e1) / 5
    out = {"workload": "approximate NMLL, Matern-5/2, %d rows x d=1024, 8192 RFFs, 26 right-hand sides, rank-512 preconditioner, "
                       "resident float32 features" % rows,
           "iterations": int(det["niter"]), "seconds": t_nmll, "ms_per_iteration": 1e3 * t_nmll / det["niter"],
           "block_matvec_ms": mv_ms, "nmll": val}
    ds._zcache = None
    del zc, ds, pre, x, y
    torch.cuda.empty_cache()
    return out


def wide_probe(device, rows=131072):
    """Padded input widths beyond 1024 (round 6: transforms of two / four wave tiles on the three-wave kernel, one
    cross-wave exchange per round) at the reference's own test shapes (tests/fht_operations_tests/test_rbf_rfgen.py:37,41:
    d = 2003 / M = 4000, d = 1076 / M = 8192) and at d = 4000 / M = 8192: the float64 feature operator, one CG matvec the
    way ``ConjugateGrad._matvec`` routes it (fused, features regenerated), the same over the resident float32 cache, z^T y and
    a rank-512 preconditioner build -- beside what the any-width LDS path + materialised float64 Z took on the round-5 tree
    (profiles/r6_generic_path_before.json)."""
    import numpy as np
    import torch
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import ConjugateGrad, calc_zty
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    before = {}
    try:
        for r in json.load(open(os.path.join(ROOT, "profiles", "r6_generic_path_before.json")))["shapes"]:
            before[(r["d"], r["num_rffs"])] = r
    except (OSError, KeyError, ValueError):
        pass

    def ev_ms(fn, reps=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    out = {"rows": rows, "shapes": []}
    g = torch.Generator(device=device).manual_seed(7)
    for d, m in ((2003, 4000), (1076, 8192), (4000, 8192)):
        x = torch.randn(rows, d, device=device, generator=g) / d ** 0.5
        y = torch.randn(rows, dtype=torch.float64, device=device, generator=g)
        kern = make_kernel("RBF", (rows, d), m, 123, device, {})
        kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
        ds = build_regression_dataset(x, y, chunk_size=16384, device=device)
        xs = ds.scaled_x(kern.hyperparams[1])
        tiles = rows * ((m // 2 + 1023) // 1024)
        z = torch.empty(32768, m, dtype=torch.float64, device=device)

        def featgen():
            for lo in range(0, rows, 32768):
                ext.hipRBFFeatureGen(xs[lo:lo + 32768], z[:min(32768, rows - lo)], kern.radem_diag, kern.chi_arr, True)
        fg = ev_ms(featgen)
        # the float64 overload (double_precision=True kernels) and the gradient operator (features + d/dsigma), 32768 rows: wave tiles
        # too since round 6 (wave_tile.inc)
        xd, chid = xs[:32768].double(), kern.chi_arr.double()
        f64 = ev_ms(lambda: ext.hipRBFFeatureGen(xd, z, kern.radem_diag, chid, True), reps=3)
        grad = torch.empty(32768, m, 1, dtype=torch.float64, device=device)
        gr32 = ev_ms(lambda: ext.hipRBFGrad(xs[:32768], z, grad, kern.radem_diag, kern.chi_arr, 1.0, True), reps=3)
        del z, grad, xd
        vec = torch.randn(m, 1, dtype=torch.float64, device=device, generator=g)
        w = torch.zeros_like(vec)
        mv = ev_ms(lambda: ConjugateGrad(cache_features=False)._matvec(ds, kern, vec, w))
        cgc = ConjugateGrad(cache_features=True)
        wv = torch.zeros(m, dtype=torch.float64, device=device)
        t0 = time.perf_counter()
        ds.feature_cache(kern)
        torch.cuda.synchronize()
        t_cache = time.perf_counter() - t0
        mvc = ev_ms(lambda: cgc._ztz(ds, kern, vec[:, 0].contiguous(), wv))
        ds._zcache = None
        ds._zcache_key = None
        zty = ev_ms(lambda: calc_zty(ds, kern), reps=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pre = RandNysPreconditioner(kern, ds, 512, False, 123, "srht")
        torch.cuda.synchronize()
        t_pre = time.perf_counter() - t0
        b = before.get((d, m), {})
        out["shapes"].append({
            "d": d, "num_rffs": m, "padded_width": 1 << int(np.ceil(np.log2(d))), "wave_tiles_per_transform": (1 << int(np.ceil(np.log2(d)))) // 1024,
            "matvec_plan": ext.ztz_matvec_plan(d, m // 2), "cache_features_auto": bool(kern.cache_pays()),
            "featgen_f64_ms": fg, "featgen_GBs": (4.0 * d + 8.0 * m) * rows / (fg * 1e-3) / 1e9,
            "float64_input_op_ms_32768_rows": f64, "gradient_op_ms_32768_rows": gr32,
            "cg_matvec_regenerating_ms": mv, "cg_matvec_ns_per_tile": mv * 1e6 / tiles,
            "cg_matvec_cached_ms": mvc, "feature_cache_build_s": t_cache, "zty_ms": zty,
            "precond_build_rank512_s": t_pre, "achieved_ratio": float(pre.achieved_ratio),
            "round5_tree": {"featgen_f64_ms": b.get("featgen_f32_ms"), "float64_input_op_ms_32768_rows": b.get("featgen_f64_ms"),
                            "cg_matvec_ms": b.get("cg_matvec_ms"), "zty_ms": b.get("zty_ms"),
                            "precond_build_rank512_s": None if "precond_build_rank512_ms" not in b else b["precond_build_rank512_ms"] * 1e-3,
                            "source": "profiles/r6_generic_path_before.json (generic_sorf_kernel + float64 Z + library GEMV; stored)"}})
        del pre, ds, kern, x, y, xs
        torch.cuda.empty_cache()
    out["note"] = ("fit() at d > 1024 runs the same fused / cached / matrix-core paths as at d <= 1024 since round 6; per 1024-frequency "
                   "tile the regenerating matvec takes ~1.2x (padded width 2048) / ~1.5x (4096) the padded-width-1024 time, so "
                   "cache_features='auto' keeps Z resident there")
    return out


def notebook_tabular(device):
    """The one published number that touches this path (/root/reference/docs/notebooks/
    tabular_example.ipynb:547-560): ``fit(mode="cg", tol=1e-6)`` of an RBF model with 8192 RFFs on the UCI CASP training split
    -- 36 584 rows x 9 standardised features, chunk_size 2000, preconditioner autoselected (the ratio check on a sample, then
    a rank-512 srht build), CG, variance with variance_rffs = 512 -- 3.18 s wall on an unnamed NVIDIA GPU with xGPR 0.4.8
    (35 CG iterations on the real data).  Here: synthetic rows of that shape (no network for the dataset), the notebook's
    hyperparameters, the same call through the drop-in harness (xgpr_amd.models.xGPRegression.fit).  Context, not the target."""
    import numpy as np
    import torch
    from xgpr_amd.models import xGPRegression
    from xgpr_amd.dataset import build_regression_dataset
    n, d = 36584, 9
    rng = np.random.default_rng(123)
    x = rng.standard_normal((n, d))
    a = rng.standard_normal(d)
    y = 6.0 * np.sin(x @ a / np.sqrt(d)) + 0.5 * (x[:, 0] * x[:, 1]) + 4.0 * rng.standard_normal(n) + 7.7
    ds = build_regression_dataset(x, y, chunk_size=2000, device=device)
    hp = np.array([-0.5406061, 0.2469573])          # tabular_example.ipynb cell 14 (log lambda, log sigma)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        warm = xGPRegression(num_rffs=8192, variance_rffs=512, kernel_choice="RBF", verbose=False, device=device, random_seed=123)
        dsw = build_regression_dataset(x[:4000], y[:4000], chunk_size=2000, device=device)
        warm.set_hyperparams(hp, dsw)
        warm.fit(dsw, mode="cg", tol=1e-6)          # first launches / library initialisation: untimed
        del warm, dsw
        model = xGPRegression(num_rffs=8192, variance_rffs=512, kernel_choice="RBF", verbose=False, device=device, random_seed=123)
        model.set_hyperparams(hp, ds)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_iter, losses = model.fit(ds, mode="cg", tol=1e-6, run_diagnostics=True)
        torch.cuda.synchronize()
        t_fit = time.perf_counter() - t0
    return {"workload": "xGPRegression.fit(mode='cg', tol=1e-6): RBF, 36584 x 9 synthetic standardised rows (the CASP training split's shape), "
                        "8192 RFFs, autoselected rank-512 srht preconditioner, variance_rffs 512, chunk_size 2000",
            "seconds": t_fit, "cg_iterations": int(n_iter), "final_err": float(losses[-1]),
            "reference": {"seconds": 3.184, "cg_iterations": 35, "hardware": "unnamed NVIDIA GPU, xGPR 0.4.8, real CASP data read from .npy chunks on disk",
                          "source": "/root/reference/docs/notebooks/tabular_example.ipynb:547-560"},
            "note": "context only: different hardware, synthetic data of the same shape (iteration counts differ), data resident on the device here"}


def valu_only_probe(kern, ds, kern_ms, device):
    """Times the fused matvec of this rank's shard through the VALU-only timing probe (tools/, xgpr_amd/build.py PROBE_LIB: the
    kernel's vector instruction stream with LDS traffic, barrier and prefetch compiled out; results meaningless, never
    used).  Returns {"ms", "frac"} or None when the probe library has not been built."""
    import ctypes as C
    import torch
    path = os.path.join(ROOT, "tools", "libxgpr_hip_valuonly_probe.so")
    if not os.path.exists(path):
        return None
    fn = C.CDLL(path).xgpr_ztz_matvec_f32
    vp, l, i, sz = C.c_void_p, C.c_long, C.c_int, C.c_size_t
    fn.argtypes = [vp, vp, vp, vp, vp, l, l, l, l, l, i, vp, sz, vp]
    fn.restype = C.c_int
    xs = ds.scaled_x(1.0)
    mm = kern.num_rffs
    v = torch.randn(mm, dtype=torch.float64, device=device)
    w = torch.empty_like(v)
    ws = torch.empty(kern.workspace_bytes(), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream().cuda_stream

    def call():
        rc = fn(xs.data_ptr(), kern.radem_diag.data_ptr(), kern.chi_arr.data_ptr(), v.data_ptr(), w.data_ptr(), xs.shape[0],
                xs.shape[1], mm, kern.num_freqs, kern.radem_diag.shape[2], int(kern.fit_intercept), ws.data_ptr(),
                ws.numel(), stream)
        if rc != 0:
            raise RuntimeError("VALU-only probe launch failed")
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return {"ms": ms, "frac": ms / kern_ms, "source": "live: libxgpr_hip_valuonly_probe.so (timing-only build, -DXGPR_ABL_VALUONLY)"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    if args.dist_check:
        # loaded by path: the package itself needs the built HIP library, this leg needs only the communicator
        import importlib.util
        spec = importlib.util.spec_from_file_location("xgpr_amd_dist", os.path.join(ROOT, "xgpr_amd", "dist.py"))
        xd = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(xd)
        on_gpu = torch.cuda.is_available()
        comm = xd.init_from_env(device_type="cuda" if on_gpu else "cpu")
        device = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        if comm.world_size != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={comm.world_size}")
        info = dist_summary(comm, device, args.rffs)
        lo, hi = comm.shard_bounds(args.rows)
        info["shard_rows_per_rank"] = [r[0] for r in gather_per_rank(comm, device, [float(hi - lo)])]
        if comm.rank == 0:
            print(json.dumps({"dist_check": True, "n_gpus": args.gpus, **info}))
        if comm.through_backend:
            torch.distributed.destroy_process_group()
        return
    from xgpr_amd import dist as xd
    comm = xd.init_from_env(device_type="cuda")
    if comm.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={comm.world_size}")
    device = torch.device("cuda", torch.cuda.current_device())
    dist_info = dist_summary(comm, device, args.rffs)
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import DeviceDataset, build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import ConjugateGrad, calc_zty
    from xgpr_amd import xgpr_hip_rfgen_ext as ext

    n, d, m = args.rows, args.dim, args.rffs
    lo, hi = comm.shard_bounds(n)
    x, y = make_shard(lo, hi, d, device)
    ds = build_regression_dataset(x, y, chunk_size=16384, device=device, comm=comm, already_sharded=True)
    kern = make_kernel("Matern", (n, d), m, 123, device, {"matern_nu": 5 / 2})
    kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)

    # rank-512 SRHT preconditioner from ALL rows (one pass: HIP feature-gen + HIP SRHT per 8192-row
    # chunk, float64 MFMA GEMM for acc += SRHT(Z)^T Z; ~0.3 s at N = 1e6) -- outside the timed region
    ds_pre = DeviceDataset(x, y, None, 8192, ds.get_ymean(), ds.get_ystd(), n, device, comm)
    # one small untimed build first: the first GEMM / eigensolver / Cholesky calls of a process pay ~0.4 s of
    # library initialisation that is not part of a build
    warm_rows = min(hi - lo, 16384)
    RandNysPreconditioner(kern, DeviceDataset(x[:warm_rows], y[:warm_rows], None, 8192, ds.get_ymean(), ds.get_ystd(),
                                               warm_rows * comm.world_size, device, comm),
                          args.rank_precond, False, 123, "srht")
    comm.barrier()
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    pre = RandNysPreconditioner(kern, ds_pre, args.rank_precond, False, 123, "srht")
    torch.cuda.synchronize()
    precond_first_s = time.perf_counter() - tp0
    # built twice: the first full-size build also pays for the allocator's first 4 GiB window / 0.5 GiB sketch blocks
    # (hipMalloc: 0 ... 0.3 s depending on the box); the second is the build itself.  Both are reported.
    del pre
    comm.barrier()
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    pre = RandNysPreconditioner(kern, ds_pre, args.rank_precond, False, 123, "srht")
    torch.cuda.synchronize()
    precond_build_s = time.perf_counter() - tp0
    zty, _ = calc_zty(ds, kern)

    cg = ConjugateGrad(comm)
    timings = []

    # wrap the fused matvec with HIP events on the launch stream (torch's current stream)
    orig = kern.ztz_matvec

    def timed_matvec(xs, vec, out, ws=None, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(xs, vec, out, ws, **kw)
        e1.record()
        timings.append((e0, e1))
    kern.ztz_matvec = timed_matvec

    # the exchange step of the iteration: torch.distributed's all-reduce of w.  ProcessGroupNCCL runs RCCL on its
    # own stream and chains it to the compute stream with events on both sides; the pair of events recorded here on
    # the compute stream brackets exactly that hand-off + the collective, so the time is what an iteration pays.
    ar_timings = []
    orig_ar = comm.all_reduce_

    def timed_all_reduce(tensor):
        if not comm.through_backend:
            return tensor
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_ar(tensor)
        e1.record()
        ar_timings.append((e0, e1))
        return tensor
    comm.all_reduce_ = timed_all_reduce

    def run(iters):
        resid = torch.zeros((m, 2, 1), dtype=torch.float64, device=device)
        resid[:, 0, 0] = zty / n
        return cg.fit(ds, kern, pre, resid, maxiter=iters, tol=0.0, verbose=False)

    if args.warmup > 0:
        run(args.warmup)
    timings.clear()
    ar_timings.clear()
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, _, niter, losses = run(args.steps)
    torch.cuda.synchronize()
    comm.barrier()
    t1 = time.perf_counter()
    assert niter == args.steps
    # the timed iterations are a real solve: its final loss against the stored value (reported in the line; a mismatch
    # makes the run exit non-zero AFTER the line is printed)
    loss_check = final_loss_check((args.rows, args.dim, args.rffs, args.rank_precond, args.steps), losses[-1], comm.world_size)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=device)
    if comm.through_backend:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    t = float(elapsed.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in timings]))
    ar_ms = float(np.mean([a.elapsed_time(b) for a, b in ar_timings])) if ar_timings else 0.0
    kern.ztz_matvec = orig
    comm.all_reduce_ = orig_ar
    per_rank = gather_per_rank(comm, device, [1e3 * (t1 - t0) / args.steps, kern_ms, ar_ms, float(hi - lo)])

    # time to tolerance of the north-star entry point's solve (xGPRegression.fit(mode="cg"): preconditioner build, then
    # cg_fit_lib_internal's solve to 1e-6 -- xgp_regression.py's fit path, cg_fitting_toolkit.py:18-70); every rank runs it
    comm.barrier()
    torch.cuda.synchronize()
    tf0 = time.perf_counter()
    resid = torch.zeros((m, 2, 1), dtype=torch.float64, device=device)
    resid[:, 0, 0] = pre.get_zty() / n
    _, tol_converged, tol_iters, tol_losses = cg.fit(ds, kern, pre, resid, maxiter=500, tol=1e-6, verbose=False)
    torch.cuda.synchronize()
    comm.barrier()
    tol_s = torch.tensor([time.perf_counter() - tf0], dtype=torch.float64, device=device)
    if comm.through_backend:
        torch.distributed.all_reduce(tol_s, op=torch.distributed.ReduceOp.MAX)
    fit_to_tol = {"tol": 1e-6, "converged": bool(tol_converged), "iterations": int(tol_iters), "cg_seconds": float(tol_s.item()),
                  "precond_build_seconds": precond_build_s, "seconds": float(tol_s.item()) + precond_build_s,
                  "final_err": tol_losses[-1], "note": "build (all rows, rank %d srht) + preconditioned CG to 1e-6, features "
                  "regenerated on every iteration; wall clock, max over ranks" % args.rank_precond}
    # the same solve the way the product's cg_fit_lib_internal runs it by default (cache_features="auto": the shard's
    # float32 feature matrix resident in HBM when it fits, generated once and streamed) -- reported beside, never the headline
    from xgpr_amd.cg import cg_fit_lib_internal
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cg_fit_lib_internal(kern, ds, 1e-6, 3, pre, False, cache_features="auto")      # first launches / first allocation, untimed
    ds._zcache = None                    # the cache is generated again inside the timed region (the allocator keeps the block)
    ds._zcache_key = None
    comm.barrier()
    torch.cuda.synchronize()
    ta0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, auto_iters, auto_losses = cg_fit_lib_internal(kern, ds, 1e-6, 500, pre, False, cache_features="auto")
    torch.cuda.synchronize()
    comm.barrier()
    auto_s = torch.tensor([time.perf_counter() - ta0], dtype=torch.float64, device=device)
    if comm.through_backend:
        torch.distributed.all_reduce(auto_s, op=torch.distributed.ReduceOp.MAX)
    fit_to_tol["product_default"] = {"iterations": int(auto_iters), "cg_seconds": float(auto_s.item()),
                                     "seconds": float(auto_s.item()) + precond_build_s, "final_err": auto_losses[-1],
                                     "note": "cg_fit_lib_internal(cache_features='auto'): resident float32 features when they fit; "
                                             "the feature-generation pass that fills the cache is inside cg_seconds, its first "
                                             "hipMalloc (0.5-0.9 s of page scrubbing, once per process) is not"}
    ds._zcache = None
    ds._zcache_key = None
    torch.cuda.empty_cache()

    # optional mode, reported beside the headline and never part of it: the shard's feature matrix kept
    # resident in HBM as float32 (32 KB per datapoint) and streamed on every CG iteration instead of
    # regenerated -- HBM-bound instead of VALU-bound.  Features are generated once, outside these steps.
    cached = None
    if kern.cache_ok() and ds.feature_cache_bytes(kern) < 0.6 * torch.cuda.get_device_properties(device).total_memory:
        cgc = ConjugateGrad(comm, cache_features=True)
        # the allocation is timed apart from the generation pass: a fresh 32.8 GB hipMalloc costs 0.5-0.9 s of
        # page scrubbing in the driver (once per process; the caching allocator hands the block back instantly
        # afterwards), the generation pass itself milliseconds
        torch.cuda.synchronize()
        talloc0 = time.perf_counter()
        probe = torch.empty((hi - lo, m), dtype=torch.float32, device=device)
        torch.cuda.synchronize()
        cache_alloc_s = time.perf_counter() - talloc0
        del probe
        tcache0 = time.perf_counter()
        zc = ds.feature_cache(kern)
        torch.cuda.synchronize()
        cache_build_s = time.perf_counter() - tcache0
        ctimes = []
        origc = kern.ztz_matvec_cached

        def timed_cached(zcache, vec, out, ws):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            origc(zcache, vec, out, ws)
            e1.record()
            ctimes.append((e0, e1))
        kern.ztz_matvec_cached = timed_cached

        def run_cached(iters):
            resid = torch.zeros((m, 2, 1), dtype=torch.float64, device=device)
            resid[:, 0, 0] = zty / n
            return cgc.fit(ds, kern, pre, resid, maxiter=iters, tol=0.0, verbose=False)
        run_cached(max(1, args.warmup))
        ctimes.clear()
        comm.barrier()
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        run_cached(args.steps)
        torch.cuda.synchronize()
        comm.barrier()
        tc = torch.tensor([time.perf_counter() - tc0], dtype=torch.float64, device=device)
        if comm.through_backend:
            torch.distributed.all_reduce(tc, op=torch.distributed.ReduceOp.MAX)
        ck_ms = float(np.mean([a.elapsed_time(b) for a, b in ctimes]))
        kern.ztz_matvec_cached = origc
        cbytes = 4.0 * m * (hi - lo)
        # PMC-measured HBM bytes / algorithmic bytes of this streaming kernel (stored; quoted while the library is the measured build)
        from xgpr_amd import _lib as xlib0
        cpm, ctraffic_src = stored_profile("pmc_traffic_cached.json", xlib0.build_id())
        ctraffic = cbytes * cpm["traffic_over_algorithmic"] if cpm and "traffic_over_algorithmic" in cpm else None
        cached = {"ms_per_step": 1e3 * float(tc.item()) / args.steps, "cg_iters_per_sec": args.steps / float(tc.item()),
                  "cache_bytes_per_gpu": cbytes, "cache_build_s": cache_build_s, "cache_alloc_s": cache_alloc_s,
                  "roofline": {"kernel": "zcache_ztz_kernel<true, 2> (+ reduce_slabs)", "bound": "hbm",
                               "achieved": cbytes / (ck_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": cbytes / (ck_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": ctraffic, "traffic_source": ctraffic_src,
                               "algorithmic_bytes": cbytes, "kernel_ms": ck_ms},
                  "note": "features generated once (cache_build_s) and streamed; NOT the headline number"}
        # the block of right-hand sides of the approximate NMLL (k = 26) over the same resident cache: the
        # two contractions of Z^T (Z V) on the float64 matrix cores (not part of the headline either)
        if kern.block_ok():
            from xgpr_amd.kernels import block_workspace_bytes
            kb = 26
            vb = torch.randn((m, kb), dtype=torch.float64, device=device)
            wb = torch.empty_like(vb)
            bws = torch.empty(block_workspace_bytes(hi - lo, m, kb), dtype=torch.uint8, device=device)
            for _ in range(2):
                kern.ztz_block_cached(zc, vb, wb, bws)
            be0, be1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            breps = 3
            be0.record()
            for _ in range(breps):
                kern.ztz_block_cached(zc, vb, wb, bws)
            be1.record()
            torch.cuda.synchronize()
            b_ms = be0.elapsed_time(be1) / breps
            bflop = 4.0 * (hi - lo) * m * kb
            cached["block_matvec_k26"] = {
                "ms_per_matvec": b_ms,
                "roofline": {"kernel": "zblock_t_kernel<1,4,3> + zblock_w_kernel<1,3> (v_mfma_f64_16x16x4_f64 for columns 0-15, "
                                       "v_mfma_f64_4x4x4_4b_f64 for columns 16-27)", "bound": "mfma",
                             "achieved": bflop / (b_ms * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": bflop / (b_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             "issued": bflop * 28 / kb / (b_ms * 1e-3) / 1e12, "traffic": None,
                             "mfma_busy": mfma_busy_of(("zblock_t_kernel", "zblock_w_kernel")),
                             "algorithmic_flops": bflop},
                "note": "26 right-hand sides run as 16 + 3 x 4 columns: 'issued' counts the 28 columns of MFMA work; "
                        "DESIGN.md section 3 (block of right-hand sides)"}
            del vb, wb, bws
        del zc
        ds._zcache = None
        ds._zcache_key = None

    # stand-alone feature-generation operator (Z materialised, float64), this rank's device
    fg_rows = min(131072, hi - lo)      # 9.1 GB per call: a ~1.7 ms window, not a launch-noise-sized one
    xs = ds.scaled_x(1.0)[:fg_rows]
    zbuf = torch.empty((fg_rows, m), dtype=torch.float64, device=device)
    for _ in range(2):
        ext.hipRBFFeatureGen(xs, zbuf, kern.radem_diag, kern.chi_arr, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ext.hipRBFFeatureGen(xs, zbuf, kern.radem_diag, kern.chi_arr, True)
    e1.record()
    torch.cuda.synchronize()
    fg_ms = e0.elapsed_time(e1) / reps
    del zbuf

    conv = conv_featgen_probe(device) if (comm.rank == 0 and args.gpus == 1) else None

    valu_probe = None
    if comm.rank == 0 and (d, m) == (1024, 8192):
        valu_probe = valu_only_probe(kern, ds, kern_ms, device)

    # the other BASELINE configs at their per-GPU shares (N = 1 only; collective-free: they run on comm SINGLE)
    configs = None
    if comm.rank == 0 and args.gpus == 1 and not args.no_configs and n == 1_000_000:
        del pre, ds_pre, ds, x, y, cg
        torch.cuda.empty_cache()
        configs = {c: config_share(c, device) for c in ("cfg2", "cfg4", "cfg5")}
        configs["nmll_k26"] = nmll_probe(device)
        configs["wide"] = wide_probe(device)
        configs["notebook_tabular"] = notebook_tabular(device)

    if comm.rank == 0:
        n_local = hi - lo
        # HBM bytes per launch of the dominant kernel from the PMC counters: a STORED measurement (separate
        # rocprofv3 --pmc passes of this same command, corrected as MI355X_MICROARCH.md prescribes;
        # profiles/r2_pmc_traffic.json), not a quantity of this run -- quoted only for the configuration it was
        # measured on, and labelled as such in the line
        from xgpr_amd import _lib as xlib
        build_id = xlib.build_id()     # sha256 of csrc/*, include/xgpr_hip.h and the compiler flags the LOADED library was built from
        traffic = None
        pm, traffic_src = stored_profile("pmc_traffic.json", build_id)
        try:
            if pm is not None:
                c = pm["config"]
                if (c["rows_per_gpu"], c["dim"], c["rffs"]) == (n_local, d, m):
                    traffic = pm["hbm_bytes_per_launch"]
                else:
                    traffic_src = "stored for another configuration: not quoted"
        except (KeyError, ValueError):
            pass
        # the resource this kernel is actually bound by: the vector pipe.  Three readings, each labelled:
        #  issue_slots -- vector instructions per wave tile (static count of the loop's hot path from the disassembly,
        #                 tools/count_loop_insts.py -> profiles/r6_ztz3_inst_table.json) x tiles / live kernel time, against
        #                 one wave-instruction per TWO cycles per SIMD at 2.4 GHz (the rate of v_add_f32; most of this
        #                 kernel's instructions -- packed, DPP, float64, conversions -- occupy the pipe for four cycles,
        #                 cos/sin for eight, so this reading cannot reach 1)
        #  priced      -- sum over instruction classes of count x measured issue cost at three waves per SIMD
        #                 (tools/valu_cost.hip -> profiles/r3_valu_cost.json), i.e. the time the vector pipe is occupied,
        #                 over the live kernel time
        #  valu_only   -- LIVE: the same launch through the VALU-only timing probe (xgpr_amd/build.py PROBE_LIB: same
        #                 instruction stream, LDS traffic / barrier / prefetch compiled out), over the live kernel time
        vector_pipe = None
        try:
            tab = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_ztz3_inst_table.json")))
            if (d, m) == (1024, 8192):
                tiles = n_local * ((m // 2 + 1023) // 1024)
                peak_inst = 256 * 4 * 2.4e9 / 2
                inst_s = tab["valu_instructions"] * tiles / (kern_ms * 1e-3)
                pipe_ms = tab["priced_vector_ns_per_tile_per_simd"] * tiles / 1024 * 1e-6
                vector_pipe = {"valu_insts_per_tile": tab["valu_instructions"], "tiles_per_launch": tiles,
                               "issue_slots": {"achieved": inst_s / 1e9, "peak": peak_inst / 1e9, "unit": "G wave-instructions/s",
                                               "frac": inst_s / peak_inst},
                               "priced": {"pipe_ms": pipe_ms, "frac": pipe_ms / kern_ms,
                                          "source": "profiles/%s_ztz3_inst_table.json x profiles/r3_valu_cost.json (stored)" % PROFILE_ROUND},
                               "valu_only": valu_probe}
        except (OSError, KeyError, ValueError):
            pass
        alg_bytes = 4.0 * d * n_local                   # 4*d bytes per row, X read once
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        fg_bytes = (4.0 * d + 8.0 * m) * fg_rows
        fg_gbs = fg_bytes / (fg_ms * 1e-3) / 1e9
        out = {
            "metric": "random-features/sec (fused CG matvec; every CG iteration regenerates all N x M features)",
            "value": n * m * args.steps / t,
            "unit": "random-features/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t / args.steps,
            "cg_iters_per_sec": args.steps / t,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (SORF + cos/sin) / f64 (Z^T Z p accumulation and CG state)",
            "data": "synthetic",
            "build_id": build_id,
            "direct_rccl_child": (None if args.gpus == 1 else
                                  "after this line rank 0 starts a fresh `bench.py --gpus %d` with XGPR_RCCL_DIRECT=1 under a watchdog; its record "
                                  "(direct_equals_torch_allreduce, final_loss_check, all-reduce times) goes to stderr and "
                                  "gpurun_out/bench_direct_rccl_child.json and cannot change this line or the exit code" % args.gpus
                                  if (not args.no_direct_child and "XGPR_BENCH_CHILD" not in os.environ
                                      and os.environ.get("XGPR_BENCH_DIRECT_CHILD", "1") != "0") else "off"),
            "config": {"workload": "BASELINE configs[2]: Matern-5/2, N=%d, d=%d, %d RFFs, rows sharded over %d GPU(s), "
                                   "rank-%d SRHT preconditioner (all rows), CG step" %
                                   (n, d, m, args.gpus, args.rank_precond),
                       "rows_per_gpu": n_local, "lambda": 0.1, "sigma": 1.0},
            "roofline": {"kernel": "ztz3_kernel<10> (+ pack_radem, reduce_slabs)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": (traffic_src + " (rocprofv3 --pmc passes of this command on this build; "
                         "stored, not re-measured in this run)") if traffic is not None else traffic_src,
                         "algorithmic_bytes": alg_bytes, "kernel_ms": kern_ms,
                         "vector_pipe": vector_pipe,
                         "note": "HBM traffic of this kernel is only the X read; the binding resource is the vector pipe "
                                 "(butterflies, cos/sin, float64 dot + rank-1 update) at 3 waves/SIMD: vector_pipe, "
                                 "profiles/r6_fused_pmc_sq.json, DESIGN.md section 3"},
            "featgen_op": {"rows": fg_rows, "ms": fg_ms, "features_per_s": fg_rows * m / (fg_ms * 1e-3),
                           "roofline": {"bound": "hbm", "achieved": fg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": fg_gbs / HBM_PEAK_GBS, "traffic": None}},
            "conv_featgen": conv,
            "cached_z_mode": cached,
            "distributed": {**dist_info,
                            "per_rank": [{"rank": i, "ms_per_step": v[0], "fused_kernel_ms": v[1],
                                          "allreduce_ms_per_iter": v[2], "rows": int(v[3])} for i, v in enumerate(per_rank)],
                            "allreduce_note": "allreduce_ms_per_iter is measured by HIP events on the compute stream around the "
                                              "all-reduce of w (allreduce_path), inside the timed CG iterations"},
            "final_loss": losses[-1], "final_loss_check": loss_check,
            "fit_to_tol": fit_to_tol,
            "configs": configs,
            "precond_build": {"seconds": precond_build_s, "first_build_seconds": precond_first_s, "rows": n,
                              "rank": args.rank_precond, "method": "srht",
                              "flops": 2.0 * n * args.rank_precond * m,
                              "roofline": {"kernel": "sketch_gemm_lds_kernel<false> (S^T Z on v_mfma_f64_16x16x4_f64, float32 Z rows, operand tiles through LDS) "
                                                     "+ srht_sample_rows_kernel + wave_rbf_kernel (cache rows)", "bound": "mfma",
                                           "achieved": 2.0 * n * args.rank_precond * m / precond_build_s / 1e12 / args.gpus,
                                           "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": 2.0 * n * args.rank_precond * m / precond_build_s / 1e12 / args.gpus / FP64_MFMA_PEAK_TFLOPS,
                                           "mfma_busy": mfma_busy_of(("sketch_gemm_lds_kernel",)),
                                           "traffic": None},
                              "note": "whole build (feature rows, SRHT + sample, contraction, all-reduce, factorizations) over the "
                                      "flops of the contraction alone; per GPU"},
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    torch.cuda.synchronize()
    comm.close()                   # the direct RCCL communicator (if any): every rank is here, its stream is idle
    if comm.through_backend:
        torch.distributed.destroy_process_group()
    if (args.gpus > 1 and comm.rank == 0 and not args.no_direct_child and "XGPR_BENCH_CHILD" not in os.environ
            and os.environ.get("XGPR_BENCH_DIRECT_CHILD", "1") != "0"):
        sys.stdout.flush()
        # this process still holds its GPU; the other ranks of this job are exiting.  Give them a moment (a GPU box admits only a few
        # processes per device at once), and do not stack N more ranks on a device that all ranks share (one-device rehearsals)
        if "XGPR_LOCAL_DEVICE" in os.environ and args.gpus > 3:
            print("direct-rccl child: skipped (%d ranks share one device in this rehearsal)" % args.gpus, file=sys.stderr)
        else:
            time.sleep(float(os.environ.get("XGPR_BENCH_CHILD_DELAY", "3")))
            direct_rccl_child(args)    # never raises, never touches stdout or the exit code
    if loss_check["ok"] is False:
        raise SystemExit(f"final loss {losses[-1]!r} differs from the stored {loss_check['expected']!r} (rtol {loss_check['rtol']}): "
                         "the timed iterations did not do the work they claim")


if __name__ == "__main__":
    main()
