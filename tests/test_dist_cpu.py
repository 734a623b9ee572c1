"""world_size-2 test of the multi-rank path on CPU (gloo): row sharding, the global y
statistics, the all-reduce of the per-shard partial matvec / z^T y, and the replicated CG
updates of xgpr_amd.cg.  The product's hot operators only exist on the GPU, so here -- in the
test only -- a kernel object with the product kernel's interface serves them from the CPU
oracle; everything else is the product's own host code.  The 2-rank solution must equal the
single-process oracle solution."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class OracleBackedKernel:
    """Test double for xgpr_amd.kernels.SORFKernel (same methods the CG driver calls)."""
    device = "cpu"

    def __init__(self, num_rffs, d, hyperparams, seed=123):
        from oracle import oracle as orc
        self.ops = orc.Oracle()
        self.num_rffs, self.num_freqs = num_rffs, num_rffs // 2
        self.hyperparams = np.asarray(hyperparams, dtype=np.float64)
        self.fit_intercept = True
        self.radem, self.chi = orc.draw_sorf_params(num_rffs, d, seed)

    def get_lambda(self):
        return self.hyperparams[0]

    def get_num_rffs(self):
        return self.num_rffs

    def fused_ok(self):
        return True

    def workspace_bytes(self):
        return 16

    def _features(self, xs):
        z = np.zeros((xs.shape[0], self.num_rffs))
        self.ops.cpuRBFFeatureGen(np.ascontiguousarray(xs.numpy()), z, self.radem, self.chi, True)
        z[:, 0] = 1.0
        return z

    def ztz_matvec(self, xs, vec, out, ws=None):
        z = self._features(xs)
        out.copy_(torch.from_numpy(z.T @ (z @ vec.numpy())))

    def zty(self, xs, y, out, ws=None):
        out.copy_(torch.from_numpy(self._features(xs).T @ y.numpy()))


def _problem():
    rng = np.random.default_rng(42)
    n, d, m = 601, 12, 128
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    y = np.sin(x @ rng.standard_normal(d)) + 0.1 * rng.standard_normal(n)
    return x, y, m, np.array([0.3, 0.8])


def _worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from xgpr_amd import dist as xd
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import cg_fit_lib_internal
    comm = xd.init_from_env(device_type="cpu")
    assert comm.world_size == world and comm.rank == rank
    x, y, m, hp = _problem()
    ds = build_regression_dataset(x, y, chunk_size=100, device="cpu", comm=comm)
    lo, hi = comm.shard_bounds(x.shape[0])
    assert ds.get_local_ndatapoints() == hi - lo and ds.get_ndatapoints() == x.shape[0]
    kern = OracleBackedKernel(m, x.shape[1], hp)
    w, niter, losses = cg_fit_lib_internal(kern, ds, 1e-10, 400, None, verbose=False)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), w=w.numpy(), niter=niter, ymean=ds.get_ymean(),
             ystd=ds.get_ystd())
    torch.distributed.destroy_process_group()


def test_two_rank_cg_equals_single_process_oracle(tmp_path):
    from oracle import oracle as orc
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["w"], r1["w"]) and int(r0["niter"]) == int(r1["niter"])   # replicated state
    x, y, m, hp = _problem()
    ods = orc.OracleDataset(x.astype(np.float64), y, chunk_size=100)
    assert np.isclose(float(r0["ymean"]), ods.y_mean, rtol=1e-12)
    assert np.isclose(float(r0["ystd"]), ods.y_std, rtol=1e-12)
    okern = orc.OracleKernel("RBF", m, x.shape, hp, 123)
    wref, nref, _, _ = orc.cg_fit_lib_internal(okern, ods, 1e-10, 400, None)
    assert int(r0["niter"]) == nref
    assert np.linalg.norm(r0["w"] - wref) <= 1e-9 * np.linalg.norm(wref)


# ---- second scenario: the rows next to the path (approximate NMLL with a 2-pass preconditioner; the
# classifier's cost function) on two ranks.  The SRHT of the preconditioner chunks is a HIP operator too,
# so the test swaps in an oracle-backed compressor with the product compressor's interface.
class OracleBackedKernel2(OracleBackedKernel):
    def transform_x(self, input_x, sequence_length=None):
        from xgpr_amd.kernels import scale_input
        xs = scale_input(input_x.to(torch.float32), self.hyperparams[1])
        return torch.from_numpy(self._features(xs))

    def transform_x_y(self, input_x, input_y, sequence_length=None):
        return self.transform_x(input_x), input_y.to(torch.float64)

    def fused_ok(self):
        return False


class OracleBackedCompressor:
    def __init__(self, compression_size, input_size, device="cpu", random_seed=123):
        from oracle import oracle as orc
        self.inner = orc.OracleSRHTCompressor(compression_size, input_size, random_seed)

    def transform_x(self, features, no_compression=False, out=None):
        res = torch.from_numpy(self.inner.transform_x(features.numpy(), no_compression))
        if out is None:
            return res
        out[:, :res.shape[1]] = res
        return out[:, :res.shape[1]]


def _class_problem():
    rng = np.random.default_rng(43)
    n, d, m, ncls = 301, 10, 64, 3
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    y = rng.integers(0, ncls, size=n).astype(np.int64)
    y[:ncls] = np.arange(ncls)
    return x, y, m, np.array([0.5, 0.7]), 0.1 * rng.standard_normal((m, ncls))


def _worker2(rank, world, port, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from xgpr_amd import dist as xd, preconditioner as xp, nmll
    from xgpr_amd.dataset import build_regression_dataset, build_classification_dataset
    from xgpr_amd.classification import NonlinearCGClassification
    xp.SRHTCompressor = OracleBackedCompressor
    comm = xd.init_from_env(device_type="cpu")
    x, y, m, hp = _problem()
    ds = build_regression_dataset(x, y, chunk_size=100, device="cpu", comm=comm)
    kern = OracleBackedKernel2(m, x.shape[1], hp)
    pre = xp.RandNysPreconditioner(kern, ds, 32, False, 123, "srht_2")
    det = {}
    val = nmll.approximate_nmll(kern, ds, pre, {"nsamples": 5, "nmll_iter": 200, "nmll_tol": 1e-8}, 123,
                                cache_features=False, details=det)
    xc, yc, mc, hpc, w0 = _class_problem()
    cds = build_classification_dataset(xc, yc, chunk_size=64, device="cpu", comm=comm)
    assert cds.get_n_classes() == 3 and cds.get_ndatapoints() == xc.shape[0]
    ckern = OracleBackedKernel2(mc, xc.shape[1], hpc)
    grad, loss = NonlinearCGClassification(cds, ckern, False, None, cache_features=False) \
        .cost_fun_classification(torch.from_numpy(w0))
    np.savez(os.path.join(outdir, f"s2_rank{rank}.npz"), nmll=val, logdet=det["logdet"], ratio=pre.achieved_ratio,
             grad=grad.numpy(), loss=loss)
    torch.distributed.destroy_process_group()


def test_two_rank_nmll_and_classifier_cost_equal_single_process_oracle(tmp_path):
    from oracle import oracle as orc
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker2, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "s2_rank0.npz"), np.load(tmp_path / "s2_rank1.npz")
    assert float(r0["nmll"]) == float(r1["nmll"]) and np.array_equal(r0["grad"], r1["grad"])
    x, y, m, hp = _problem()
    ods = orc.OracleDataset(x.astype(np.float64), y, chunk_size=100)
    okern = orc.OracleKernel("RBF", m, x.shape, hp, 123)
    opre = orc.OracleRandNysPreconditioner(okern, ods, 32, 123, "srht_2")
    assert np.isclose(float(r0["ratio"]), opre.achieved_ratio, rtol=1e-6)
    det = {}
    ref = orc.approximate_nmll(okern, ods, opre, 5, 200, 1e-8, 123, details=det)
    assert np.isclose(float(r0["logdet"]), det["logdet"], rtol=1e-7)
    assert np.isclose(float(r0["nmll"]), ref, rtol=1e-8)
    xc, yc, mc, hpc, w0 = _class_problem()
    cds = orc.OracleClassificationDataset(xc.astype(np.float64), yc, chunk_size=64)
    ckern = orc.OracleKernel("RBF", mc, xc.shape, hpc, 123)
    gref, lref = orc.classification_cost(cds, ckern, w0)
    assert np.isclose(float(r0["loss"]), lref, rtol=1e-10)
    assert np.linalg.norm(r0["grad"] - gref) <= 1e-10 * np.linalg.norm(gref)
