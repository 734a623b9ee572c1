"""How far the un-preconditioned CG iterates of the g7 problem (BASELINE cfg1 size) move when the feature
matrix is perturbed at rounding level -- measured on the CPU oracle (reference algorithm:
fitting_toolkit/cg_tools.py:255-287, restated in oracle/oracle.py:cg_fit).

That solve passes through a near-breakdown (two almost dependent search directions): iterates 8 and 9 (1-based)
are determined by the data only to ~1e-3..1e-2, whatever the precision of Z, while every iterate outside the
window 7..10 follows the perturbation size.  ``tests/test_cg_sensitivity.py`` (CPU) pins that behaviour;
``tests/test_gpu_cg.py`` bounds the HIP path's un-preconditioned iterates by the envelope measured here at the
HIP path's own feature error instead of by a literal tolerance.  Test infrastructure only."""
import numpy as np

from oracle import oracle as orc

WINDOW = (6, 7, 8, 9)          # 0-based iterate indices of the near-breakdown window


class _FixedFeatures:
    """Kernel + dataset pair over a precomputed (perturbed) feature matrix: the chunk iterators hand out row
    ranges and transform_x slices the matrix, so that CG sees the same perturbed Z in every iteration."""

    def __init__(self, z, y_chunks, lam, chunk):
        self.z, self.y_chunks, self.lam, self.chunk = z, y_chunks, lam, chunk

    def get_lambda(self):
        return self.lam

    def get_num_rffs(self):
        return self.z.shape[1]

    def get_ndatapoints(self):
        return self.z.shape[0]

    def transform_x(self, rows, _lengths=None):
        return self.z[rows[0]:rows[1]]

    def get_chunked_x_data(self):
        for i in range(0, self.z.shape[0], self.chunk):
            yield (i, min(i + self.chunk, self.z.shape[0])), None

    def get_chunked_data(self):
        for i, yc in zip(range(0, self.z.shape[0], self.chunk), self.y_chunks):
            yield (i, min(i + self.chunk, self.z.shape[0])), yc, None


def oracle_problem(g, kname):
    """(Z from the oracle, standardised y chunks, lambda, chunk size) of the g7 problem for kernel ``kname``."""
    x, y = g["x"], g["y"]
    chunk = int(g["chunk_size"])
    ods = orc.OracleDataset(x.astype(np.float64), y, None, chunk_size=chunk)
    ok = orc.OracleKernel(kname, int(g["num_rffs"]), x.shape, g["hyperparams"], 123, matern_nu=2.5)
    z = ok.transform_x(x)
    y_chunks = [yc for _, yc, _ in ods.get_chunked_data()]
    return z, y_chunks, float(ok.get_lambda()), chunk


def iterate_errors(g, kname, z, y_chunks, lam, chunk, niter=None, tol=1e-30):
    """Relative error of every un-preconditioned CG iterate over the feature matrix ``z`` against the
    reference's iterates in ``g`` (and the iteration count of the run)."""
    ref = g[f"{kname}_none_iterates"]
    prob = _FixedFeatures(z, y_chunks, lam, chunk)
    trace = {}
    _, n_it, _, _ = orc.cg_fit_lib_internal(prob, prob, tol, ref.shape[0] if niter is None else niter, None, trace=trace)
    n = z.shape[0]
    errs = np.array([np.linalg.norm(trace["x_k"][j][:, 0] * n - ref[j]) / np.linalg.norm(ref[j])
                     for j in range(min(ref.shape[0], len(trace["x_k"])))])
    return errs, n_it


def envelope(g, kname, eps, seeds=range(6)):
    """Per-iterate maximum, over ``seeds``, of the iterate error when every feature is multiplied by
    1 + eps * U(-1, 1); also the range of iteration counts of the full solve (tol 1e-8) under the same
    perturbations."""
    z, y_chunks, lam, chunk = oracle_problem(g, kname)
    env = None
    counts = []
    for seed in seeds:
        rng = np.random.default_rng(seed)
        zp = z * (1.0 + eps * rng.uniform(-1.0, 1.0, size=z.shape))
        errs, _ = iterate_errors(g, kname, zp, y_chunks, lam, chunk)
        env = errs if env is None else np.maximum(env, errs)
        _, n_it = iterate_errors(g, kname, zp, y_chunks, lam, chunk, niter=500, tol=1e-8)
        counts.append(n_it)
    return env, (min(counts), max(counts))
