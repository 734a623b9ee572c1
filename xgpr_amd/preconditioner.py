"""Randomized-Nystrom (SRHT) preconditioner on the device, mirroring the reference.

  * ``RandNysPreconditioner``       <-> preconditioners/rand_nys_preconditioners.py:18-72
  * ``single_pass_srht_zty``        <-> preconditioners/rand_nys_constructors.py:96-123
  * ``single_pass_gauss``           <-> rand_nys_constructors.py:18-36
  * ``initialize_srht``             <-> rand_nys_constructors.py:221-296
  * ``initialize_srht_multipass``   <-> rand_nys_constructors.py:127-218

The accumulation passes run per chunk on the device (SORF feature generation and the SRHT of
the chunk are HIP kernels of libxgpr_hip.so; the dense ``[rank x n] @ [n x M]`` float64
contractions are plain library GEMMs, rocBLAS through torch).  Per-rank partial sums
(``acc``, ``Z^T y``, ``y^T y``) are combined with one RCCL all-reduce per pass; the small
factorizations (SVD / QR / Cholesky of M x rank matrices) then run redundantly -- and
identically -- on every rank (rocSOLVER through torch.linalg), which is cheaper than
broadcasting U.
"""
import numpy as np
import torch

from .kernels import SRHTCompressor


def _feature_chunks(dataset, kernel, with_y, from_cache):
    if hasattr(dataset, "get_chunked_features"):
        return dataset.get_chunked_features(kernel, with_y, from_cache)
    if with_y:
        return (kernel.transform_x_y(xin, yin, ldata) for xin, yin, ldata in dataset.get_chunked_data())
    return (kernel.transform_x(xin, ldata) for xin, ldata in dataset.get_chunked_x_data())


def single_pass_srht_zty(dataset, kernel, compressor, acc_results, z_trans_y, verbose, from_cache=False):
    """rand_nys_constructors.py:96-123.  The compressed chunk and the chunk's z^T y come out of one read of
    Z (hipSRHTSample) instead of a copy + in-place SRHT + gather and a separate library GEMV; the
    accumulation itself is the float64 MFMA library GEMM."""
    y_trans_y = torch.zeros(1, dtype=torch.float64, device=acc_results.device)
    zty_chunk = torch.empty_like(z_trans_y)
    for j, (xdata, ydata) in enumerate(_feature_chunks(dataset, kernel, True, from_cache)):
        ydata = ydata.to(torch.float64)
        if hasattr(compressor, "transform_x_zty"):
            compressed = compressor.transform_x_zty(xdata, ydata, zty_chunk)
            z_trans_y += zty_chunk
        else:
            z_trans_y += xdata.T @ ydata
            compressed = compressor.transform_x(xdata)
        y_trans_y += ydata @ ydata
        acc_results.addmm_(compressed.T, xdata)
        if j % 10 == 0 and verbose:
            print(f"Chunk {j} complete.")
    return y_trans_y


def single_pass_gauss(dataset, kernel, q_mat, acc_results, verbose, from_cache=False):
    for j, xdata in enumerate(_feature_chunks(dataset, kernel, False, from_cache)):
        acc_results += xdata.T @ (xdata @ q_mat)
        if j % 10 == 0 and verbose:
            print(f"Chunk {j} complete.")


def single_pass_srht(dataset, kernel, compressor, acc_results, verbose, from_cache=False):
    """rand_nys_constructors.py:39-56 (classification: no z^T y)."""
    for j, xdata in enumerate(_feature_chunks(dataset, kernel, False, from_cache)):
        acc_results += compressor.transform_x(xdata).T @ xdata
        if j % 10 == 0 and verbose:
            print(f"Chunk {j} complete.")


def _first_pass(dataset, rank, kernel, random_state, verbose, is_regression=True, from_cache=False):
    comm = dataset.comm
    m = kernel.get_num_rffs()
    compressor = SRHTCompressor(rank, m, device=kernel.device, random_seed=random_state)
    if not is_regression:
        acc_results = torch.zeros((rank, m), dtype=torch.float64, device=kernel.device)
        single_pass_srht(dataset, kernel, compressor, acc_results, verbose, from_cache)
        comm.all_reduce_(acc_results)
        return acc_results, None, 0, compressor
    acc_results = torch.zeros((rank, m), dtype=torch.float64, device=kernel.device)
    z_trans_y = torch.zeros(m, dtype=torch.float64, device=kernel.device)
    y_trans_y = single_pass_srht_zty(dataset, kernel, compressor, acc_results, z_trans_y, verbose, from_cache)
    comm.all_reduce_(acc_results)
    comm.all_reduce_(z_trans_y)
    comm.all_reduce_(y_trans_y)
    return acc_results, z_trans_y, float(y_trans_y.item()), compressor


# ---- the small factorizations.  rocSOLVER's SVD / QR of an [M x rank] matrix are Jacobi / unblocked
# Householder sweeps (0.08 s at rank 512, 2.5 s at rank 2048 -- longer than the accumulation passes they
# follow), while GEMM, Cholesky, triangular solves and the symmetric eigensolver of a rank x rank matrix are
# fast.  Each helper takes the GEMM-rich route when the matrix is safely conditioned for it and otherwise
# falls back to the factorization the reference calls, so degenerate problems (fewer datapoints than rank)
# behave exactly as before.  The routes differ from the reference's LAPACK calls only in column signs / basis
# of intermediate factors, which cancel in U diag(eig) U^T.
GRAM_COND_LIMIT = 1e7        # eigenvalue ratio up to which the Gram-matrix route keeps ~1e-9 relative accuracy


def _inv_sqrt_apply(acc_t, c_mat):
    """acc_t @ C^(-1/2) for the symmetric positive semi-definite sketch C (rand_nys_constructors.py:275-285,
    where the reference takes an SVD of C)."""
    c_sym = 0.5 * (c_mat + c_mat.T)
    if float((c_mat - c_mat.T).abs().max().item()) <= 1e-9 * float(c_mat.abs().max().item()):
        evals, evecs = torch.linalg.eigh(c_sym)
        if float(evals[0].item()) * GRAM_COND_LIMIT > float(evals[-1].item()) > 0:
            return ((acc_t @ evecs) * torch.rsqrt(evals)[None, :]) @ evecs.T
    _, c_s1, c_v1 = torch.linalg.svd(c_mat, full_matrices=False)
    mask = c_s1 < 1e-14
    c_s1 = 1 / torch.sqrt(c_s1.clip(min=1e-14))
    c_s1[mask] = 0
    return acc_t @ c_v1.T @ (c_s1[:, None] * c_v1)


def _tall_svd(b_mat):
    """(U, s) of the thin SVD of b_mat [M, rank] (rand_nys_constructors.py:212, :290)."""
    gram = b_mat.T @ b_mat
    evals, evecs = torch.linalg.eigh(gram)
    if float(evals[0].item()) * GRAM_COND_LIMIT > float(evals[-1].item()) > 0:
        s_mat = torch.sqrt(evals.flip(0))
        u_mat = (b_mat @ evecs.flip(1)) / s_mat[None, :]
        return u_mat, s_mat
    u_mat, s_mat, _ = torch.linalg.svd(b_mat, full_matrices=False)
    return u_mat, s_mat


def _orthonormal_basis(a_mat):
    """Q of the thin QR of a_mat [M, rank] up to column signs (rand_nys_constructors.py:187): two rounds of
    Cholesky QR, checked, with Householder QR as the fallback."""
    q_mat = a_mat
    ok = True
    for _ in range(2):
        chol, info = torch.linalg.cholesky_ex(q_mat.T @ q_mat)
        if int(info.item()) != 0:
            ok = False
            break
        q_mat = torch.linalg.solve_triangular(chol, q_mat.T, upper=False).T
    if ok:
        eye_err = q_mat.T @ q_mat
        eye_err.diagonal().sub_(1.0)
        if float(eye_err.abs().max().item()) < 1e-11:
            return q_mat.contiguous()
    return torch.linalg.qr(a_mat)[0]


def initialize_srht(dataset, rank, kernel, random_state, verbose=False, is_regression=True, from_cache=False):
    acc_results, z_trans_y, y_trans_y, compressor = _first_pass(dataset, rank, kernel, random_state, verbose,
                                                                is_regression, from_cache)
    c_mat = compressor.transform_x(acc_results)
    acc_results = _inv_sqrt_apply(acc_results.T, c_mat)
    u_mat, s_mat = _tall_svd(acc_results)
    s_mat = s_mat ** 2
    return u_mat, s_mat, z_trans_y, y_trans_y


def initialize_srht_multipass(dataset, rank, kernel, random_state, verbose=False, n_passes=1, is_regression=True,
                              from_cache=False):
    comm = dataset.comm
    acc_results, z_trans_y, y_trans_y, _ = _first_pass(dataset, rank, kernel, random_state, verbose, is_regression,
                                                       from_cache)
    acc_results = acc_results.T.contiguous()
    q_mat = None
    for _ in range(n_passes - 1):
        q_mat = _orthonormal_basis(acc_results)
        acc_results.zero_()
        single_pass_gauss(dataset, kernel, q_mat, acc_results, verbose, from_cache)
        comm.all_reduce_(acc_results)
    norm = float(torch.sqrt((acc_results ** 2).sum()).item())
    shift = float(np.spacing(norm))
    acc_results += shift * q_mat
    q_mat = q_mat.T @ acc_results
    q_mat = torch.linalg.cholesky(q_mat)
    acc_results = torch.linalg.solve_triangular(q_mat, acc_results.T, upper=False).T
    u_mat, s_mat = _tall_svd(acc_results)
    s_mat = (s_mat ** 2 - shift).clip(min=0)
    return u_mat, s_mat, z_trans_y, y_trans_y


class RandNysPreconditioner:
    """Preconditioner from the randomized Nystrom approximation of (Z^T Z + lambda^2)^-1."""

    def __init__(self, kernel, dataset, max_rank, verbose=False, random_state=123, method="srht",
                 is_regression=True, cache_features="auto"):
        """``cache_features``: take the feature chunks of the accumulation passes from the dataset's resident
        float32 feature cache (building it if needed) instead of regenerating them.  "auto": only for kernels
        whose features are expensive to regenerate (the convolution kernels) and when the cache fits; the
        solve that follows then reuses the same cache."""
        if method not in ["srht_2", "srht_3", "srht"]:
            raise RuntimeError("Unknown method supplied for tuning preconditioner construction.")
        if cache_features == "auto":
            from .cg import _resolve_cache_mode
            from_cache = (hasattr(kernel, "fused_ok") and not kernel.fused_ok()
                          and hasattr(kernel, "cache_rows_to_features")
                          and _resolve_cache_mode("auto", kernel, dataset, block=True))
        else:
            from_cache = bool(cache_features)
        if method.startswith("srht_"):
            n_passes = int(method.split("_")[1])
            self.u_mat, self.eig, self.z_trans_y, self.y_trans_y = initialize_srht_multipass(
                dataset, max_rank, kernel, random_state, verbose, n_passes, is_regression, from_cache)
        else:
            self.u_mat, self.eig, self.z_trans_y, self.y_trans_y = initialize_srht(
                dataset, max_rank, kernel, random_state, verbose, is_regression, from_cache)
        lambda_ = float(kernel.get_lambda())
        min_eig = float(self.eig.min().item())
        self.eig = self.eig + lambda_ ** 2
        self.inv_eig = self.eig.clone()
        mask = self.inv_eig > 1e-14
        self.inv_eig[mask] = 1 / self.inv_eig[mask]
        self.inv_eig[~mask] = 0.0
        self.achieved_ratio = min_eig / lambda_ ** 2
        self.prefactor = float(min_eig + lambda_ ** 2)
        self.device = kernel.device
        self.u_mat = self.u_mat.contiguous()

    def batch_matvec(self, xvec):
        """rand_nys_preconditioners.py:66-72."""
        xprod = self.u_mat.T @ xvec
        xprod1 = self.u_mat @ (self.inv_eig[:, None] * self.prefactor * xprod)
        xprod2 = xvec - (self.u_mat @ xprod)
        return xprod2 + xprod1

    def get_logdet(self):
        """rand_nys_preconditioners.py:96-102."""
        logdet = 1 + (self.eig - self.prefactor) / self.prefactor
        return float(torch.log(logdet.clamp(min=1e-12)).sum().item())

    def matvec_for_sampling(self, xvec):
        """rand_nys_preconditioners.py:105-119: the square root of the preconditioner's inverse applied
        to probe vectors, so that they are drawn from N(0, P)."""
        eigvals = self.eig.clamp(min=0).sqrt()
        prefactor = (1 / self.prefactor) ** 0.5
        xprod = self.u_mat.T @ xvec
        xprod1 = self.u_mat @ (eigvals[:, None] * prefactor * xprod)
        xprod2 = xvec - (self.u_mat @ xprod)
        return xprod1 + xprod2

    def get_rank(self):
        return self.inv_eig.shape[0]

    def get_zty(self):
        return self.z_trans_y

    def get_yty(self):
        return float(self.y_trans_y)
