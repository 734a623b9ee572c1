"""Randomized-Nystrom (SRHT) preconditioner on the device, mirroring the reference.

  * ``RandNysPreconditioner``       <-> preconditioners/rand_nys_preconditioners.py:18-72
  * ``single_pass_srht_zty``        <-> preconditioners/rand_nys_constructors.py:96-123
  * ``single_pass_gauss``           <-> rand_nys_constructors.py:18-36
  * ``initialize_srht``             <-> rand_nys_constructors.py:221-296
  * ``initialize_srht_multipass``   <-> rand_nys_constructors.py:127-218
  * ``subsampled_srht`` / ``srht_ratio_check`` <-> rand_nys_constructors.py:60-93, :301-357
  * ``check_rank_ratio`` / ``autoselect_preconditioner`` <-> model_baseclass.py:438-480, :376-436

The accumulation passes run on the device over windows of FLOAT32 feature rows (the resident cache, or rows
regenerated window by window): the compressor step reads them directly (hipSRHTSampleRows: pad + SRHT + gather
+ z^T y in one pass, any padded width up to 32768) and the dense float64 contractions ``S^T Z``, ``Z Q``,
``Z^T T`` run on the matrix cores with Z widened next to the MFMA (hipSketchGemm) -- a float64 copy of Z is never
written.  Kernels without float32 rows (or odd shapes the operators do not cover) take the reference's
formulation: float64 chunks and library GEMMs.  Per-rank partial sums
(``acc``, ``Z^T y``, ``y^T y``) are combined with one RCCL all-reduce per pass; the small
factorizations (SVD / QR / Cholesky of M x rank matrices) then run redundantly -- and
identically -- on every rank (rocSOLVER through torch.linalg), which is cheaper than
broadcasting U.
"""
import numpy as np
import torch

from . import xgpr_hip_rfgen_ext as ext
from .kernels import SRHTCompressor

ROW_WINDOW_BYTES = 4 << 30        # float32 feature rows regenerated per window of the accumulation passes


def _rows_path_ok(dataset, kernel, rank, from_cache, need_bt=False):
    """Whether the accumulation passes can run on float32 feature rows (module docstring)."""
    if not hasattr(kernel, "row_cache_params") or torch.device(kernel.device).type != "cuda":
        return False
    if not from_cache and not (hasattr(kernel, "fused_ok") and kernel.fused_ok() and hasattr(kernel, "fill_feature_cache")):
        return False
    m = kernel.get_num_rffs()
    if m % 2 != 0 or (need_bt and m % 4 != 0) or not hasattr(dataset, "scaled_x"):
        return False
    from .kernels import padded_dims
    return ext.srht_sample_rows_ok(padded_dims(m), rank, m)


def _row_windows(dataset, kernel, from_cache, with_y):
    """(zc [w, M] float32 feature rows, standardised y [w] or None) over the shard."""
    m = kernel.get_num_rffs()
    n = dataset.get_local_ndatapoints()
    yall = dataset.normalized_y() if with_y else None
    step = max(8192, ROW_WINDOW_BYTES // (4 * m)) // 4 * 4       # whole groups of 4 rows: window bases stay 16-byte aligned for any even m
    if from_cache:
        zc = dataset.feature_cache(kernel)
        for lo in range(0, n, step):
            yield zc[lo:lo + step], (None if yall is None else yall[lo:lo + step])
        return
    xs = dataset.scaled_x(kernel.hyperparams[1])
    step = min(step, max(n, 1))
    zwin = torch.empty((step, m), dtype=torch.float32, device=xs.device)
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        kernel.fill_feature_cache(xs[lo:hi], zwin[:hi - lo])
        yield zwin[:hi - lo], (None if yall is None else yall[lo:hi])


def _srht_pass_rows(dataset, kernel, compressor, acc_results, z_trans_y, from_cache):
    """single_pass_srht_zty / single_pass_srht (rand_nys_constructors.py:96-123, :39-56) over float32 rows;
    z_trans_y None: classification, no z^T y.  Returns y^T y (device scalar)."""
    icpt, scale = kernel.row_cache_params()
    rank, m = acc_results.shape
    lds = (rank + 127) // 128 * 128          # whole 128-row tiles: the LDS-staged contraction kernel
    dev = acc_results.device
    y_trans_y = torch.zeros(1, dtype=torch.float64, device=dev)
    zty_chunk = None if z_trans_y is None else torch.empty_like(z_trans_y)
    sws = None if z_trans_y is None else torch.empty(ext.srht_sample_workspace_bytes(m), dtype=torch.uint8, device=dev)
    sbuf = gws = None
    for zc, yw in _row_windows(dataset, kernel, from_cache, z_trans_y is not None):
        w = zc.shape[0]
        if sbuf is None or sbuf.shape[0] < w:
            sbuf = torch.empty((w, lds), dtype=torch.float64, device=dev)
            gws = torch.empty(ext.sketch_gemm_workspace_bytes(rank, m, w, m, False), dtype=torch.uint8, device=dev)
        sk = sbuf[:w]
        if z_trans_y is None:
            ext.hipSRHTSampleRows(zc, compressor.radem, compressor.truncated_sampler, sk, rank, icpt, scale)
        else:
            ext.hipSRHTSampleRows(zc, compressor.radem, compressor.truncated_sampler, sk, rank, icpt, scale, yw,
                                  zty_chunk, sws)
            z_trans_y += zty_chunk
            y_trans_y += yw @ yw
        ext.hipSketchGemm(sk, zc, acc_results, rank, False, False, icpt, scale, accumulate=True, workspace=gws)
    return y_trans_y


def _gauss_pass_rows(dataset, kernel, q_mat, acc_results, from_cache):
    """single_pass_gauss (rand_nys_constructors.py:18-36): acc[M, rank] += Z^T (Z Q) over float32 rows."""
    icpt, scale = kernel.row_cache_params()
    m, rank = acc_results.shape
    ldq = (rank + 127) // 128 * 128
    dev = acc_results.device
    qpad = torch.zeros((m, ldq), dtype=torch.float64, device=dev)
    qpad[:, :rank] = q_mat
    tbuf = ws1 = ws2 = None
    for zc, _ in _row_windows(dataset, kernel, from_cache, False):
        w = zc.shape[0]
        if tbuf is None or tbuf.shape[0] < w:
            tbuf = torch.zeros((w, ldq), dtype=torch.float64, device=dev)       # padding columns stay zero
            ws1 = torch.empty(ext.sketch_gemm_workspace_bytes(rank, w, m, ldq, True), dtype=torch.uint8, device=dev)
            ws2 = torch.empty(ext.sketch_gemm_workspace_bytes(rank, m, w, rank, True), dtype=torch.uint8, device=dev)
        tk = tbuf[:w]
        ext.hipSketchGemm(qpad, zc, tk, rank, True, True, icpt, scale, workspace=ws1)                    # T = Z Q
        ext.hipSketchGemm(tk, zc, acc_results, rank, False, True, icpt, scale, accumulate=True, workspace=ws2)   # acc += Z^T T


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
    rows = _rows_path_ok(dataset, kernel, rank, from_cache)
    if not is_regression:
        acc_results = torch.zeros((rank, m), dtype=torch.float64, device=kernel.device)
        if rows:
            _srht_pass_rows(dataset, kernel, compressor, acc_results, None, from_cache)
        else:
            single_pass_srht(dataset, kernel, compressor, acc_results, verbose, from_cache)
        comm.all_reduce_(acc_results)
        return acc_results, None, 0, compressor
    acc_results = torch.zeros((rank, m), dtype=torch.float64, device=kernel.device)
    z_trans_y = torch.zeros(m, dtype=torch.float64, device=kernel.device)
    if rows:
        y_trans_y = _srht_pass_rows(dataset, kernel, compressor, acc_results, z_trans_y, from_cache)
    else:
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


CHOL_COND_LIMIT = 1e6        # estimated condition number of the sketch up to which the triangular route is taken
CHOL_COND_ITERS = 12


def _chol_cond_estimate(chol):
    """Estimate of cond_2(C) for C = L L^T from its Cholesky factor: largest eigenvalue of C by power iteration on
    L L^T, smallest by inverse power iteration (two triangular solves per step) -- a few dozen rank x rank launches, no
    host synchronisation before the final read.  Both iterations approach their eigenvalue from inside the spectrum,
    so the product is a lower bound that a dozen steps from a generic start bring within a small factor; the caller
    applies a safety factor.  (The ratio of the extreme diagonal entries of L, which this replaces, also only bounds
    cond(C) from below -- but by orders of magnitude for rotated spectra.)"""
    n = chol.shape[0]
    gen = torch.Generator(device="cpu").manual_seed(1234)
    v = torch.randn(n, 1, generator=gen, dtype=torch.float64).to(chol.device)
    w = v.clone()
    top = bot = None
    # L^-1 once (one blocked triangular solve with n right-hand sides), then two products per step: the 2 x 12
    # single-vector triangular solves this replaces are ~100 us each at n = 512 (rocBLAS trsv), 5 ms of a 150 ms build
    linv = torch.linalg.solve_triangular(chol, torch.eye(n, dtype=torch.float64, device=chol.device), upper=False)
    for _ in range(CHOL_COND_ITERS):
        v = chol @ (chol.T @ v)
        top = torch.linalg.vector_norm(v)
        v = v / top
        w = linv.T @ (linv @ w)
        bot = torch.linalg.vector_norm(w)
        w = w / bot
    return float((top * bot).item())


def _inv_sqrt_apply(acc_t, c_mat):
    """acc_t @ C^(-1/2) for the symmetric positive semi-definite sketch C (rand_nys_constructors.py:275-285,
    where the reference takes an SVD of C) -- up to an orthogonal factor on the right: the only consumer is the thin
    SVD of the product, whose U and singular values do not see that factor.  With C = L L^T, acc_t @ L^(-T) equals
    acc_t @ C^(-1/2) @ (C^(1/2) L^(-T)) and the last factor is orthogonal, so a well-conditioned sketch takes one
    blocked Cholesky and a triangular solve (a dozen launches) instead of a symmetric eigendecomposition (rocSOLVER:
    ~2200 launch-bound kernels for a 512 x 512 problem, ~22 ms)."""
    c_sym = 0.5 * (c_mat + c_mat.T)
    if float((c_mat - c_mat.T).abs().max().item()) <= 1e-9 * float(c_mat.abs().max().item()):
        chol, info = torch.linalg.cholesky_ex(c_sym)
        if int(info.item()) == 0:
            cond = _chol_cond_estimate(chol)
            if np.isfinite(cond) and 4.0 * cond < CHOL_COND_LIMIT:           # (4: the estimate is a lower bound)
                return torch.linalg.solve_triangular(chol, acc_t.T, upper=False).T
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


def _tall_svals(b_mat):
    """Singular values of b_mat [M, rank], descending -- what the rank check needs of the thin SVD: the eigenvalues of
    the Gram matrix alone (no eigenvectors, no back-transformation: roughly half the symmetric eigensolver's work)."""
    evals = torch.linalg.eigvalsh(b_mat.T @ b_mat)
    if float(evals[0].item()) * GRAM_COND_LIMIT > float(evals[-1].item()) > 0:
        return torch.sqrt(evals.flip(0))
    return torch.linalg.svdvals(b_mat)


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
        if _rows_path_ok(dataset, kernel, rank, from_cache, need_bt=True):
            _gauss_pass_rows(dataset, kernel, q_mat, acc_results, from_cache)
        else:
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


def subsampled_srht(dataset, kernel, compressor, acc_results, verbose, sample_frac=0.1, random_seed=123):
    """rand_nys_constructors.py:60-93: the SRHT accumulation pass over a random sample of every chunk (the
    host draws the row indices with the reference's generator calls)."""
    rng = np.random.default_rng(random_seed)
    for j, (xdata, ldata) in enumerate(dataset.get_chunked_x_data()):
        cutoff = max(int(sample_frac * float(xdata.shape[0])), 1)
        idx = rng.permutation(xdata.shape[0])[:cutoff]
        tidx = torch.from_numpy(idx).to(xdata.device)
        zdata = kernel.transform_x(xdata[tidx, ...], None if ldata is None else ldata[idx])
        acc_results += compressor.transform_x(zdata).T @ zdata
        if j % 10 == 0 and verbose:
            print(f"Chunk {j} complete.")


def srht_ratio_check(dataset, rank, kernel, random_state, verbose=False, sample_frac=0.1):
    """rand_nys_constructors.py:301-357 -> the eigenvalues of the Nystrom approximation built from the sample."""
    acc_results = torch.zeros((rank, kernel.get_num_rffs()), dtype=torch.float64, device=kernel.device)
    compressor = SRHTCompressor(rank, kernel.get_num_rffs(), device=kernel.device, random_seed=random_state)
    subsampled_srht(dataset, kernel, compressor, acc_results, verbose, sample_frac, random_state)
    dataset.comm.all_reduce_(acc_results)
    c_mat = compressor.transform_x(acc_results)
    acc_results = _inv_sqrt_apply(acc_results.T, c_mat)
    return _tall_svals(acc_results) ** 2


def check_rank_ratio(kernel, dataset, sample_frac=0.1, max_rank=512, random_seed=123, verbose=False):
    """model_baseclass.py:438-480: min eigenvalue / lambda^2 / sample_frac for a preconditioner of rank
    ``max_rank`` estimated from a sample; like the reference, kernels with more than 8192 random features are
    checked with an 8192-feature kernel of the same family, seed and hyperparameters."""
    if sample_frac < 0.01 or sample_frac > 1:
        raise RuntimeError("sample_frac must be in the range [0.01, 1]")
    check_kernel = kernel
    if kernel.get_num_rffs() > 8192:
        from .kernels import make_kernel
        check_kernel = make_kernel(kernel.kernel_choice, dataset.get_xdim(), 8192, kernel.random_seed, kernel.device,
                                   kernel.kernel_spec_parms)
        check_kernel.set_hyperparams(kernel.get_hyperparams(logspace=False), logspace=False)
    s_mat = srht_ratio_check(dataset, max_rank, check_kernel, random_seed, verbose, sample_frac)
    return float(s_mat.min().item() / kernel.get_lambda() ** 2) / sample_frac


def autoselect_preconditioner(kernel, dataset, min_rank=512, max_rank=3000, increment_size=512,
                              always_use_srht2=False, ratio_target=30., random_seed=123, is_regression=True,
                              verbose=False):
    """model_baseclass.py:376-436: grow the rank until the sampled ratio drops below ``ratio_target`` (falling
    back to a 2-pass build at ``max_rank``), then build the preconditioner -> (preconditioner, rank, method)."""
    sample_frac, method, ratio, rank = 0.2, "srht", np.inf, min_rank
    actual_num_rffs = kernel.get_num_rffs()
    if rank >= actual_num_rffs:
        rank = actual_num_rffs - 1
        ratio = 0.5 * ratio_target
    if dataset.get_ndatapoints() < 5000:
        sample_frac = 1
    while ratio > ratio_target and rank < max_rank:
        ratio = check_rank_ratio(kernel, dataset, sample_frac, rank, random_seed, verbose)
        if ratio > ratio_target:
            if (rank + increment_size) < max_rank and (rank + increment_size) < actual_num_rffs:
                rank += increment_size
            else:
                rank = max_rank
                if rank > actual_num_rffs:
                    rank = actual_num_rffs - 1
                method = "srht_2"
                break
    if verbose:
        print(f"Using rank: {rank}")
    if always_use_srht2:
        method = "srht_2"
    return RandNysPreconditioner(kernel, dataset, rank, verbose, random_seed, method, is_regression), rank, method


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
        """rand_nys_preconditioners.py:66-72.  Blocks of up to 32 float64 right-hand sides on the device take
        x + U ((inv_eig * prefactor - 1) .* (U^T x)) -- the same operator in two products -- through
        hipPrecondApplyBlock (float64 matrix cores; the library's skinny GEMM for U^T x takes 220 us at rank 512)."""
        if (xvec.is_cuda and xvec.dim() == 2 and xvec.dtype == torch.float64
                and 1 <= xvec.shape[1] <= ext.PRECOND_UTR_BLOCK_MAX_K and xvec.shape[0] == self.u_mat.shape[0]):
            xc = xvec.contiguous()
            out = torch.empty_like(xc)
            ext.hipPrecondApplyBlock(self.u_mat, self.inv_eig, self.prefactor, xc, out)
            return out
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
