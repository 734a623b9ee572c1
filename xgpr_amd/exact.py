"""Closed-form pieces that share the path's dense accumulations ("next" rows of SURVEY.md section 8f):
the design matrix Z^T Z, exact weights, the variance matrix and the predictive mean.

Z^T Z is accumulated by ``xgpr_ztz_gram_f64`` (csrc/gram.inc): windows of float32 feature rows -- regenerated, or
the resident cache for the convolution kernels -- contracted with themselves on the float64 matrix cores
(v_mfma_f64_16x16x4_f64, tiles on or above the diagonal only); a float64 copy of Z is never written.  Shapes the
kernel does not cover (feature counts that are not a multiple of 128, kernels without float32 feature rows, CPU
tensors in the gloo tests) take the reference's formulation, ``xfeatures.T @ xfeatures`` on float64 features.

  * ``calc_design_mat``     <-> scoring_toolkit/exact_nmll_calcs.py:42-78
  * ``direct_weight_calc``  <-> scoring_toolkit/exact_nmll_calcs.py:80-110
  * ``calc_weights_exact``  <-> fitting_toolkit/exact_fitting_toolkit.py:16-40
  * ``calc_variance_exact`` <-> fitting_toolkit/exact_fitting_toolkit.py:43-72, exact_nmll_calcs.py:116-139
  * ``predict_mean``        <-> xgp_regression.py:77-145 (mean only)

The M x M factorizations are library calls (rocSOLVER through torch).  Partial sums are all-reduced over ranks.
"""
import torch


def gram_route(dataset, kernel, msub):
    """How Z[:, :msub]^T Z[:, :msub] is accumulated: False = regenerated float32 windows, True = rows of the resident
    float32 cache, None = the float64 formulation (module docstring)."""
    from . import xgpr_hip_rfgen_ext as ext
    if not hasattr(kernel, "row_cache_params") or torch.device(kernel.device).type != "cuda" or not hasattr(dataset, "scaled_x"):
        return None
    if not ext.gram_ok(kernel.get_num_rffs(), msub):
        return None
    if hasattr(kernel, "fused_ok") and kernel.fused_ok() and hasattr(kernel, "fill_feature_cache"):
        return False
    if hasattr(kernel, "cache_ok") and kernel.cache_ok() and hasattr(kernel, "build_feature_cache"):
        # the resident route pins n_local x M x 4 bytes on the dataset: taken only when the cache is already there or
        # fits in free HBM with the same headroom rule the solver uses (cg._resolve_cache_mode); otherwise the bounded
        # chunked float64 formulation
        from .cg import _resolve_cache_mode
        held = getattr(dataset, "_zcache_key", None) == (id(kernel), float(kernel.hyperparams[1])) \
            and getattr(dataset, "_zcache", None) is not None
        if held or _resolve_cache_mode("auto", kernel, dataset, block=False):
            return True
    return None


def accumulate_gram_rows(dataset, kernel, z_trans_z, from_cache, z_trans_y=None):
    """z_trans_z[msub, msub] += Z[:, :msub]^T Z[:, :msub] over this rank's rows from float32 feature rows; with
    z_trans_y [M] also z_trans_y += Z^T y (one-column back-projection of the same rows).  Returns y^T y (device
    scalar) when z_trans_y is given."""
    from . import xgpr_hip_rfgen_ext as ext
    from .preconditioner import _row_windows
    icpt, scale = kernel.row_cache_params()
    m = kernel.get_num_rffs()
    dev = z_trans_z.device
    gws = bws = None
    y_trans_y = torch.zeros(1, dtype=torch.float64, device=dev)
    for zc, yw in _row_windows(dataset, kernel, from_cache, z_trans_y is not None):
        gws = ext.hipZtZGram(zc, z_trans_z, icpt, scale, accumulate=True, workspace=gws)
        if z_trans_y is not None:
            need = ext.zcache_block_workspace_bytes(zc.shape[0], m, 1)
            if bws is None or bws.numel() < need:
                bws = torch.empty(need, dtype=torch.uint8, device=dev)
            ext.hipZCacheBlockBackproject(zc, yw.reshape(-1, 1).contiguous(), z_trans_y.reshape(-1, 1), icpt, bws, scale,
                                          accumulate=True)
            y_trans_y += yw @ yw
    return y_trans_y


def calc_design_mat(dataset, kernel):
    comm = dataset.comm
    m = kernel.get_num_rffs()
    z_trans_z = torch.zeros((m, m), dtype=torch.float64, device=kernel.device)
    z_trans_y = torch.zeros(m, dtype=torch.float64, device=kernel.device)
    y_trans_y = torch.zeros(1, dtype=torch.float64, device=kernel.device)
    route = gram_route(dataset, kernel, m)
    if route is not None:
        y_trans_y = accumulate_gram_rows(dataset, kernel, z_trans_z, route, z_trans_y)
    else:
        for xin, yin, ldata in dataset.get_chunked_data():
            xfeatures, ydata = kernel.transform_x_y(xin, yin, ldata)
            z_trans_y += xfeatures.T @ ydata
            z_trans_z += xfeatures.T @ xfeatures
            y_trans_y += ydata @ ydata
    comm.all_reduce_(z_trans_z)
    comm.all_reduce_(z_trans_y)
    comm.all_reduce_(y_trans_y)
    return z_trans_z, z_trans_y, float(y_trans_y.item())


def direct_weight_calc(chol_z_trans_z, z_trans_y, kernel):
    """exact_nmll_calcs.py:80-110 -- adds lambda^2 to the diagonal (in place), factorizes, solves."""
    lambda_p = kernel.get_hyperparams(logspace=False)[0]
    chol_z_trans_z.diagonal().add_(float(lambda_p) ** 2)
    chol = torch.linalg.cholesky(chol_z_trans_z)
    weights = torch.cholesky_solve(z_trans_y[:, None], chol)[:, 0]
    return chol, weights


def calc_weights_exact(dataset, kernel):
    """exact_fitting_toolkit.py:16-40.  Faithful to the reference, lambda^2 reaches the diagonal
    TWICE here (once in this function, :36, and again inside direct_weight_calc,
    exact_nmll_calcs.py:100-101): the exact-mode weights solve (Z^T Z + 2 lambda^2) w = Z^T y."""
    z_trans_z, z_trans_y, _ = calc_design_mat(dataset, kernel)
    lambda_p = kernel.get_hyperparams(logspace=False)[0]
    z_trans_z.diagonal().add_(float(lambda_p) ** 2)
    _, weights = direct_weight_calc(z_trans_z, z_trans_y, kernel)
    return weights, 1, []


def calc_variance_exact(kernel, dataset, variance_rffs):
    """exact_fitting_toolkit.py:43-72: pinv of the leading variance_rffs x variance_rffs block of
    Z^T Z + lambda^2."""
    comm = dataset.comm
    z_trans_z = torch.zeros((variance_rffs, variance_rffs), dtype=torch.float64, device=kernel.device)
    route = gram_route(dataset, kernel, variance_rffs)
    if route is not None:                 # the leading variance_rffs x variance_rffs block, from float32 rows
        accumulate_gram_rows(dataset, kernel, z_trans_z, route)
    else:
        for xdata, ldata in dataset.get_chunked_x_data():
            xfeatures = kernel.transform_x(xdata, ldata)
            z_trans_z += xfeatures[:, :variance_rffs].T @ xfeatures[:, :variance_rffs]
    comm.all_reduce_(z_trans_z)
    z_trans_z.diagonal().add_(float(kernel.get_lambda()) ** 2)
    # the matrix is symmetric positive definite (Gram block + lambda^2): the pseudo-inverse through the symmetric
    # eigendecomposition (same cut-off rule as the SVD form the reference calls) -- rocSOLVER's Jacobi SVD of a
    # 512 x 512 matrix takes 74 ms, its syevd 11
    return torch.linalg.pinv(0.5 * (z_trans_z + z_trans_z.T), hermitian=True)


def predict_mean(kernel, weights, input_x, trainy_mean, trainy_std, sequence_lengths=None, chunk_size=2000):
    """xgp_regression.py:77-145, mean only: ``(Z * w).sum(1)`` chunk by chunk, then un-standardise.  For the
    fixed-vector kernels float64 Z is never written: a chunk's float32 feature rows (exactly the values the float64
    operator output is the widening of) go through the one-column projection on the float64 matrix cores
    (``xgpr_zcache_block_project_f32``, k = 1), which applies scale and intercept; other kernels multiply
    ``transform_x`` output."""
    from . import xgpr_hip_rfgen_ext as ext
    preds = []
    fused = (getattr(kernel, "supports_fused", False) and kernel.block_ok() and hasattr(kernel, "fill_feature_cache")
             and torch.device(kernel.device).type == "cuda")
    if fused:
        from .kernels import scale_input
        wcol = weights.to(torch.float64).reshape(-1, 1).contiguous()
    for i in range(0, input_x.shape[0], chunk_size):
        if fused:
            xs = scale_input(kernel._as_device_f32(input_x[i:i + chunk_size]), kernel.hyperparams[1])
            zc = torch.empty((xs.shape[0], kernel.get_num_rffs()), dtype=torch.float32, device=kernel.device)
            kernel.fill_feature_cache(xs, zc)
            pred = torch.empty((xs.shape[0], 1), dtype=torch.float64, device=kernel.device)
            ext.hipZCacheBlockProject(zc, wcol, pred, kernel.fit_intercept, 0.0)
            preds.append(pred[:, 0])
            continue
        sl = None if sequence_lengths is None else sequence_lengths[i:i + chunk_size]
        z = kernel.transform_x(input_x[i:i + chunk_size], sl)
        preds.append((z * weights[None, :]).sum(dim=1))
    return torch.cat(preds) * trainy_std + trainy_mean
