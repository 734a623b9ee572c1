"""Negative marginal log likelihood: exact, its gradient, and the approximate (stochastic Lanczos
quadrature) form that drives hyperparameter tuning -- the "next" rows 1 and 2 of SURVEY.md
section 8f.  They reuse the hot path unchanged: the approximate NMLL is one preconditioned CG solve
with k = nsamples + 1 = 26 right-hand sides, whose block matvec runs on the float64 matrix cores
(hipZCacheBlockMatvec); the gradient terms are the gradient operators (hipRBFGrad / hipConvGrad)
followed by dense M x M accumulations.

  * ``optimize_alpha_beta``     <-> scoring_toolkit/alpha_beta_optimizer.py:13-39
  * ``generate_normal_probes``  <-> scoring_toolkit/probe_generators.py:9-30 (gpu) / :54-75 (cpu)
  * ``estimate_logdet``         <-> scoring_toolkit/approximate_nmll_calcs.py:12-50
  * ``approximate_nmll``        <-> xgp_regression.py:264-367
  * ``exact_nmll``              <-> xgp_regression.py:152-205
  * ``calc_gradient_terms``     <-> scoring_toolkit/nmll_gradient_tools.py:12-93 (subsample = 1)
  * ``exact_nmll_reg_grad``     <-> scoring_toolkit/nmll_gradient_tools.py:97-162
  * ``exact_nmll_gradient``     <-> xgp_regression.py:209-260

The tridiagonal eigenproblems of the quadrature (niter x niter, per probe) are solved on the host
with LAPACK ``stev`` exactly as the reference does; everything M-sized stays on the device.
"""
import warnings

import numpy as np
import torch
from scipy.linalg import eigh_tridiagonal

from .cg import ConjugateGrad, _resolve_cache_mode
from .exact import calc_design_mat, direct_weight_calc
from .preconditioner import RandNysPreconditioner

DEFAULT_SCORE_IF_PROBLEM = 1e40              # constants.py:13
DEFAULT_NMLL_PARAMS = {"max_rank": 1024, "preconditioner_mode": "srht_2", "nsamples": 25,
                       "nmll_iter": 500, "nmll_tol": 1e-6}          # constants.py:15-16


def optimize_alpha_beta(lambda_, nll_terms, ndatapoints, nrffs, beta_max=10., beta_min=0.1):
    beta = np.sqrt(2 * nll_terms[0] / (ndatapoints * lambda_ ** 2))
    beta = max(min(beta, beta_max), beta_min)
    score = nll_terms[0] / (beta * lambda_) ** 2 + (ndatapoints - nrffs) * np.log(lambda_)
    score += nll_terms[1] + ndatapoints * np.log(beta)
    return score + 0.5 * ndatapoints * np.log(2 * np.pi), beta


def generate_normal_probes(nsamples, num_rffs, random_seed=123, preconditioner=None, device="cuda"):
    """Probe vectors N(0, I) -- or N(0, P) through the preconditioner -- drawn on the host with the
    same generator calls as the reference (they are inputs to the device path, not part of it)."""
    rng = np.random.default_rng(random_seed)
    probes = torch.from_numpy(rng.standard_normal(size=(num_rffs, nsamples))).to(device)
    if preconditioner is not None:
        probes = preconditioner.matvec_for_sampling(probes)
    return probes


def estimate_logdet(alphas, betas, num_rffs, preconditioner=None):
    """log det (Z^T Z + lambda^2) from the CG coefficients of the probe solves: each probe's
    (alpha, beta) sequence defines the Lanczos tridiagonal, whose Gauss quadrature of log gives the
    probe's estimate; the preconditioner's log-determinant is added back."""
    alphas = alphas.detach().cpu().numpy() if isinstance(alphas, torch.Tensor) else np.asarray(alphas)
    betas = betas.detach().cpu().numpy() if isinstance(betas, torch.Tensor) else np.asarray(betas)
    mat_diag = 1 / alphas
    mat_diag[1:, :] += betas[:-1, :] / alphas[:-1, :]
    upper_diag = np.sqrt(betas) / alphas
    logdets = np.zeros((mat_diag.shape[1]))
    for i in range(mat_diag.shape[1]):
        eigvals, eigvecs = eigh_tridiagonal(mat_diag[:, i], upper_diag[:-1, i], lapack_driver="stev")
        weights = eigvecs[0, :] ** 2
        logdets[i] += (weights * np.log(eigvals)).sum()
    logdet = num_rffs * logdets.sum() / alphas.shape[1]
    if preconditioner is not None:
        logdet += preconditioner.get_logdet()
    return float(logdet)


def approximate_nmll(kernel, dataset, preconditioner=None, manual_settings=None, random_seed=123,
                     cache_features="auto", details=None):
    """xgp_regression.py:264-367 for a kernel whose hyperparameters are already set.  With no
    preconditioner given one is built from ``manual_settings`` merged over the reference defaults
    (the reference's rank autoselection is outside the path; pass a preconditioner to control it).
    ``details``: optional dict that receives alphas, betas, logdet, the weights and the iteration count."""
    settings = dict(DEFAULT_NMLL_PARAMS)
    if manual_settings is not None:
        for key in settings:
            if key in manual_settings:
                settings[key] = manual_settings[key]
    num_rffs = kernel.get_num_rffs()
    if settings["max_rank"] >= num_rffs:
        settings["max_rank"] = num_rffs - 1
    if preconditioner is None:
        preconditioner = RandNysPreconditioner(kernel, dataset, settings["max_rank"], False, random_seed,
                                               settings["preconditioner_mode"])
    ndatapoints = dataset.get_ndatapoints()
    cg_operator = ConjugateGrad(dataset.comm, _resolve_cache_mode(cache_features, kernel, dataset, block=True))
    resid = torch.zeros((num_rffs, 2, settings["nsamples"] + 1), dtype=torch.float64, device=kernel.device)
    probes = generate_normal_probes(settings["nsamples"], num_rffs, random_seed, preconditioner, kernel.device)
    z_trans_y = preconditioner.get_zty()
    y_trans_y = preconditioner.get_yty()
    resid[:, 0, 0] = z_trans_y / ndatapoints
    resid[:, 0, 1:] = probes
    x_k, alphas, betas = cg_operator.fit(dataset, kernel, preconditioner, resid, settings["nmll_iter"],
                                         settings["nmll_tol"], verbose=False, nmll_settings=True)
    x_k[:, 0] *= ndatapoints
    logdet = estimate_logdet(alphas, betas, num_rffs, preconditioner)
    nll1 = float(0.5 * (y_trans_y - float((z_trans_y @ x_k[:, 0]).item())))
    negloglik, _ = optimize_alpha_beta(kernel.get_lambda(), np.array([nll1, 0.5 * logdet]), ndatapoints, num_rffs)
    if details is not None:
        details.update(alphas=alphas, betas=betas, logdet=logdet, weights=x_k[:, 0], niter=alphas.shape[0],
                       probes=probes)
    return float(negloglik)


def exact_nmll(kernel, dataset):
    """xgp_regression.py:152-205 for a kernel whose hyperparameters are already set."""
    ndatapoints = dataset.get_ndatapoints()
    z_trans_z, z_trans_y, y_trans_y = calc_design_mat(dataset, kernel)
    try:
        chol_z_trans_z, weights = direct_weight_calc(z_trans_z, z_trans_y, kernel)
    except Exception:          # singular design matrix for extreme hyperparameters
        warnings.warn("Near-singular matrix encountered when calculating score.")
        return DEFAULT_SCORE_IF_PROBLEM
    nll1 = float(0.5 * (y_trans_y - float((z_trans_y @ weights).item())))
    nll2 = float(torch.log(torch.diagonal(chol_z_trans_z)).sum().item())
    negloglik, _ = optimize_alpha_beta(kernel.get_lambda(), np.array([nll1, nll2]), ndatapoints,
                                       kernel.get_num_rffs())
    if np.isnan(negloglik):
        warnings.warn("Near-singular matrix encountered when calculating score.")
        return DEFAULT_SCORE_IF_PROBLEM
    return float(negloglik)


def calc_gradient_terms(dataset, kernel):
    """nmll_gradient_tools.py:12-93 with subsample = 1; partial sums are all-reduced over ranks."""
    comm = dataset.comm
    num_rffs = kernel.get_num_rffs()
    nkern = kernel.get_hyperparams().shape[0] - 1
    f64 = dict(dtype=torch.float64, device=kernel.device)
    z_trans_z = torch.zeros((num_rffs, num_rffs), **f64)
    z_trans_y = torch.zeros(num_rffs, **f64)
    dz_dsigma_ty = torch.zeros((num_rffs, nkern), **f64)
    inner_deriv = torch.zeros((num_rffs, num_rffs, nkern), **f64)
    y_trans_y = torch.zeros(1, **f64)
    for xin, yin, ldata in dataset.get_chunked_data():
        xfeatures, dz_dsigma, ydata = kernel.gradient_x_y(xin, yin, ldata)
        z_trans_y += xfeatures.T @ ydata
        z_trans_z += xfeatures.T @ xfeatures
        y_trans_y += ydata @ ydata
        for i in range(dz_dsigma.shape[2]):
            dz_dsigma_ty[:, i] += dz_dsigma[:, :, i].T @ ydata
            inner_deriv[:, :, i] += dz_dsigma[:, :, i].T @ xfeatures
    for t in (z_trans_z, z_trans_y, dz_dsigma_ty, inner_deriv, y_trans_y):
        comm.all_reduce_(t)
    inner_deriv += inner_deriv.transpose(0, 1).clone()
    return z_trans_z, z_trans_y, float(y_trans_y.item()), dz_dsigma_ty, inner_deriv, dataset.get_ndatapoints()


def exact_nmll_reg_grad(z_trans_z, z_trans_y, y_trans_y, hparams, ndatapoints, dz_dsigma_ty, inner_deriv):
    """nmll_gradient_tools.py:97-162 -> (negloglik, grad w.r.t. log hyperparameters, beta)."""
    lam = float(hparams[0])
    z_trans_z.diagonal().add_(lam ** 2)
    chol = torch.linalg.cholesky(z_trans_z)
    weights = torch.cholesky_solve(z_trans_y[:, None], chol)[:, 0]
    z_trans_z.diagonal().sub_(lam ** 2)
    eye = torch.eye(chol.shape[0], dtype=torch.float64, device=chol.device)
    chol_inv = torch.linalg.solve_triangular(chol, eye, upper=False)
    zty_w = float((z_trans_y @ weights).item())
    nll1 = float(0.5 * (y_trans_y - zty_w))
    nll2 = float(torch.log(torch.diagonal(chol)).sum().item())
    nrffs = float(z_trans_z.shape[0])
    negloglik, beta = optimize_alpha_beta(lam, np.array([nll1, nll2]), float(ndatapoints), nrffs)
    grad = np.zeros((hparams.shape[0]))
    alpha = lam * beta
    dnll_dlambda = (1 / (beta ** 2 * lam ** 3)) * (zty_w - y_trans_y)
    dnll_dlambda += (1 / (beta ** 2 * lam)) * float((weights @ weights).item())
    dnll_dlambda += (ndatapoints - chol.shape[1]) / lam
    dnll_dlambda += lam * float((chol_inv ** 2).sum().item())
    grad[0] = float(dnll_dlambda)
    for i in range(grad.shape[0] - 1):
        trace_term = torch.cholesky_solve(inner_deriv[:, :, i], chol)
        dnll_dsigma = -2 * float((weights @ dz_dsigma_ty[:, i]).item())
        dnll_dsigma += float((weights @ (inner_deriv[:, :, i] @ weights)).item())
        dnll_dsigma *= (0.5 / alpha ** 2)
        dnll_dsigma += 0.5 * float(torch.trace(trace_term).item())
        grad[i + 1] = float(dnll_dsigma)
    grad *= hparams
    return negloglik, grad, beta


def exact_nmll_gradient(kernel, dataset):
    """xgp_regression.py:209-260 for a kernel whose hyperparameters are already set."""
    hparams = kernel.get_hyperparams(logspace=False)
    terms = calc_gradient_terms(dataset, kernel)
    z_trans_z, z_trans_y, y_trans_y, dz_dsigma_ty, inner_deriv, nsamples = terms
    negloglik, grad, _ = exact_nmll_reg_grad(z_trans_z, z_trans_y, y_trans_y, hparams, nsamples,
                                             dz_dsigma_ty, inner_deriv)
    return float(negloglik), grad
