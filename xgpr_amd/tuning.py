"""Hyperparameter tuning driver: xgp_regression.py:564-727 (``tune_hyperparams``) as a function over a kernel and a
dataset.  It is scipy's ``minimize`` around the NMLL functions of ``xgpr_amd.nmll`` -- every cost evaluation is
the hot path (feature generation + dense accumulations for the exact NMLL; preconditioner build + the
26-column CG solve on the matrix cores for the approximate NMLL) -- with the reference's options, restart rule
and generator calls.  Default bounds are the kernels' (kernel_baseclass / sorf_kernel_baseclass:
lambda in [1e-3, 1e2], sigma in [1e-6, 1e2]; MiniARD: one sigma bound per group; Linear: lambda in [1e-3, 1e1])."""
import warnings

import numpy as np
from scipy.optimize import minimize

from . import nmll


# the (lambda, sigma) optimization bounds the reference's kernel classes are built with
_LAMBDA_SIGMA_BOUNDS = {
    "RBF": [[1e-3, 4], [1e-6, 1e2]],                        # basic_kernels/rbf.py:39
    "Matern": [[1e-3, 1e1], [1e-6, 1e2]],                   # matern.py:57
    "Cauchy": [[1e-3, 1e1], [1e-6, 1e2]],                   # cauchy.py:44
    "Conv1dRBF": [[1e-3, 5], [1e-6, 1e2]],                  # conv1d_rbf.py:53
    "Conv1dMatern": [[1e-3, 5], [1e-6, 1e2]],               # conv1d_matern.py:65
    "Conv1dCauchy": [[1e-3, 1e1], [1e-6, 1e2]],             # conv1d_cauchy.py:57
    "Conv1dTwoLayer": [[1e-3, 5], [1e-6, 1e2]],             # l2_conv1d.py:98
    "GraphRBF": [[1e-3, 1e2], [1e-2, 1e2]],                 # graph_rbf.py:47
    "GraphMatern": [[1e-3, 1e1], [1e-6, 1e2]],              # graph_matern.py:58
    "GraphCauchy": [[1e-3, 1e1], [1e-6, 1e2]],              # graph_cauchy.py:50
    "Linear": [[1e-3, 1e1]],                                # linear.py:47
}


def default_bounds(kernel, logspace=True):
    name = getattr(kernel, "kernel_choice", "")
    if name == "MiniARD":                                   # mini_ard.py:86-88
        nh = kernel.get_hyperparams().shape[0]
        bounds = np.asarray([[1e-3, 1e2]] + [[1e-6, 1e2] for _ in range(nh - 1)])
    elif name in _LAMBDA_SIGMA_BOUNDS:
        bounds = np.asarray(_LAMBDA_SIGMA_BOUNDS[name], dtype=np.float64)
    else:
        raise RuntimeError(f"no default bounds for kernel '{name}'")
    return np.log(bounds) if logspace else bounds


def tune_hyperparams(kernel, dataset, bounds=None, max_iter=50, tuning_method="Powell", starting_hyperparams=None,
                     tol=1e-2, n_restarts=1, nmll_method="exact", manual_settings=None, random_seed=123,
                     verbose=False):
    """-> (hyperparams (log space), n_feval, best_score); the kernel is left at the best hyperparameters."""
    if tuning_method == "Powell":
        options = {"maxfev": max_iter, "xtol": 1e-1, "ftol": tol}
    elif tuning_method == "Nelder-Mead":
        options = {"maxfev": max_iter, "ftol": tol}
    elif tuning_method == "L-BFGS-B":
        if nmll_method == "approximate":
            raise RuntimeError("Approximate NMLL is not supported for L-BFGS-B at this time.")
        options = {"maxiter": max_iter, "ftol": tol}
    else:
        raise RuntimeError("Invalid tuning method supplied.")
    optim_bounds = default_bounds(kernel) if bounds is None else np.asarray(bounds, dtype=np.float64)
    init_hparams = kernel.get_hyperparams().copy()

    def exact(hp):
        kernel.set_hyperparams(hp, logspace=True)
        return nmll.exact_nmll(kernel, dataset)

    def exact_grad(hp):
        kernel.set_hyperparams(hp, logspace=True)
        try:
            score, grad = nmll.exact_nmll_gradient(kernel, dataset)
        except Exception:                  # non-positive-definite design matrix for extreme hyperparameters
            return nmll.DEFAULT_SCORE_IF_PROBLEM, hp - init_hparams
        if np.isnan(score):
            return nmll.DEFAULT_SCORE_IF_PROBLEM, hp - init_hparams
        return score, grad

    def approximate(hp):
        kernel.set_hyperparams(hp, logspace=True)
        return nmll.approximate_nmll(kernel, dataset, None, manual_settings, random_seed)

    if nmll_method == "approximate":
        cost_fun = approximate
    elif nmll_method == "exact":
        cost_fun = exact_grad if tuning_method == "L-BFGS-B" else exact
    else:
        raise RuntimeError("Invalid nmll method supplied.")
    bounds_tuples = list(map(tuple, optim_bounds))
    rng = np.random.default_rng(random_seed)
    if starting_hyperparams is None:
        x0 = kernel.get_hyperparams()
        if (x0 - optim_bounds[:, 0]).min() < 0 or (optim_bounds[:, 1] - x0).min() < 0:
            x0 = optim_bounds.mean(axis=1)
            warnings.warn("The kernel hyperparameters were outside the optimization boundaries. The mean of the "
                          "optimization boundaries will be used as a starting point.", UserWarning)
    elif isinstance(starting_hyperparams, np.ndarray) and starting_hyperparams.shape[0] == init_hparams.shape[0]:
        x0 = starting_hyperparams
    else:
        raise RuntimeError("Invalid starting hyperparams were supplied.")
    best_score, n_feval, hyperparams = np.inf, 0, np.asarray(x0, dtype=np.float64)
    for _ in range(n_restarts):
        res = minimize(cost_fun, x0=x0, options=options, method=tuning_method, bounds=bounds_tuples,
                       jac=(tuning_method == "L-BFGS-B"))
        n_feval += res.nfev
        if res.fun < best_score:
            n_feval, hyperparams, best_score = res.nfev, res.x, res.fun
        if verbose:
            print(f"Best score: {best_score}")
        x0 = np.asarray([rng.uniform(low=optim_bounds[j, 0], high=optim_bounds[j, 1])
                         for j in range(optim_bounds.shape[0])])
    kernel.set_hyperparams(hyperparams, logspace=True)
    return hyperparams, n_feval, float(best_score)
