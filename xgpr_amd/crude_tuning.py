"""The "crude" tuner: xgp_regression.py:497-561 (``tune_hyperparams_crude``), scoring_toolkit/lb_optimizer.py
(``shared_hparam_search`` and helpers) and scoring_toolkit/bayes_grid.py (``bayes_grid_tuning``).

For fixed kernel-specific hyperparameters the dense Z^T Z is accumulated once (HIP feature generation + float64
library GEMMs, all-reduced over ranks) and decomposed; the exact NMLL of every lambda on a grid then costs O(M) --
so lambda is searched exhaustively and only sigma is explored, by Bayesian optimisation (scikit-learn Gaussian
process surrogate, Thompson sampling) exactly as the reference does.  The decomposition uses the symmetric
eigensolver (Z^T Z + 1e-5 I is symmetric positive definite, its SVD and eigendecomposition coincide) because
rocSOLVER's Jacobi SVD of an M x M matrix is an order of magnitude slower.
"""
import warnings

import numpy as np
import torch

from .tuning import default_bounds


def get_grid_pts(bounds, n_pts_per_dim, device):
    """lb_optimizer.py:173-192 -> (lambda grid on the device, spacing)."""
    lambda_pts = np.exp(np.linspace(bounds[0, 0], bounds[0, 1], n_pts_per_dim))
    spacing = 1.05 * np.abs(bounds[0, 0] - bounds[0, 1]) / n_pts_per_dim
    return torch.from_numpy(lambda_pts).to(device), spacing


def get_eigvals(kernel, dataset, subsample=1):
    """lb_optimizer.py:68-117 -> (eigvals [M], eigvecs^T z^T y [M], y^T y, ndatapoints)."""
    comm = dataset.comm
    m = kernel.get_num_rffs()
    f64 = dict(dtype=torch.float64, device=kernel.device)
    z_trans_z, z_trans_y = torch.zeros((m, m), **f64), torch.zeros(m, **f64)
    stats = torch.zeros(2, **f64)                    # y^T y, number of datapoints used
    rng = np.random.default_rng(123)
    for xin, yin, ldata in dataset.get_chunked_data():
        if subsample != 1:
            idx_size = max(1, int(subsample * xin.shape[0]))
            idx = rng.choice(xin.shape[0], idx_size, replace=False)
            tidx = torch.from_numpy(idx).to(xin.device)
            xin, yin = xin[tidx, ...], yin[tidx]
            ldata = None if ldata is None else ldata[idx]
        xtrans, ydata = kernel.transform_x_y(xin, yin, ldata)
        z_trans_z += xtrans.T @ xtrans
        z_trans_y += xtrans.T @ ydata
        stats[0] += ydata @ ydata
        stats[1] += xtrans.shape[0]
    for t in (z_trans_z, z_trans_y, stats):
        comm.all_reduce_(t)
    z_trans_z.diagonal().add_(1e-5)
    evals, evecs = torch.linalg.eigh(z_trans_z)
    eigvals, eigvecs = evals.flip(0) - 1e-5, evecs.flip(1)           # descending, as an SVD returns them
    mask = eigvals >= 1e-7
    cut_point = max(int(mask.sum().item()), 1)
    eigvals[cut_point:] = 1e-7
    eigvecs[:, cut_point:] = 0
    return eigvals, eigvecs.T @ z_trans_y, float(stats[0].item()), int(round(stats[1].item()))


def generate_scoregrid(kernel, eigvals, eigvecs, lambda_, y_trans_y, ndatapoints):
    """lb_optimizer.py:120-170 -> numpy array of NMLL scores, one per lambda."""
    eigval_batch = eigvals[:, None] + lambda_[None, :] ** 2
    scoregrid = y_trans_y - eigvecs @ (eigvecs[:, None] / eigval_batch)
    scoregrid[scoregrid < 0] = 0
    scoregrid *= 0.5
    beta = torch.sqrt(2 * scoregrid / (ndatapoints * lambda_ ** 2)).clamp(min=0.1, max=10)
    scoregrid /= (beta * lambda_) ** 2
    scoregrid += 0.5 * torch.log(eigval_batch).sum(dim=0)
    scoregrid += (ndatapoints - kernel.get_num_rffs()) * torch.log(lambda_)
    scoregrid += ndatapoints * 0.5 * np.log(2 * np.pi) + ndatapoints * torch.log(beta)
    return scoregrid.cpu().numpy()


def shared_hparam_search(sigma_vals, kernel, dataset, init_bounds, n_pts_per_dim=100, n_cycles=1, subsample=1):
    """lb_optimizer.py:12-65: telescoping grid over lambda for fixed kernel-specific hyperparameters ->
    (score rounded to 3 decimals, log lambda rounded to 7)."""
    bounds = np.array(init_bounds, dtype=np.float64, copy=True)
    if np.exp(bounds[0, 0]) < 1e-3:
        bounds[0, 0] = np.log(1e-3)
    hparams = np.zeros((np.asarray(sigma_vals).shape[0] + 1))
    if hparams.shape[0] > 1:
        hparams[1:] = sigma_vals
    kernel.set_hyperparams(hparams, logspace=True)
    eigvals, eigvecs, y_trans_y, ndatapoints = get_eigvals(kernel, dataset, subsample=subsample)
    best_score, best_lb = np.inf, bounds[0, 0]
    for _ in range(n_cycles):
        lambda_, spacing = get_grid_pts(bounds, n_pts_per_dim, kernel.device)
        scoregrid = generate_scoregrid(kernel, eigvals, eigvecs, lambda_, y_trans_y, ndatapoints)
        min_pt = scoregrid.argmin()
        best_score, best_lb = scoregrid[min_pt], np.log(float(lambda_[min_pt].item()))
        bounds[0, 0] = max(best_lb - spacing, init_bounds[0, 0])
        bounds[0, 1] = min(best_lb + spacing, init_bounds[0, 1])
    return np.round(float(best_score), 3), np.round(np.asarray([best_lb]), 7)


def _sigma_grid_pts(num_pts_per_sigma, bounds):
    """bayes_grid.py:166-200."""
    if bounds.shape[0] == 2:
        return np.linspace(bounds[1, 0], bounds[1, 1], num_pts_per_sigma)
    if bounds.shape[0] == 3:
        s1 = np.linspace(bounds[1, 0], bounds[1, 1], num_pts_per_sigma)
        s2 = np.linspace(bounds[2, 0], bounds[2, 1], num_pts_per_sigma)
        s1, s2 = np.meshgrid(s1, s2)
        return np.array((s1.ravel(), s2.ravel())).T
    raise RuntimeError("This routine is only applicable for kernels with < 4 hyperparameters.")


def _random_starting_pts(num_sigma_vals, bounds, random_seed=123):
    """bayes_grid.py:143-163."""
    rng = np.random.default_rng(random_seed)
    sigma_grid = np.empty((num_sigma_vals, bounds.shape[0] - 1))
    for i in range(sigma_grid.shape[1]):
        sigma_grid[:, i] = rng.uniform(size=num_sigma_vals, low=bounds[i + 1, 0], high=bounds[i + 1, 1])
    return sigma_grid


def _propose_new_point(sigma_vals, scores, surrogate, bounds, random_seed, num_cand=500):
    """bayes_grid.py:103-140: refit the surrogate, Thompson-sample 15 draws over 500 candidates."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xvals = np.vstack(sigma_vals)
        surrogate.fit(xvals, scores)
    rng = np.random.default_rng(random_seed)
    candidates = np.round(rng.uniform(low=bounds[:, 0], high=bounds[:, 1], size=(num_cand, bounds.shape[0])), 7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y_candidates = surrogate.sample_y(candidates, n_samples=15, random_state=random_seed)
    best_idx = np.unravel_index(y_candidates.argmin(), y_candidates.shape)
    best_cand = candidates[best_idx[0], :]
    min_dist = np.min(np.linalg.norm(best_cand[None, :] - xvals, axis=1))
    return best_cand, min_dist, surrogate


def bayes_grid_tuning(kernel, dataset, bounds, random_seed, max_iter, verbose, tol=1e-1, n_pts_per_dim=100, n_cycles=1,
                      n_init_pts=10, subsample=1):
    """bayes_grid.py:12-100 -> (best hyperparameters (log), (sigma points, scores), best score, iterations)."""
    from sklearn.gaussian_process import GaussianProcessRegressor as GPR
    from sklearn.gaussian_process.kernels import RBF
    if bounds.shape[0] >= 4 or bounds.shape[0] < 2:
        raise RuntimeError("Bayesian optimization is only allowed for kernels with 2 - 3 hyperparameters.")
    sigma_grid = _sigma_grid_pts(n_init_pts, bounds) if bounds.shape[0] == 2 \
        else _random_starting_pts(n_init_pts, bounds, random_seed)
    sigma_grid = np.round(sigma_grid, 7)
    if len(sigma_grid.shape) == 1:
        sigma_grid = sigma_grid.reshape(-1, 1)
    sigma_grid = list(sigma_grid)
    lb_vals, scores = [], []
    for i, sigma_pt in enumerate(sigma_grid):
        score, lb_val = shared_hparam_search(sigma_pt, kernel, dataset, bounds[:1, :], n_pts_per_dim, n_cycles, subsample)
        scores.append(score)
        lb_vals.append(lb_val)
        if verbose:
            print(f"Grid point {i} acquired.")
    scores = np.asarray(scores)
    smallest_non_inf_val = np.max(scores[scores < np.inf])
    scores[scores == np.inf] = smallest_non_inf_val
    scores = scores.tolist()
    surrogate = GPR(kernel=RBF(), normalize_y=True, alpha=1e-6, random_state=random_seed, n_restarts_optimizer=4)
    sigma_bounds = bounds[1:, :]
    iternum = len(sigma_grid)
    for iternum in range(len(sigma_grid), max_iter):
        new_sigma, min_dist, surrogate = _propose_new_point(sigma_grid, scores, surrogate, sigma_bounds,
                                                            random_seed + iternum)
        if verbose:
            print(f"New hparams: {new_sigma}")
        score, lb_val = shared_hparam_search(new_sigma, kernel, dataset, bounds[:1, :], n_pts_per_dim, n_cycles, subsample)
        sigma_grid.append(new_sigma)
        lb_vals.append(lb_val)
        scores.append(min(score, smallest_non_inf_val))
        if min_dist < tol:
            break
    best_hparams = np.empty((bounds.shape[0]))
    best_hparams[1:] = sigma_grid[np.argmin(scores)]
    best_hparams[:1] = lb_vals[np.argmin(scores)]
    return best_hparams, (sigma_grid, scores), np.min(scores), iternum


def tune_hyperparams_crude(kernel, dataset, bounds=None, random_seed=123, max_bayes_iter=30, subsample=1, verbose=False):
    """xgp_regression.py:497-561 -> (hyperparams (log), n_feval, best_score); the kernel is left at the best point."""
    if subsample < 0.01 or subsample > 1:
        raise RuntimeError("subsample must be in the range [0.01, 1].")
    optim_bounds = default_bounds(kernel) if bounds is None else np.asarray(bounds, dtype=np.float64)
    num_hparams = kernel.get_hyperparams().shape[0]
    if num_hparams == 1:
        best_score, hyperparams = shared_hparam_search(np.array([]), kernel, dataset, optim_bounds, subsample=subsample)
        n_feval = 1
    elif 4 > num_hparams > 1:
        hyperparams, _, best_score, n_feval = bayes_grid_tuning(kernel, dataset, optim_bounds, random_seed,
                                                                max_bayes_iter, verbose, subsample=subsample)
    else:
        raise RuntimeError("The crude procedure is only appropriate for kernels with 1-3 hyperparameters.")
    kernel.set_hyperparams(hyperparams, logspace=True)
    return hyperparams, n_feval, best_score
