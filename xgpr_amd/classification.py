"""Kernel classifier on the hot path: row 4 of SURVEY.md section 8f.

  * ``NonlinearCGClassification``  <-> fitting_toolkit/nonlinear_cg_toolkit.py:13-275
    (preconditioned nonlinear CG, Polak-Ribiere with restart, quadratic-interpolation /
    backtracking line search; ``cost_fun_classification`` :231-275)
  * ``fit_classifier``             <-> xgp_classification.py:111-200 (preconditioner supplied by caller)
  * ``predict_proba``              <-> xgp_classification.py:59-109

The cost function is where the data is touched: per chunk ``pred = Z @ W``, a row-wise softmax, and
``grad += Z^T (pred - onehot)``.  Both contractions are the two halves of the block matvec
(hipZCacheBlockProject / hipZCacheBlockBackproject: float64 MFMA over float32 feature rows); with
``cache_features`` the shard's feature rows stay resident in HBM across the many cost-function
evaluations of a fit, otherwise they are regenerated window by window.  The softmax on the [n, classes]
logits is elementwise torch code.  Partial sums (gradient, loss) are all-reduced over ranks.
"""
import numpy as np
import torch

from . import xgpr_hip_rfgen_ext as ext
from .cg import _resolve_cache_mode
from .kernels import block_workspace_bytes


class NonlinearCGClassification:
    WINDOW_BYTES = 8 << 30

    def __init__(self, dataset, kernel, verbose=False, preconditioner=None, cache_features="auto"):
        self.dataset, self.kernel = dataset, kernel
        self.lambda_ = float(kernel.get_lambda())
        self.verbose = verbose
        self.preconditioner = preconditioner
        self.n_iter = 0
        self.losses = []
        self.last_grad = None
        self.last_search_direction = None
        self.cache_features = _resolve_cache_mode(cache_features, kernel, dataset, block=True)
        self._ws = None
        self._zwin = None

    # ---- feature rows of the shard as float32 windows: (cache rows, labels, fit_intercept flag, scale)
    def _windows(self):
        kernel, ds = self.kernel, self.dataset
        labels = ds._ydata
        if self.cache_features:
            yield ds.feature_cache(kernel), labels
            return
        if kernel.fused_ok():
            xs = ds.scaled_x(kernel.hyperparams[1])
            n, m = xs.shape[0], kernel.get_num_rffs()
            win = max(1024, min(n, self.WINDOW_BYTES // (4 * m)))
            if self._zwin is None or self._zwin.shape != (win, m):
                self._zwin = torch.empty((win, m), dtype=torch.float32, device=xs.device)
            for lo in range(0, n, win):
                hi = min(n, lo + win)
                zc = self._zwin[:hi - lo]
                kernel.fill_feature_cache(xs[lo:hi], zc)
                yield zc, labels[lo:hi]
            return
        row = 0
        for x, lengths in ds.get_chunked_x_data():
            zc = kernel.transform_x(x, lengths).to(torch.float32)
            yield zc, labels[row:row + zc.shape[0]]
            row += zc.shape[0]

    def _block_args(self):
        """(fit_intercept, scale) the block operators need for this kernel's cache rows."""
        if getattr(self.kernel, "supports_fused", False):
            return self.kernel.fit_intercept, 0.0
        return False, 1.0          # convolution kernels cache complete feature rows

    def cost_fun_classification(self, wvec):
        """nonlinear_cg_toolkit.py:231-275 -> (grad [M, classes], loss)."""
        dev = wvec.device
        m, ncls = wvec.shape
        wvec = wvec.contiguous()
        grad = torch.zeros_like(wvec)
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        if not (wvec.is_cuda and hasattr(self.kernel, "block_ok") and self.kernel.block_ok()):
            # num_rffs not a multiple of 4: float64 feature chunks and library GEMMs, as the reference does
            for xd, yd, ld in self.dataset.get_chunked_data():
                z = self.kernel.transform_x(xd, ld)
                pred = z @ wvec
                pred -= pred.max(dim=1, keepdim=True).values
                pred = 2.71828 ** pred
                pred /= pred.sum(dim=1, keepdim=True)
                idx = yd.to(torch.int64)
                loss -= torch.log(pred.clamp(min=1e-16)).gather(1, idx[:, None]).sum()
                pred.scatter_add_(1, idx[:, None], torch.full((z.shape[0], 1), -1.0, dtype=torch.float64, device=dev))
                grad += z.T @ pred
            return self._finish_cost(grad, loss, wvec)
        icpt, scale = self._block_args()
        for zc, labels in self._windows():
            n = zc.shape[0]
            need = block_workspace_bytes(n, m, ncls)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            pred = torch.empty((n, ncls), dtype=torch.float64, device=dev)
            for j0 in range(0, ncls, 32):                     # 32 columns per call of the block operators
                j1 = min(ncls, j0 + 32)
                if j0 == 0 and j1 == ncls:
                    ext.hipZCacheBlockProject(zc, wvec, pred, icpt, scale)
                else:
                    part = torch.empty((n, j1 - j0), dtype=torch.float64, device=dev)
                    ext.hipZCacheBlockProject(zc, wvec[:, j0:j1].contiguous(), part, icpt, scale)
                    pred[:, j0:j1] = part
            pred -= pred.max(dim=1, keepdim=True).values
            pred = 2.71828 ** pred                            # the reference's constant, not e
            pred /= pred.sum(dim=1, keepdim=True)
            idx = labels.to(torch.int64)
            logpred = torch.log(pred.clamp(min=1e-16))
            loss -= logpred.gather(1, idx[:, None]).sum()
            pred.scatter_add_(1, idx[:, None], torch.full((n, 1), -1.0, dtype=torch.float64, device=dev))
            for j0 in range(0, ncls, 32):
                j1 = min(ncls, j0 + 32)
                if j0 == 0 and j1 == ncls:
                    ext.hipZCacheBlockBackproject(zc, pred, grad, icpt, self._ws, scale, accumulate=True)
                else:
                    g = grad[:, j0:j1].contiguous()
                    ext.hipZCacheBlockBackproject(zc, pred[:, j0:j1].contiguous(), g, icpt, self._ws, scale,
                                                  accumulate=True)
                    grad[:, j0:j1] = g
        return self._finish_cost(grad, loss, wvec)

    def _finish_cost(self, grad, loss, wvec):
        comm = self.dataset.comm
        comm.all_reduce_(grad)
        comm.all_reduce_(loss)
        grad[1:, :] += self.lambda_ ** 2 * wvec[1:, :]
        total = float(loss.item()) + 0.5 * self.lambda_ ** 2 * float((wvec ** 2)[1:, :].sum().item())
        if self.verbose and comm.rank == 0:
            print(f"        Func eval loss {total}", flush=True)
        return grad, total

    def fit_model(self, max_iter=500, tol=1e-4):
        """nonlinear_cg_toolkit.py:72-110."""
        wvec = torch.zeros((self.kernel.get_num_rffs(), self.dataset.get_n_classes()), dtype=torch.float64,
                           device=self.kernel.device)
        self.n_iter = 0
        grad, loss = self.cost_fun_classification(wvec)
        self.losses = [loss]
        last_alpha = None
        while self.n_iter < max_iter:
            grad, loss, wvec, _ = self.update_params(grad, wvec, loss, last_alpha, tol)
            self.losses.append(loss)
            if np.abs(np.abs(self.losses[-1] - self.losses[-2]) / self.losses[-2]) < tol:
                break
            self.n_iter += 1
            last_alpha = self.losses[self.n_iter - 1]
        return wvec, self.n_iter, self.losses

    def update_params(self, grad, wvec, loss, previous_loss, tol):
        """nonlinear_cg_toolkit.py:115-226."""
        if self.preconditioner is not None:
            search_direction = self.preconditioner.batch_matvec(grad)
        else:
            search_direction = grad      # the SAME array, as in the reference (:136): the in-place update
                                         # below then also changes ``grad`` in the un-preconditioned case
        if self.last_grad is not None:
            polak_ribiere = float((search_direction * (grad - self.last_grad)).sum().item())
            polak_ribiere /= float((self.last_grad * self.last_search_direction).sum().item())
            polak_ribiere = max(0., polak_ribiere)
            course_correction = polak_ribiere * self.last_search_direction
            self.last_grad = grad.clone()
            self.last_search_direction = search_direction.clone()
            search_direction += course_correction
        else:
            self.last_grad = grad.clone()
            self.last_search_direction = search_direction.clone()
        search_direction = -search_direction
        alpha0_prime = float((grad * search_direction).sum().item())
        if previous_loss is None:
            alpha_init = 1
        else:
            alpha_init = 2 * (loss - previous_loss) / alpha0_prime
        new_wvec = wvec + alpha_init * search_direction
        full_step_grad, full_step_loss = self.cost_fun_classification(new_wvec)
        if self.n_iter >= 10:
            if np.abs(np.abs(full_step_loss - loss) / loss) > tol:
                if full_step_loss < (loss + alpha_init * 1e-4 * alpha0_prime):
                    return full_step_grad, full_step_loss, new_wvec, alpha_init
        alpha_quad = -(alpha0_prime * alpha_init ** 2) / (2 * (full_step_loss - loss - alpha0_prime * alpha_init))
        quad_wvec = wvec + alpha_quad * search_direction
        quad_grad, quad_loss = self.cost_fun_classification(quad_wvec)
        if quad_loss < full_step_loss:
            if quad_loss < (loss + alpha_quad * 1e-4 * alpha0_prime):
                return quad_grad, quad_loss, quad_wvec, alpha_quad
        elif full_step_loss < (loss + alpha_init * 1e-4 * alpha0_prime):
            return full_step_grad, full_step_loss, new_wvec, alpha_init
        losses = [loss, full_step_loss, quad_loss]
        grads = [grad, full_step_grad, quad_grad]
        wvecs = [wvec, new_wvec, quad_wvec]
        alphas = [0, alpha_init, alpha_quad]
        alpha_max = alpha_init
        if quad_loss < full_step_loss:
            alpha_max = alpha_quad
        rfactor = 0.5
        for _ in range(10):
            alpha = rfactor * alpha_max
            candidate_wvec = wvec + alpha * search_direction
            candidate_grad, candidate_loss = self.cost_fun_classification(candidate_wvec)
            if candidate_loss < (loss + alpha * 1e-4 * alpha0_prime):
                return candidate_grad, candidate_loss, candidate_wvec, alpha
            losses.append(candidate_loss)
            grads.append(candidate_grad)
            wvecs.append(candidate_wvec)
            alphas.append(alpha)
            rfactor *= 0.5
        best_idx = int(np.argmin(losses))
        return grads[best_idx], losses[best_idx], wvecs[best_idx], alphas[best_idx]


def fit_classifier(kernel, dataset, preconditioner=None, tol=1e-3, max_iter=500, verbose=False,
                   cache_features="auto"):
    """xgp_classification.py:111-200 with the caller's preconditioner -> (weights [M, classes], gamma
    [classes] = 0, n_iter, losses)."""
    op = NonlinearCGClassification(dataset, kernel, verbose, preconditioner, cache_features)
    weights, n_iter, losses = op.fit_model(max_iter, tol)
    gamma = torch.zeros(dataset.get_n_classes(), dtype=torch.float64, device=kernel.device)
    return weights, gamma, n_iter, losses


def predict_proba(kernel, weights, gamma, input_x, sequence_lengths=None, chunk_size=2000):
    """xgp_classification.py:59-109: class probabilities, ``softmax(Z @ weights + gamma)`` chunk by chunk
    (same 2.71828 base as the reference)."""
    preds = []
    weights = weights.contiguous()
    icpt, scale = (kernel.fit_intercept, 0.0) if getattr(kernel, "supports_fused", False) else (False, 1.0)
    for i in range(0, input_x.shape[0], chunk_size):
        sl = None if sequence_lengths is None else sequence_lengths[i:i + chunk_size]
        if getattr(kernel, "supports_fused", False) and kernel.block_ok() and weights.shape[1] <= 32:
            from .kernels import scale_input
            xs = scale_input(kernel._as_device_f32(input_x[i:i + chunk_size]), kernel.hyperparams[1])
            zc = torch.empty((xs.shape[0], kernel.get_num_rffs()), dtype=torch.float32, device=kernel.device)
            kernel.fill_feature_cache(xs, zc)
            pred = torch.empty((xs.shape[0], weights.shape[1]), dtype=torch.float64, device=kernel.device)
            ext.hipZCacheBlockProject(zc, weights, pred, icpt, scale)
        else:
            pred = kernel.transform_x(input_x[i:i + chunk_size], sl) @ weights
        pred = pred + gamma[None, :]
        pred -= pred.max(dim=1, keepdim=True).values
        pred = 2.71828 ** pred
        pred /= pred.sum(dim=1, keepdim=True)
        preds.append(pred)
    return torch.cat(preds)
