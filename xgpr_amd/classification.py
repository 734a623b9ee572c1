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
import enum
from dataclasses import dataclass

import torch

from . import xgpr_hip_rfgen_ext as ext
from .cg import _resolve_cache_mode
from .kernels import block_workspace_bytes


class StepKind(enum.Enum):
    """Which rule of the line search produced the accepted step."""
    FULL = "full"                # the first trial step
    QUADRATIC = "quadratic"      # the minimiser of the interpolating parabola
    BACKTRACK = "backtrack"      # a halved step that met the sufficient-decrease test
    BEST_SEEN = "best_seen"      # nothing met it: the lowest loss evaluated on the ray


@dataclass
class _Trial:
    """An evaluated point on the search ray: step length, weights, gradient, loss."""
    step: float
    wvec: torch.Tensor
    grad: torch.Tensor
    loss: float


class NonlinearCGClassification:
    WINDOW_BYTES = 8 << 30

    def __init__(self, dataset, kernel, verbose=False, preconditioner=None, cache_features="auto"):
        self.dataset, self.kernel = dataset, kernel
        self.lambda_ = float(kernel.get_lambda())
        self.verbose = verbose
        self.preconditioner = preconditioner
        self.n_iter = 0
        self.losses = []
        self.step_kinds = []
        self._prev = None            # (gradient, preconditioned gradient) of the previous iteration
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
        # labels are validated where the dataset is built (zero category, global maximum = n_classes - 1); the weight
        # block must cover them: the fused softmax / residual kernel indexes classes by label (an uncovered label would
        # drop out of the loss; the kernel returns NaN for it, this is the message)
        if self.dataset.get_n_classes() > ncls:
            raise RuntimeError(f"labels run to {self.dataset.get_n_classes() - 1} but the weights have {ncls} class columns")
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
                    ext.hipZCacheBlockProject(zc, wvec, pred, icpt, scale, self._ws)
                else:
                    part = torch.empty((n, j1 - j0), dtype=torch.float64, device=dev)
                    ext.hipZCacheBlockProject(zc, wvec[:, j0:j1].contiguous(), part, icpt, scale, self._ws)
                    pred[:, j0:j1] = part
            # row maximum, 2.71828 ** (.) (the reference's constant, not e), normalisation, loss and residual: one launch
            loss += ext.hipSoftmaxResidual(pred, labels.to(torch.int64).contiguous())
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

    # ---- the optimiser.  Same arithmetic as nonlinear_cg_toolkit.py:73-226 (the loss sequence of the
    # reference's own classifier test reproduces, tests/golden/g11_classifier.npz), organised as: a search
    # direction object, a list of evaluated trial points, and a step-selection rule that names its outcome.
    def _evaluate(self, origin, direction, step):
        wvec = origin.wvec + step * direction
        grad, loss = self.cost_fun_classification(wvec)
        return _Trial(step, wvec, grad, loss)

    def _search_direction(self, grad):
        """Preconditioned Polak-Ribiere direction with restart (beta clipped at 0), nonlinear_cg_toolkit.py:131-152.
        Returns (descent direction d, vector s whose product with d is the slope used by the line search).
        The correction is beta times the previous PRECONDITIONED GRADIENT, not the previous direction, and
        without a preconditioner the reference's slope is -|d|^2 rather than g.d (its search direction is the
        gradient array itself, updated in place, :136/:148) -- both kept, they decide which steps are accepted."""
        pgrad = grad if self.preconditioner is None else self.preconditioner.batch_matvec(grad)
        combined = pgrad
        if self._prev is not None:
            prev_grad, prev_pgrad = self._prev
            beta = float((pgrad * (grad - prev_grad)).sum().item()) / float((prev_grad * prev_pgrad).sum().item())
            combined = pgrad + max(0.0, beta) * prev_pgrad
        self._prev = (grad.clone(), pgrad.clone())
        slope_vec = combined if self.preconditioner is None else grad
        return -combined, slope_vec

    def _line_search(self, origin, direction, slope, first_step, tol):
        """One accepted point along ``direction`` from ``origin`` -> (trial, StepKind); nonlinear_cg_toolkit.py:154-226.
        Sufficient decrease: loss(t) < loss(0) + 1e-4 t slope."""
        def decreases_enough(trial):
            return trial.loss < origin.loss + 1e-4 * trial.step * slope

        full = self._evaluate(origin, direction, first_step)
        # late in the fit a full step that still changes the loss is taken without the interpolation step
        if self.n_iter >= 10 and abs(abs(full.loss - origin.loss) / origin.loss) > tol and decreases_enough(full):
            return full, StepKind.FULL
        # minimiser of the parabola through loss(0), slope(0) and loss(first_step)
        quad_step = -(slope * first_step ** 2) / (2 * (full.loss - origin.loss - slope * first_step))
        quad = self._evaluate(origin, direction, quad_step)
        better, kind = (quad, StepKind.QUADRATIC) if quad.loss < full.loss else (full, StepKind.FULL)
        if decreases_enough(better):
            return better, kind
        # halve from the better of the two until the decrease is sufficient; after ten halvings settle for the
        # lowest loss seen anywhere on the ray, the starting point included
        seen = [origin, full, quad]
        for halvings in range(1, 11):
            trial = self._evaluate(origin, direction, better.step * 0.5 ** halvings)
            if decreases_enough(trial):
                return trial, StepKind.BACKTRACK
            seen.append(trial)
        return min(seen, key=lambda t: t.loss), StepKind.BEST_SEEN

    def fit_model(self, max_iter=500, tol=1e-4):
        """nonlinear_cg_toolkit.py:73-110 -> (weights [M, classes], iterations, loss after every step)."""
        wvec = torch.zeros((self.kernel.get_num_rffs(), self.dataset.get_n_classes()), dtype=torch.float64,
                           device=self.kernel.device)
        grad, loss = self.cost_fun_classification(wvec)
        point = _Trial(0.0, wvec, grad, loss)
        self.n_iter, self.losses, self.step_kinds, self._prev = 0, [loss], [], None
        while self.n_iter < max_iter:
            direction, slope_vec = self._search_direction(point.grad)
            slope = float((slope_vec * direction).sum().item())
            # first trial step: 1 at the start, afterwards the step that would repeat the previous decrease
            # (the reference takes the loss from before the previous step, nonlinear_cg_toolkit.py:107)
            first_step = 1 if self.n_iter == 0 else 2 * (point.loss - self.losses[self.n_iter - 1]) / slope
            point, kind = self._line_search(point, direction, slope, first_step, tol)
            point = _Trial(0.0, point.wvec, point.grad, point.loss)
            self.losses.append(point.loss)
            self.step_kinds.append(kind)
            if abs(abs(self.losses[-1] - self.losses[-2]) / self.losses[-2]) < tol:
                break
            self.n_iter += 1
        return point.wvec, self.n_iter, self.losses


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
