"""Preconditioned conjugate gradients on the device, mirroring the reference solver.

  * ``ConjugateGrad._matvec`` / ``.fit``  <-> CPU/GPU_ConjugateGrad (fitting_toolkit/cg_tools.py:173-302 / :26-156)
  * ``cg_fit_lib_internal``               <-> fitting_toolkit/cg_fitting_toolkit.py:18-70
  * ``calc_zty``                          <-> scoring_toolkit/exact_nmll_calcs.py:13-39

What differs from the reference, on purpose: the matvec never materialises Z when the kernel
has a fused HIP path (``kernel.ztz_matvec``: feature generation + Z^T(Zv) in one launch over
the whole HBM-resident shard), and the per-rank partial ``w`` is summed over ranks with one
RCCL all-reduce per iteration before ``lambda^2 v`` is added -- all ranks then run identical
float64 vector updates, so alpha / beta / err agree everywhere with no further communication.
Kept from the reference: float64 CG state, the two-column ping-pong buffers, solving for
``zty / N`` and rescaling, and the lagging ``err`` (computed from the current column after the
next one was written, cg_tools.py:265), so iteration counts match.
"""
import time
import warnings

import torch

from .dist import SINGLE


def calc_zty(dataset, kernel):
    """exact_nmll_calcs.py:13-39 -> (z^T y [M] f64 device, y^T y float), summed over ranks."""
    comm = dataset.comm
    m = kernel.get_num_rffs()
    z_trans_y = torch.zeros(m, dtype=torch.float64, device=kernel.device)
    if kernel.fused_ok():
        y = dataset.normalized_y()
        kernel.zty(dataset.scaled_x(kernel.hyperparams[1]), y, z_trans_y)
        y_trans_y = (y ** 2).sum().reshape(1)
    else:
        y_trans_y = torch.zeros(1, dtype=torch.float64, device=kernel.device)
        for xin, yin, ldata in dataset.get_chunked_data():
            zdata, ydata = kernel.transform_x_y(xin, yin, ldata)
            z_trans_y += zdata.T @ ydata
            y_trans_y += (ydata ** 2).sum()
    comm.all_reduce_(z_trans_y)
    comm.all_reduce_(y_trans_y)
    return z_trans_y, float(y_trans_y.item())


class ConjugateGrad:
    """Batched-RHS preconditioned CG for (Z^T Z + lambda^2) b = rhs."""

    def __init__(self, comm=SINGLE, cache_features=False):
        """``cache_features``: keep the shard's feature matrix resident in HBM as float32 and stream
        it each iteration instead of regenerating it (faster per iteration when it fits; off by
        default -- the reference regenerates, and so does the benchmark's headline number)."""
        self.comm = comm
        self.cache_features = cache_features
        self._ws = None
        self._ws_masks_of = None     # the radem tensor whose sign masks the workspace holds
        self._bws = None
        self._zwin = None
        self._err_pinned = None      # pinned host slots the step kernels write their errors into (reused across solves)

    def _pinned_errors(self, *shape):
        """A pinned float64 array of ``shape`` filled with -1 (the "not written yet" mark).  One allocation per solver
        object, grown in powers of two: hipHostMalloc takes from a fraction of a millisecond to a few milliseconds
        depending on the host, and a solve of 20 iterations that pays it inside its timed region loses 1-4 % to it."""
        need = 1
        for dim in shape:
            need *= int(dim)
        if self._err_pinned is None or self._err_pinned.numel() < need:
            cap = 1024
            while cap < need:
                cap *= 2
            self._err_pinned = torch.empty(cap, dtype=torch.float64).pin_memory()
        view = self._err_pinned[:need].view(*shape)
        view.fill_(-1.0)
        return view

    def _matvec(self, dataset, kernel, vec, matvec, add_ridge=True):
        """cg_tools.py:173-200 (regression branch): matvec <- (Z^T Z + lambda^2) vec (``add_ridge=False``: the
        caller's vector-update kernel adds lambda^2 vec)."""
        matvec.zero_()
        # k <= 2 right-hand sides: one fused pass per column (Z never written).  More columns (the
        # NMLL probes, k = 26): the two contractions [n x M][M x k], [M x n][n x k] run on the
        # float64 matrix cores over float32 feature rows -- the resident cache, or windows of rows
        # regenerated into scratch (hipZCacheBlockMatvec).
        if kernel.fused_ok() and vec.shape[1] <= 2:
            xs = dataset.scaled_x(kernel.hyperparams[1])
            if self._ws is None or self._ws.numel() < kernel.workspace_bytes() or self._ws.device != xs.device:
                self._ws = torch.empty(kernel.workspace_bytes(), dtype=torch.uint8, device=xs.device)
                self._ws_masks_of = None
            self._matvec_cols(kernel, xs, vec, matvec)
        elif self.BLOCK_KERNELS and vec.is_cuda and hasattr(kernel, "block_ok") and kernel.block_ok():
            self._matvec_block(dataset, kernel, vec, matvec)
        else:
            for x, lengths in dataset.get_chunked_x_data():
                z = kernel.transform_x(x, lengths)
                matvec += z.T @ (z @ vec)
        self.comm.all_reduce_(matvec)
        if add_ridge:
            matvec += kernel.get_lambda() ** 2 * vec

    BLOCK_KERNELS = True                # False: chunked float64 Z + library GEMMs (kept for timing comparisons)
    BLOCK_DEVICE_SOLVE = True           # False: the generic loop of torch operations for k > 1 (tests compare the two)
    BLOCK_WINDOW_BYTES = 8 << 30        # scratch for regenerated float32 feature rows (a window fills the GPU twice over)

    def _block_ws(self, nrows, kernel, k, dev):
        from .kernels import block_workspace_bytes
        need = block_workspace_bytes(nrows, kernel.get_num_rffs(), k)
        if self._bws is None or self._bws.numel() < need or self._bws.device != dev:
            self._bws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._bws

    def _matvec_block(self, dataset, kernel, vec, matvec):
        """matvec += Z^T (Z vec) for a block of right-hand sides (matvec is zero on entry)."""
        m, k = vec.shape
        if not vec.is_contiguous():
            vec = vec.contiguous()
        out = matvec if matvec.is_contiguous() else torch.zeros_like(vec)
        if self.cache_features and dataset.get_local_ndatapoints() > 0:
            zc = dataset.feature_cache(kernel)
            kernel.ztz_block_cached(zc, vec, out, self._block_ws(zc.shape[0], kernel, k, vec.device))
        elif kernel.fused_ok():
            xs = dataset.scaled_x(kernel.hyperparams[1])
            n = xs.shape[0]
            win = max(1024, min(n, self.BLOCK_WINDOW_BYTES // (4 * m)))
            if self._zwin is None or self._zwin.shape != (win, m) or self._zwin.device != vec.device:
                self._zwin = torch.empty((win, m), dtype=torch.float32, device=vec.device)
            for lo in range(0, n, win):
                hi = min(n, lo + win)
                zc = self._zwin[:hi - lo]
                kernel.fill_feature_cache(xs[lo:hi], zc)
                # (per window: a SHORTER last window can need a larger workspace -- the split projection's partials are reserved
                # for short launches only; _block_ws only grows)
                kernel.ztz_block_cached(zc, vec, out, self._block_ws(hi - lo, kernel, k, vec.device), accumulate=True)
        else:
            for x, lengths in dataset.get_chunked_x_data():
                zc = kernel.transform_x(x, lengths).to(torch.float32)
                kernel.ztz_block_cached(zc, vec, out, self._block_ws(zc.shape[0], kernel, k, vec.device),
                                        accumulate=True)
        if out is not matvec:
            matvec.copy_(out)

    def _matvec_cols(self, kernel, xs, vec, matvec):
        tmp = torch.empty(vec.shape[0], dtype=torch.float64, device=vec.device)
        for j in range(vec.shape[1]):
            kernel.ztz_matvec(xs, vec[:, j].contiguous(), tmp, self._ws)
            matvec[:, j] = tmp

    def _ztz(self, dataset, kernel, vec, out):
        """out <- sum over ranks of Z^T (Z vec) for one right-hand side (fused kernel +
        all-reduce); lambda^2 vec is added by the caller."""
        if self._ws is None or self._ws.numel() < kernel.workspace_bytes() or self._ws.device != out.device:
            self._ws = torch.empty(kernel.workspace_bytes(), dtype=torch.uint8, device=out.device)
            self._ws_masks_of = None
        if self._use_cache(kernel):
            kernel.ztz_matvec_cached(dataset.feature_cache(kernel), vec, out, self._ws)
        else:
            # the Rademacher sign masks are packed into the workspace by the first call only
            radem = getattr(kernel, "radem_diag", None)
            if radem is None:
                kernel.ztz_matvec(dataset.scaled_x(kernel.hyperparams[1]), vec, out, self._ws)
            else:
                kernel.ztz_matvec(dataset.scaled_x(kernel.hyperparams[1]), vec, out, self._ws,
                                  masks_packed=self._ws_masks_of is radem)
                self._ws_masks_of = radem
        self.comm.all_reduce_(out)

    def _use_cache(self, kernel):
        return self.cache_features and hasattr(kernel, "cache_ok") and kernel.cache_ok()

    def _fit_one_rhs_device(self, dataset, kernel, preconditioner, resid, maxiter, tol, verbose, trace):
        """The k = 1 regression solve on the device with the vector updates of
        cg_tools.py:255-274 fused into two small kernels (hipCGStep1 / hipCGStep2) and the
        preconditioner apply (hipPrecondApply) written as r + U ((inv_eig * prefactor - 1) .* (U^T r)).  Same
        recurrences, same lagging error, same iteration count as ``fit``; the host only reads one
        scalar (err) per iteration, and reads it *after* queueing the next matvec, so the device
        never waits for the host."""
        from . import xgpr_hip_rfgen_ext as ext
        dev = resid.device
        m = resid.shape[0]
        f64 = dict(dtype=torch.float64, device=dev)
        r = [resid[:, 0, 0].clone(), torch.empty(m, **f64)]
        z = [torch.empty(m, **f64), torch.empty(m, **f64)]
        p = [torch.empty(m, **f64), torch.empty(m, **f64)]
        x_k = torch.zeros(m, **f64)
        w = torch.zeros(m, **f64)
        scal = torch.zeros(4, **f64)
        init_norm = float(torch.linalg.norm(r[0]).item())
        lam2 = float(kernel.get_lambda()) ** 2
        if preconditioner is not None:
            u_mat, inv_eig, pref = preconditioner.u_mat, preconditioner.inv_eig, preconditioner.prefactor
            pws = torch.empty(ext.precond_workspace_bytes(u_mat.shape[1]), dtype=torch.uint8, device=dev)

        def precond(src, dst):
            if preconditioner is None:
                dst.copy_(src)
            else:
                ext.hipPrecondApply(u_mat, inv_eig, pref, src, dst, pws)

        precond(r[0], z[0])
        p[0].copy_(z[0])
        if trace is None and self._graph_ok(dataset, kernel):
            return self._replay_iterations(dataset, kernel, precond, r, z, p, x_k, w, lam2, init_norm, maxiter, tol,
                                           verbose)
        # errors arrive in pinned host memory, written by cg_step1_kernel itself (system-scope fence); the host
        # polls the slot instead of recording an event per iteration -- no copy command and no barrier packet
        # between one iteration's last kernel and the next one's first (~15 us per iteration on a small shard)
        err_host = self._pinned_errors(maxiter)
        err_np = err_host.numpy()
        losses, converged = [], False
        cur, nxt = 0, 1
        done = 0            # iterations fully queued
        checked = 0         # iterations whose error the host has read
        last_err = float("inf")

        def read_err(i):
            t0 = time.perf_counter()
            while err_np[i] < 0.0:                      # errors are >= 0 (NaN also ends the wait)
                if time.perf_counter() - t0 > 0.02:     # long matvec (or no coherent view): wait on the stream
                    torch.cuda.current_stream(dev).synchronize()
                    break
            if err_np[i] < 0.0:
                raise RuntimeError("CG error of iteration %d never reached the host" % i)
            return float(err_np[i])

        for niter in range(maxiter):
            # near convergence read the pending error first (no wasted matvec); otherwise queue
            # this iteration's matvec before looking at the previous error
            if checked < done and last_err < 100.0 * tol:
                last_err = read_err(checked); losses.append(last_err); checked += 1
                if last_err < tol:
                    converged = True
                    break
            self._ztz(dataset, kernel, p[cur], w)
            if checked < done:
                last_err = read_err(checked); losses.append(last_err); checked += 1
                if last_err < tol:
                    converged = True
                    break
            # the kernel writes this iteration's error straight into the pinned host array (no copy command)
            ext.hipCGStep1(w, p[cur], x_k, r[cur], r[nxt], z[cur], scal, lam2, init_norm, 0.0,
                           err_host[niter:niter + 1])
            precond(r[nxt], z[nxt])
            ext.hipCGStep2(r[nxt], z[nxt], p[cur], p[nxt], scal)
            done += 1
            if trace is not None:
                trace.setdefault("x_k", []).append(x_k.clone()[:, None])
                trace.setdefault("alpha", []).append(scal[1:2].clone())
                trace.setdefault("beta", []).append(scal[3:4].clone())
            cur, nxt = nxt, cur
            if niter % 5 == 0 and verbose and self.comm.rank == 0:
                print(f"{niter} iterations complete.")
        while checked < done:
            last_err = read_err(checked); losses.append(last_err); checked += 1
            if last_err < tol:
                converged = True
        return x_k, converged, done, losses

    def _fit_block_device(self, dataset, kernel, preconditioner, resid, maxiter, tol, verbose, nmll_settings):
        """The batched solve (k > 1 right-hand sides: the approximate NMLL's probes, cg_tools.py:121-141 / :255-274) with
        its vector updates in two kernels per iteration (hipCGStep1Block / hipCGStep2Block: one workgroup per column,
        alpha / beta / err written straight into [iterations, k] tables) and the preconditioner as two products,
        z = r + (U diag(inv_eig * prefactor - 1)) (U^T r).  Same recurrences, lagging error and iteration count as
        ``fit``'s generic loop, which issued ~30 small launches and one host synchronisation per iteration (0.5 ms --
        as much as the block matvec itself on a 32 768-row shard); the host reads the errors one iteration behind from
        pinned memory, as in the one-column solve."""
        from . import xgpr_hip_rfgen_ext as ext
        dev = resid.device
        m, _, k = resid.shape
        f64 = dict(dtype=torch.float64, device=dev)
        r = [resid[:, 0, :].contiguous(), torch.empty((m, k), **f64)]
        z = [torch.empty((m, k), **f64), torch.empty((m, k), **f64)]
        p = [torch.empty((m, k), **f64), torch.empty((m, k), **f64)]
        x_k = torch.zeros((m, k), **f64)
        w = torch.zeros((m, k), **f64)
        rz = torch.zeros(k, **f64)
        init_norms = torch.linalg.norm(r[0], dim=0).contiguous()
        alphas = torch.zeros((maxiter, k), **f64)
        betas = torch.zeros((maxiter, k), **f64)
        lam2 = float(kernel.get_lambda()) ** 2
        cws = torch.empty(ext.cg_block_workspace_bytes(m, k), dtype=torch.uint8, device=dev)
        if preconditioner is not None:
            u_mat, inv_eig, pref = preconditioner.u_mat, preconditioner.inv_eig, preconditioner.prefactor
            pws = torch.empty(ext.precond_apply_block_workspace_bytes(m, u_mat.shape[1], k), dtype=torch.uint8, device=dev)

        def precond(src, dst):
            if preconditioner is None:
                dst.copy_(src)
            else:
                ext.hipPrecondApplyBlock(u_mat, inv_eig, pref, src, dst, pws)

        precond(r[0], z[0])
        p[0].copy_(z[0])
        err_host = self._pinned_errors(maxiter, k)
        err_np = err_host.numpy()
        losses, converged = [], False
        cur, nxt = 0, 1
        done = checked = 0
        last_err = float("inf")

        def read_err(i):
            t0 = time.perf_counter()
            while (err_np[i] < 0.0).any():                  # errors are >= 0 (NaN also ends the wait)
                if time.perf_counter() - t0 > 0.05:         # long matvec (or no coherent view): wait on the stream
                    torch.cuda.current_stream(dev).synchronize()
                    break
            if (err_np[i] < 0.0).any():
                raise RuntimeError("CG errors of iteration %d never reached the host" % i)
            losses.append(float(err_np[i, 0]))
            return float(err_np[i].max()) if not bool((err_np[i] != err_np[i]).any()) else float("nan")

        for niter in range(maxiter):
            if checked < done and last_err < 100.0 * tol:
                last_err = read_err(checked); checked += 1
                if last_err < tol:
                    converged = True
                    break
            self._matvec(dataset, kernel, p[cur], w, add_ridge=False)
            if checked < done:
                last_err = read_err(checked); checked += 1
                if last_err < tol:
                    converged = True
                    break
            ext.hipCGStep1Block(w, p[cur], x_k, r[cur], r[nxt], z[cur], rz, alphas[niter], err_host[niter], init_norms, lam2, cws)
            precond(r[nxt], z[nxt])
            ext.hipCGStep2Block(r[nxt], z[nxt], p[cur], p[nxt], rz, betas[niter], cws)
            done += 1
            cur, nxt = nxt, cur
            if niter % 5 == 0 and verbose and self.comm.rank == 0:
                print(f"{niter} iterations complete.")
        while checked < done:
            last_err = read_err(checked); checked += 1
            if last_err < tol:
                converged = True
        if nmll_settings:
            return x_k, alphas[:done, 1:].clone(), betas[:done, 1:].clone()
        return x_k, converged, done, losses

    # ---- launch-bound solves: the iteration as a HIP graph (OFF by default -- measured slower, see below).
    # One CG iteration is 8-9 dependent kernels; on a small shard they take a few microseconds each and the
    # iteration costs ~70 us whatever the arithmetic.  Here the iteration (matvec, step 1, preconditioner,
    # step 2) is captured once per parity of the ping-pong buffers and replayed; the convergence test runs
    # on the device too (hipCGStep1 stop_tol), so an iteration queued before the host has seen the previous
    # error cannot move x past the iterate the host-checked loop returns (results are identical,
    # tests/test_gpu_cg.py).  Measured on MI355X / ROCm 7.2 (tools/bench_small_cg.py, 60 iterations):
    # N=2000, M=512: 77 us per iteration with plain launches, 86 us replayed; N=20000, M=1024 cached: 74 vs
    # 92; N=1e5, M=4096: 585 vs 600.  The cost is the dependent-dispatch latency on the device, which a
    # graph replay does not shorten on this stack, plus the replay call itself -- so plain launches stay
    # the default and the host never blocks the device anyway (the error is read one iteration behind).
    GRAPH_MAX_WORK = 1 << 31            # rows x features of the shard below which an iteration is launch-bound
    USE_GRAPHS = False

    def _graph_ok(self, dataset, kernel):
        return (self.USE_GRAPHS and self.comm.world_size == 1
                and dataset.get_local_ndatapoints() * kernel.get_num_rffs() <= self.GRAPH_MAX_WORK)

    def _replay_iterations(self, dataset, kernel, precond, r, z, p, x_k, w, lam2, init_norm, maxiter, tol, verbose):
        from . import xgpr_hip_rfgen_ext as ext
        dev = x_k.device
        scal = torch.zeros(8 + maxiter, dtype=torch.float64, device=dev)
        scal[2] = float("inf")

        def iteration(cur, nxt):
            self._ztz(dataset, kernel, p[cur], w)
            ext.hipCGStep1(w, p[cur], x_k, r[cur], r[nxt], z[cur], scal, lam2, init_norm, tol)
            precond(r[nxt], z[nxt])
            ext.hipCGStep2(r[nxt], z[nxt], p[cur], p[nxt], scal, tol)

        # everything lazy (scaled inputs / feature cache, workspaces, kernel attributes) happens in this
        # un-captured first iteration
        iteration(0, 1)
        graphs = {}
        if maxiter > 1:
            # capturing records, it does not execute: the two captures leave the state untouched
            # (capture_begin / capture_end on a side stream rather than the torch.cuda.graph context, whose
            # garbage collection and cache flush cost more than a short solve)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for cur in (1, 0):
                    graphs[cur] = torch.cuda.CUDAGraph()
                    graphs[cur].capture_begin()
                    iteration(cur, 1 - cur)
                    graphs[cur].capture_end()
            torch.cuda.current_stream(dev).wait_stream(side)
        err_host = torch.zeros(maxiter, dtype=torch.float64).pin_memory()
        events = []

        def queue_read(i):
            err_host[i:i + 1].copy_(scal[8 + i:9 + i], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            events.append(ev)

        queue_read(0)
        losses, converged = [], False
        queued, cur = 1, 1
        while True:
            # keep one iteration queued ahead of the error being read (the device stops itself)
            if queued < maxiter and queued <= len(losses) + 1:
                graphs[cur].replay()
                queue_read(queued)
                queued += 1
                cur = 1 - cur
                continue
            i = len(losses)
            if i >= queued:
                break
            events[i].synchronize()
            losses.append(float(err_host[i]))
            if verbose and i % 5 == 0 and self.comm.rank == 0:
                print(f"{i} iterations complete.")
            if losses[-1] < tol:
                converged = True
                break
        return x_k, converged, len(losses), losses

    def fit(self, dataset, kernel, preconditioner, resid, maxiter=200, tol=1e-4, verbose=True,
            nmll_settings=False, trace=None):
        """cg_tools.py:203-302.  ``resid`` is [M, 2, k] float64 with column 0 holding the
        right-hand side; starting weights are zero.  Returns (x_k, converged, niter, losses),
        or (x_k, alphas, betas) with ``nmll_settings``."""
        dev = resid.device
        if (resid.shape[2] == 1 and not nmll_settings and dev.type == "cuda"
                and (kernel.fused_ok() or self._use_cache(kernel))
                and (preconditioner is None or hasattr(preconditioner, "u_mat"))):
            return self._fit_one_rhs_device(dataset, kernel, preconditioner, resid, maxiter, tol, verbose, trace)
        if (self.BLOCK_DEVICE_SOLVE and 1 < resid.shape[2] <= 32 and dev.type == "cuda" and trace is None
                and (preconditioner is None or hasattr(preconditioner, "u_mat"))):
            return self._fit_block_device(dataset, kernel, preconditioner, resid, maxiter, tol, verbose, nmll_settings)
        converged = False
        target = resid[:, 0, :].clone()
        init_norms = torch.linalg.norm(target, dim=0)
        m, k = resid.shape[0], resid.shape[2]
        z_k = torch.zeros((m, 2, k), dtype=torch.float64, device=dev)
        p_k = torch.zeros((m, 2, k), dtype=torch.float64, device=dev)
        alphas, betas, losses = [], [], []
        x_k = torch.zeros((m, k), dtype=torch.float64, device=dev)
        w = torch.zeros((m, k), dtype=torch.float64, device=dev)

        if preconditioner is None:
            z_k[:, 0, :] = resid[:, 0, :]
        else:
            z_k[:, 0, :] = preconditioner.batch_matvec(resid[:, 0, :])
        p_k[:, 0, :] = z_k[:, 0, :]

        next_col, current_col = 1, 0
        niter = 0
        for niter in range(maxiter):
            p_cur = p_k[:, current_col, :].contiguous()
            self._matvec(dataset, kernel, p_cur, w)
            r_cur, z_cur = resid[:, current_col, :], z_k[:, current_col, :]
            rz = (r_cur * z_cur).sum(dim=0)
            alpha = rz / (p_cur * w).sum(dim=0)
            x_k += alpha[None, :] * p_cur
            resid[:, next_col, :] = r_cur - alpha[None, :] * w
            err = torch.linalg.norm(r_cur, dim=0) / init_norms
            r_next = resid[:, next_col, :]
            if preconditioner is None:
                z_k[:, next_col, :] = r_next
            else:
                z_k[:, next_col, :] = preconditioner.batch_matvec(r_next)
            beta = (r_next * z_k[:, next_col, :]).sum(dim=0) / rz
            p_k[:, next_col, :] = z_k[:, next_col, :] + beta[None, :] * p_cur

            err_host = err.cpu()          # the one host sync of the iteration
            if nmll_settings:
                alphas.append(alpha.clone())
                betas.append(beta.clone())
            else:
                losses.append(float(err_host[0]))
            if trace is not None:
                trace.setdefault("x_k", []).append(x_k.clone())
                trace.setdefault("alpha", []).append(alpha.clone())
                trace.setdefault("beta", []).append(beta.clone())

            next_col, current_col = abs(next_col - 1), abs(current_col - 1)
            if niter % 5 == 0 and verbose and self.comm.rank == 0:
                print(f"{niter} iterations complete.")
            if float(err_host.max()) < tol:
                converged = True
                break

        if nmll_settings:
            alphas, betas = torch.stack(alphas), torch.stack(betas)
            return x_k, alphas[:, 1:], betas[:, 1:]
        if x_k.shape[1] > 1:
            return x_k, converged, niter + 1, losses
        return x_k[:, 0], converged, niter + 1, losses


SMALL_SHARD_ROWS = 200_000


def _resolve_cache_mode(cache_features, kernel, dataset, block=False):
    """"auto": keep Z resident when the kernel supports it, when streaming it is faster than regenerating it
    (``kernel.cache_pays``; always for a block of right-hand sides) and the float32 cache of this shard fits
    comfortably in free HBM (it then needs cache + 2 GB with 1.5x headroom).  ``block``: the solve has
    a block of right-hand sides (matrix-core matvec), which caches under ``block_ok``."""
    if cache_features != "auto":
        return bool(cache_features)
    supported = getattr(kernel, "block_ok" if block else "cache_ok", None)
    if supported is None or not supported() or torch.device(kernel.device).type != "cuda":
        return False
    if not block and hasattr(kernel, "cache_pays") and not kernel.cache_pays():
        # One right-hand side on the single-pass three-wave kernel: regenerating is at least as fast as the stream for a
        # long shard (kernels.py) -- but the stream has the smaller cost per launch, and below ~250 000 rows it wins:
        # 0.746 against 0.776 ms per iteration on a 125 000-row shard (cfg3 over 8 GPUs), 1.401 / 1.400 at 250 000,
        # 2.731 / 2.636 at 500 000 (gpurun_out/r4/shard_*.json, `cached_z_mode`).  Only where that was measured: below padded
        # width 128 the three-wave kernel regenerates a tile in 1.0-1.2 ns against the stream's 1.32-1.39 on a 125 000-row shard
        # too (d = 16 / 32 / 64 at 8192 RFFs: 0.51 / 0.51 / 0.54 against 0.66 ms; profiles/r6_cache_rule_125k.json).
        from .kernels import padded_dims
        narrow = padded_dims(dataset.get_xdim()[-1]) < 128
        if dataset.get_local_ndatapoints() > SMALL_SHARD_ROWS or narrow:
            return False
    free, _total = torch.cuda.mem_get_info(torch.device(kernel.device))
    return 1.5 * dataset.feature_cache_bytes(kernel) + 2e9 < free


def cg_fit_lib_internal(kernel, dataset, cg_tol=1e-4, max_iter=500, preconditioner=None,
                        verbose=True, trace=None, cache_features="auto"):
    """cg_fitting_toolkit.py:18-70 -> (weights [M] f64 device, n_iter, losses).
    ``cache_features``: False = regenerate the features on every iteration (what the reference does);
    True = keep the shard's Z resident in HBM as float32 and stream it; "auto" (default) = resident
    when it fits.  The solve is the same either way (same iteration count, weights to ~1e-9)."""
    comm = dataset.comm
    cg_operator = ConjugateGrad(comm, _resolve_cache_mode(cache_features, kernel, dataset))
    resid = torch.zeros((kernel.get_num_rffs(), 2, 1), dtype=torch.float64, device=kernel.device)
    if preconditioner is None:
        z_trans_y, _ = calc_zty(dataset, kernel)
    else:
        z_trans_y = preconditioner.get_zty()
    resid[:, 0, :] = z_trans_y[:, None] / dataset.get_ndatapoints()
    weights, converged, n_iter, losses = cg_operator.fit(dataset, kernel, preconditioner, resid,
                                                         max_iter, cg_tol, verbose,
                                                         nmll_settings=False, trace=trace)
    weights *= dataset.get_ndatapoints()
    if not converged:
        warnings.warn("Conjugate gradients failed to converge! Try refitting "
                      "the model with updated settings.")
    if verbose and comm.rank == 0:
        print(f"CG iterations: {n_iter}")
    return weights, n_iter, losses
