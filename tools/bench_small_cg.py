"""Per-iteration wall time of the k = 1 CG solve on small problems (launch-bound regime).
    python tools/bench_small_cg.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd.dataset import build_regression_dataset
from xgpr_amd.preconditioner import RandNysPreconditioner
from xgpr_amd.cg import cg_fit_lib_internal, ConjugateGrad

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(123)
for n, d, m, rank in ((2000, 32, 512, 64), (20000, 64, 1024, 128), (100000, 256, 4096, 512)):
    x = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
    y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    ds = build_regression_dataset(x, y, chunk_size=8192, device=dev)
    kern = make_kernel("RBF", (n, d), m, 123, dev, {})
    kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
    pre = RandNysPreconditioner(kern, ds, rank, False, 123, "srht")
    for cache in (False, True):
        for graphs in (False, True):
            ConjugateGrad.USE_GRAPHS = graphs
            best, res = None, None
            for rep in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                w, niter, losses = cg_fit_lib_internal(kern, ds, 1e-10, 60, pre, False, cache_features=cache)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print(f"n={n} d={d} M={m} rank={rank} cache={cache} graphs={graphs}: {niter} iterations, "
                  f"{best * 1e3:.2f} ms per solve, {best / niter * 1e6:.1f} us per iteration, |w|={float(w.norm()):.12e}", flush=True)
