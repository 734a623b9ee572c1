"""End-to-end runs of the BASELINE configs that are not the bench headline, at the per-GPU share of their
8-GPU size (one GPU here): preconditioner build + CG fit to tolerance, with a timing breakdown.
    python tools/run_configs.py cfg2|cfg4|cfg5 [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd.dataset import build_regression_dataset
from xgpr_amd.preconditioner import RandNysPreconditioner
from xgpr_amd.cg import cg_fit_lib_internal

dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
g = torch.Generator(device=dev).manual_seed(123)


def sync_time(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, time.perf_counter() - t0


if which == "cfg2":      # RBF, N=1e5 d=256, 4096 RFFs, single GPU
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    d, m, rank, method, chunk = 256, 4096, 512, "srht", 8192
    x = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
    sl = None
    kern = make_kernel("RBF", (n, d), m, 123, dev, {})
elif which == "cfg4":    # Conv1d RBF, L<=512, 21 channels one-hot, 16384 RFFs; N=5e5 over 8 GPUs -> 62500 per GPU
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 62500
    L, C, m, rank, method, chunk = 512, 21, 16384, 512, "srht", 1024
    idx = torch.randint(0, C, (n, L), device=dev, generator=g)
    x = torch.nn.functional.one_hot(idx, C).to(torch.float32)
    sl = torch.randint(64, L + 1, (n,), generator=torch.Generator().manual_seed(5)).numpy().astype(np.int32)
    kern = make_kernel("Conv1dRBF", (n, L, C), m, 123, dev, {"conv_width": 9, "averaging": "sqrt"})
else:                    # cfg5: RBF, d=512, 32768 RFFs, rank-2048 srht_2; N=2e6 over 8 GPUs -> 250000 per GPU
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
    d, m, rank, method, chunk = 512, 32768, 2048, "srht_2", 8192
    x = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
    sl = None
    kern = make_kernel("RBF", (n, d), m, 123, dev, {})
y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
ds = build_regression_dataset(x, y, sl, chunk_size=chunk, device=dev)
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
# one small untimed build first: the first GEMM / eigensolver / Cholesky calls of a process pay ~0.4 s of library
# initialisation that is not part of a build
nw = min(n, 4096)
ds_w = build_regression_dataset(x[:nw], y[:nw], None if sl is None else sl[:nw], chunk_size=chunk, device=dev)
pre_w = RandNysPreconditioner(kern, ds_w, min(rank, 256), False, 123, method)
for cache in (False, True) if which != "cfg4" else (True,):      # first launches of the fit's kernels, untimed as well
    cg_fit_lib_internal(kern, ds_w, 1e-6, 5, pre_w, False, cache_features=cache)
del ds_w, pre_w
if hasattr(ds, "_zcache"):
    ds._zcache = None
pre, t_pre = sync_time(lambda: RandNysPreconditioner(kern, ds, rank, False, 123, method))
print(f"{which}: n={n} M={m} rank={rank} {method}: preconditioner build {t_pre:.3f} s warm (achieved ratio {pre.achieved_ratio:.3g})", flush=True)
for cache in (False, True) if which != "cfg4" else (True,):
    t_cache = 0.0
    if cache:
        torch.cuda.empty_cache()          # the build's scratch goes back to the driver before the big allocation
        _, t_cache = sync_time(lambda: ds.feature_cache(kern))
    (w, niter, losses), t_fit = sync_time(lambda: cg_fit_lib_internal(kern, ds, 1e-6, 200, pre, False, cache_features=cache))
    print(f"   CG fit cache_features={cache}: {niter} iterations in {t_fit:.2f} s ({t_fit/niter*1e3:.1f} ms/iteration)"
          + (f" + feature cache {t_cache:.2f} s" if cache else " (features regenerated every iteration)")
          + f", final err {losses[-1]:.2e}", flush=True)
