"""Block CG matvec (k = 26 right-hand sides, the approximate-NMLL solve) at BASELINE cfg3 shape on one GPU:
per-iteration time of (a) chunked float64 Z + library GEMMs, (b) regenerated float32 windows + the
float64-MFMA block kernels, (c) resident float32 cache + the block kernels.
Usage: python tools/bench_nmll.py [rows] [dim] [num_rffs] [k] [iters]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd.dataset import build_regression_dataset
from xgpr_amd.cg import ConjugateGrad

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
k = int(sys.argv[4]) if len(sys.argv) > 4 else 26
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(123)
x = torch.randn(rows, d, dtype=torch.float32, device=dev, generator=g) / d ** 0.5
y = torch.randn(rows, dtype=torch.float64, device=dev, generator=g)
ds = build_regression_dataset(x, y, chunk_size=16384, device=dev)
kern = make_kernel("Matern", (rows, d), m, 123, dev, {"matern_nu": 2.5})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
V = torch.randn(m, k, dtype=torch.float64, device=dev, generator=g)
ref = None
for name, blockk, cache in (("library GEMMs on chunked float64 Z", False, False),
                            ("MFMA block kernels, regenerated windows", True, False),
                            ("MFMA block kernels, resident cache", True, True)):
    cg = ConjugateGrad(cache_features=cache)
    cg.BLOCK_KERNELS = blockk
    W = torch.zeros_like(V)
    cg._matvec(ds, kern, V, W)          # warm-up (also builds the cache)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        cg._matvec(ds, kern, V, W)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    if ref is None:
        ref = W.clone()
    err = ((W - ref).norm() / ref.norm()).item()
    print(f"{name:45s} {dt*1e3:9.2f} ms/matvec  {4*rows*m*k/dt/1e12:6.1f} useful TFLOP/s  rel diff vs library {err:.2e}")
