"""Fused CG matvec (features regenerated) at the headline shape on a bounded number of rows:
    python tools/bench_fused.py [rows [d [num_rffs]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
xs = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
kern = make_kernel("Matern", (n, d), m, 123, dev, {"matern_nu": 2.5})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
v = torch.randn(m, dtype=torch.float64, device=dev, generator=g)
w = torch.empty_like(v)
ws = torch.empty(kern.workspace_bytes(), dtype=torch.uint8, device=dev)
for _ in range(3):
    kern.ztz_matvec(xs, v, w, ws)
best = 1e9
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        kern.ztz_matvec(xs, v, w, ws)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
print(f"fused matvec, {n} rows, d={d}, M={m}: {best:.3f} ms  ({n * m / best / 1e6:.1f} G features/s)  checksum {float(w.sum()):.12e}")
