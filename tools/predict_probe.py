"""predict on chunks of 2000 rows (the reference's chunking, xgp_regression.py:77-145 / xgp_classification.py:59-109):
features of a chunk -> float32 rows -> block projection.  XGPR_ZB_SPLIT=1 in the environment: the unsplit projection."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd.classification import predict_proba
from xgpr_amd.exact import predict_mean
dev = "cuda"
n, d, m, ncls = 64000, 256, 8192, 10
g = torch.Generator(device=dev).manual_seed(4)
x = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
kern = make_kernel("RBF", (n, d), m, 123, dev, {})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
wc = torch.randn(m, ncls, dtype=torch.float64, device=dev, generator=g)
gam = torch.zeros(ncls, dtype=torch.float64, device=dev)
w1 = torch.randn(m, dtype=torch.float64, device=dev, generator=g)
for name, fn in (("predict_proba (10 classes)", lambda: predict_proba(kern, wc, gam, x, chunk_size=2000)),
                 ("predict_mean", lambda: predict_mean(kern, w1, x, 0.0, 1.0, chunk_size=2000))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {n} rows in chunks of 2000: {dt * 1e3:.2f} ms = {dt / (n / 2000) * 1e6:.0f} us per chunk, checksum {float(out.sum()):.10e}")
