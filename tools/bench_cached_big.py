"""Resident-cache CG matvec (default: cfg5 per-GPU share, num_freqs > 8192; `1000000 1024 8192` is the headline
shape): cache build time and per-matvec time.  usage: bench_cached_big.py [rows [d [num_rffs]]]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd.dataset import build_regression_dataset
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
ds = build_regression_dataset(x, y, chunk_size=8192, device=dev)
kern = make_kernel("RBF", (n, d), m, 123, dev, {})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
zc = ds.feature_cache(kern)
torch.cuda.synchronize(); print(f"cache build {time.perf_counter() - t0:.3f} s ({zc.numel() * 4 / 1e9:.1f} GB)")
v = torch.randn(m, dtype=torch.float64, device=dev, generator=g)
w = torch.empty_like(v)
ws = torch.empty(kern.workspace_bytes(), dtype=torch.uint8, device=dev)
xs = ds.scaled_x(1.0)
from xgpr_amd.kernels import block_workspace_bytes
from xgpr_amd import xgpr_hip_rfgen_ext as ext
bws = torch.empty(block_workspace_bytes(n, m, 1), dtype=torch.uint8, device=dev)
for name, fn in (("cached, single pass (two tiles per wave)", lambda: kern.ztz_matvec_cached(zc, v, w, ws)),
                 ("cached, block contractions with 1 column", lambda: ext.hipZCacheBlockMatvec(zc, v[:, None], w[:, None], kern.fit_intercept, bws)),
                 ("regenerating two-pass fused", lambda: kern.ztz_matvec(xs, v, w, ws))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name}: {ms:.2f} ms per matvec ({zc.numel() * 4 / ms / 1e9:.2f} TB/s of cache)" if "cached" in name else f"{name}: {ms:.2f} ms per matvec")
