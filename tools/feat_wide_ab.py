"""Feature rows at padded width 4096 (float64 operator in 32768-row windows, float32 cache rows) through the library XGPR_HIP_LIB names:
    python tools/feat_wide_ab.py          (A/B of the aliased row image for the feature modes: profiles/r6_xalias_ab.txt)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext
dev="cuda"
for d, m in ((4000, 8192), (4096, 16384), (3000, 4096)):
    n = 131072 if m <= 8192 else 65536
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
    k = make_kernel("RBF", (n, d), m, 123, dev, {})
    z = torch.empty(32768, m, dtype=torch.float64, device=dev)
    zc = torch.empty(n, m, dtype=torch.float32, device=dev)
    def f64():
        for lo in range(0, n, 32768):
            ext.hipRBFFeatureGen(x[lo:lo+32768], z, k.radem_diag, k.chi_arr, True)
    def f32():
        ext.hipRBFFeatureCache(x, zc, k.radem_diag, k.chi_arr)
    for name, fn in (("float64 rows", f64), ("float32 cache rows", f32)):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{os.environ.get('XGPR_HIP_LIB','current'):32s} d={d} M={m} n={n} {name}: {dt*1e3:.3f} ms  checksum {float(z.sum()) if name.startswith('float64') else float(zc.double().sum()):.10e}")
