#!/usr/bin/env python3
"""The float64 overload of the feature operator (hipRBFFeatureGen on float64 input: double_precision=True kernels,
kernel_baseclass.py:278-285) through whichever library XGPR_HIP_LIB names: time per 32768 rows and a checksum.
    python tools/f64_op_probe.py [d] [num_rffs]          (A/B of the double-precision cos/sin forms: tools/ablate_build.sh f64sep / f64nosc)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext
d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n = 32768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
x = (torch.randn(n, d, device=dev, generator=g) / d ** 0.5).double()
k = make_kernel("RBF", (n, d), m, 123, dev, {})
chi = k.chi_arr.double()
z = torch.empty(n, m, dtype=torch.float64, device=dev)
for _ in range(2):
    ext.hipRBFFeatureGen(x, z, k.radem_diag, chi, True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    ext.hipRBFFeatureGen(x, z, k.radem_diag, chi, True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"{os.environ.get('XGPR_HIP_LIB', 'current'):40s} float64 operator d={d} M={m} {n} rows: {dt*1e3:.3f} ms  ({(8.0*d+8.0*m)*n/dt/1e9:.0f} GB/s)  checksum {float(z.sum()):.15e}")
