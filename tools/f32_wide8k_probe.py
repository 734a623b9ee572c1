#!/usr/bin/env python3
"""The float32 feature operator at padded width 8192 (wave tiles, eight waves per transform: wave_tile.inc) through whichever library
XGPR_HIP_LIB names; XGPR_F64_PLAN=generic: the any-width path."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext
n = 32768
g = torch.Generator(device="cuda").manual_seed(3)
for d, m in ((5000, 8192), (8192, 16384)):
    x = torch.randn(n, d, device="cuda", generator=g) / d ** 0.5
    k = make_kernel("RBF", (n, d), m, 123, "cuda", {})
    z = torch.empty(n, m, dtype=torch.float64, device="cuda")
    for _ in range(2):
        ext.hipRBFFeatureGen(x, z, k.radem_diag, k.chi_arr, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ext.hipRBFFeatureGen(x, z, k.radem_diag, k.chi_arr, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{os.environ.get('XGPR_F64_PLAN', 'wave tiles'):12s} float32 feature operator d={d} M={m} {n} rows: {dt*1e3:.3f} ms  ({(4.0*d+8.0*m)*n/dt/1e9:.0f} GB/s)  checksum {float(z.sum()):.15e}")
