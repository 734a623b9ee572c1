#!/usr/bin/env python3
"""The gradient operator (hipRBFGrad: features and d/dsigma, cudaRBFGrad's drop-in) on float32 and float64 input through whichever
library XGPR_HIP_LIB names: time per 32768 rows and checksums.
    python tools/grad_op_probe.py [d] [num_rffs]      (XGPR_F64_PLAN=generic: the any-width path for the shapes wave_tile.inc serves)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext
d = int(sys.argv[1]) if len(sys.argv) > 1 else 2003
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
n = 32768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
x32 = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
k = make_kernel("RBF", (n, d), m, 123, dev, {})
z = torch.empty(n, m, dtype=torch.float64, device=dev)
gr = torch.empty(n, m, 1, dtype=torch.float64, device=dev)
for name, x, chi in (("float32", x32, k.chi_arr), ("float64", x32.double(), k.chi_arr.double())):
    for _ in range(2):
        ext.hipRBFGrad(x, z, gr, k.radem_diag, chi, 0.7, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ext.hipRBFGrad(x, z, gr, k.radem_diag, chi, 0.7, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{os.environ.get('XGPR_F64_PLAN', 'wave tiles'):12s} {name} gradient operator d={d} M={m} {n} rows: {dt*1e3:.3f} ms  ({(x.element_size()*d+16.0*m)*n/dt/1e9:.0f} GB/s)  checksums {float(z.sum()):.15e} {float(gr.sum()):.15e}")
