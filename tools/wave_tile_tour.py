#!/usr/bin/env python3
"""One call of every operator / shape family the wave-tile kernels of wave_tile.inc serve (for a rocprofv3 --kernel-trace --stats summary:
profiles/r6_wave_tile_kernel_stats.csv): float64 features and gradient at d = 9 / 1024 / 2003 / 4000 / 5000, float32 gradient and
features at the widths they take there, the convolution operators on float64 input and on float32 windows of 2048 / 4096 elements."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
n = 32768
for d, m in ((9, 8192), (1024, 8192), (2003, 4000), (4000, 8192), (5000, 8192)):
    k = make_kernel("RBF", (n, d), m, 123, dev, {})
    x = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
    z = torch.empty(n, m, dtype=torch.float64, device=dev)
    gr = torch.empty(n, m, 1, dtype=torch.float64, device=dev)
    for _ in range(3):
        ext.hipRBFFeatureGen(x.double(), z, k.radem_diag, k.chi_arr.double(), True)
        ext.hipRBFGrad(x.double(), z, gr, k.radem_diag, k.chi_arr.double(), 1.0, True)
        ext.hipRBFGrad(x, z, gr, k.radem_diag, k.chi_arr, 1.0, True)
        ext.hipRBFFeatureGen(x, z, k.radem_diag, k.chi_arr, True)
    del x, z, gr
rng = np.random.default_rng(5)
for dt, nseq, L, C, cw, m in ((torch.float64, 1024, 512, 21, 9, 16384), (torch.float32, 512, 256, 128, 9, 8192), (torch.float32, 512, 128, 64, 40, 8192),
                              (torch.float64, 256, 256, 128, 9, 8192)):
    P = 1 << int(np.ceil(np.log2(cw * C)))
    F = m // 2
    R = -(-F // P) * P
    radem = torch.from_numpy(rng.choice(np.array([-1, 1], dtype=np.int8), size=(3, 1, R))).to(dev)
    chi = (torch.rand(F, device=dev, dtype=torch.float64, generator=g) + 0.5).to(dt)
    x = torch.randn(nseq, L, C, device=dev, dtype=torch.float64, generator=g).to(dt)
    sl = rng.integers(max(cw, L // 8), L + 1, size=nseq).astype(np.int32)
    out = torch.zeros(nseq, m, dtype=torch.float64, device=dev)
    grad = torch.zeros(nseq, m, 1, dtype=torch.float64, device=dev)
    for _ in range(3):
        ext.hipConv1dFGen(x, out, radem, chi, sl, cw, 1)
        ext.hipConvGrad(x, out, radem, chi, sl, grad, 0.9, cw, 1)
torch.cuda.synchronize()
print("done")
