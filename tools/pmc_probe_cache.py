#!/usr/bin/env python3
"""One launch of the resident-Z streaming matvec over 8 GiB of cache (262144 x 8192 float32), for
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext

dev = torch.device("cuda", 0)
n, d, m = 262144, 1024, 8192
x = torch.randn(n, d, device=dev) / np.sqrt(d)
k = make_kernel("Matern", (n, d), m, 123, dev, {"matern_nu": 2.5})
zc = torch.empty((n, m), dtype=torch.float32, device=dev)
ext.hipRBFFeatureCache(x, zc, k.radem_diag, k.chi_arr)
v = torch.randn(m, dtype=torch.float64, device=dev)
out = torch.zeros(m, dtype=torch.float64, device=dev)
ws = torch.empty(k.workspace_bytes(), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
k.ztz_matvec_cached(zc, v, out, ws)
torch.cuda.synchronize()
print("probe done")
