#!/usr/bin/env python3
"""Launches, once each and on > 256 MiB of data, (1) a calibration kernel with a known byte count and
the same dword-per-lane access pattern (the generic FHT over [n, 1024] float32: reads and writes
4 KiB per row), (2) the fused ZtZ matvec, (3) the stand-alone feature-generation operator.  Run under
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python tools/pmc_probe.py
and again with --pmc WRITE_SIZE (the TCC counters do not fit one pass, MI355X_MICROARCH.md)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgpr_amd.kernels import make_kernel   # noqa: E402
from xgpr_amd import xgpr_hip_rfgen_ext as ext   # noqa: E402

dev = torch.device("cuda", 0)
n, d, m = 262144, 1024, 8192
x = torch.randn(n, d, device=dev) / np.sqrt(d)
k = make_kernel("Matern", (n, d), m, 123, dev, {"matern_nu": 2.5})
v = torch.randn(m, dtype=torch.float64, device=dev)
out = torch.zeros(m, dtype=torch.float64, device=dev)
ws = torch.empty(k.workspace_bytes(), dtype=torch.uint8, device=dev)
z = torch.empty(16384, m, dtype=torch.float64, device=dev)
xc = x.clone()
torch.cuda.synchronize()
ext.hipFastHadamardTransform2D(xc)                       # calibration: 1 GiB read + 1 GiB written
k.ztz_matvec(x, v, out, ws)                              # algorithmic: 1 GiB read (X)
ext.hipRBFFeatureGen(x[:16384], z, k.radem_diag, k.chi_arr, True)   # 64 MiB read + 1 GiB written
torch.cuda.synchronize()
print("probe done: n =", n)
