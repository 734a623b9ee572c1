#!/usr/bin/env python3
"""One launch of the fused matvec at a given shape (for the SQ counter passes of the three layouts):
    rocprofv3 --pmc ... --kernel-trace --output-format csv -d <dir> -- python tools/pmc_probe_shapes.py rows d num_rffs"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgpr_amd.kernels import make_kernel   # noqa: E402
n, d, m = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda", 0)
x = torch.randn(n, d, device=dev) / np.sqrt(d)
k = make_kernel("RBF", (n, d), m, 123, dev, {})
k.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
v = torch.randn(m, dtype=torch.float64, device=dev)
out = torch.zeros(m, dtype=torch.float64, device=dev)
ws = torch.empty(k.workspace_bytes(), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
k.ztz_matvec(x, v, out, ws)
k.ztz_matvec(x, v, out, ws)
torch.cuda.synchronize()
