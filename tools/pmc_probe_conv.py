"""One launch of the convolution feature operator at BASELINE cfg4 shape (for rocprofv3 --pmc passes)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgpr_amd.kernels import make_kernel
dev = "cuda"
n, L, C, m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 512, 21, 16384
g = torch.Generator(device=dev).manual_seed(3)
idx = torch.randint(0, C, (n, L), device=dev, generator=g)
x = torch.nn.functional.one_hot(idx, C).to(torch.float32)
sl = torch.randint(64, L + 1, (n,), generator=torch.Generator().manual_seed(5)).numpy().astype(np.int32)
kern = make_kernel("Conv1dRBF", (n, L, C), m, 123, dev, {"conv_width": 9, "averaging": "sqrt"})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
import time
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z = kern.transform_x(x, sl)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"conv features: {n} sequences in {dt*1e3:.1f} ms = {n/dt:.3e} sequences/s; k-mers {int((sl - 8).sum())}")
