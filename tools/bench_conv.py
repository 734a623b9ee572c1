"""cudaConv1dFGen-equivalent at cfg4's shape (L = 512, C = 21, conv_width 9, 16384 RFFs):  python tools/bench_conv.py [nseq] [conv_width]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = int(sys.argv[2]) if len(sys.argv) > 2 else 9            # conv_width: 9 -> padded window 256 (cfg4), 24 -> 512, 48 -> 1024
L, C, m = 512, 21, 16384
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
x = torch.nn.functional.one_hot(torch.randint(0, C, (n, L), device=dev, generator=g), C).to(torch.float32)
sl = torch.randint(64, L + 1, (n,), generator=torch.Generator().manual_seed(5)).numpy().astype(np.int32)
kern = make_kernel("Conv1dRBF", (n, L, C), m, 123, dev, {"conv_width": w, "averaging": "sqrt"})
kern.set_hyperparams(np.array([1.0, 0.8]), logspace=False)
z = kern.transform_x(x, sl)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    z = kern.transform_x(x, sl)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"conv features (conv_width {w}): {n} sequences in {dt*1e3:.2f} ms = {n/dt:.3e} sequences/s, checksum {float(z.sum()):.10e}")
