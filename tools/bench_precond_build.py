"""Whole preconditioner build (rand_nys_constructors.py) timed at a bounded size, with the kernel breakdown
left to rocprofv3:  python tools/bench_precond_build.py [rows] [dim] [num_rffs] [rank] [method]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd.dataset import build_regression_dataset
from xgpr_amd.preconditioner import RandNysPreconditioner

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 512
method = sys.argv[5] if len(sys.argv) > 5 else "srht"
chunk = int(sys.argv[6]) if len(sys.argv) > 6 else 16384
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(123)
x = torch.randn(rows, d, dtype=torch.float32, device=dev, generator=g) / d ** 0.5
y = torch.randn(rows, dtype=torch.float64, device=dev, generator=g)
ds = build_regression_dataset(x, y, chunk_size=chunk, device=dev)
kern = make_kernel("RBF", (rows, d), m, 123, dev, {})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pre = RandNysPreconditioner(kern, ds, rank, False, 123, method)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    passes = 1 if method == "srht" else int(method.split("_")[1])
    fl = 2.0 * rows * rank * m * (1 if method == "srht" else 1 + 2 * (passes - 1))
    print(f"rows={rows} d={d} M={m} rank={rank} {method}: {dt*1e3:.1f} ms  ({dt/rows*1e6:.3f} us/row, GEMM work alone {fl/dt/1e12:.1f} TFLOP/s) ratio={pre.achieved_ratio:.6g} "
          f"checksums eig {float(pre.inv_eig.sum()):.15e} |U| {float(pre.u_mat.abs().sum()):.15e} zty {float(pre.get_zty().sum()):.15e}  "
          f"(XGPR_PRECOND_PIPELINE={os.environ.get('XGPR_PRECOND_PIPELINE', '1')})")
