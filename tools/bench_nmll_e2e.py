"""approximate_nmll end to end at cfg3 shape (one GPU): preconditioner build, the 26-column CG solve (iterations,
time per iteration against the block matvec alone), the rest.   python tools/bench_nmll_e2e.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel, block_workspace_bytes
from xgpr_amd.dataset import build_regression_dataset
from xgpr_amd.preconditioner import RandNysPreconditioner
from xgpr_amd.nmll import approximate_nmll

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
d, m = 1024, 8192
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(123)
x = torch.randn(rows, d, device=dev, generator=g) / d ** 0.5
w = torch.randn(d, device=dev, generator=g)
y = (torch.sin(x @ w) + 0.1 * torch.randn(rows, device=dev, generator=g)).double()
ds = build_regression_dataset(x, y, chunk_size=16384, device=dev)
kern = make_kernel("Matern", (rows, d), m, 123, dev, {"matern_nu": 2.5})
kern.set_hyperparams(np.array([0.3, 1.0]), logspace=False)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, time.perf_counter() - t0


pre, t_pre = timed(lambda: RandNysPreconditioner(kern, ds, 512, False, 123, "srht"))
pre, t_pre = timed(lambda: RandNysPreconditioner(kern, ds, 512, False, 123, "srht"))
zc, t_cache = timed(lambda: ds.feature_cache(kern))
det = {}
for rep in range(2):
    det = {}
    val, t_nmll = timed(lambda: approximate_nmll(kern, ds, pre, None, 123, True, det))
k = 26
V = torch.randn(m, k, dtype=torch.float64, device=dev, generator=g)
W = torch.empty_like(V)
bws = torch.empty(block_workspace_bytes(rows, m, k), dtype=torch.uint8, device=dev)
kern.ztz_block_cached(zc, V, W, bws)
_, t_mv = timed(lambda: [kern.ztz_block_cached(zc, V, W, bws) for _ in range(5)])
t_mv /= 5
print(f"rows={rows}: preconditioner {t_pre*1e3:.0f} ms, feature cache {t_cache*1e3:.0f} ms, approximate_nmll {t_nmll*1e3:.0f} ms "
      f"for {det['niter']} iterations = {t_nmll/det['niter']*1e3:.2f} ms per iteration; block matvec alone {t_mv*1e3:.2f} ms; NMLL {val:.6e}")
