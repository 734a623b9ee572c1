"""Where a warm preconditioner build's time goes (round 4): the phases of initialize_srht timed one by one with
synchronisation between them (cfg3 shape).   python tools/build_phases.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd.dataset import build_regression_dataset
from xgpr_amd import preconditioner as xp
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d, m, rank = 1024, 8192, 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(123)
x = torch.randn(rows, d, dtype=torch.float32, device=dev, generator=g) / d ** 0.5
y = torch.randn(rows, dtype=torch.float64, device=dev, generator=g)
ds = build_regression_dataset(x, y, chunk_size=16384, device=dev)
kern = make_kernel("RBF", (rows, d), m, 123, dev, {})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
xp.RandNysPreconditioner(kern, ds, rank, False, 123, "srht")        # warm
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, 1e3 * (time.perf_counter() - t0)
for rep in range(2):
    (acc, zty, yty, comp), t1 = T(lambda: xp._first_pass(ds, rank, kern, 123, False, True, False))
    c_mat, t2 = T(lambda: comp.transform_x(acc))
    acc_t = acc.T
    c_sym = 0.5 * (c_mat + c_mat.T)
    (chol, info), t3 = T(lambda: torch.linalg.cholesky_ex(c_sym))
    cond, t4 = T(lambda: xp._chol_cond_estimate(chol))
    b, t5 = T(lambda: torch.linalg.solve_triangular(chol, acc_t.T, upper=False).T)
    gram, t6 = T(lambda: b.T @ b)
    (ev, evec), t7 = T(lambda: torch.linalg.eigh(gram))
    u, t8 = T(lambda: (b @ evec.flip(1)) / torch.sqrt(ev.flip(0))[None, :])
    whole, t9 = T(lambda: xp.RandNysPreconditioner(kern, ds, rank, False, 123, "srht"))
    print(f"first pass {t1:.1f} | SRHT(acc) {t2:.2f} | cholesky {t3:.2f} | cond estimate {t4:.2f} | trsm {t5:.2f} | gram {t6:.2f} | eigh {t7:.2f} | U {t8:.2f} | sum {t1+t2+t3+t4+t5+t6+t7+t8:.1f} | whole build {t9:.1f} ms")
