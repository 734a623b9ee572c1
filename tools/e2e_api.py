import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.dataset import build_regression_dataset
from xgpr_amd.models import xGPRegression
rng = np.random.default_rng(0)
n, d, m = 100000, 256, 4096
x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
y = np.sin(x @ rng.standard_normal(d)) + 0.1 * rng.standard_normal(n)
def T(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, time.perf_counter() - t0
data, t = T(lambda: build_regression_dataset(x, y, chunk_size=8192)); print(f"dataset {t:.3f} s")
model = xGPRegression(num_rffs=m, kernel_choice="RBF", variance_rffs=512, kernel_settings={"intercept": True})
_, t = T(lambda: model.set_hyperparams(np.log([0.1, 1.0]), data)); print(f"set_hyperparams {t:.3f} s")
for rep in range(2):
    _, t = T(lambda: model.fit(data, tol=1e-6)); print(f"fit (autoselected preconditioner) {t:.3f} s")
(pre, ratio), t = T(lambda: model.build_preconditioner(data, max_rank=512, method="srht")); print(f"build_preconditioner {t:.3f} s ratio {ratio:.3g}")
_, t = T(lambda: model.fit(data, preconditioner=pre, tol=1e-6)); print(f"fit with preconditioner {t:.3f} s")
xt = x[:50000]
for rep in range(2):
    (mean, var), t = T(lambda: model.predict(xt, get_var=True)); print(f"predict 50000 rows with variance {t:.3f} s")
    mean2, t = T(lambda: model.predict(xt, get_var=False)); print(f"predict 50000 rows mean only {t:.3f} s")
_, t = T(lambda: model.exact_nmll(np.log([0.1, 1.0]), data)) if False else (None, 0)
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    if "--auto" in sys.argv:
        model.fit(data, tol=1e-6)
    else:
        model.fit(data, preconditioner=pre, tol=1e-6)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
