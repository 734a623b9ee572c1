"""One evaluation of the classifier's cost function (nonlinear_cg_toolkit.py:231-275) against its two block contractions
alone (round 4).   python tools/classifier_probe.py [rows] [classes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel, block_workspace_bytes
from xgpr_amd.dataset import build_classification_dataset
from xgpr_amd.classification import NonlinearCGClassification
from xgpr_amd import xgpr_hip_rfgen_ext as ext
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ncls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d, m = 256, 8192
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(rows, d, device=dev, generator=g) / d ** 0.5
y = torch.randint(0, ncls, (rows,), device=dev, generator=g)
ds = build_classification_dataset(x, y, chunk_size=16384, device=dev)
kern = make_kernel("RBF", (rows, d), m, 123, dev, {})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
op = NonlinearCGClassification(ds, kern, False, None, cache_features=True)
w = 0.01 * torch.randn(m, ncls, dtype=torch.float64, device=dev, generator=g)
def T(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps
t_cost = T(lambda: op.cost_fun_classification(w))
zc = ds.feature_cache(kern)
pred = torch.empty(rows, ncls, dtype=torch.float64, device=dev)
grad = torch.zeros(m, ncls, dtype=torch.float64, device=dev)
ws = torch.empty(block_workspace_bytes(rows, m, ncls), dtype=torch.uint8, device=dev)
t_proj = T(lambda: ext.hipZCacheBlockProject(zc, w, pred, True, 0.0))
t_back = T(lambda: ext.hipZCacheBlockBackproject(zc, pred, grad, True, ws, 0.0, accumulate=True))
print(f"rows={rows} classes={ncls}: cost function {t_cost:.2f} ms; project {t_proj:.2f} + backproject {t_back:.2f} = {t_proj+t_back:.2f} ms; the rest {t_cost-t_proj-t_back:.2f} ms")
