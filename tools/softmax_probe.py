import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
dev = "cuda"
for n, ncls in ((262144, 10), (32768, 3), (1000000, 10)):
    g = torch.Generator(device=dev).manual_seed(1)
    pred0 = torch.randn(n, ncls, dtype=torch.float64, device=dev, generator=g)
    labels = torch.randint(0, ncls, (n,), device=dev, generator=g)
    def chain():
        pred = pred0.clone()
        pred -= pred.max(dim=1, keepdim=True).values
        pred = 2.71828 ** pred
        pred /= pred.sum(dim=1, keepdim=True)
        logpred = torch.log(pred.clamp(min=1e-16))
        loss = -logpred.gather(1, labels[:, None]).sum()
        pred.scatter_add_(1, labels[:, None], torch.full((n, 1), -1.0, dtype=torch.float64, device=dev))
        return loss
    def fused():
        pred = pred0.clone()
        return ext.hipSoftmaxResidual(pred, labels)
    def T(fn, reps=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / reps
    print(f"n={n} classes={ncls}: torch chain {T(chain):.1f} us, fused kernel {T(fused):.1f} us (both incl. a clone of pred)")
