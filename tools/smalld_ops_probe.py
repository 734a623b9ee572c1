"""The stand-alone operators at small input widths (d = 8 .. 128): float64 feature operator, float32 cache rows, z^T y, the
two-pass matvec (num_freqs > 8192) -- which of them still run the register-only transform of the two-wave generation.
    python tools/smalld_ops_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
n = 131072


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for d in (8, 16, 32, 64, 128, 256, 1024):
    x = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
    for m in (8192, 32768):
        kern = make_kernel("RBF", (n, d), m, 123, dev, {})
        kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
        v = torch.randn(m, dtype=torch.float64, device=dev, generator=g)
        y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
        w = torch.empty_like(v)
        ws = torch.empty(kern.workspace_bytes(), dtype=torch.uint8, device=dev)
        out = {}
        out["matvec_ms"] = timed(lambda: kern.ztz_matvec(x, v, w, ws))
        out["zty_ms"] = timed(lambda: kern.zty(x, y, w, ws))
        zc = torch.empty((n, m), dtype=torch.float32, device=dev)
        out["cache_rows_ms"] = timed(lambda: ext.hipRBFFeatureCache(x, zc, kern.radem_diag, kern.chi_arr))
        del zc
        if m == 8192:
            z = torch.empty((n, m), dtype=torch.float64, device=dev)
            out["feature_op_ms"] = timed(lambda: ext.hipRBFFeatureGen(x, z, kern.radem_diag, kern.chi_arr, True))
            del z
        print(f"d={d:5d} M={m:6d} " + " ".join(f"{k}={v_:.3f}" for k, v_ in out.items()), flush=True)
