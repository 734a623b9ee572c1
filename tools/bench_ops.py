#!/usr/bin/env python3
"""Quick per-operator timings on one MI355X (development aid; bench.py is the contract benchmark).
    python tools/bench_ops.py [--rows 262144]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgpr_amd.kernels import make_kernel   # noqa: E402
from xgpr_amd import xgpr_hip_rfgen_ext as ext   # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=262144)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for (d, m, name, parms) in [(1024, 8192, "Matern", {"matern_nu": 2.5}), (256, 4096, "RBF", {}),
                                (512, 16384, "RBF", {}), (512, 32768, "RBF", {})]:
        n = args.rows
        x = torch.randn(n, d, device=dev) / np.sqrt(d)
        k = make_kernel(name, (n, d), m, 123, dev, parms)
        v = torch.randn(m, dtype=torch.float64, device=dev)
        out = torch.zeros(m, dtype=torch.float64, device=dev)
        ws = torch.empty(k.workspace_bytes(), dtype=torch.uint8, device=dev)
        ms = timeit(lambda: k.ztz_matvec(x, v, out, ws))
        print(f"fused ZtZ matvec  d={d:5d} M={m:6d} n={n}: {ms:8.3f} ms  {n / ms / 1e3:8.1f} Mrows/s  "
              f"{n * m / ms / 1e6:8.1f} Gfeat/s  X-read {4 * d * n / ms / 1e6:7.1f} GB/s")
        if k.cache_ok():
            from xgpr_amd import xgpr_hip_rfgen_ext as _e
            zc = torch.empty((n, m), dtype=torch.float32, device=dev)
            ms = timeit(lambda: _e.hipRBFFeatureCache(x, zc, k.radem_diag, k.chi_arr), reps=2, warm=1)
            print(f"feature cache     d={d:5d} M={m:6d} n={n}: {ms:8.3f} ms  {n / ms / 1e3:8.1f} Mrows/s  "
                  f"HBM {(4 * d + 4 * m) * n / ms / 1e6:7.1f} GB/s")
            ms = timeit(lambda: k.ztz_matvec_cached(zc, v, out, ws))
            print(f"cached ZtZ matvec d={d:5d} M={m:6d} n={n}: {ms:8.3f} ms  {n / ms / 1e3:8.1f} Mrows/s  "
                  f"HBM {4 * m * n / ms / 1e6:7.1f} GB/s")
            o2 = torch.zeros_like(out)
            k.ztz_matvec(x, v, o2, ws)
            k.ztz_matvec_cached(zc, v, out, ws)
            print(f"   cached vs fused max rel diff {float((out - o2).abs().max() / o2.abs().max()):.2e}")
            del zc
        fr = min(n, (1 << 30) // (8 * m))
        z = torch.empty(fr, m, dtype=torch.float64, device=dev)
        ms = timeit(lambda: ext.hipRBFFeatureGen(x[:fr], z, k.radem_diag, k.chi_arr, True))
        print(f"featgen op        d={d:5d} M={m:6d} n={fr}: {ms:8.3f} ms  {fr / ms / 1e3:8.1f} Mrows/s  "
              f"{fr * m / ms / 1e6:8.1f} Gfeat/s  HBM {(4 * d + 8 * m) * fr / ms / 1e6:7.1f} GB/s")
        del z
    # conv: cfg4-like
    n, L, C, m = 2048, 512, 21, 16384
    x = torch.zeros(n, L, C, device=dev)
    idx = torch.randint(0, C, (n, L), device=dev)
    x.scatter_(2, idx[..., None], 1.0)
    sl = np.random.default_rng(0).integers(64, L + 1, size=n).astype(np.int32)
    k = make_kernel("Conv1dRBF", (n, L, C), m, 123, dev, {"conv_width": 9, "averaging": "sqrt"})
    z = torch.zeros(n, m, dtype=torch.float64, device=dev)
    ms = timeit(lambda: ext.hipConv1dFGen(x, z, k.radem_diag, k.chi_arr, sl, 9, 1), reps=2, warm=1)
    kmers = int((sl - 8).sum())
    print(f"conv1d featgen    L<={L} C={C} M={m} n={n}: {ms:8.3f} ms  {n / ms * 1e3:8.1f} seqs/s  "
          f"{kmers * m / 2 / ms / 1e6:8.1f} G kmer-freqs/s")


if __name__ == "__main__":
    main()
