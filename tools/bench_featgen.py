#!/usr/bin/env python3
"""Stand-alone feature operator (float64 Z materialised) next to plain write-bandwidth references on the same buffer
(development aid).   python tools/bench_featgen.py [--rows 131072]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgpr_amd.kernels import make_kernel   # noqa: E402
from xgpr_amd import xgpr_hip_rfgen_ext as ext   # noqa: E402
from bench_ops import timeit   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=131072)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for (d, m, name, parms) in [(1024, 8192, "Matern", {"matern_nu": 2.5}), (256, 4096, "RBF", {})]:
        n = args.rows
        x = torch.randn(n, d, device=dev) / np.sqrt(d)
        k = make_kernel(name, (n, d), m, 123, dev, parms)
        z = torch.empty(n, m, dtype=torch.float64, device=dev)
        ms = timeit(lambda: ext.hipRBFFeatureGen(x, z, k.radem_diag, k.chi_arr, True), reps=10)
        print(f"featgen op d={d} M={m} n={n}: {ms:.3f} ms  {n * m / ms / 1e6:.1f} Gfeat/s  HBM {(4 * d + 8 * m) * n / ms / 1e6:.1f} GB/s")
        ms = timeit(lambda: z.fill_(1.0), reps=10)
        print(f"   torch fill_ of the same buffer: {ms:.3f} ms  {8 * m * n / ms / 1e6:.1f} GB/s")
        zc = torch.empty(n, m, dtype=torch.float32, device=dev)
        ms = timeit(lambda: torch.mul(zc, 2.0, out=z), reps=10)
        print(f"   f32 -> f64 widen (read 4, write 8 B/elt): {ms:.3f} ms  {12 * m * n / ms / 1e6:.1f} GB/s, write {8 * m * n / ms / 1e6:.1f}")
        ms = timeit(lambda: ext.hipRBFFeatureCache(x, zc, k.radem_diag, k.chi_arr), reps=10)
        print(f"   float32 cache rows: {ms:.3f} ms  HBM {(4 * d + 4 * m) * n / ms / 1e6:.1f} GB/s")
        del z, zc


if __name__ == "__main__":
    main()
