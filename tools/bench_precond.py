#!/usr/bin/env python3
"""Breakdown of one preconditioner accumulation chunk (rand_nys_constructors.py:115-119) on one MI355X."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgpr_amd.kernels import make_kernel, SRHTCompressor

dev = torch.device("cuda", 0)


def timeit(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (n, d, m, rank) in [(8192, 1024, 8192, 512), (8192, 512, 32768, 2048)]:
    x = torch.randn(n, d, device=dev) / np.sqrt(d)
    k = make_kernel("RBF", (n, d), m, 123, dev, {})
    comp = SRHTCompressor(rank, m, device=dev, random_seed=123)
    z = k.transform_x(x)
    acc = torch.zeros(rank, m, dtype=torch.float64, device=dev)
    t_f = timeit(lambda: k.transform_x(x))
    t_s = timeit(lambda: comp.transform_x(z))
    s = comp.transform_x(z)
    st = s.T.contiguous()
    t_g = timeit(lambda: acc.addmm_(s.T, z))
    t_g2 = timeit(lambda: acc.addmm_(st, z))
    fl = 2.0 * n * rank * m
    print(f"n={n} d={d} M={m} rank={rank}: transform_x {t_f:.3f} ms | SRHT(+pad/gather) {t_s:.3f} ms | "
          f"S^T Z GEMM {t_g:.3f} ms = {fl / t_g / 1e9:.1f} TFLOP/s (contig S^T: {t_g2:.3f} ms = {fl / t_g2 / 1e9:.1f} TFLOP/s)")
