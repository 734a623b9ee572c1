#!/usr/bin/env python3
"""How the per-feature cos/sin error (hardware v_sin_f32 / v_cos_f32 against libm; <= 4e-7 x scale by the parity tests)
shows in the fused matvec against the oracle's float64 product, by padded width: python tools/matvec_err_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
from oracle import oracle as orc
orc.build(ref=False)
oracle = orc.Oracle()
dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for d, rffs in ((1024, 8192), (2048, 8192), (4000, 8192), (4000, 16384), (512, 8192), (4096, 2048)):
    for seed in (1, 2, 3):
        rng = np.random.default_rng(seed)
        n, icpt = 252, True
        radem, chi = orc.draw_sorf_params(rffs, d, seed + 10)
        x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
        z = np.zeros((n, rffs))
        oracle.cpuRBFFeatureGen(x.copy(), z, radem, chi, icpt)
        F = rffs // 2
        scale = np.sqrt(1.0 / (F - 0.5))
        out = torch.zeros((n, rffs), dtype=torch.float64, device=dev)
        ext.hipRBFFeatureGen(T(x), out, T(radem), T(chi), icpt)
        dz = (out.cpu().numpy() - z) / scale
        z[:, 0] = 1.0
        v = rng.standard_normal(rffs)
        w = torch.zeros(rffs, dtype=torch.float64, device=dev)
        ext.hipZtZMatvec(T(x), T(radem), T(chi), T(v), w, icpt)
        ref = z.T @ (z @ v)
        err = np.abs(w.cpu().numpy() - ref)
        j = int(err.argmax())
        print(f"d={d} M={rffs} seed={seed}: matvec err/max|ref| {err.max() / np.abs(ref).max():.2e} (component {j}, |ref_j| {abs(ref[j]):.3g}, max|ref| {np.abs(ref).max():.3g}, v0 {v[0]:.2f}) | "
              f"feature err/scale: max {np.abs(dz).max():.2e} rms {np.sqrt((dz**2).mean()):.2e} mean cos {dz[:, 0::2].mean():+.2e} mean sin {dz[:, 1::2].mean():+.2e}", flush=True)
