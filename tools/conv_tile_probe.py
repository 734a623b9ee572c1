#!/usr/bin/env python3
"""The convolution feature operator on the shapes wave_tile_conv_kernel serves (float64 input; float32 input with windows of 2048 / 4096
elements) through whichever library XGPR_HIP_LIB names: time per call and a checksum.
    python tools/conv_tile_probe.py          (XGPR_F64_PLAN=generic: the any-width path)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
dev = "cuda"
rng = np.random.default_rng(5)
tag = os.environ.get("XGPR_F64_PLAN", "wave tiles")
torch.manual_seed(11)
for name, dt, nseq, L, C, cw, m in (("float64 C=21 w=9 (P=256)", torch.float64, 1024, 512, 21, 9, 16384),
                                    ("float64 C=4 w=5 (P=32)", torch.float64, 2048, 200, 4, 5, 2048),
                                    ("float64 C=4 w=16 (P=64)", torch.float64, 2048, 200, 4, 16, 4096),
                                    ("float64 C=21 w=5 (P=128)", torch.float64, 1024, 300, 21, 5, 8192),
                                    ("float64 C=21 w=20 (P=512)", torch.float64, 1024, 300, 21, 20, 8192),
                                    ("float64 C=64 w=16 (P=1024)", torch.float64, 512, 200, 64, 16, 8192),
                                    ("float32 C=128 w=9 (P=2048)", torch.float32, 512, 256, 128, 9, 8192),
                                    ("float32 C=64 w=40 (P=4096)", torch.float32, 512, 128, 64, 40, 8192),
                                    ("float64 C=128 w=9 (P=2048)", torch.float64, 256, 256, 128, 9, 8192)):
    P = 1 << int(np.ceil(np.log2(cw * C)))
    F = m // 2
    R = -(-F // P) * P
    radem = torch.from_numpy(rng.choice(np.array([-1, 1], dtype=np.int8), size=(3, 1, R))).to(dev)
    chi = (torch.rand(F, device=dev, dtype=torch.float64) + 0.5).to(dt)
    x = torch.randn(nseq, L, C, device=dev, dtype=torch.float64).to(dt)
    sl = rng.integers(max(cw, L // 8), L + 1, size=nseq).astype(np.int32)
    out = torch.zeros(nseq, m, dtype=torch.float64, device=dev)
    ext.hipConv1dFGen(x, out, radem, chi, sl, cw, 1)
    torch.cuda.synchronize()
    chk = float(out.sum())
    t0 = time.perf_counter()
    for _ in range(3):
        ext.hipConv1dFGen(x, out, radem, chi, sl, cw, 1)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / 3
    kmers = int((sl - cw + 1).sum())
    print(f"{tag:12s} conv operator {name}: {nseq} sequences, {kmers} k-mers, M={m}: {dt_s*1e3:.3f} ms  ({kmers * (-(-F // 1024)) / dt_s / 1e6:.1f} M k-mer tiles/s)  checksum {chk:.15e}")
