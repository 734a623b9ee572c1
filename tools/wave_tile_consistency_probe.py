#!/usr/bin/env python3
"""Many launches of the wave-tile operators whose wide transforms exchange tiles between the waves of a workgroup (wave_tile.inc): the
float64 feature operator, the float32 / float64 gradient operators and the float32 / float64 convolution operator at padded widths
2048 / 4096, each launch compared bit for bit with the first one (a cross-wave ordering that fails does so in a few launches per hundred,
under load -- tools/wide_consistency_probe.py found the fused kernel's that way).
    python tools/wave_tile_consistency_probe.py [launches]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(5)
bad = 0
for d, m, n in ((4000, 8192, 24000), (2003, 4000, 40001), (3000, 2050, 30000)):
    k = make_kernel("RBF", (n, d), m, 123, dev, {})
    x32 = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
    xd, chid = x32.double(), k.chi_arr.double()
    first = {}
    o, gr = torch.empty(n, m, dtype=torch.float64, device=dev), torch.empty(n, m, 1, dtype=torch.float64, device=dev)
    for it in range(launches):
        for tag, fn in (("f64 features", lambda: ext.hipRBFFeatureGen(xd, o, k.radem_diag, chid, True)),
                        ("f64 gradient", lambda: ext.hipRBFGrad(xd, o, gr, k.radem_diag, chid, 1.1, True)),
                        ("f32 gradient", lambda: ext.hipRBFGrad(x32, o, gr, k.radem_diag, k.chi_arr, 1.1, True))):
            fn()
            key = (float(o.sum()), float((o * o).sum()), float(gr.sum()) if "grad" in tag else 0.0)
            if tag not in first:
                first[tag] = (key, o.clone())
            elif key != first[tag][0] or not torch.equal(o, first[tag][1]):
                bad += 1
                rows = (o != first[tag][1]).any(dim=1).nonzero().flatten()
                print(f"MISMATCH d={d} {tag} launch {it}: {rows.numel()} rows differ, first {rows[:4].tolist()}", flush=True)
    print(f"d={d} M={m} n={n}: {launches} launches x 3 operators done, mismatches so far {bad}", flush=True)
    del o, gr, first, xd, x32
rng = np.random.default_rng(3)
for dt, C, cw, nseq, L, m in ((torch.float32, 128, 9, 2000, 60, 4096), (torch.float32, 64, 40, 1500, 70, 4096), (torch.float64, 128, 12, 1000, 50, 4096)):
    P = 1 << int(np.ceil(np.log2(cw * C)))
    F = m // 2
    R = -(-F // P) * P
    radem = torch.from_numpy(rng.choice(np.array([-1, 1], dtype=np.int8), size=(3, 1, R))).to(dev)
    chi = (torch.rand(F, device=dev, dtype=torch.float64, generator=g) + 0.5).to(dt)
    x = torch.randn(nseq, L, C, device=dev, dtype=torch.float64, generator=g).to(dt)
    sl = rng.integers(cw, L + 1, size=nseq).astype(np.int32)
    ref = None
    for it in range(launches):
        out = torch.zeros(nseq, m, dtype=torch.float64, device=dev)
        ext.hipConv1dFGen(x, out, radem, chi, sl, cw, 1)
        if ref is None:
            ref = out
        elif not torch.equal(out, ref):
            bad += 1
            print(f"MISMATCH conv {dt} C={C} w={cw} launch {it}", flush=True)
    print(f"conv {dt} C={C} w={cw} (P={P}): {launches} launches done, mismatches so far {bad}", flush=True)
print("mismatching launches:", bad)
sys.exit(1 if bad else 0)
