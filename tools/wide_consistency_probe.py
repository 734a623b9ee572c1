"""Whole-launch consistency of the wide transforms (padded width 2048 / 4096): the float64 feature operator must equal the float32 cache
rows x scale BIT FOR BIT in every row, and z^T y / the fused matvec must equal float64 products of those cache rows to 1e-10 -- over many
launches with varying row counts and other work in between.  Sampled-row checks against the CPU oracle cannot see ONE wrong row in 32768;
this can, and did: round 6's first wide build dropped the LDS wait in front of its cross-wave barriers (fused_ztz.inc, Z3_CROSS_BARRIER) and
produced one wrong row in ~3 % of launches at d = 4000.  On a mismatch the row blocks at fault are listed.
    python tools/wide_consistency_probe.py [d] [launches]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext
DEV = "cuda"
N, d, m = 120000, int(sys.argv[1]) if len(sys.argv) > 1 else 4000, 8192
k = make_kernel("RBF", (N, d), m, 123, DEV, {})
k.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
g = torch.Generator(device=DEV); g.manual_seed(d)
x = torch.randn(N, d, generator=g, device=DEV) / np.sqrt(d)
y = torch.randn(N, dtype=torch.float64, device=DEV, generator=g)
v = torch.randn(m, dtype=torch.float64, device=DEV, generator=g)
ws = torch.empty(k.workspace_bytes(), dtype=torch.uint8, device=DEV)
scale = float(np.float32(np.sqrt(1.0 / (m // 2 - 0.5))))
CH = 32768
zt = torch.empty_like(v)
junk = torch.empty(64 << 20, device=DEV)
bad = 0
rng = np.random.default_rng(0)
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    lo = int(rng.integers(0, N - CH))
    n = int(rng.choice([CH, 20000, 7777, 32768, 12345, 30001]))
    xs, ys = x[lo:lo + n], y[lo:lo + n]
    zc = torch.empty((n, m), dtype=torch.float32, device=DEV)
    ext.hipRBFFeatureCache(xs, zc, k.radem_diag, k.chi_arr)
    if trial % 3 == 0:
        junk.normal_()                      # other traffic in front of the launch
    if trial % 2 == 0:
        k.ztz_matvec(xs, v, zt, ws)         # a matvec in front (as in the failing test)
    k.zty(xs, ys, zt, ws)
    z2 = zc.double() * scale
    z64 = torch.zeros((n, m), dtype=torch.float64, device=DEV)
    ext.hipRBFFeatureGen(xs, z64, k.radem_diag, k.chi_arr, True)
    if not torch.equal(z64, z2):
        rows = ((z64 - z2).abs().max(dim=1).values > 0).nonzero().flatten()
        print(f"   FEAT64 != FEAT32 x scale in {rows.numel()} rows, first {rows[:8].tolist()}, max diff {float((z64 - z2).abs().max()):.3e}  <-- FEATURE MISMATCH", flush=True)
        bad += 1
    del z64
    z2[:, 0] = 1.0
    r2 = z2.T @ ys
    err = float((zt - r2).abs().max() / r2.abs().max())
    mv = torch.empty_like(v)
    k.ztz_matvec(xs, v, mv, ws)
    rm = z2.T @ (z2 @ v)
    errm = float((mv - rm).abs().max() / rm.abs().max())
    flag = "  <-- MISMATCH" if err > 1e-10 or errm > 1e-10 else ""
    print(f"trial {trial}: lo={lo} n={n} zty err {err:.2e} matvec err {errm:.2e}{flag}", flush=True)
    if err > 1e-10:
        bad += 1
        # which rows: contribution of row blocks
        for blk in range(0, n, 2048):
            yy = torch.zeros_like(ys); yy[blk:blk + 2048] = ys[blk:blk + 2048]
            k.zty(xs, yy, zt, ws)
            rr = z2[blk:blk + 2048].T @ ys[blk:blk + 2048]
            e2 = float((zt - rr).abs().max() / rr.abs().max())
            if e2 > 1e-10:
                print(f"     rows {blk}..{blk + 2047}: {e2:.2e}")
    del zc, z2
print("mismatching launches:", bad)
