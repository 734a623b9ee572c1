#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (development aid, not part of the test suite): random shapes through the fused
matvec, z^T y, the feature operator (float32 and float64 input), the gradient operator (both), the cache rows, the block (k right-hand
sides) matvec / projection and the convolution operator, each against the CPU oracle or a float64 torch product on the same inputs.
    python tools/stress_parity.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
from oracle import oracle as orc      # (the checker: development tooling, like tests/)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
orc.build(ref=False)
oracle = orc.Oracle()
dev = "cuda"
worst = {}


def note(tag, err, bar):
    worst[tag] = max(worst.get(tag, 0.0), err / bar)
    assert err <= bar, (tag, err, bar)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


for case in range(cases):
    # (round 5: row lengths that are not multiples of 4, every padded width from 2 up, 5 / 7 / 8 tiles per datapoint)
    # (round 6: padded widths 2048 / 4096 -- transforms of two / four wave tiles)
    d = int(rng.choice([2, 3, 7, 16, 20, 32, 33, 64, 100, 128, 130, 256, 300, 512, 617, 700, 1023, 1024, 1025, 1076, 1500, 2003, 2048, 2500, 3001,
                        4000, 4096]))
    rffs = int(rng.choice([64, 512, 2048, 3000, 4096, 6144, 8192, 10000, 12288, 14000, 16384])) // 2 * 2
    n = int(rng.integers(1, 700))
    if d > 1024:
        n = int(rng.integers(1, 400))
    if case % 4 == 3 and rffs <= 4096:
        n = int(rng.integers(20000, 60000) // (4 if d > 1024 else 1))      # a launch that fills the chip: loads and stores in flight everywhere
    icpt = bool(rng.integers(0, 2))
    print(f"case {case} starts: d={d} M={rffs} n={n} icpt={icpt}", flush=True)
    radem, chi = orc.draw_sorf_params(rffs, d, int(rng.integers(1, 1000)))
    x = (rng.standard_normal((n, d)) / np.sqrt(d) * rng.choice([1.0, 1.0, 30.0, 3e5, 1e15])).astype(np.float32)     # (the last two: the rare cos/sin branch, beyond 2^31)
    z = np.zeros((n, rffs))
    oracle.cpuRBFFeatureGen(x.copy(), z, radem, chi, icpt)
    F = rffs // 2
    scale = np.sqrt(1.0 / (F - 0.5 if icpt else F))
    out = torch.zeros((n, rffs), dtype=torch.float64, device=dev)
    ext.hipRBFFeatureGen(T(x), out, T(radem), T(chi), icpt)
    note("features", float(np.abs(out.cpu().numpy() - z).max()), 4e-7 * scale)
    if icpt:
        z[:, 0] = 1.0
    v = rng.standard_normal(rffs)
    y = rng.standard_normal(n)
    w = torch.zeros(rffs, dtype=torch.float64, device=dev)
    ext.hipZtZMatvec(T(x), T(radem), T(chi), T(v), w, icpt)
    ref = z.T @ (z @ v)
    fm_err = float(np.abs(w.cpu().numpy() - ref).max()); print(f"   fused matvec err / max|ref| = {fm_err / float(np.abs(ref).max()):.3e}", flush=True) if os.environ.get("STRESS_VERBOSE") else None
    # Two bars.  (1) The kernel against the float64 product of the GPU's OWN feature matrix (checked entry by entry against the oracle
    # above): the same float32 features, float64 sums -- 1e-9.  (2) Against the oracle's product: 2e-6 of the largest entry PLUS what
    # 4e-7 x scale of independent error per feature adds up to over n x M terms (4 sigma).  Without that floor the bar is a lottery on the
    # conditioning of the intercept component: w_0 = sum_i (z_i . v) can cancel to a few units (seed 601 case 19, d = 4000: |w_0| = 3.7
    # where 250-500 is typical; tools/matvec_err_probe.py shows the same at d = 1024) while the accumulated cos/sin error does not shrink
    # with it.  (d = 3: seed 406 case 13 reaches 1.38e-6 of the largest entry, the same on every run; the path's requirement is 1e-5.)
    zg = out.clone()
    if icpt:
        zg[:, 0] = 1.0
    wg = zg.T @ (zg @ T(v))
    note("fused matvec vs own features", float((w - wg).abs().max()), 1e-9 * float(wg.abs().max()))
    floor = 4.0 * 4e-7 * scale * np.sqrt(float(n) * rffs) * float(np.sqrt((v ** 2).mean()))
    note("fused matvec", fm_err, 2e-6 * float(np.abs(ref).max()) + floor)
    zty = torch.zeros(rffs, dtype=torch.float64, device=dev)
    ext.hipZtY(T(x), T(radem), T(chi), T(y), zty, icpt)
    refy = z.T @ y
    note("zty", float(np.abs(zty.cpu().numpy() - refy).max()), 1e-6 * float(np.abs(refy).max()) + 1e-12)
    if rffs % 4 == 0:
        zc = torch.empty((n, rffs), dtype=torch.float32, device=dev)
        ext.hipRBFFeatureCache(T(x), zc, T(radem), T(chi))
        zs = zc.double() * float(np.float32(scale))
        if icpt:
            zs[:, 0] = 1.0
        for k in (1, int(rng.integers(2, 17)), 26, 32):
            V = torch.from_numpy(rng.standard_normal((rffs, k))).to(dev)
            W = torch.zeros_like(V)
            ws = torch.empty(ext.zcache_block_workspace_bytes(n, rffs, k), dtype=torch.uint8, device=dev)
            ext.hipZCacheBlockMatvec(zc, V, W, icpt, ws)
            refw = zs.T @ (zs @ V)
            note(f"block matvec", float((W - refw).abs().max()), 1e-10 * float(refw.abs().max()) + 1e-300)
            P = torch.zeros((n, k), dtype=torch.float64, device=dev)
            ext.hipZCacheBlockProject(zc, V, P, icpt)
            refp = zs @ V
            note("block project", float((P - refp).abs().max()), 1e-11 * float(refp.abs().max()) + 1e-300)
    # gradient operator, float32 (wave tiles at every padded width up to 4096) -- on un-amplified rows (sigma scales the argument)
    ng = min(n, 300)
    xg = (rng.standard_normal((ng, d)) / np.sqrt(d)).astype(np.float32)
    sigma = float(rng.uniform(0.2, 3.0))
    ro, rg = np.zeros((ng, rffs)), np.zeros((ng, rffs, 1))
    oracle.cpuRBFGrad(xg.copy(), ro, rg, radem, chi, sigma, icpt)
    og = torch.full((ng, rffs), 7.0, dtype=torch.float64, device=dev)
    gg = torch.full((ng, rffs, 1), 7.0, dtype=torch.float64, device=dev)
    ext.hipRBFGrad(T(xg), og, gg, T(radem), T(chi), sigma, icpt)
    gscale = np.sqrt(2.0 / rffs)
    note("grad features f32", float(np.abs(og.cpu().numpy() - ro).max()), 4e-7 * gscale)
    note("grad f32", float(np.abs(gg.cpu().numpy() - rg).max()), 1e-6 * max(float(np.abs(rg).max()), gscale))
    # float64 overloads: feature operator and gradient (float64 wave tiles at 64 <= padded width <= 4096, any-width path otherwise)
    radem_d, chi_d = orc.draw_sorf_params(rffs, d, int(rng.integers(1, 1000)), double_precision=True)
    xd = rng.standard_normal((ng, d)) / np.sqrt(d) * float(rng.choice([1.0, 1.0, 40.0, 1e5]))
    zd = np.zeros((ng, rffs))
    oracle.cpuRBFFeatureGen(xd.copy(), zd, radem_d, chi_d, icpt)
    od = torch.full((ng, rffs), 7.0, dtype=torch.float64, device=dev)
    ext.hipRBFFeatureGen(T(xd), od, T(radem_d), T(chi_d), icpt)
    # cos / sin of a bit-identical argument by two double-precision libms: 1e-13 x scale at every amplitude
    note("features f64", float(np.abs(od.cpu().numpy() - zd).max()), 1e-13 * scale)
    xd1 = rng.standard_normal((ng, d)) / np.sqrt(d)
    ro, rg = np.zeros((ng, rffs)), np.zeros((ng, rffs, 1))
    oracle.cpuRBFGrad(xd1.copy(), ro, rg, radem_d, chi_d, sigma, icpt)
    ext.hipRBFGrad(T(xd1), og, gg, T(radem_d), T(chi_d), sigma, icpt)
    note("grad features f64", float(np.abs(og.cpu().numpy() - ro).max()), 1e-13 * gscale)
    note("grad f64", float(np.abs(gg.cpu().numpy() - rg).max()), 1e-13 * max(float(np.abs(rg).max()), gscale))
    # convolution operator
    C = int(rng.choice([4, 21, 64])); cw = int(rng.integers(1, 17)); L = cw + int(rng.integers(0, 40))
    m2 = int(rng.choice([64, 600, 1024, 2048]))
    ns = int(rng.integers(1, 9))
    if case % 4 == 1:
        ns = int(rng.integers(300, 900))          # many sequences in one launch (see tests/test_gpu_conv_long_windows.py)
    radem2, chi2 = orc.draw_sorf_params(m2, cw * C, 77, conv=True)
    xs = rng.standard_normal((ns, L, C)).astype(np.float32)
    sl = rng.integers(cw, L + 1, size=ns).astype(np.int32)
    sc = int(rng.integers(0, 3))
    refc = np.zeros((ns, m2))
    oracle.cpuConv1dFGen(xs, refc, radem2, chi2, sl, cw, sc)
    oc = torch.zeros((ns, m2), dtype=torch.float64, device=dev)
    ext.hipConv1dFGen(T(xs), oc, T(radem2), T(chi2), sl, cw, sc)
    kmax = int(sl.max()) - cw + 1
    cscale = np.sqrt(2.0 / m2) * {0: kmax, 1: np.sqrt(kmax), 2: 1.0}[sc]
    note("conv features", float(np.abs(oc.cpu().numpy() - refc).max()), 4e-7 * cscale)
    # the same on float64 input (wave tiles at every window width, wave_tile.inc) and, every third case, a float32 window of 2048 / 4096
    # elements (C = 128)
    radem3, chi3 = orc.draw_sorf_params(m2, cw * C, 78, conv=True, double_precision=True)
    refd = np.zeros((ns, m2))
    oracle.cpuConv1dFGen(xs.astype(np.float64), refd, radem3, chi3, sl, cw, sc)
    od = torch.zeros((ns, m2), dtype=torch.float64, device=dev)
    ext.hipConv1dFGen(T(xs.astype(np.float64)), od, T(radem3), T(chi3), sl, cw, sc)
    note("conv features f64", float(np.abs(od.cpu().numpy() - refd).max()), 1e-13 * cscale)
    if case % 3 == 0:
        Cw = 128; cww = int(rng.integers(9, 33)); Lw = cww + int(rng.integers(0, 12)); nsw = int(rng.integers(1, 12))
        radem4, chi4 = orc.draw_sorf_params(m2, cww * Cw, 79, conv=True)
        xw = rng.standard_normal((nsw, Lw, Cw)).astype(np.float32)
        slw = rng.integers(cww, Lw + 1, size=nsw).astype(np.int32)
        refw = np.zeros((nsw, m2))
        oracle.cpuConv1dFGen(xw, refw, radem4, chi4, slw, cww, sc)
        ow = torch.zeros((nsw, m2), dtype=torch.float64, device=dev)
        ext.hipConv1dFGen(T(xw), ow, T(radem4), T(chi4), slw, cww, sc)
        kw = int(slw.max()) - cww + 1
        note("conv features wide f32", float(np.abs(ow.cpu().numpy() - refw).max()), 4e-7 * np.sqrt(2.0 / m2) * {0: kw, 1: np.sqrt(kw), 2: 1.0}[sc])
    print(f"case {case}: d={d} M={rffs} n={n} icpt={icpt} | conv C={C} w={cw} L={L} M={m2} ok", flush=True)
print("worst error / bar per check:", {k: round(v, 3) for k, v in worst.items()})
