#!/usr/bin/env python3
"""Second randomised parity sweep on the GPU (development aid; tools/stress_parity.py covers the feature operators, the
fused matvec and the block matvec): random shapes through the operators that one did not reach --
  FHT / SRHT (float32 and float64)                 vs the oracle, bit for bit
  RBF gradient, convolution gradient, max-pool     vs the oracle
  SRHT + sample from float32 rows                  vs pad + oracle SRHT + gather
  sketch GEMM (both orientations), Gram            vs float64 torch products
  preconditioner apply (one and k right-hand sides), the CG step kernels (one column and block)   vs their formulas
    python tools/stress_parity2.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from scipy.stats import chi as chi_dist
from xgpr_amd import xgpr_hip_rfgen_ext as ext
from oracle import oracle as orc      # (the checker: development tooling, like tests/)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
orc.build(ref=False)
oracle = orc.Oracle()
dev = "cuda"
worst = {}


def note(tag, err, bar):
    worst[tag] = max(worst.get(tag, 0.0), err / bar if bar > 0 else (0.0 if err == 0 else np.inf))
    assert err <= bar, (tag, err, bar)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-300))


for case in range(cases):
    big = case % 4 == 2
    # ---- FHT / SRHT, bit exact
    p = int(2 ** rng.integers(1, 14))
    n = int(rng.integers(1, 40)) if not big else int(rng.integers(2000, 6000) * max(1, 4096 // p))
    n = min(n, 200000)
    for dt in (np.float32, np.float64):
        a = rng.standard_normal((n, p)).astype(dt)
        radem = rng.choice(np.asarray([-1, 1], dtype=np.int8), size=p)
        ref = a.copy(); oracle.cpuFastHadamardTransform2D(ref)
        g = T(a); ext.hipFastHadamardTransform2D(g)
        note("fht2d", float(np.abs(g.cpu().numpy() - ref).max()), 0.0)
        ref = a.copy(); oracle.cpuSRHT(ref, radem)
        g = T(a); ext.hipSRHT(g, T(radem))
        note("srht", float(np.abs(g.cpu().numpy() - ref).max()), 0.0)
    # ---- RBF gradient
    d = int(rng.choice([3, 20, 64, 100, 256, 300, 1024]))
    m = int(rng.choice([64, 512, 2048, 3000, 4096])) // 2 * 2
    n = int(rng.integers(1, 300)) if not big else int(rng.integers(20000, 40000))
    icpt = bool(rng.integers(0, 2))
    sigma = float(rng.uniform(0.3, 2.0))
    radem, chi = orc.draw_sorf_params(m, d, int(rng.integers(1, 1000)))
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    oref, gref = np.zeros((n, m)), np.zeros((n, m, 1))
    oracle.cpuRBFGrad(x.copy(), oref, gref, radem, chi, sigma, icpt)
    o, gg = torch.zeros((n, m), dtype=torch.float64, device=dev), torch.zeros((n, m, 1), dtype=torch.float64, device=dev)
    ext.hipRBFGrad(T(x), o, gg, T(radem), T(chi), sigma, icpt)
    scale = np.sqrt(1.0 / (m // 2 - 0.5 if icpt else m // 2))
    note("rbf grad: features", float(np.abs(o.cpu().numpy() - oref).max()), 4e-7 * scale)
    note("rbf grad: gradient", float(np.abs(gg.cpu().numpy() - gref).max()), 4e-7 * scale * max(1.0, float(np.abs(gref).max() / scale)))
    del o, gg
    # ---- convolution gradient and max-pool
    C = int(rng.choice([4, 21, 64])); cw = int(rng.integers(1, 17)); L = cw + int(rng.integers(0, 40))
    m2 = int(rng.choice([64, 600, 1024, 2048]))
    ns = int(rng.integers(1, 9)) if not big else int(rng.integers(300, 700))
    radem2, chi2 = orc.draw_sorf_params(m2, cw * C, 77, conv=True)
    xs = rng.standard_normal((ns, L, C)).astype(np.float32)
    sl = rng.integers(cw, L + 1, size=ns).astype(np.int32)
    sc = int(rng.integers(0, 3))
    oref, gref = np.zeros((ns, m2)), np.zeros((ns, m2, 1))
    oracle.cpuConvGrad(xs, oref, radem2, chi2, sl, gref, sigma, cw, sc)
    o, gg = torch.zeros((ns, m2), dtype=torch.float64, device=dev), torch.zeros((ns, m2, 1), dtype=torch.float64, device=dev)
    ext.hipConvGrad(T(xs), o, T(radem2), T(chi2), sl, gg, sigma, cw, sc)
    kmax = int(sl.max()) - cw + 1
    cscale = np.sqrt(2.0 / m2) * {0: kmax, 1: np.sqrt(kmax), 2: 1.0}[sc]
    note("conv grad: features", float(np.abs(o.cpu().numpy() - oref).max()), 4e-7 * cscale)
    note("conv grad: gradient", float(np.abs(gg.cpu().numpy() - gref).max()), 4e-7 * cscale * max(1.0, float(np.abs(gref).max() / max(float(np.abs(oref).max()), 1e-300))))
    pw = int(2 ** np.ceil(np.log2(max(cw * C, 2))))
    reps = int(rng.integers(1, 5))
    mp = reps * pw
    if mp <= 8192:
        prng = np.random.default_rng(int(rng.integers(1, 1000)))
        radem3 = prng.choice(np.asarray([-1, 1], dtype=np.int8), size=(3, 1, mp), replace=True)
        chi3 = chi_dist.rvs(df=pw, size=mp, random_state=5).astype(np.float32)
        oref3 = np.zeros((ns, mp), dtype=np.float32)
        oracle.cpuConv1dMaxpool(xs, oref3, radem3, chi3, sl, cw)
        o3 = torch.zeros((ns, mp), dtype=torch.float32, device=dev)
        ext.hipConv1dMaxpool(T(xs), o3, T(radem3), T(chi3), sl, cw)
        note("maxpool", float(np.abs(o3.cpu().numpy() - oref3).max()), 0.0)
    # ---- float32 rows -> dense products on the matrix cores
    n = int(rng.integers(1, 500)) if not big else int(rng.integers(8000, 20000))
    m = int(rng.choice([128, 512, 1000, 2048, 4096, 8192])) // 4 * 4
    rank = int(rng.integers(1, 300))
    icpt = bool(rng.integers(0, 2))
    scale = float(rng.uniform(0.01, 0.2))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1, 10000)))
    zc = (torch.rand(n, m, device=dev, generator=g) * 2 - 1)
    zs = zc.double() * scale
    if icpt:
        zs[:, 0] = 1.0
    lda = (rank + 63) // 64 * 64
    amat = torch.zeros(n, lda, dtype=torch.float64, device=dev)
    amat[:, :rank] = torch.randn(n, rank, dtype=torch.float64, device=dev, generator=g)
    outm = torch.zeros(rank, m, dtype=torch.float64, device=dev)
    ext.hipSketchGemm(amat, zc, outm, rank, False, False, icpt, scale)
    ref = amat[:, :rank].T @ zs
    note("sketch gemm (contract datapoints)", relmax(outm, ref), 1e-12 * np.sqrt(n) + 1e-300)
    qmat = torch.zeros(m, lda, dtype=torch.float64, device=dev)
    qmat[:, :rank] = torch.randn(m, rank, dtype=torch.float64, device=dev, generator=g)
    outt = torch.zeros(n, lda, dtype=torch.float64, device=dev)
    ext.hipSketchGemm(qmat, zc, outt, rank, True, True, icpt, scale)
    ref = zs @ qmat[:, :rank]
    note("sketch gemm (contract features)", relmax(outt[:, :rank], ref), 1e-12 * np.sqrt(m))
    msub = int(rng.integers(1, m // 128 + 1)) * 128
    gram = torch.zeros(msub, msub, dtype=torch.float64, device=dev)
    ext.hipZtZGram(zc, gram, icpt, scale)
    ref = zs[:, :msub].T @ zs[:, :msub]
    note("gram", relmax(gram, ref), 1e-12 * np.sqrt(n) + 1e-300)
    # ---- preconditioner apply and the CG steps
    mm = int(rng.choice([64, 500, 2100, 8192, 12288]))
    rk = int(rng.integers(1, min(mm, 600)))
    k = int(rng.integers(1, 33))
    u = torch.linalg.qr(torch.randn(mm, rk, dtype=torch.float64, device=dev, generator=g))[0].contiguous()
    ie = 1.0 / (torch.rand(rk, dtype=torch.float64, device=dev, generator=g) * 50 + 0.1)
    pref = float(rng.uniform(0.1, 2.0))
    r1 = torch.randn(mm, dtype=torch.float64, device=dev, generator=g)
    z1 = torch.empty_like(r1)
    ext.hipPrecondApply(u, ie, pref, r1, z1)
    xp = u.T @ r1
    ref = (r1 - u @ xp) + u @ (ie * pref * xp)
    note("precond apply (one column)", relmax(z1, ref), 1e-12 * np.sqrt(mm))
    rb = torch.randn(mm, k, dtype=torch.float64, device=dev, generator=g)
    zb = torch.empty_like(rb)
    ext.hipPrecondApplyBlock(u, ie, pref, rb, zb)
    xp = u.T @ rb
    ref = (rb - u @ xp) + u @ (ie[:, None] * pref * xp)
    note("precond apply (block)", relmax(zb, ref), 1e-12 * np.sqrt(mm))
    w, p_, x_, r_, z_ = (torch.randn(mm, k, dtype=torch.float64, device=dev, generator=g) for _ in range(5))
    lam2 = float(rng.uniform(0.0, 2.0))
    wv = w + lam2 * p_
    rz = (r_ * z_).sum(0); alpha = rz / (p_ * wv).sum(0)
    xr, rn = x_ + alpha[None, :] * p_, r_ - alpha[None, :] * wv
    nrm = torch.rand(k, dtype=torch.float64, device=dev, generator=g) + 0.5
    err_ref = torch.linalg.norm(r_, dim=0) / nrm
    rzb, alb, beb = (torch.zeros(k, dtype=torch.float64, device=dev) for _ in range(3))
    errb = torch.zeros(k, dtype=torch.float64, device=dev)
    cws = torch.empty(ext.cg_block_workspace_bytes(mm, k), dtype=torch.uint8, device=dev)
    wk, xk, rnk = w.clone(), x_.clone(), torch.empty_like(r_)
    ext.hipCGStep1Block(wk, p_, xk, r_, rnk, z_, rzb, alb, errb, nrm, lam2, cws)
    note("block step 1", max(relmax(xk, xr), relmax(rnk, rn), relmax(alb, alpha), relmax(errb, err_ref), relmax(rzb, rz)), 1e-11 * np.sqrt(mm))
    zn = torch.randn(mm, k, dtype=torch.float64, device=dev, generator=g)
    beta = (rn * zn).sum(0) / rz
    pn_ref = zn + beta[None, :] * p_
    pn = torch.empty_like(p_)
    ext.hipCGStep2Block(rnk, zn, p_, pn, rzb, beb, cws)
    note("block step 2", max(relmax(pn, pn_ref), relmax(beb, beta)), 1e-10 * np.sqrt(mm))
    # one column
    scal = torch.zeros(4, dtype=torch.float64, device=dev)
    w1, x1, rn1 = w[:, 0].contiguous(), x_[:, 0].contiguous(), torch.empty(mm, dtype=torch.float64, device=dev)
    p1, r1c, z1c = p_[:, 0].contiguous(), r_[:, 0].contiguous(), z_[:, 0].contiguous()
    ext.hipCGStep1(w1, p1, x1, r1c, rn1, z1c, scal, lam2, float(nrm[0]))
    note("step 1", max(relmax(x1, xr[:, 0]), relmax(rn1, rn[:, 0]), abs(float(scal[1]) / float(alpha[0]) - 1.0), abs(float(scal[2]) / float(err_ref[0]) - 1.0)), 1e-11 * np.sqrt(mm))
    pn1 = torch.empty(mm, dtype=torch.float64, device=dev)
    ext.hipCGStep2(rn1, zn[:, 0].contiguous(), p1, pn1, scal)
    note("step 2", max(relmax(pn1, pn_ref[:, 0]), abs(float(scal[3]) / float(beta[0]) - 1.0)), 1e-10 * np.sqrt(mm))
    print(f"case {case}: fht p={p} | grad d={d} M={m2} | conv C={C} w={cw} ns={ns} | rows n={n} M={m} rank={rank} | cg M={mm} rank={rk} k={k} ok", flush=True)
print("worst error / bar per check:", {k: round(float(v), 3) for k, v in worst.items()})
