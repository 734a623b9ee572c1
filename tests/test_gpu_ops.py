"""GPU parity tests of the native operators (through the C ABI, via the reference's own
operator surface xgpr_amd.xgpr_hip_rfgen_ext) against the golden vectors the reference
produced and against the CPU oracle on seeded inputs.

Bars
  * FHT / SRHT / max-pool: bit-exact (adds, subtracts and single multiplies in the
    reference's order).
  * cos/sin features, float path: the f32 argument of every cos/sin is bit-identical to
    the reference's; device sincos is <= 1.6 ulp, glibc's < 1 ulp, so
    |gpu - ref| <= 4e-7 * scale elementwise (scale = sqrt(1/F)); the north-star bar
    (1e-5 relative) is asserted as allclose(rtol=1e-5, atol=1e-5*scale).
  * double path: 1e-13 * scale (ocml vs glibc sin/cos in double).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ext():
    from xgpr_amd import xgpr_hip_rfgen_ext as e
    return e


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def check_features(got, ref, scale, double=False):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    tight = 1e-13 if double else 4e-7
    err = np.abs(got - ref).max()
    assert err <= tight * scale, f"max abs err {err:.3e} vs {tight * scale:.3e}"
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-5 * scale)


def test_selftest_cross_lane(ext):
    out = ext.selftest_lane_xor(DEV)
    lane = np.arange(64)
    for q, h in enumerate([1, 2, 4, 8, 16, 32]):
        for r in range(16):
            exp = np.where(lane & h, -h, 2 * lane + h + 128 * r)
            assert np.array_equal(out[q, r], exp), (h, r, out[q, r])


def test_g1_fht_bit_exact(ext):
    g = load_golden("g1_fht.npz")
    for P in [2, 4, 32, 1024, 4096]:
        x32 = dev(g[f"x_{P}"])
        x64 = dev(g[f"x_{P}"].astype(np.float64))
        ext.hipFastHadamardTransform2D(x32)
        ext.hipFastHadamardTransform2D(x64)
        assert np.array_equal(x32.cpu().numpy(), g[f"y32_{P}"]), P
        assert np.array_equal(x64.cpu().numpy(), g[f"y64_{P}"]), P
    x = dev(g["x3d"])
    x64 = dev(g["x3d"].astype(np.float64))
    ext.hipFastHadamardTransform(x)
    ext.hipFastHadamardTransform(x64)
    assert np.array_equal(x.cpu().numpy(), g["y3d32"])
    assert np.array_equal(x64.cpu().numpy(), g["y3d64"])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,P", [(3, 8), (37, 64), (5, 512), (3, 16384), (2, 32768), (2, 65536), (1, 131072)])
def test_fht_vs_oracle(ext, oracle, dtype, n, P):
    rng = np.random.default_rng(P + n)
    x = rng.standard_normal((n, P)).astype(dtype)
    ref = x.copy()
    oracle.cpuFastHadamardTransform2D(ref)
    xd = dev(x)
    ext.hipFastHadamardTransform2D(xd)
    assert np.array_equal(xd.cpu().numpy(), ref)
    radem = rng.choice(np.asarray([-1, 1], np.int8), size=P)
    ref = x.copy()
    oracle.cpuSRHT(ref, radem)
    xd = dev(x)
    ext.hipSRHT(xd, dev(radem))
    assert np.array_equal(xd.cpu().numpy(), ref)


def test_g5_srht_bit_exact(ext):
    g = load_golden("g5_srht.npz")
    for P in [256, 512, 2048, 8192, 32768]:
        x32 = dev(g[f"x_{P}"])
        x64 = dev(g[f"x_{P}"].astype(np.float64))
        r = dev(g[f"radem_{P}"])
        ext.hipSRHT(x32, r)
        ext.hipSRHT(x64, r)
        assert np.array_equal(x32.cpu().numpy(), g[f"y32_{P}"]), P
        assert np.array_equal(x64.cpu().numpy(), g[f"y64_{P}"]), P


def test_g2_rbf_golden(ext):
    g = load_golden("g2_rbf.npz")
    for si in range(int(g["n_settings"])):
        x32, radem, chi32 = g[f"x_{si}"], g[f"radem_{si}"], g[f"chi_{si}"]
        icpt = bool(g[f"intercept_{si}"])
        F = chi32.shape[0]
        scale = np.sqrt(1.0 / (F - 0.5 if icpt else F))
        o32 = torch.zeros(g[f"out32_{si}"].shape, dtype=torch.float64, device=DEV)
        o64 = torch.zeros_like(o32)
        ext.hipRBFFeatureGen(dev(x32), o32, dev(radem), dev(chi32), icpt)
        ext.hipRBFFeatureGen(dev(x32.astype(np.float64)), o64, dev(radem), dev(chi32.astype(np.float64)), icpt)
        check_features(o32, g[f"out32_{si}"], scale)
        check_features(o64, g[f"out64_{si}"], scale, double=True)
        if f"sigma_{si}" in g:
            sigma = float(g[f"sigma_{si}"])
            for tag, dt in (("32", np.float32), ("64", np.float64)):
                o = torch.zeros_like(o32)
                gr = torch.zeros(o32.shape + (1,), dtype=torch.float64, device=DEV)
                ext.hipRBFGrad(dev(x32.astype(dt)), o, gr, dev(radem), dev(chi32.astype(dt)), sigma, icpt)
                check_features(o, g[f"gout{tag}_{si}"], scale, double=False)
                gref = g[f"grad{tag}_{si}"]
                gs = np.abs(gref).max()
                assert np.abs(gr.cpu().numpy() - gref).max() <= 1e-5 * gs


@pytest.mark.parametrize("d,rffs,icpt,n", [
    (2, 16, False, 9), (3, 64, True, 33), (7, 512, False, 17), (16, 100, True, 5), (32, 512, True, 200),
    (50, 128, False, 11), (64, 2048, True, 9), (100, 300, False, 13), (128, 4096, True, 7),
    (256, 4096, True, 300), (300, 1000, False, 6), (512, 16384, False, 3), (513, 4096, True, 5),
    (1000, 8192, True, 4), (1024, 8192, True, 64), (1024, 2050, False, 3), (1025, 4096, True, 3),
    (2003, 4000, False, 3),
    # padded widths 2048 / 4096: transforms of two / four wave tiles (cross-wave stages); the reference's own test shapes
    # (tests/fht_operations_tests/test_rbf_rfgen.py:37,41) among them; ragged last tiles, tile groups, rows off the 16-byte grid
    (1076, 8192, True, 40), (2003, 4000, True, 70), (2048, 8192, False, 33), (1025, 2048, True, 9), (1500, 1000, False, 21),
    (2000, 16384, True, 25), (1999, 12290, False, 14), (4000, 8192, True, 30), (4096, 4096, False, 19), (3001, 16384, True, 11),
    (2500, 2, False, 5), (4096, 20482, True, 7),
    # beyond 4096: the any-width LDS path
    (5000, 8192, True, 3), (8192, 16384, False, 2)])
def test_rbf_vs_oracle(ext, oracle, d, rffs, icpt, n):
    from oracle import oracle as orc
    rng = np.random.default_rng(d * 7 + rffs)
    radem, chi = orc.draw_sorf_params(rffs, d, 321)
    x = (rng.standard_normal((n, d)) * 2.0).astype(np.float32)
    ref = np.zeros((n, rffs))
    oracle.cpuRBFFeatureGen(x.copy(), ref, radem, chi, icpt)
    out = torch.full((n, rffs), 7.0, dtype=torch.float64, device=DEV)   # overwritten, not accumulated
    ext.hipRBFFeatureGen(dev(x), out, dev(radem), dev(chi), icpt)
    F = rffs // 2
    check_features(out, ref, np.sqrt(1.0 / (F - 0.5 if icpt else F)))


@pytest.mark.parametrize("d,rffs,icpt,n", [
    # float64 wave tiles (padded width <= 4096, wave_tile.inc): every width, ragged rows and tiles, several transforms per tile
    (7, 512, False, 17), (32, 512, True, 200), (9, 8192, True, 77), (3, 2048, False, 31), (2, 128, True, 5), (20, 4096, True, 130), (16, 1000, False, 9),
    (40, 3000, True, 21),
    (33, 64, False, 9), (64, 2048, True, 70), (100, 300, False, 13), (128, 4096, True, 40), (200, 1026, False, 21), (256, 4096, True, 300),
    (300, 1000, False, 6), (512, 16384, False, 30), (513, 4096, True, 50), (1000, 8192, True, 40), (1024, 8192, True, 260), (1024, 2050, False, 3),
    # padded widths 2048 / 4096: two / four waves per transform (a last workgroup with a spare pair: 3 rows x 2 tiles; tiles past the last
    # frequency; several transforms per row)
    (1025, 4096, True, 3), (2003, 4000, False, 3), (1076, 8192, True, 37), (2048, 2050, False, 5), (1500, 100, True, 11), (2049, 8192, True, 21),
    (4000, 8192, False, 19), (4096, 16384, True, 6), (3000, 1000, False, 9), (2500, 10000, True, 2),
    # padded width 8192: eight waves = one workgroup per transform
    (5000, 8192, True, 2), (8192, 16384, False, 5), (4097, 1000, True, 7), (6000, 20000, True, 3),
    # the any-width path: diagonals shorter than 64, padded width > 8192
    (2, 16, False, 9), (5, 48, True, 12), (9000, 16384, True, 2)])
def test_rbf_float64_vs_oracle(ext, oracle, d, rffs, icpt, n):
    """The float64 overload of the feature operator (double_precision = True kernels, kernel_baseclass.py:278-285) against the oracle
    in double: same butterfly order and per-round `radem * norm` product as shared_rfgen_ops.cpp:51-78, so the cos / sin arguments are
    bit-identical and the features agree to the last digits of the two double-precision libms (1e-13 x scale)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(d * 11 + rffs)
    radem, chi = orc.draw_sorf_params(rffs, d, 321, double_precision=True)
    x = rng.standard_normal((n, d)) * 2.0
    ref = np.zeros((n, rffs))
    oracle.cpuRBFFeatureGen(x.copy(), ref, radem, chi, icpt)
    out = torch.full((n, rffs), 7.0, dtype=torch.float64, device=DEV)
    ext.hipRBFFeatureGen(dev(x), out, dev(radem), dev(chi), icpt)
    F = rffs // 2
    check_features(out, ref, np.sqrt(1.0 / (F - 0.5 if icpt else F)), double=True)
    out2 = torch.zeros_like(out)
    ext.hipRBFFeatureGen(dev(x), out2, dev(radem), dev(chi), icpt)
    assert torch.equal(out, out2)
    # un-normalised rows: arguments beyond 2^20 leave the kernel's own reduction for the library's (rows on both sides of it in one launch)
    xb = x * 3e6
    xb[::3] = x[::3]
    with np.errstate(all="ignore"):
        oracle.cpuRBFFeatureGen(xb.copy(), ref := np.zeros((n, rffs)), radem, chi, icpt)
    ext.hipRBFFeatureGen(dev(xb), out, dev(radem), dev(chi), icpt)
    check_features(out, ref, np.sqrt(1.0 / (F - 0.5 if icpt else F)), double=True)


@pytest.mark.parametrize("d,rffs,amp", [(50, 128, 3e4), (1024, 8192, 2e4), (256, 4096, 5e4), (20, 64, 1e6), (512, 2048, 3e9),
                                        (512, 2048, 1e15), (1024, 8192, 1e24), (40, 256, 1e28),
                                        # padded widths 2048 / 4096 (wide transforms: the argument is still bit-identical)
                                        (1500, 4096, 1e5), (1500, 4096, 1e22), (2003, 4000, 3e4), (4000, 8192, 1e5), (3000, 2048, 1e9),
                                        (2048, 16384, 1e4),
                                        # any-width LDS path (padded width 8192: generic_sorf_kernel, Cephes kernels)
                                        (5000, 4096, 1e5), (5000, 4096, 1e22)])
def test_large_arguments_take_the_rare_path(ext, oracle, d, rffs, amp):
    """Un-normalised inputs: cos/sin arguments at and beyond 2^18 take the kernels' rare branch (common.inc turns_fixed:
    the angle in revolutions from the float's integer significand and a table of frac(2^k / (2 pi))), good for EVERY
    finite float -- the reference evaluates libm's cos / sin there (shared_rfgen_ops.cpp:105-111; glibc: Payne-Hanek).
    The float32 argument is bit-identical to the reference's, so the feature bar is the usual one at any magnitude;
    where the reference's transform overflowed (inf - inf) both sides hold NaN.  Operator, fused matvec and cache build
    agree."""
    from oracle import oracle as orc
    rng = np.random.default_rng(d + rffs)
    radem, chi = orc.draw_sorf_params(rffs, d, 5)
    n = 24
    x = (rng.standard_normal((n, d)) * amp).astype(np.float32)
    x[::3] *= 1e-4                                  # rows on the common path next to rows on the rare one
    x[1::6] *= np.float32(1e-4 if amp < 1e10 else 1e-12)
    ref = np.zeros((n, rffs))
    with np.errstate(all="ignore"):
        oracle.cpuRBFFeatureGen(x.copy(), ref, radem, chi, False)
    tiny = np.float32(2.0 ** -100) if amp > 1e12 else np.float32(2.0 ** -40)
    arg = np.zeros((n, rffs))
    oracle.cpuRBFFeatureGen(x.copy() * tiny, arg, radem, chi, False)   # tiny arguments: sin ~ argument
    F = rffs // 2
    scale = np.sqrt(1.0 / F)
    mag = np.abs(arg[:, 1::2]) / scale / float(tiny)                                # |argument| per (row, frequency)
    assert (mag >= 262144.0).mean() > 0.05, "the case must exercise the rare branch"
    if amp >= 3e9:
        assert (mag >= 2.0 ** 31).mean() > 0.05, "the case must exercise arguments beyond 2^31"
    out = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
    ext.hipRBFFeatureGen(dev(x), out, dev(radem), dev(chi), False)
    got = out.cpu().numpy()
    finite = np.isfinite(ref)
    assert finite.mean() > 0.9
    assert np.isnan(got[~finite]).all() and np.isfinite(got[finite]).all()
    err = np.abs(got - ref)[finite].max()
    assert err <= 4e-7 * scale, f"max abs err {err:.3e} vs {4e-7 * scale:.3e}"
    if d > 4096:
        return                                      # the cache build and the fused matvec serve padded widths <= 4096
    zc = torch.empty((n, rffs), dtype=torch.float32, device=DEV)
    ext.hipRBFFeatureCache(dev(x), zc, dev(radem), dev(chi))
    same = zc.double().cpu().numpy() * float(np.float32(scale))
    assert np.array_equal(same[finite], got[finite]) and np.isnan(same[~finite]).all()
    if not finite.all():
        return
    v = rng.standard_normal(rffs)
    w = torch.zeros(rffs, dtype=torch.float64, device=DEV)
    ext.hipZtZMatvec(dev(x), dev(radem), dev(chi), dev(v), w, False)
    refw = ref.T @ (ref @ v)
    assert np.abs(w.cpu().numpy() - refw).max() <= 1e-6 * np.abs(refw).max()


def test_rbf_deterministic(ext):
    from oracle import oracle as orc
    radem, chi = orc.draw_sorf_params(8192, 1024, 123)
    x = torch.randn(128, 1024, device=DEV, dtype=torch.float32)
    a = torch.zeros(128, 8192, dtype=torch.float64, device=DEV)
    b = torch.zeros_like(a)
    ext.hipRBFFeatureGen(x, a, dev(radem), dev(chi), True)
    ext.hipRBFFeatureGen(x, b, dev(radem), dev(chi), True)
    assert torch.equal(a, b)


def test_g3_conv_golden(ext):
    g = load_golden("g3_conv.npz")
    for si in range(int(g["n_settings"])):
        x32, radem, chi32, sl = g[f"x_{si}"], g[f"radem_{si}"], g[f"chi_{si}"], g[f"seqlen_{si}"]
        cw, sc = int(g[f"conv_width_{si}"]), int(g[f"scaling_{si}"])
        F = chi32.shape[0]
        kmax = int(sl.max()) - cw + 1
        scale = np.sqrt(1.0 / F) * {0: kmax, 1: np.sqrt(kmax), 2: 1.0}[sc]   # sum of up to kmax terms
        o32 = torch.zeros(g[f"out32_{si}"].shape, dtype=torch.float64, device=DEV)
        o64 = torch.zeros_like(o32)
        ext.hipConv1dFGen(dev(x32), o32, dev(radem), dev(chi32), sl, cw, sc)
        ext.hipConv1dFGen(dev(x32.astype(np.float64)), o64, dev(radem), dev(chi32.astype(np.float64)), sl, cw, sc)
        check_features(o32, g[f"out32_{si}"], scale)
        check_features(o64, g[f"out64_{si}"], scale, double=True)
    for tag, dt in (("32", np.float32), ("64", np.float64)):
        o = torch.zeros(g["g_out32"].shape, dtype=torch.float64, device=DEV)
        gr = torch.zeros(g["g_grad32"].shape, dtype=torch.float64, device=DEV)
        ext.hipConvGrad(dev(g["g_x"].astype(dt)), o, dev(g["g_radem"]), dev(g["g_chi"].astype(dt)),
                        g["g_seqlen"], gr, float(g["g_sigma"]), int(g["g_conv_width"]), int(g["g_scaling"]))
        ref_o, ref_g = g[f"g_out{tag}"], g[f"g_grad{tag}"]
        assert np.abs(o.cpu().numpy() - ref_o).max() <= 1e-5 * np.abs(ref_o).max()
        assert np.abs(gr.cpu().numpy() - ref_g).max() <= 1e-5 * np.abs(ref_g).max()


@pytest.mark.parametrize("L,C,cw,rffs,sc,n", [(30, 21, 9, 1024, 0, 9), (17, 4, 1, 64, 1, 21), (40, 21, 5, 600, 2, 7),
                                              (64, 21, 9, 4096, 1, 6), (12, 300, 4, 512, 1, 4), (25, 8, 3, 2050, 0, 5),
                                              # long windows (P = 512, 1024), ragged and full
                                              (40, 21, 30, 2048, 1, 5), (30, 21, 15, 1024, 0, 6), (20, 64, 8, 1024, 2, 5),
                                              (24, 64, 16, 2048, 1, 4)])
@pytest.mark.parametrize("amp", [1.0, 4e4, 1e13])
def test_conv_vs_oracle(ext, oracle, L, C, cw, rffs, sc, n, amp):
    """amp = 4e4: un-normalised inputs, part of the cos/sin arguments beyond 2^18 (the kernels' rare branch);
    1e13: beyond 2^31 (the table-driven reduction, any finite float)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(L * C + rffs)
    radem, chi = orc.draw_sorf_params(rffs, cw * C, 77, conv=True)
    x = (rng.standard_normal((n, L, C)) * amp).astype(np.float32)
    x[::2] *= np.float32(1.0 / amp)
    sl = rng.integers(cw, L + 1, size=n).astype(np.int32)
    ref = np.zeros((n, rffs))
    oracle.cpuConv1dFGen(x, ref, radem, chi, sl, cw, sc)
    out = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
    ext.hipConv1dFGen(dev(x), out, dev(radem), dev(chi), sl, cw, sc)
    kmax = int(sl.max()) - cw + 1
    scale = np.sqrt(2.0 / rffs) * {0: kmax, 1: np.sqrt(kmax), 2: 1.0}[sc]
    check_features(out, ref, scale)
    # accumulate semantics: a second call adds on top (reference rbf_convolution.cu:140-146)
    ext.hipConv1dFGen(dev(x), out, dev(radem), dev(chi), sl, cw, sc)
    check_features(out, 2 * ref, 2 * scale)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("L,C,cw,rffs,sc,n", [
    # float64: every padded window width (wave tiles, wave_tile.inc); float32 narrower than 2048: wave_conv_kernel (covered above too)
    (30, 21, 9, 1024, 0, 9), (17, 4, 1, 64, 1, 21), (40, 21, 5, 600, 2, 7), (25, 8, 3, 2050, 0, 5), (20, 64, 8, 1024, 2, 5), (12, 3, 2, 4096, 1, 33),
    # padded windows of 2048 / 4096 elements (two / four waves per transform; workgroups spanning two sequences of different lengths;
    # a last workgroup with a spare pair: 3 sequences x 2 tiles)
    (30, 128, 9, 2048, 1, 3), (40, 64, 20, 4096, 0, 7), (24, 128, 16, 1000, 2, 5), (50, 100, 30, 8192, 1, 4), (33, 21, 60, 2048, 1, 1),
    (70, 64, 33, 4100, 0, 6)])
def test_wave_tile_conv_operators_vs_oracle(ext, oracle, dtype, L, C, cw, rffs, sc, n):
    """The convolution feature operator, its gradient and the max-pool operator on float64 input (every padded window width up to 4096)
    and on float32 input with windows of 2048 / 4096 elements (wave_tile_conv_kernel): against the oracle; accumulate semantics (a second
    call adds on top, the max-pool keeps the maximum); bit-reproducible."""
    from oracle import oracle as orc
    if L < cw:
        L = cw + 3
    rng = np.random.default_rng(L * C + rffs + cw)
    dp = dtype == np.float64
    radem, chi = orc.draw_sorf_params(rffs, cw * C, 77, conv=True, double_precision=dp)
    x = rng.standard_normal((n, L, C)).astype(dtype)
    sl = rng.integers(cw, L + 1, size=n).astype(np.int32)
    sl[0] = L
    kmax = int(sl.max()) - cw + 1
    scale = np.sqrt(2.0 / rffs) * {0: kmax, 1: np.sqrt(kmax), 2: 1.0}[sc]
    tol = 1e-13 if dp else 4e-7
    ref = np.zeros((n, rffs))
    oracle.cpuConv1dFGen(x, ref, radem, chi, sl, cw, sc)
    out = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
    ext.hipConv1dFGen(dev(x), out, dev(radem), dev(chi), sl, cw, sc)
    assert np.abs(out.cpu().numpy() - ref).max() <= tol * scale
    again = torch.zeros_like(out)
    ext.hipConv1dFGen(dev(x), again, dev(radem), dev(chi), sl, cw, sc)
    assert torch.equal(out, again)
    ext.hipConv1dFGen(dev(x), out, dev(radem), dev(chi), sl, cw, sc)
    assert np.abs(out.cpu().numpy() - 2 * ref).max() <= 2 * tol * scale
    # gradient
    sigma = 0.8
    ro, rg = np.zeros((n, rffs)), np.zeros((n, rffs, 1))
    oracle.cpuConvGrad(x, ro, radem, chi, sl, rg, sigma, cw, sc)
    o = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
    g = torch.zeros((n, rffs, 1), dtype=torch.float64, device=DEV)
    ext.hipConvGrad(dev(x), o, dev(radem), dev(chi), sl, g, sigma, cw, sc)
    gt = 1e-12 if dp else 1e-6
    assert np.abs(o.cpu().numpy() - ro).max() <= gt * np.abs(ro).max()
    assert np.abs(g.cpu().numpy() - rg).max() <= gt * np.abs(rg).max()
    # max-pool: F == M, the diagonal exactly reps * P long
    P = 1 << max(1, int(np.ceil(np.log2(cw * C))))
    m2 = max(P, (rffs // P) * P)
    rng2 = np.random.default_rng(7)
    radem2 = rng2.choice(np.array([-1, 1], dtype=np.int8), size=(3, 1, m2))
    chi2 = np.abs(rng2.standard_normal(m2)).astype(dtype) + dtype(0.5)
    refm = np.zeros((n, m2), dtype=np.float32)
    oracle.cpuConv1dMaxpool(x, refm, radem2, chi2, sl, cw)
    om = torch.zeros((n, m2), dtype=torch.float32, device=DEV)
    ext.hipConv1dMaxpool(dev(x), om, dev(radem2), dev(chi2), sl, cw)
    got = om.cpu().numpy()
    if dp:
        assert np.array_equal(got, refm)
    else:
        assert np.abs(got - refm).max() <= 1e-5 * np.abs(refm).max()


def test_g4_maxpool_bit_exact(ext):
    g = load_golden("g4_maxpool.npz")
    for si in range(int(g["n_settings"])):
        x32, radem, chi32, sl = g[f"x_{si}"], g[f"radem_{si}"], g[f"chi_{si}"], g[f"seqlen_{si}"]
        cw = int(g[f"conv_width_{si}"])
        o32 = torch.zeros(g[f"out32_{si}"].shape, dtype=torch.float32, device=DEV)
        o64 = torch.zeros_like(o32)
        ext.hipConv1dMaxpool(dev(x32), o32, dev(radem), dev(chi32), sl, cw)
        ext.hipConv1dMaxpool(dev(x32.astype(np.float64)), o64, dev(radem), dev(chi32.astype(np.float64)), sl, cw)
        assert np.array_equal(o32.cpu().numpy(), g[f"out32_{si}"]), si
        assert np.array_equal(o64.cpu().numpy(), g[f"out64_{si}"]), si


def test_error_behaviour(ext):
    """RuntimeError where the reference throws; TypeError for un-converted arguments
    (nanobind .noconvert()); reference tests/fht_operations_tests/
    test_variable_length_seq_handling.py:74-95."""
    from oracle import oracle as orc
    radem, chi = orc.draw_sorf_params(64, 10, 123)
    x = torch.zeros(4, 10, device=DEV)
    with pytest.raises(RuntimeError):
        ext.hipRBFFeatureGen(x, torch.zeros(3, 64, dtype=torch.float64, device=DEV), dev(radem), dev(chi), False)
    with pytest.raises(RuntimeError):
        ext.hipRBFFeatureGen(x, torch.zeros(4, 62, dtype=torch.float64, device=DEV), dev(radem), dev(chi), False)
    with pytest.raises(TypeError):
        ext.hipRBFFeatureGen(x, torch.zeros(4, 64, dtype=torch.float32, device=DEV), dev(radem), dev(chi), False)
    with pytest.raises(TypeError):
        ext.hipRBFFeatureGen(x.cpu(), torch.zeros(4, 64, dtype=torch.float64, device=DEV), dev(radem), dev(chi), False)
    with pytest.raises(TypeError):
        ext.hipRBFFeatureGen(x.double(), torch.zeros(4, 64, dtype=torch.float64, device=DEV), dev(radem), dev(chi), False)
    radem, chi = orc.draw_sorf_params(64, 12, 123, conv=True)
    xc = torch.zeros(3, 10, 4, device=DEV)
    out = torch.zeros(3, 64, dtype=torch.float64, device=DEV)
    ext.hipConv1dFGen(xc, out, dev(radem), dev(chi), np.array([10, 5, 3], np.int32), 3, 0)
    for bad in (np.array([11, 5, 3], np.int32), np.array([10, 5, 2], np.int32), np.array([10, 5], np.int32)):
        with pytest.raises(RuntimeError):
            ext.hipConv1dFGen(xc, out, dev(radem), dev(chi), bad, 3, 0)
    with pytest.raises(RuntimeError):
        ext.hipConv1dFGen(xc, out, dev(radem), dev(chi), np.array([10, 5, 3], np.int32), 11, 0)
    with pytest.raises(TypeError):
        ext.hipConv1dFGen(xc, out, dev(radem), dev(chi), np.array([10, 5, 3], np.int64), 3, 0)
    with pytest.raises(RuntimeError):
        ext.hipFastHadamardTransform2D(torch.zeros(3, 12, device=DEV))
    with pytest.raises(RuntimeError):
        ext.hipSRHT(torch.zeros(3, 16, device=DEV), torch.ones(8, dtype=torch.int8, device=DEV))


@pytest.mark.parametrize("d,rffs,icpt,n", [(32, 512, True, 2000), (20, 64, False, 100), (256, 4096, True, 3000),
                                           (100, 3000, True, 777), (1024, 8192, True, 1500),
                                           (512, 16384, False, 300), (8, 2048, True, 50),
                                           (512, 32768, True, 700), (64, 20000, True, 333), (300, 18434, False, 65),
                                           # the two passes on the three-wave plan: tile groups of 2, 4 and 6, a ragged last tile
                                           (256, 18434, True, 130), (128, 22530, False, 90), (512, 26626, True, 77),
                                           (1024, 24576, True, 50),
                                           # rows that are not whole aligned 16-byte groups (fetched float by float), five tiles per
                                           # datapoint (ten of twelve waves), eight tiles (two passes), and the same beyond 8192 frequencies
                                           (1022, 8192, True, 150), (130, 4096, False, 77), (513, 6144, True, 133), (250, 10240, True, 250),
                                           (1000, 10000, False, 90), (1024, 16384, True, 140), (257, 16384, False, 60), (511, 20480, True, 70),
                                           # 16 <= padded width < 128 on the three-wave kernel (two or more tiles per datapoint)
                                           (32, 4096, True, 300), (64, 8192, False, 200), (20, 6144, True, 150), (16, 4096, False, 170),
                                           (9, 10240, True, 120), (50, 4096, False, 130), (33, 6000, True, 90), (64, 16384, True, 80),
                                           # padded width <= 16: the tile never leaves the rows layout (no exchange); rows shorter than 16 floats
                                           (2, 4096, True, 210), (3, 6144, False, 100), (5, 4096, True, 160), (8, 8192, False, 140),
                                           (12, 4096, True, 110), (7, 10240, False, 75), (4, 2050, True, 65), (16, 12288, True, 55),
                                           # ... and the two passes beyond 8192 frequencies at those widths
                                           (32, 32768, True, 90), (64, 18434, False, 70), (8, 20480, True, 60), (50, 16384, True, 100),
                                           # one tile per datapoint below padded width 128 (twelve one-wave slots)
                                           (32, 2048, True, 400), (64, 1024, False, 333), (10, 512, True, 257), (3, 1500, False, 129),
                                           # padded widths 2048 / 4096: two / four wave tiles per transform, one cross-wave exchange per round;
                                           # two and four computed tiles, ragged ones, the two passes in groups of 2 / 4, rows off the 16-byte grid
                                           (1076, 8192, True, 300), (2003, 4000, False, 250), (2048, 8192, True, 200), (1025, 2048, False, 150),
                                           (1500, 6146, True, 170), (2000, 12288, False, 120), (2047, 16384, True, 90), (1030, 20482, False, 60),
                                           (4000, 8192, True, 160), (4096, 4096, False, 140), (2049, 2048, True, 100), (3000, 6000, False, 80),
                                           (4000, 16384, True, 70), (3333, 24578, False, 40), (2050, 50, True, 30)])
def test_fused_matvec_vs_oracle(ext, oracle, d, rffs, icpt, n):
    """hipZtZMatvec == Z.T @ (Z @ v) with Z = transform_x(x) from the oracle (incl. Z[:,0] = 1);
    f64 accumulation, so 1e-9 relative in the max norm; and bit-reproducible run to run."""
    from oracle import oracle as orc
    rng = np.random.default_rng(d + rffs)
    radem, chi = orc.draw_sorf_params(rffs, d, 11)
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    z = np.zeros((n, rffs))
    oracle.cpuRBFFeatureGen(x.copy(), z, radem, chi, icpt)
    if icpt:
        z[:, 0] = 1.0
    v = rng.standard_normal(rffs)
    y = rng.standard_normal(n)
    ref = z.T @ (z @ v)
    out = torch.zeros(rffs, dtype=torch.float64, device=DEV)
    ext.hipZtZMatvec(dev(x), dev(radem), dev(chi), dev(v), out, icpt)
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-6 * np.abs(ref).max()
    out2 = torch.zeros_like(out)
    ext.hipZtZMatvec(dev(x), dev(radem), dev(chi), dev(v), out2, icpt)
    assert torch.equal(out, out2)
    # a caller-held workspace keeps the packed sign masks: later calls may skip the packing launch
    ws = torch.empty(ext.ztz_workspace_bytes(rffs, radem.shape[2]), dtype=torch.uint8, device=DEV)
    out3, out4 = torch.zeros_like(out), torch.zeros_like(out)
    ext.hipZtZMatvec(dev(x), dev(radem), dev(chi), dev(v), out3, icpt, ws)
    ext.hipZtZMatvec(dev(x), dev(radem), dev(chi), dev(v), out4, icpt, ws, masksPacked=True)
    assert torch.equal(out, out3) and torch.equal(out, out4)
    zty = torch.zeros(rffs, dtype=torch.float64, device=DEV)
    ext.hipZtY(dev(x), dev(radem), dev(chi), dev(y), zty, icpt)
    refy = z.T @ y
    assert np.abs(zty.cpu().numpy() - refy).max() <= 1e-6 * np.abs(refy).max()
    # a base pointer off the 16-byte boundary (a view one float into a buffer): same numbers as the aligned copy
    if rffs % 4 == 0 and d <= 4096:
        buf = torch.empty(n * d + 1, dtype=torch.float32, device=DEV)
        xo = buf[1:].view(n, d)
        xo.copy_(dev(x))
        assert xo.data_ptr() % 16 != 0
        out5 = torch.zeros_like(out)
        ext.hipZtZMatvec(xo, dev(radem), dev(chi), dev(v), out5, icpt)
        assert np.abs(out5.cpu().numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
        zc, zo = (torch.empty((n, rffs), dtype=torch.float32, device=DEV) for _ in range(2))
        ext.hipRBFFeatureCache(dev(x), zc, dev(radem), dev(chi))
        ext.hipRBFFeatureCache(xo, zo, dev(radem), dev(chi))
        assert torch.equal(zc, zo)


@pytest.mark.parametrize("d,rffs,icpt,n", [(32, 512, True, 2000), (20, 64, False, 100), (256, 4096, True, 3000),
                                           (100, 3000, True, 777), (1024, 8192, True, 1500), (512, 16384, False, 300),
                                           (7, 10, True, 33), (64, 6146, True, 5), (3, 2, False, 1),
                                           (512, 32768, True, 300), (256, 20002, False, 77), (64, 16386, True, 40),
                                           (1024, 32768, False, 3), (128, 24580, True, 1),
                                           # padded widths 2048 / 4096
                                           (1076, 8192, True, 200), (2003, 4000, False, 50), (4000, 8192, True, 90), (2500, 12290, False, 20),
                                           (1100, 2, True, 4), (3000, 32768, True, 6)])
def test_feature_cache_and_cached_matvec(ext, oracle, d, rffs, icpt, n):
    """The resident float32 feature cache holds exactly the float32 cos/sin the float64 operator
    widens (bit-for-bit: cache * scale == hipRBFFeatureGen output), and the matvec streamed from it
    equals Z.T @ (Z @ v) from the oracle's features."""
    from oracle import oracle as orc
    rng = np.random.default_rng(d + rffs)
    radem, chi = orc.draw_sorf_params(rffs, d, 11)
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    zc = torch.empty((n, rffs), dtype=torch.float32, device=DEV)
    ext.hipRBFFeatureCache(dev(x), zc, dev(radem), dev(chi))
    zf = torch.empty((n, rffs), dtype=torch.float64, device=DEV)
    ext.hipRBFFeatureGen(dev(x), zf, dev(radem), dev(chi), icpt)
    F = rffs // 2
    scale = float(np.float32(np.sqrt(1.0 / (F - 0.5 if icpt else F))))
    assert torch.equal(zc.double() * scale, zf)
    z = np.zeros((n, rffs))
    oracle.cpuRBFFeatureGen(x.copy(), z, radem, chi, icpt)
    if icpt:
        z[:, 0] = 1.0
    v = rng.standard_normal(rffs)
    ref = z.T @ (z @ v)
    out = torch.zeros(rffs, dtype=torch.float64, device=DEV)
    ws = torch.empty(ext.ztz_workspace_bytes(rffs, radem.shape[2]), dtype=torch.uint8, device=DEV)
    ext.hipZCacheMatvec(zc, dev(v), out, icpt, ws)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
    zz = zc.double() * scale                    # the same float32 values in a float64 product: 1e-12
    if icpt:
        zz[:, 0] = 1.0
    ref2 = zz.T @ (zz @ dev(v))
    assert float((out - ref2).abs().max()) <= 1e-12 * float(ref2.abs().max())
    out2 = torch.zeros_like(out)
    ext.hipZCacheMatvec(zc, dev(v), out2, icpt, ws)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("d,rffs,icpt,n", [(50, 128, False, 11), (256, 4096, True, 30), (856, 4000, True, 7),
                                           (1024, 8192, False, 9), (1100, 2048, True, 5),
                                           # padded widths 2048 / 4096: two / four waves of a workgroup per transform (wave_tile.inc, T = float)
                                           (2003, 4000, False, 7), (1076, 8192, True, 3), (1500, 100, False, 11), (4000, 8192, True, 5),
                                           (4096, 4100, False, 6), (3000, 16384, True, 2),
                                           # padded width 8192 (eight waves per transform)
                                           (5000, 8192, True, 2), (8192, 16384, False, 3), (6000, 300, True, 5),
                                           # the any-width path
                                           (9000, 16384, True, 2)])
def test_rbf_grad_vs_oracle(ext, oracle, d, rffs, icpt, n):
    """cudaRBFGrad on the wave kernels (P <= 4096) and the any-width path: features and d/dsigma against
    the oracle, including the reference's roundings back to float (shared_rfgen_ops.cpp:140-155)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(d + rffs)
    radem, chi = orc.draw_sorf_params(rffs, d, 17)
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    sigma = 1.7
    ro, rg = np.zeros((n, rffs)), np.zeros((n, rffs, 1))
    oracle.cpuRBFGrad(x.copy(), ro, rg, radem, chi, sigma, icpt)
    o = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
    g = torch.zeros((n, rffs, 1), dtype=torch.float64, device=DEV)
    ext.hipRBFGrad(dev(x), o, g, dev(radem), dev(chi), sigma, icpt)
    scale = np.sqrt(2.0 / rffs)
    assert np.abs(o.cpu().numpy() - ro).max() <= 4e-7 * scale
    assert np.abs(g.cpu().numpy() - rg).max() <= 1e-6 * np.abs(rg).max()


@pytest.mark.parametrize("d,rffs,icpt,n", [(64, 2048, True, 70), (100, 300, False, 13), (200, 1026, False, 21), (513, 4096, True, 50),
                                           (1024, 8192, True, 33), (1076, 8192, True, 7), (2003, 4000, False, 3), (4000, 8192, False, 9),
                                           (3000, 1000, True, 6), (33, 64, False, 9), (5000, 8192, True, 2),
                                           (9, 8192, True, 77), (3, 2048, False, 31), (2, 128, True, 5), (20, 4096, True, 130), (32, 512, False, 9),
                                           (7, 16, False, 4), (8000, 16384, False, 3), (4100, 8192, True, 4), (9000, 16384, True, 2)])
def test_rbf_grad_float64_vs_oracle(ext, oracle, d, rffs, icpt, n):
    """The float64 overload of cudaRBFGrad (double_precision = True kernels) on the float64 wave tiles (P <= 4096) and on the any-width
    path beyond: same stage order and per-round `radem * norm` product, so the argument and `grad_val` are bit-identical to the oracle's in
    double and the outputs agree to the last digits of the two libms."""
    from oracle import oracle as orc
    rng = np.random.default_rng(d * 7 + rffs)
    radem, chi = orc.draw_sorf_params(rffs, d, 19, double_precision=True)
    x = rng.standard_normal((n, d)) / np.sqrt(d)
    sigma = 1.3
    ro, rg = np.zeros((n, rffs)), np.zeros((n, rffs, 1))
    oracle.cpuRBFGrad(x.copy(), ro, rg, radem, chi, sigma, icpt)
    o = torch.full((n, rffs), 7.0, dtype=torch.float64, device=DEV)
    g = torch.full((n, rffs, 1), 7.0, dtype=torch.float64, device=DEV)
    ext.hipRBFGrad(dev(x), o, g, dev(radem), dev(chi), sigma, icpt)
    scale = np.sqrt(2.0 / rffs)
    assert np.abs(o.cpu().numpy() - ro).max() <= 1e-13 * scale
    assert np.abs(g.cpu().numpy() - rg).max() <= 1e-13 * max(np.abs(rg).max(), scale)
    o2, g2 = torch.zeros_like(o), torch.zeros_like(g)
    ext.hipRBFGrad(dev(x), o2, g2, dev(radem), dev(chi), sigma, icpt)
    assert torch.equal(o, o2) and torch.equal(g, g2)


@pytest.mark.parametrize("L,C,cw,rffs,sc,n", [(30, 21, 9, 1024, 1, 9), (17, 4, 1, 64, 0, 21), (40, 21, 5, 600, 2, 7)])
def test_conv_grad_vs_oracle(ext, oracle, L, C, cw, rffs, sc, n):
    from oracle import oracle as orc
    rng = np.random.default_rng(L * C + rffs)
    radem, chi = orc.draw_sorf_params(rffs, cw * C, 77, conv=True)
    x = rng.standard_normal((n, L, C)).astype(np.float32)
    sl = rng.integers(cw, L + 1, size=n).astype(np.int32)
    sigma = 0.8
    ro, rg = np.zeros((n, rffs)), np.zeros((n, rffs, 1))
    oracle.cpuConvGrad(x, ro, radem, chi, sl, rg, sigma, cw, sc)
    o = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
    g = torch.zeros((n, rffs, 1), dtype=torch.float64, device=DEV)
    ext.hipConvGrad(dev(x), o, dev(radem), dev(chi), sl, g, sigma, cw, sc)
    assert np.abs(o.cpu().numpy() - ro).max() <= 1e-6 * np.abs(ro).max()
    assert np.abs(g.cpu().numpy() - rg).max() <= 1e-6 * np.abs(rg).max()


def test_g12_mini_ard_grad_and_kernel_vs_reference_ground_truth(oracle):
    """cudaMiniARDGrad drop-in (hipMiniARDGrad) and the MiniARD kernel object against the reference's own
    ground truth (tests/golden/g12_mini_ard.npz, reference test_ARD_kernel_gradient.py:120-162) with the
    tolerances of that test (:77-80), and against the oracle's operator on the same inputs."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel, MiniARDKernel
    g = load_golden("g12_mini_ard.npz")
    for ci in range(int(g["ncases"])):
        x = g[f"c{ci}_x"]
        nf, icpt = int(g[f"c{ci}_num_freqs"]), bool(g[f"c{ci}_intercept"])
        sp = [int(v) for v in g[f"c{ci}_split_points"]]
        for dp, rtol_f, atol_f, rtol_g, atol_g in ((True, 1e-5, 1e-8, 1e-5, 1e-8), (False, 1e-5, 1e-5, 1e-4, 1e-3)):
            kern = MiniARDKernel(x.shape, 2 * nf, 123, DEV, dp, {"split_points": sp, "intercept": icpt})
            kern.set_hyperparams(g[f"c{ci}_hyperparams"], logspace=False)
            kern.precompute_weights()
            okern = orc.OracleMiniARDKernel(2 * nf, x.shape, sp, g[f"c{ci}_hyperparams"], 123,
                                            double_precision=dp, fit_intercept=icpt, ops=oracle)
            okern.precompute_weights()
            # FHT on the identity is exact integer arithmetic until the chi multiply: weights are bit-identical
            assert np.array_equal(kern.precomputed_weights.cpu().numpy(), okern.precomputed_weights)
            feats, grad = kern.gradient_x(x)
            ofeats, ograd = okern.gradient_x(x)
            assert np.allclose(feats.cpu().numpy(), ofeats, rtol=1e-9, atol=1e-11)     # cos/sin of a float64 argument
            assert np.allclose(grad.cpu().numpy(), ograd, rtol=1e-9, atol=1e-9)
            ref_f, ref_g = g[f"c{ci}_features"].copy(), g[f"c{ci}_grad"].copy()
            if icpt:
                ref_g[:, 0, :] = 0
                ref_f[:, 0] = 1.0
            assert np.allclose(feats.cpu().numpy(), ref_f, rtol=rtol_f, atol=atol_f)
            assert np.allclose(grad.cpu().numpy(), ref_g, rtol=rtol_g, atol=atol_g)
            if dp:
                assert np.allclose(kern.transform_x(x).cpu().numpy(), g[f"c{ci}_transform_x"], rtol=1e-6, atol=1e-7)
    assert isinstance(make_kernel("MiniARD", (10, 20), 64, 123, DEV, {"split_points": [5]}), MiniARDKernel)
    with pytest.raises(ValueError):
        make_kernel("MiniARD", (10, 20), 64, 123, DEV, {})
    # validation mirrors the reference's throws (ard_ops.cpp:68-84)
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    xt = torch.zeros((4, 6), dtype=torch.float32, device=DEV)
    w = torch.zeros((8, 6), dtype=torch.float32, device=DEV)
    mp = torch.zeros(6, dtype=torch.int32, device=DEV)
    sv = torch.ones(6, dtype=torch.float64, device=DEV)
    with pytest.raises(RuntimeError):
        ext.hipMiniARDGrad(xt, torch.zeros((4, 14), dtype=torch.float64, device=DEV), w, mp, sv,
                           torch.zeros((4, 14, 1), dtype=torch.float64, device=DEV), True)
    with pytest.raises(RuntimeError):
        ext.hipMiniARDGrad(xt, torch.zeros((4, 16), dtype=torch.float64, device=DEV), w, mp[:5], sv,
                           torch.zeros((4, 16, 1), dtype=torch.float64, device=DEV), True)


@pytest.mark.parametrize("n,m,rank", [(37, 8192, 512), (5, 300, 17), (3, 4, 2), (64, 16384, 2048), (9, 4097, 100)])
def test_srht_sample_equals_pad_srht_gather(oracle, n, m, rank):
    """hipSRHTSample == SRHTCompressor's reference formulation (srht_compressor.py:87-97: zero-pad, cudaSRHT
    in place, gather) bit for bit, and == the oracle's compressor; the input is left untouched."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import SRHTCompressor
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(n + m)
    z = rng.standard_normal((n, m))
    comp = SRHTCompressor(rank, m, device=DEV, random_seed=123)
    zd = torch.from_numpy(z).to(DEV)
    zcopy = zd.clone()
    fused = comp.transform_x(zd)
    assert torch.equal(zd, zcopy)
    # the reference formulation with the separate operators
    xf = torch.zeros((n, comp.padded_dims), dtype=torch.float64, device=DEV)
    xf[:, :m] = zd
    ext.hipSRHT(xf, comp.radem)
    assert torch.equal(fused, xf[:, comp.truncated_sampler])
    ocomp = orc.OracleSRHTCompressor(rank, m, random_seed=123, ops=oracle)
    assert np.array_equal(fused.cpu().numpy(), ocomp.transform_x(z))
    # the same pass can also deliver z^T y (the preconditioner's first pass)
    y = rng.standard_normal(n)
    zty = torch.full((m,), 3.0, dtype=torch.float64, device=DEV)
    both = comp.transform_x_zty(zd, torch.from_numpy(y).to(DEV), zty)
    assert torch.equal(both, fused) and torch.equal(zd, zcopy)
    ref = z.T @ y
    assert np.linalg.norm(zty.cpu().numpy() - ref) <= 1e-13 * max(np.linalg.norm(ref), 1e-300)
    # into a wider preallocated buffer (row pitch rank + 1)
    buf = torch.full((n, rank + 1), -7.0, dtype=torch.float64, device=DEV)
    view = comp.transform_x(zd, out=buf)
    assert torch.equal(view, fused) and bool((buf[:, rank] == -7.0).all())


def test_g14_two_layer_kernel_vs_reference():
    """Conv1dTwoLayer on the device (hipConv1dMaxpool -> sigma -> hipRBFFeatureGen / hipRBFGrad) against the
    reference's kernel class (tests/golden/g14_two_layer.npz)."""
    from xgpr_amd.kernels import make_kernel
    g = load_golden("g14_two_layer.npz")
    kern = make_kernel("Conv1dTwoLayer", g["x"].shape, int(g["num_rffs"]), 123, DEV,
                       {"conv_width": int(g["conv_width"]), "init_rffs": int(g["init_rffs"]), "intercept": True})
    kern.set_hyperparams(g["hyperparams"], logspace=False)
    scale = np.sqrt(1.0 / (kern.num_freqs - 0.5))
    feats = kern.transform_x(g["x"], g["seqlen"]).cpu().numpy()
    assert np.abs(feats - g["features"]).max() <= 4e-7 * scale
    f2, grad = kern.gradient_x(g["x"], g["seqlen"])
    assert np.abs(f2.cpu().numpy() - g["grad_features"]).max() <= 4e-7 * scale
    assert np.allclose(grad.cpu().numpy(), g["grad"], rtol=1e-5, atol=1e-5 * scale * np.abs(g["grad"]).max() / scale)
    with pytest.raises(ValueError):
        make_kernel("Conv1dTwoLayer", g["x"].shape, 128, 123, DEV, {"conv_width": 5})
