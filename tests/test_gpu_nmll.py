"""GPU parity of the "next" rows 1-2 of SURVEY.md section 8f: the block (k right-hand sides) CG
matvec on the float64 matrix cores, the approximate NMLL built on it, and the exact NMLL with its
gradient -- against the oracle and against values the REFERENCE's own Python produced
(tests/golden/g10_nmll.npz, reference xgp_regression.py:152-367).  Tolerances: 1e-5 relative
(BASELINE.json north_star); the contraction itself is held to 1e-12 against a float64 product over
the same cached features."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else a
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("n,d,m,k,icpt", [(700, 20, 256, 5, True), (257, 33, 2100, 26, False),
                                          (1, 8, 64, 1, True), (15, 8, 4, 3, True), (513, 64, 1024, 40, True),
                                          (600, 16, 36, 32, False), (2000, 32, 512, 26, True)])
def test_block_matvec_vs_oracle(oracle, n, d, m, k, icpt):
    """hipZCacheBlockMatvec over a cache built by hipRBFFeatureCache == the oracle's
    Z^T (Z V) (reference cg_tools.py:189-191 with V of k columns), ragged sizes included."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel, scale_input, block_workspace_bytes
    rng = np.random.default_rng(n + m + k)
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    hp = np.array([0.3, 0.8])
    v = rng.standard_normal((m, k))
    kern = make_kernel("RBF", x.shape, m, 123, DEV, {"intercept": icpt})
    kern.set_hyperparams(hp, logspace=False)
    xs = scale_input(torch.from_numpy(x).to(DEV), hp[1])
    zc = torch.empty((n, m), dtype=torch.float32, device=DEV)
    kern.fill_feature_cache(xs, zc)
    vd = torch.from_numpy(v).to(DEV)
    out = torch.full((m, k), 7.0, dtype=torch.float64, device=DEV)
    ws = torch.empty(block_workspace_bytes(n, m, k), dtype=torch.uint8, device=DEV)
    kern.ztz_block_cached(zc, vd, out, ws)
    okern = orc.OracleKernel("RBF", m, x.shape, hp, 123, fit_intercept=icpt, ops=oracle)
    z = okern.transform_x(x.astype(np.float64))
    ref = z.T @ (z @ v)
    assert rel(out, ref) < 1e-6        # the float32 features themselves agree to ~4e-7 of their scale
    # the contraction alone: float64 torch product over the very same cached float32 features
    scale = float(np.float32(np.sqrt(1.0 / (m // 2 - 0.5 if icpt else m // 2))))
    zd = zc.double() * scale
    if icpt:
        zd[:, 0] = 1.0
    ref_same = (zd.T @ (zd @ vd)).cpu().numpy()
    assert rel(out, ref_same) < 1e-12
    # accumulate: a second call adds the same product
    kern.ztz_block_cached(zc, vd, out, ws, accumulate=True)
    assert rel(out, 2 * ref_same) < 1e-12
    # deterministic
    out2 = torch.empty_like(out)
    kern.ztz_block_cached(zc, vd, out2, ws)
    kern.ztz_block_cached(zc, vd, out, ws)
    assert torch.equal(out, out2)


def test_block_matvec_rejects_bad_arguments():
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    zc = torch.zeros((8, 6), dtype=torch.float32, device=DEV)       # num_rffs not a multiple of 4
    v = torch.zeros((6, 2), dtype=torch.float64, device=DEV)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError):
        ext.hipZCacheBlockMatvec(zc, v, torch.empty_like(v), True, ws)
    zc = torch.zeros((8, 8), dtype=torch.float32, device=DEV)
    v = torch.zeros((8, 33), dtype=torch.float64, device=DEV)       # more than 32 columns per call
    with pytest.raises(RuntimeError):
        ext.hipZCacheBlockMatvec(zc, v, torch.empty_like(v), True, ws)
    v = torch.zeros((8, 4), dtype=torch.float64, device=DEV)
    with pytest.raises(RuntimeError):                               # workspace too small
        ext.hipZCacheBlockMatvec(zc, v, torch.empty_like(v), True, ws[:16])
    with pytest.raises(TypeError):                                  # float32 right-hand sides
        ext.hipZCacheBlockMatvec(zc, v.float(), torch.empty_like(v), True, ws)


def test_block_cg_conv_kernel_matches_oracle(oracle):
    """Batched right-hand sides with a convolution kernel: per-chunk float32 feature rows through the
    block matvec (scale 1) vs the oracle's batched CG."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import ConjugateGrad
    rng = np.random.default_rng(15)
    n, L, C, m, k = 300, 20, 6, 128, 4
    x = rng.standard_normal((n, L, C)).astype(np.float32)
    sl = rng.integers(5, L + 1, size=n).astype(np.int32)
    y = rng.standard_normal(n)
    hp = np.array([0.6, 0.7])
    rhs = rng.standard_normal((m, k))
    for cache in (False, True):
        ds = build_regression_dataset(x, y, sl, chunk_size=100, device=DEV)
        kern = make_kernel("Conv1dRBF", x.shape, m, 123, DEV, {"conv_width": 5, "averaging": "full"})
        kern.set_hyperparams(hp, logspace=False)
        resid = torch.zeros((m, 2, k), dtype=torch.float64, device=DEV)
        resid[:, 0, :] = torch.from_numpy(rhs).to(DEV)
        xk, conv, niter, _ = ConjugateGrad(cache_features=cache).fit(ds, kern, None, resid, 300, 1e-9, False)
        ods = orc.OracleDataset(x.astype(np.float64), y, sl, chunk_size=100)
        okern = orc.OracleKernel("Conv1dRBF", m, x.shape, hp, 123, conv_width=5, averaging="full", ops=oracle)
        oresid = np.zeros((m, 2, k))
        oresid[:, 0, :] = rhs
        xref, oconv, oniter, _ = orc.cg_fit(ods, okern, None, oresid, 300, 1e-9)
        assert conv and oconv and abs(niter - oniter) <= 1
        assert rel(xk, xref) < 1e-5


@pytest.mark.parametrize("tag", ["easy", "hard"])
def test_g10_nmll_vs_reference(tag):
    """exact_nmll, exact_nmll_gradient and approximate_nmll on the reference fixture against the
    reference's own numbers; the 26-column solve runs on the block matvec (regenerating windows and
    resident cache both)."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd import nmll
    g8, g = load_golden("g8_e2e.npz"), load_golden("g10_nmll.npz")
    x, y = g8["xtrain"], g8["ytrain"]
    ds = build_regression_dataset(x, y, chunk_size=2000, device=DEV)
    kern = make_kernel("RBF", x.shape, 512, 123, DEV, {"intercept": True})
    kern.set_hyperparams(g[f"{tag}_hparam_log"], logspace=True)
    exact = float(g[f"{tag}_exact_nmll"])
    assert np.isclose(nmll.exact_nmll(kern, ds), exact, rtol=1e-6)
    nll, grad = nmll.exact_nmll_gradient(kern, ds)
    assert np.isclose(nll, float(g[f"{tag}_grad_nmll"]), rtol=1e-6)
    assert np.allclose(grad, g[f"{tag}_grad"], rtol=2e-4, atol=1e-4)
    pre = RandNysPreconditioner(kern, ds, 64, False, 123, "srht_2")
    assert np.isclose(pre.get_logdet(), float(g[f"{tag}_precond_logdet"]), rtol=1e-6)
    settings = {"nsamples": 25, "nmll_iter": 500, "nmll_tol": 1e-6}
    for cache in (False, True):
        det = {}
        approx = nmll.approximate_nmll(kern, ds, pre, settings, 123, cache_features=cache, details=det)
        assert rel(det["probes"], g[f"{tag}_probes"]) < 1e-6
        na = g[f"{tag}_alphas"].shape[0]
        assert abs(det["niter"] - na) <= 1
        nc = min(na, det["niter"], 8)
        assert np.allclose(det["alphas"].cpu().numpy()[:nc], g[f"{tag}_alphas"][:nc], rtol=1e-5)
        assert np.allclose(det["betas"].cpu().numpy()[:nc], g[f"{tag}_betas"][:nc], rtol=1e-5, atol=1e-12)
        assert rel(det["weights"], g[f"{tag}_xk0"]) < 1e-5
        assert np.isclose(det["logdet"], float(g[f"{tag}_logdet"]), rtol=1e-5)
        assert np.isclose(approx, float(g[f"{tag}_approx_nmll"]), rtol=1e-6)
        assert 100 * abs(approx - exact) / exact < 1.0     # the reference's acceptance bar (test_slq_nmll.py:73-79)


def test_nmll_gradient_matches_finite_difference_of_exact_nmll():
    """Property test at a size the goldens do not cover (Matern, no intercept): d NMLL / d log sigma
    from exact_nmll_gradient vs a central difference of exact_nmll."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd import nmll
    rng = np.random.default_rng(21)
    n, d, m = 600, 12, 128
    x = rng.uniform(-1, 1, size=(n, d))
    y = np.sin(x @ rng.standard_normal(d)) + 0.1 * rng.standard_normal(n)
    ds = build_regression_dataset(x, y, chunk_size=250, device=DEV)
    kern = make_kernel("Matern", x.shape, m, 123, DEV, {"matern_nu": 1.5, "intercept": False})
    hp = np.array([-0.5, 0.2])
    kern.set_hyperparams(hp, logspace=True)
    _, grad = nmll.exact_nmll_gradient(kern, ds)
    eps = 1e-3
    fd = np.zeros(2)
    for i in range(2):
        hi, lo = hp.copy(), hp.copy()
        hi[i] += eps
        lo[i] -= eps
        kern.set_hyperparams(hi, logspace=True)
        fhi = nmll.exact_nmll(kern, ds)
        kern.set_hyperparams(lo, logspace=True)
        flo = nmll.exact_nmll(kern, ds)
        fd[i] = (fhi - flo) / (2 * eps)
    assert np.allclose(grad, fd, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("method,max_iter", [("Nelder-Mead", 100), ("Powell", 100), ("L-BFGS-B", 100)])
def test_tuning_reaches_the_reference_bar(method, max_iter):
    """tune_hyperparams (reference xgp_regression.py:564-727) on the reference fixture with the settings of its
    tests/tuning_tests/test_tuning.py:18-40 (RBF, 512 RFFs, start at (0, 0), exact NMLL): best score < 430."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.tuning import tune_hyperparams
    g8 = load_golden("g8_e2e.npz")
    x, y = g8["xtrain"], g8["ytrain"]
    ds = build_regression_dataset(x, y, chunk_size=2000, device=DEV)
    kern = make_kernel("RBF", x.shape, 512, 123, DEV, {"intercept": True})
    hp, nfev, best = tune_hyperparams(kern, ds, tuning_method=method, n_restarts=1,
                                      starting_hyperparams=np.array([0., 0.]), max_iter=max_iter, nmll_method="exact")
    assert best < 430
    assert np.allclose(kern.get_hyperparams(), hp)


def test_g15_crude_tuning_vs_reference():
    """shared_hparam_search (lambda grid from one eigendecomposition of Z^T Z) at fixed sigmas, and the whole
    tune_hyperparams_crude loop, against the reference's values on its fixture (tests/golden/g15_crude_tuning.npz).
    Scores are rounded to 3 decimals by the algorithm, grid points are discrete: exact grid agreement is expected."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.crude_tuning import shared_hparam_search, tune_hyperparams_crude
    from xgpr_amd.tuning import default_bounds
    g8, g = load_golden("g8_e2e.npz"), load_golden("g15_crude_tuning.npz")
    ds = build_regression_dataset(g8["xtrain"], g8["ytrain"], chunk_size=2000, device=DEV)
    kern = make_kernel("RBF", g8["xtrain"].shape, 512, 123, DEV, {"intercept": True})
    bounds = default_bounds(kern)
    assert np.allclose(bounds, g["bounds"])
    for s, score, lb in zip(g["sigmas"], g["scores"], g["best_lbs"]):
        got_score, got_lb = shared_hparam_search(np.array([s]), kern, ds, bounds[:1, :])
        assert abs(got_score - float(score)) <= 2e-3
        assert np.isclose(got_lb[0], float(lb), atol=1e-6)
    hp, nfev, best = tune_hyperparams_crude(kern, ds)
    assert abs(best - float(g["crude_best"])) < 0.5 and best < 430
    assert np.allclose(kern.get_hyperparams(), hp)
    # the acquisitions themselves may differ (a score that differs in the third decimal changes the surrogate);
    # the optimum found must be as good and in the same place
    assert np.abs(hp - g["crude_hparams"]).max() < 0.2


def test_windowed_block_matvec_with_a_ragged_last_window(monkeypatch):
    """A block of right-hand sides without a resident cache, over more rows than one window of regenerated feature rows: the
    last, shorter window needs a LARGER workspace than a full one (the split projection's partials are reserved for short
    launches only: xgpr_zcache_block_workspace_bytes(65536, ., 8) < (33920, ., 8)), which the round-5 tree sized once per solve.
    Windows of 65536 rows over 65536 + 33920 rows, against the same product on one window."""
    from xgpr_amd.kernels import make_kernel, block_workspace_bytes
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import ConjugateGrad
    n, d, m, k = 65536 + 33920, 24, 2048, 8
    assert block_workspace_bytes(33920, m, k) > block_workspace_bytes(65536, m, k), "the case must be the non-monotone one"
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, d, device=DEV, generator=g) / d ** 0.5
    y = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
    ds = build_regression_dataset(x, y, chunk_size=16384, device=DEV)
    kern = make_kernel("RBF", (n, d), m, 123, DEV, {})
    kern.set_hyperparams(np.array([0.5, 1.0]), logspace=False)
    vec = torch.randn(m, k, dtype=torch.float64, device=DEV, generator=g)
    outs = []
    for wbytes in (65536 * 4 * m, 8 << 30):
        monkeypatch.setattr(ConjugateGrad, "BLOCK_WINDOW_BYTES", wbytes)
        cg = ConjugateGrad(cache_features=False)
        out = torch.zeros_like(vec)
        cg._matvec(ds, kern, vec, out, add_ridge=False)
        outs.append(out)
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-10 * float(outs[1].abs().max())
    z = kern.transform_x(x[:4096])
    ref_part = z.T @ (z @ vec)            # sanity on a slice: same operator
    cg = ConjugateGrad(cache_features=False)
    ds2 = build_regression_dataset(x[:4096], y[:4096], chunk_size=4096, device=DEV)
    o2 = torch.zeros_like(vec)
    cg._matvec(ds2, kern, vec, o2, add_ridge=False)
    assert float((o2 - ref_part).abs().max()) <= 1e-6 * float(ref_part.abs().max())


@pytest.mark.parametrize("m,k,with_pre", [(512, 26, True), (2100, 3, False), (12288, 26, True), (4096, 2, True)])
def test_block_device_solve_equals_the_generic_loop(m, k, with_pre, monkeypatch):
    """The batched solve with its vector updates in two kernels per iteration (hipCGStep1Block / hipCGStep2Block, errors
    read one iteration behind) against the generic loop of torch operations: same iteration count, alphas / betas /
    iterates to rounding (the two forms differ in the summation order of the dot products and in the algebraic form
    of the preconditioner apply), with and without a preconditioner, M below and above one batch of the kernels, in
    both return conventions of ``fit``."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import ConjugateGrad
    rng = np.random.default_rng(m + k)
    n, d = 900, 40
    x = rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d)
    y = np.sin(x @ rng.standard_normal(d)) + 0.1 * rng.standard_normal(n)
    ds = build_regression_dataset(x, y, chunk_size=300, device=DEV)
    kern = make_kernel("RBF", x.shape, m, 123, DEV, {})
    # (without a preconditioner a larger ridge keeps the recurrence from amplifying the 1e-16 differences between the two
    # forms within the compared iterations: 900 rows against up to 12288 features is rank-deficient at lambda = 0.4)
    kern.set_hyperparams(np.array([0.4 if with_pre else 4.0, 1.1]), logspace=False)
    pre = RandNysPreconditioner(kern, ds, 48, False, 123, "srht") if with_pre else None
    rhs = torch.from_numpy(rng.standard_normal((m, k))).to(DEV)
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(ConjugateGrad, "BLOCK_DEVICE_SOLVE", fused)
        for nm in (True, False):
            resid = torch.zeros((m, 2, k), dtype=torch.float64, device=DEV)
            resid[:, 0, :] = rhs
            out[(fused, nm)] = ConjugateGrad(cache_features=True).fit(ds, kern, pre, resid, 400, 1e-8, False, nmll_settings=nm)
    xa, al_a, be_a = out[(True, True)]
    xb, al_b, be_b = out[(False, True)]
    # (iteration counts: equal, or one apart when the last error sits on the tolerance -- the two forms round differently)
    assert abs(al_a.shape[0] - al_b.shape[0]) <= 1 and al_a.shape[1] == al_b.shape[1] == k - 1
    assert be_a.shape == al_a.shape and be_b.shape == al_b.shape
    # Once a column's residual reaches its rounding floor its alpha / beta are quotients of noise (measured: the two forms
    # agree to 1e-16 for five iterations, then 1e-11, 1e-6, 0.2 as the 3-column solve at lambda = 4 converges): compare
    # the iterations in which every residual is still resolved
    early = min(al_a.shape[0], al_b.shape[0], 5)
    assert rel(al_a[:early], al_b[:early].cpu().numpy()) < 1e-9 and rel(be_a[:early], be_b[:early].cpu().numpy()) < 1e-8
    assert rel(xa, xb.cpu().numpy()) < 1e-7
    xc, conv_c, nit_c, loss_c = out[(True, False)]
    xd, conv_d, nit_d, loss_d = out[(False, False)]
    assert conv_c and conv_d and nit_c == al_a.shape[0] and nit_d == al_b.shape[0]
    assert len(loss_c) == nit_c and len(loss_d) == nit_d
    assert rel(xc, xd.cpu().numpy()) < 1e-7 and np.allclose(loss_c[:early], loss_d[:early], rtol=1e-6)
    assert torch.equal(xc, xa)                         # the two return conventions are one solve


@pytest.mark.parametrize("m,rank,k", [(8192, 512, 26), (2100, 48, 3), (1000, 513, 32), (64, 7, 1), (12288, 1030, 9)])
def test_utr_block_kernel_equals_the_product(m, rank, k):
    """hipPrecondUtRBlock (U^T R for a block of right-hand sides, the first product of
    rand_nys_preconditioners.py:66-72) against torch's float64 product; odd ranks, more columns than one pass of the
    kernel's 512, row counts that do not divide the blocks."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = torch.Generator(device=DEV).manual_seed(m + rank + k)
    u = torch.randn(m, rank, dtype=torch.float64, device=DEV, generator=g)
    r = torch.randn(m, k, dtype=torch.float64, device=DEV, generator=g)
    t = torch.full((rank, k), float("nan"), dtype=torch.float64, device=DEV)
    ext.hipPrecondUtRBlock(u, r, t)
    ref = u.T @ r
    assert float((t - ref).abs().max()) <= 1e-12 * float(ref.abs().max()) * np.sqrt(m)
    t2 = torch.empty_like(t)
    ext.hipPrecondUtRBlock(u, r, t2)
    assert torch.equal(t, t2)                           # fixed summation order
    with pytest.raises(RuntimeError):
        ext.hipPrecondUtRBlock(u, torch.zeros(m, 33, dtype=torch.float64, device=DEV), torch.zeros(rank, 33, dtype=torch.float64, device=DEV))


@pytest.mark.parametrize("m,rank,k", [(8192, 512, 26), (2100, 48, 3), (1000, 513, 32), (70, 7, 1), (12288, 1030, 9), (4097, 130, 16)])
def test_block_preconditioner_apply_equals_the_reference_form(m, rank, k):
    """hipPrecondApplyBlock (both products on the float64 matrix cores) against batch_matvec as the reference writes it
    (rand_nys_preconditioners.py:66-72: xprod2 + xprod1, three float64 products); ranks that are not multiples of 4 or
    16, row counts that are not multiples of 16, one to thirty-two right-hand sides."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = torch.Generator(device=DEV).manual_seed(m + rank + k)
    u = torch.linalg.qr(torch.randn(m, rank, dtype=torch.float64, device=DEV, generator=g))[0].contiguous()
    inv_eig = 1.0 / (torch.rand(rank, dtype=torch.float64, device=DEV, generator=g) * 50 + 0.1)
    pref = 0.37
    r = torch.randn(m, k, dtype=torch.float64, device=DEV, generator=g)
    z = torch.full((m, k), float("nan"), dtype=torch.float64, device=DEV)
    ext.hipPrecondApplyBlock(u, inv_eig, pref, r, z)
    xprod = u.T @ r
    ref = (r - u @ xprod) + u @ (inv_eig[:, None] * pref * xprod)
    assert float((z - ref).abs().max()) <= 1e-12 * float(ref.abs().max()) * np.sqrt(m)
    z2 = torch.empty_like(z)
    ext.hipPrecondApplyBlock(u, inv_eig, pref, r, z2)
    assert torch.equal(z, z2)
