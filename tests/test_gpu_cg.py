"""GPU parity of the solver side of the path -- transform_x, z^T y, the randomized-Nystrom
preconditioner and preconditioned CG -- against the values the REFERENCE's own Python
produced (tests/golden/g7_cg.npz: BASELINE cfg1-sized problem; g8_e2e.npz: the reference's
381x84 fixture with the settings of its tests/fitting_tests/test_cg_fit.py:26-40).
Bar (BASELINE.json north_star): CG iterates within 1e-5 relative; iteration counts equal."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.cpu().numpy() if isinstance(b, torch.Tensor) else b
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("kname,parms", [("RBF", {}), ("Matern", {"matern_nu": 5 / 2})])
def test_g7_cg_iterates(kname, parms):
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import cg_fit_lib_internal, calc_zty
    g = load_golden("g7_cg.npz")
    x, y = g["x"], g["y"]
    ds = build_regression_dataset(x, y, chunk_size=int(g["chunk_size"]), device=DEV)
    assert np.isclose(ds.get_ymean(), float(g["y_mean"]), rtol=1e-12)
    assert np.isclose(ds.get_ystd(), float(g["y_std"]), rtol=1e-12)
    kern = make_kernel(kname, x.shape, int(g["num_rffs"]), 123, DEV, parms)
    kern.set_hyperparams(g["hyperparams"], logspace=False)
    z8 = kern.transform_x(x[:8]).cpu().numpy()
    scale = np.sqrt(1.0 / (kern.num_freqs - 0.5))
    assert np.abs(z8 - g[f"{kname}_z_first8"]).max() <= 4e-7 * scale
    zty, yty = calc_zty(ds, kern)          # fused z^T y kernel
    assert rel(zty, g[f"{kname}_zty"]) < 1e-6
    assert np.isclose(yty, float(g[f"{kname}_yty"]), rtol=1e-10)
    for ptag, method in [("none", None), ("srht", "srht"), ("srht2", "srht_2")]:
        pre = None
        if method is not None:
            pre = RandNysPreconditioner(kern, ds, 64, False, 123, method)
            assert np.allclose(pre.eig.cpu().numpy(), g[f"{kname}_{ptag}_eig"], rtol=1e-5)
            assert np.isclose(pre.achieved_ratio, float(g[f"{kname}_{ptag}_ratio"]), rtol=1e-4)
            assert rel(pre.get_zty(), g[f"{kname}_{ptag}_zty"]) < 1e-6
            u_ref = g[f"{kname}_{ptag}_u"]
            v = np.linspace(-1, 1, u_ref.shape[0])
            pu = (pre.u_mat @ (pre.u_mat.T @ torch.from_numpy(v).to(DEV))).cpu().numpy()
            assert np.allclose(pu, u_ref @ (u_ref.T @ v), atol=1e-5)
        ref_it = g[f"{kname}_{ptag}_iterates"]
        trace = {}
        w, niter, losses = cg_fit_lib_internal(kern, ds, 1e-30, ref_it.shape[0], pre, False, trace=trace,
                                               cache_features=False)
        n = ds.get_ndatapoints()
        errs = np.array([rel(trace["x_k"][j][:, 0] * n, ref_it[j]) for j in range(ref_it.shape[0])])
        print(f"{kname} {ptag}: iterate rel err " + " ".join(f"{e:.1e}" for e in errs))
        if pre is not None:
            assert errs.max() <= 1e-5, (ptag, errs)
        else:
            # Un-preconditioned CG on this problem passes through a near-breakdown (two almost dependent
            # search directions) that tests/test_cg_sensitivity.py pins on the CPU oracle: iterates outside
            # the window follow the size of a perturbation of Z, the ones inside move by 1e-4..2e-2 whatever
            # its size.  The bound is the envelope measured on the oracle at THIS run's feature error.
            import cg_sensitivity as cs
            zall = kern.transform_x(x).cpu().numpy()
            zor, _, _, _ = cs.oracle_problem(g, kname)
            eps = max(float(np.abs(zall - zor).max() / scale), 3e-8)
            assert eps <= 4e-7, eps
            env, count_range = cs.envelope(g, kname, eps, seeds=range(12))
            print(f"{kname} none: feature error {eps:.1e} -> oracle envelope " + " ".join(f"{e:.1e}" for e in env))
            outside = [j for j in range(len(errs)) if j not in cs.WINDOW]
            assert errs[outside].max() <= 1e-5, (ptag, errs)
            # inside the window the error is 1 / (a near-zero denominator): heavy-tailed, hence the factor on the 12-sample maximum
            assert np.all(errs <= np.maximum(1e-5, 10.0 * env)), (ptag, errs, env)
        nl = min(len(losses), len(g[f"{kname}_{ptag}_losses"]))
        if pre is None:
            nl = min(nl, cs.WINDOW[0])      # before the near-breakdown window
        assert np.allclose(losses[:nl], g[f"{kname}_{ptag}_losses"][:nl], rtol=1e-4)
        # full solve to the reference's tolerance: same iteration count, same weights
        w, niter, losses = cg_fit_lib_internal(kern, ds, 1e-8, 500, pre, False)
        # preconditioned counts are exact +-1; the un-preconditioned solve (100+ iterations through the
        # near-breakdown) must end inside the range of counts the perturbed oracle solves end in, +-1
        if pre is not None:
            assert abs(niter - int(g[f"{kname}_{ptag}_niter"])) <= 1, (ptag, niter)
        else:
            assert count_range[0] - 1 <= niter <= count_range[1] + 1, (ptag, niter, count_range)
        assert rel(w, g[f"{kname}_{ptag}_weights"]) < 1e-5


def test_g8_reference_fixture_end_to_end():
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import cg_fit_lib_internal
    g = load_golden("g8_e2e.npz")
    x, y = g["xtrain"], g["ytrain"]
    ds = build_regression_dataset(x, y, chunk_size=2000, device=DEV)
    kern = make_kernel("RBF", x.shape, 4096, 123, DEV, {"intercept": True})
    kern.set_hyperparams(g["hparam_log"], logspace=True)
    pre = RandNysPreconditioner(kern, ds, 256, False, 123, "srht")
    assert np.isclose(pre.achieved_ratio, float(g["ratio"]), rtol=1e-4)
    w, niter, losses = cg_fit_lib_internal(kern, ds, 1e-6, 500, pre, False)
    assert niter == int(g["niter"]) and niter < 10
    assert rel(w, g["weights"]) < 1e-5
    z = kern.transform_x(g["xtest"])
    preds = (z @ w).cpu().numpy() * ds.get_ystd() + ds.get_ymean()
    assert np.allclose(preds, g["preds"], rtol=1e-5, atol=1e-6)
    # the fused predictor (float32 feature rows -> one-column projection on the matrix cores, no float64 Z) against
    # the reference's predictions and against the materialised product, over ragged chunks
    from xgpr_amd.exact import predict_mean
    for chunk in (2000, 17):
        pm = predict_mean(kern, w, g["xtest"], ds.get_ymean(), ds.get_ystd(), chunk_size=chunk).cpu().numpy()
        assert np.allclose(pm, g["preds"], rtol=1e-5, atol=1e-6)
        assert np.abs(pm - preds).max() <= 1e-12 * max(1.0, np.abs(preds).max())


def test_conv_kernel_cg_matches_oracle(oracle):
    """Conv1d kernel through the un-fused chunk path (Z materialised per chunk) vs the oracle."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import cg_fit_lib_internal
    rng = np.random.default_rng(5)
    n, L, C = 300, 24, 8
    x = rng.standard_normal((n, L, C)).astype(np.float32)
    sl = rng.integers(5, L + 1, size=n).astype(np.int32)
    y = rng.standard_normal(n)
    hp = np.array([0.5, 0.7])
    ds = build_regression_dataset(x, y, sl, chunk_size=128, device=DEV)
    kern = make_kernel("Conv1dRBF", x.shape, 256, 123, DEV, {"conv_width": 5, "averaging": "sqrt"})
    kern.set_hyperparams(hp, logspace=False)
    w, niter, _ = cg_fit_lib_internal(kern, ds, 1e-9, 300, None, False)
    ods = orc.OracleDataset(x.astype(np.float64), y, sl, chunk_size=128)
    okern = orc.OracleKernel("Conv1dRBF", 256, x.shape, hp, 123, conv_width=5, averaging="sqrt", ops=oracle)
    wref, nref, _, _ = orc.cg_fit_lib_internal(okern, ods, 1e-9, 300, None)
    assert abs(niter - nref) <= 1
    assert rel(w, wref) < 1e-5


def test_precond_apply_and_fused_cg_steps():
    """hipPrecondApply == RandNysPreconditioner.batch_matvec (rand_nys_preconditioners.py:66-72);
    hipCGStep1/2 == the vector updates of cg_tools.py:256-274 -- float64, 1e-12 relative."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(3)
    for m, rank in [(8192, 512), (1000, 37), (300, 64)]:
        u, _ = np.linalg.qr(rng.standard_normal((m, rank)))
        inv_eig = 1.0 / rng.uniform(0.1, 5.0, size=rank)
        pref = 0.37
        r = rng.standard_normal(m)
        ref = u @ (inv_eig * pref * (u.T @ r)) + (r - u @ (u.T @ r))
        z = torch.zeros(m, dtype=torch.float64, device=DEV)
        ext.hipPrecondApply(torch.from_numpy(u).to(DEV), torch.from_numpy(inv_eig).to(DEV), pref,
                            torch.from_numpy(r).to(DEV), z)
        assert rel(z, ref) < 1e-12
        w, p, x, zz = (rng.standard_normal(m) for _ in range(4))
        lam2, init_norm = 0.01, 3.3
        wd, pd, xd, rd, zd = (torch.from_numpy(a.copy()).to(DEV) for a in (w, p, x, r, zz))
        rn = torch.empty(m, dtype=torch.float64, device=DEV)
        scal = torch.zeros(4, dtype=torch.float64, device=DEV)
        ext.hipCGStep1(wd, pd, xd, rd, rn, zd, scal, lam2, init_norm)
        w2 = w + lam2 * p
        rz = (r * zz).sum()
        alpha = rz / (p * w2).sum()
        assert rel(wd, w2) < 1e-14 and rel(xd, x + alpha * p) < 1e-12 and rel(rn, r - alpha * w2) < 1e-12
        s = scal.cpu().numpy()
        assert np.isclose(s[0], rz, rtol=1e-12) and np.isclose(s[1], alpha, rtol=1e-12)
        assert np.isclose(s[2], np.linalg.norm(r) / init_norm, rtol=1e-12)
        znext = rng.standard_normal(m)
        pn = torch.empty(m, dtype=torch.float64, device=DEV)
        ext.hipCGStep2(rn, torch.from_numpy(znext).to(DEV), pd, pn, scal)
        beta = ((r - alpha * w2) * znext).sum() / rz
        assert rel(pn, znext + beta * p) < 1e-12
        assert np.isclose(scal.cpu().numpy()[3], beta, rtol=1e-10)


def test_g9_exact_fit_and_variance():
    """mode="exact" (reference fitting_toolkit/exact_fitting_toolkit.py:16-72) on the reference
    fixture: weights, variance matrix and predictions against the reference's own run."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.exact import calc_weights_exact, calc_variance_exact, predict_mean
    g8, g9 = load_golden("g8_e2e.npz"), load_golden("g9_exact.npz")
    x, y = g8["xtrain"], g8["ytrain"]
    ds = build_regression_dataset(x, y, chunk_size=2000, device=DEV)
    kern = make_kernel("RBF", x.shape, 512, 123, DEV, {"intercept": True})
    kern.set_hyperparams(g9["hparam_log"], logspace=True)
    from xgpr_amd.exact import gram_route, calc_design_mat
    assert gram_route(ds, kern, 512) is False          # Z^T Z on the matrix cores from regenerated float32 windows
    w, _, _ = calc_weights_exact(ds, kern)
    assert rel(w, g9["weights"]) < 1e-5
    # ... and the accumulated design matrix itself against the reference's formulation on float64 features
    ztz, zty, yty = calc_design_mat(ds, kern)
    z = kern.transform_x(x)
    yn = ds.normalized_y()
    assert rel(ztz, z.T @ z) < 1e-12 and rel(zty, z.T @ yn) < 1e-12 and abs(yty - float(yn @ yn)) < 1e-9 * yty
    assert gram_route(ds, kern, 12) is None            # 12 variance features: not whole tiles, the float64 formulation
    var = calc_variance_exact(kern, ds, 12)
    assert rel(var, g9["var"]) < 1e-5
    v128 = calc_variance_exact(kern, ds, 128)          # a whole tile: the leading block through the gram kernel
    blk = z[:, :128].T @ z[:, :128]
    blk.diagonal().add_(float(kern.get_lambda()) ** 2)
    assert rel(v128, torch.linalg.pinv(blk, hermitian=True)) < 1e-8
    preds = predict_mean(kern, w, torch.from_numpy(g9["xtest"]).to(DEV), ds.get_ymean(), ds.get_ystd())
    assert np.allclose(preds.cpu().numpy(), g9["preds"], rtol=1e-5, atol=1e-6)


def test_batched_rhs_cg_matches_oracle(oracle):
    """k > 1 right-hand sides (the shape of the reference's NMLL probes, cg_tools.py:203-302 with
    resid [M, 2, k]): chunked block-GEMM matvec on the device vs the oracle's batched CG."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import ConjugateGrad
    rng = np.random.default_rng(8)
    n, d, m, k = 1500, 20, 256, 5
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    y = rng.standard_normal(n)
    hp = np.array([0.4, 0.9])
    rhs = rng.standard_normal((m, k))
    ds = build_regression_dataset(x, y, chunk_size=400, device=DEV)
    kern = make_kernel("RBF", x.shape, m, 123, DEV, {})
    kern.set_hyperparams(hp, logspace=False)
    resid = torch.zeros((m, 2, k), dtype=torch.float64, device=DEV)
    resid[:, 0, :] = torch.from_numpy(rhs).to(DEV)
    xk, conv, niter, losses = ConjugateGrad().fit(ds, kern, None, resid, 300, 1e-9, False)
    ods = orc.OracleDataset(x.astype(np.float64), y, chunk_size=400)
    okern = orc.OracleKernel("RBF", m, x.shape, hp, 123, ops=oracle)
    oresid = np.zeros((m, 2, k))
    oresid[:, 0, :] = rhs
    xref, oconv, oniter, _ = orc.cg_fit(ods, okern, None, oresid, 300, 1e-9)
    assert conv and oconv and abs(niter - oniter) <= 1
    assert rel(xk, xref) < 1e-6


def test_auto_cache_mode_keeps_features_resident_only_where_streaming_them_is_faster(monkeypatch):
    """cache_features="auto" follows the matvec launcher's own plan (xgpr_ztz_matvec_plan): one right-hand side on the
    single-pass three-wave kernel regenerates (a long shard: at least as fast as the HBM stream of the cache); every
    other plan -- the two-wave kernel (one tile per datapoint at padded width >= 128, seven tiles), two feature
    passes (num_freqs > 8192, or eight tiles per datapoint) --, a short shard, a block of right-hand sides or a convolution kernel keep the float32
    features resident (when they fit).  Measured: tools/cache_rule_probe.py."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd import cg as cgmod, xgpr_hip_rfgen_ext as ext
    from xgpr_amd.cg import _resolve_cache_mode
    rng = np.random.default_rng(0)
    assert [ext.ztz_matvec_plan(d, f) for d, f in ((1024, 4096), (256, 2048), (256, 1024), (1024, 8192), (512, 5120),
                                                    (64, 2048), (20, 1024), (1022, 4096), (512, 16384), (2000, 4096),
                                                    (2000, 2000), (1025, 100), (4000, 4096), (2000, 4097), (4000, 8192), (5000, 4096))] == \
        [1, 1, 2, 3, 1, 1, 1, 1, 3, 1, 1, 1, 1, 3, 3, 0]
    monkeypatch.setattr(cgmod, "SMALL_SHARD_ROWS", 100)          # the 256-row shards below count as long ones
    pays = {}
    for d, m in ((256, 4096), (256, 2048), (1024, 16384), (64, 4096), (20, 2048), (254, 4096), (256, 32768), (2003, 4000), (4000, 8192)):
        x = rng.standard_normal((256, d)).astype(np.float32)
        ds = build_regression_dataset(x, rng.standard_normal(256), chunk_size=128, device=DEV)
        k = make_kernel("RBF", x.shape, m, 123, DEV, {})
        k.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
        assert k.cache_ok()
        pays[(d, m)] = k.cache_pays()
        assert _resolve_cache_mode("auto", k, ds) is k.cache_pays()
        assert _resolve_cache_mode("auto", k, ds, block=True) is True
        assert _resolve_cache_mode(True, k, ds) is True and _resolve_cache_mode(False, k, ds) is False
    assert pays == {(256, 4096): False, (256, 2048): True, (1024, 16384): True, (64, 4096): False, (20, 2048): False,
                    (254, 4096): False, (256, 32768): True,
                    # padded widths 2048 / 4096 (wide transforms: 1.8-2.4 ns per tile regenerated against 1.33 streamed)
                    (2003, 4000): True, (4000, 8192): True}
    monkeypatch.setattr(cgmod, "SMALL_SHARD_ROWS", 200_000)      # ... and as short ones: the stream's smaller cost per launch wins
    x = rng.standard_normal((256, 256)).astype(np.float32)
    ds = build_regression_dataset(x, rng.standard_normal(256), chunk_size=128, device=DEV)
    k = make_kernel("RBF", x.shape, 4096, 123, DEV, {})
    k.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
    assert not k.cache_pays() and _resolve_cache_mode("auto", k, ds) is True
    # ... but not below padded width 128, where regenerating wins on short shards too (profiles/r6_cache_rule_125k.json)
    x = rng.standard_normal((256, 64)).astype(np.float32)
    ds = build_regression_dataset(x, rng.standard_normal(256), chunk_size=128, device=DEV)
    k = make_kernel("RBF", x.shape, 8192, 123, DEV, {})
    k.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
    assert not k.cache_pays() and _resolve_cache_mode("auto", k, ds) is False


@pytest.mark.parametrize("d,m,method", [(1500, 4096, "srht"), (2600, 8192, "srht_2"), (1100, 2048, "srht")])
def test_fit_beyond_1024_features_equals_the_materialised_float64_path(d, m, method, monkeypatch):
    """d > 1024 (padded width 2048 / 4096): the whole solve -- z^T y, the preconditioner's passes over float32 feature rows, the
    fused / cached CG matvec, and the k = 26 block matvec -- on the wave-tile kernels (round 6) against the SAME solve with the
    kernel's fused paths switched off, i.e. what rounds 1-5 ran there and what the reference does: float64 Z materialised chunk by
    chunk by the any-width operator (an independent implementation: one workgroup per row, butterflies in LDS), library GEMV /
    GEMM.  Same iteration count, weights to 1e-6 (the two feature matrices agree to 4e-7 x scale per entry), z^T y to 1e-7."""
    from xgpr_amd.kernels import make_kernel, SORFKernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import cg_fit_lib_internal, calc_zty, ConjugateGrad
    rng = np.random.default_rng(d)
    n = 6000
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    y = np.sin(3.0 * x @ rng.standard_normal(d)) + 0.1 * rng.standard_normal(n)
    ds = build_regression_dataset(x, y, chunk_size=2000, device=DEV)
    out = {}
    for fused in (True, False):
        kern = make_kernel("RBF", x.shape, m, 123, DEV, {})
        kern.set_hyperparams(np.array([0.3, 1.2]), logspace=False)
        if not fused:
            monkeypatch.setattr(SORFKernel, "fused_ok", lambda self: False)
            monkeypatch.setattr(SORFKernel, "block_ok", lambda self: False)
            assert not kern.fused_ok() and not kern.cache_ok()
        else:
            assert kern.fused_ok() and kern.cache_ok() and kern.block_ok()
        zty, yty = calc_zty(ds, kern)
        pre = RandNysPreconditioner(kern, ds, 128, False, 123, method)
        w, niter, losses = cg_fit_lib_internal(kern, ds, 1e-7, 300, pre, False, cache_features="auto" if fused else False)
        vec = torch.from_numpy(np.random.default_rng(1).standard_normal((m, 26))).to(DEV)
        mv = torch.zeros_like(vec)
        ConjugateGrad(cache_features=False)._matvec(ds, kern, vec, mv, add_ridge=False)
        out[fused] = (zty, yty, float(pre.achieved_ratio), w, niter, mv)
        monkeypatch.undo()
    (za, ya, ra, wa, na, ma), (zb, yb, rb, wb, nb, mb) = out[True], out[False]
    assert float((za - zb).abs().max() / zb.abs().max()) < 1e-7 and abs(ya - yb) <= 1e-12 * abs(yb)
    assert abs(ra - rb) <= 1e-4 * rb
    assert abs(na - nb) <= 1
    assert float((wa - wb).abs().max() / wb.abs().max()) < 1e-5
    assert float((ma - mb).abs().max() / mb.abs().max()) < 1e-6


def test_cg_with_resident_feature_cache_matches_regenerating_cg():
    """cache_features=True (Z kept in HBM as float32, streamed each iteration) gives the same solve as
    the default (features regenerated each iteration): iteration counts within one at tol 1e-8, weights to 1e-7."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import cg_fit_lib_internal
    g = load_golden("g7_cg.npz")
    x, y = g["x"], g["y"]
    ds = build_regression_dataset(x, y, chunk_size=500, device=DEV)
    kern = make_kernel("Matern", x.shape, int(g["num_rffs"]), 123, DEV, {"matern_nu": 2.5})
    kern.set_hyperparams(g["hyperparams"], logspace=False)
    pre = RandNysPreconditioner(kern, ds, 64, False, 123, "srht")
    w0, n0, _ = cg_fit_lib_internal(kern, ds, 1e-8, 500, pre, False, cache_features=False)
    w1, n1, _ = cg_fit_lib_internal(kern, ds, 1e-8, 500, pre, False, cache_features=True)
    # one iteration of slack, between the two modes and against the reference's count: at 1e-8 the stopping test
    # sits below the float32 rounding of the features (the reference's own last errors are 1.24e-8, 1.33e-8,
    # 8.98e-9 -- not monotone, 10 % under the tolerance when it stops at 79), and after ~80 Lanczos steps the
    # 1e-16 difference in float64 summation order between the two kernels shows in the fourth digit of the error
    nref = int(g["Matern_srht_niter"])
    assert abs(n0 - n1) <= 1 and abs(n0 - nref) <= 1 and abs(n1 - nref) <= 1
    assert rel(w1, w0.cpu().numpy()) < 1e-7
    assert rel(w1, g["Matern_srht_weights"]) < 1e-5


@pytest.mark.parametrize("m,tol", [(256, 1e-9), (20000, 1e-6)])
def test_conv_cg_with_feature_cache_matches_oracle(oracle, m, tol):
    """Conv1d kernel, CG with the resident (float32) feature cache -- the configuration that matters
    for sequence kernels, where regenerating Z costs K k-mers x SORF per sequence per iteration --
    against the oracle's CG on regenerated float64 features (m = 20000: the two-tiles-per-wave
    streaming kernel; n << m there, and below 1e-7 the residual of that solve wanders at the level of
    the float32 feature rounding, so the iteration counts are compared at 1e-6)."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import cg_fit_lib_internal
    rng = np.random.default_rng(5)
    n, L, C = 400, 24, 8
    x = rng.standard_normal((n, L, C)).astype(np.float32)
    sl = rng.integers(5, L + 1, size=n).astype(np.int32)
    y = rng.standard_normal(n)
    hp = np.array([0.5, 0.7])
    ds = build_regression_dataset(x, y, sl, chunk_size=128, device=DEV)
    kern = make_kernel("Conv1dRBF", x.shape, m, 123, DEV, {"conv_width": 5, "averaging": "sqrt"})
    kern.set_hyperparams(hp, logspace=False)
    w, niter, _ = cg_fit_lib_internal(kern, ds, tol, 300, None, False, cache_features=True)
    ods = orc.OracleDataset(x.astype(np.float64), y, sl, chunk_size=128)
    okern = orc.OracleKernel("Conv1dRBF", m, x.shape, hp, 123, conv_width=5, averaging="sqrt", ops=oracle)
    wref, nref, _, _ = orc.cg_fit_lib_internal(okern, ods, tol, 300, None)
    assert abs(niter - nref) <= 1
    assert rel(w, wref) < 1e-5


@pytest.mark.parametrize("method", ["srht", "srht_2"])
def test_conv_preconditioned_cg_sharing_one_feature_pass_matches_oracle(oracle, method):
    """Convolution kernel: the preconditioner passes take their feature chunks from the resident float32
    cache (cache_features="auto" for kernels that cannot regenerate cheaply) and the CG solve streams the same
    cache -- one generation pass in total -- against the oracle's preconditioned CG on float64 features."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import cg_fit_lib_internal
    rng = np.random.default_rng(77)
    n, L, C, m = 500, 20, 6, 256
    x = rng.standard_normal((n, L, C)).astype(np.float32)
    sl = rng.integers(6, L + 1, size=n).astype(np.int32)
    y = rng.standard_normal(n)
    hp = np.array([0.4, 0.6])
    ds = build_regression_dataset(x, y, sl, chunk_size=128, device=DEV)
    kern = make_kernel("Conv1dRBF", x.shape, m, 123, DEV, {"conv_width": 5, "averaging": "sqrt"})
    kern.set_hyperparams(hp, logspace=False)
    calls = {"n": 0}
    orig = kern.transform_x

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    kern.transform_x = counting
    pre = RandNysPreconditioner(kern, ds, 32, False, 123, method)
    w, niter, _ = cg_fit_lib_internal(kern, ds, 1e-9, 300, pre, False, cache_features=True)
    step = min(kern.CACHE_BUILD_ROWS, (1 << 30) // (8 * m))
    assert calls["n"] == -(-n // step)              # every sequence generated exactly once (in cache-build windows)
    ods = orc.OracleDataset(x.astype(np.float64), y, sl, chunk_size=128)
    okern = orc.OracleKernel("Conv1dRBF", m, x.shape, hp, 123, conv_width=5, averaging="sqrt", ops=oracle)
    opre = orc.OracleRandNysPreconditioner(okern, ods, 32, 123, method)
    assert np.isclose(pre.achieved_ratio, opre.achieved_ratio, rtol=1e-5)
    wref, nref, _, _ = orc.cg_fit_lib_internal(okern, ods, 1e-9, 300, opre)
    assert abs(niter - nref) <= 1
    assert rel(w, wref) < 1e-5


@pytest.mark.parametrize("m", [32768, 40000])
def test_resident_cache_beyond_8192_frequencies(m):
    """num_freqs = 16384 (BASELINE cfg5's M = 32768): the k = 1 streaming kernel takes two tiles per wave;
    num_freqs = 20000: the resident cache goes through the two block contractions with one column.  Same solve as
    the two-pass regenerating matvec either way."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import cg_fit_lib_internal
    rng = np.random.default_rng(9)
    n, d = 700, 40
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    y = np.sin(x @ rng.standard_normal(d)) + 0.1 * rng.standard_normal(n)
    ds = build_regression_dataset(x, y, chunk_size=256, device=DEV)
    kern = make_kernel("RBF", x.shape, m, 123, DEV, {})
    kern.set_hyperparams(np.array([0.8, 0.5]), logspace=False)
    assert kern.cache_ok() and kern.num_freqs > 8192
    w0, n0, _ = cg_fit_lib_internal(kern, ds, 1e-9, 300, None, False, cache_features=False)
    w1, n1, _ = cg_fit_lib_internal(kern, ds, 1e-9, 300, None, False, cache_features=True)
    assert abs(n0 - n1) <= 1
    assert rel(w1, w0.cpu().numpy()) < 1e-7


def test_g13_rank_selection_vs_reference():
    """check_rank_ratio / autoselect_preconditioner on the device against the reference's values
    (tests/golden/g13_rank_selection.npz)."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import check_rank_ratio, autoselect_preconditioner
    g8, g = load_golden("g8_e2e.npz"), load_golden("g13_rank_selection.npz")
    x, y = g8["xtrain"], g8["ytrain"]
    ds = build_regression_dataset(x, y, chunk_size=int(g["chunk_size"]), device=DEV)
    kern = make_kernel("RBF", x.shape, 512, 123, DEV, {"intercept": True})
    kern.set_hyperparams(g["hparam_log"], logspace=True)
    for frac, rank, ratio in zip(g["sample_fracs"], g["ranks"], g["ratios"]):
        assert np.isclose(check_rank_ratio(kern, ds, float(frac), int(rank), 123), float(ratio), rtol=1e-4)
    for tag, target in (("t30", 30.), ("t3", 3.)):
        pre, rank, method = autoselect_preconditioner(kern, ds, min_rank=16, max_rank=200, increment_size=48,
                                                      ratio_target=target, random_seed=123)
        assert rank == int(g[f"{tag}_rank"]) and pre.get_rank() == rank
        assert np.isclose(pre.achieved_ratio, float(g[f"{tag}_achieved_ratio"]), rtol=1e-4)


def test_linear_kernel_fit_equals_ridge_regression():
    """The Linear kernel (reference kernels/basic_kernels/linear.py): features are the float32-rounded input
    plus an intercept column; the CG fit equals the closed-form ridge solution in float64."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.cg import cg_fit_lib_internal
    from xgpr_amd.exact import calc_weights_exact
    rng = np.random.default_rng(51)
    n, d = 3000, 37
    x = rng.standard_normal((n, d))
    y = x @ rng.standard_normal(d) + 0.3 * rng.standard_normal(n) + 2.0
    ds = build_regression_dataset(x, y, chunk_size=700, device=DEV)
    kern = make_kernel("Linear", x.shape, None, 123, DEV, {})
    assert kern.get_num_rffs() == d + 1
    lam = 0.37
    kern.set_hyperparams(np.array([lam]), logspace=False)
    z = kern.transform_x(x[:5]).cpu().numpy()
    assert np.array_equal(z[:, 1:], x[:5].astype(np.float32).astype(np.float64)) and np.all(z[:, 0] == 1.0)
    w, niter, _ = cg_fit_lib_internal(kern, ds, 1e-12, 500, None, False, cache_features=False)
    zf = np.hstack([np.ones((n, 1)), x.astype(np.float32).astype(np.float64)])
    yn = (y - y.mean()) / y.std()
    ref = np.linalg.solve(zf.T @ zf + lam ** 2 * np.eye(d + 1), zf.T @ yn)
    assert rel(w, ref) < 1e-8
    xt, grad = kern.gradient_x(x[:4])
    assert tuple(grad.shape) == (4, 0, 0) and xt.shape[1] == d + 1


def test_graph_replayed_iterations_equal_plain_launches():
    """ConjugateGrad.USE_GRAPHS: the iteration captured as a HIP graph and replayed, with the convergence test
    applied on the device (hipCGStep1/2 stop_tol) -- same iterate, same losses, same iteration count as the
    host-checked loop, both when the solve converges and when it runs out of iterations."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import ConjugateGrad, cg_fit_lib_internal
    rng = np.random.default_rng(11)
    n, d, m = 3000, 20, 512
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    y = np.sin(x @ rng.standard_normal(d)) + 0.1 * rng.standard_normal(n)
    ds = build_regression_dataset(x, y, chunk_size=512, device=DEV)
    kern = make_kernel("RBF", x.shape, m, 123, DEV, {})
    kern.set_hyperparams(np.array([0.5, 0.6]), logspace=False)
    pre = RandNysPreconditioner(kern, ds, 48, False, 123, "srht")
    try:
        for tol, maxiter, precond, cache in ((1e-7, 200, pre, False), (1e-7, 200, None, True), (1e-12, 7, pre, False),
                                             (1e-3, 200, pre, True), (1e-7, 1, pre, False)):
            import warnings
            out = []
            for graphs in (False, True):
                ConjugateGrad.USE_GRAPHS = graphs
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out.append(cg_fit_lib_internal(kern, ds, tol, maxiter, precond, False, cache_features=cache))
            (w0, n0, l0), (w1, n1, l1) = out
            assert n0 == n1 and l0 == l1, (tol, maxiter)
            assert torch.equal(w0, w1)
    finally:
        ConjugateGrad.USE_GRAPHS = False
