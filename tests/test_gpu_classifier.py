"""GPU parity of the classifier row: the two halves of the block matvec
(projection / back-projection on the float64 matrix cores), the classification cost function, the
preconditioned nonlinear CG fit and the predicted probabilities, against values the REFERENCE's own
Python produced on its own classifier test data (tests/golden/g11_classifier.npz) and against the
oracle.  Tolerance 1e-5 relative (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else a
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("n,m,k,icpt", [(133, 1024, 3, True), (1, 8, 1, False), (700, 260, 17, True),
                                        (4097, 512, 32, False)])
def test_project_and_backproject_vs_float64_product(n, m, k, icpt):
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = torch.Generator(device=DEV).manual_seed(n + m)
    zc = torch.rand((n, m), dtype=torch.float32, device=DEV, generator=g) * 2 - 1
    v = torch.randn((m, k), dtype=torch.float64, device=DEV, generator=g)
    r = torch.randn((n, k), dtype=torch.float64, device=DEV, generator=g)
    scale = float(np.float32(np.sqrt(1.0 / (m // 2 - 0.5 if icpt else m // 2))))
    z = zc.double() * scale
    if icpt:
        z[:, 0] = 1.0
    t = torch.full((n, k), 3.0, dtype=torch.float64, device=DEV)
    ext.hipZCacheBlockProject(zc, v, t, icpt)
    assert rel(t, (z @ v).cpu().numpy()) < 1e-13
    gout = torch.full((m, k), 5.0, dtype=torch.float64, device=DEV)
    ws = torch.empty(ext.zcache_block_workspace_bytes(n, m, k), dtype=torch.uint8, device=DEV)
    ext.hipZCacheBlockBackproject(zc, r, gout, icpt, ws)
    ref = (z.T @ r).cpu().numpy()
    assert rel(gout, ref) < 1e-13
    ext.hipZCacheBlockBackproject(zc, r, gout, icpt, ws, accumulate=True)
    assert rel(gout, 2 * ref) < 1e-13
    # explicit scale (complete feature rows, as the convolution kernels cache them)
    ext.hipZCacheBlockProject(zc, v, t, False, 0.5)
    assert rel(t, (0.5 * zc.double() @ v).cpu().numpy()) < 1e-13


@pytest.mark.parametrize("n,m,k,icpt", [(133, 1024, 3, True), (2048, 8192, 26, True), (2000, 8192, 10, False), (5000, 4100, 32, True),
                                        (16384, 2048, 17, True), (300, 640, 1, True)])
def test_short_projection_splits_the_contraction_and_stays_exact(n, m, k, icpt, monkeypatch):
    """Launches of fewer than 65536 rows (the reference's chunk is ~2000 rows, cg_tools.py:41-44) split the contraction
    over the features across workgroups (zblock_t_kernel, blockIdx.y) and add the partials in split order: the result
    is the float64 product to rounding, the same bits on every call, and equal to the unsplit kernel's (workspace
    None at the C ABI) to a few ulps of the row sums."""
    import ctypes as C
    from xgpr_amd import xgpr_hip_rfgen_ext as ext, _lib
    g = torch.Generator(device=DEV).manual_seed(n + m + k)
    zc = torch.rand((n, m), dtype=torch.float32, device=DEV, generator=g) * 2 - 1
    v = torch.randn((m, k), dtype=torch.float64, device=DEV, generator=g)
    scale = float(np.float32(np.sqrt(1.0 / (m // 2 - 0.5 if icpt else m // 2))))
    z = zc.double() * scale
    if icpt:
        z[:, 0] = 1.0
    ref = (z @ v).cpu().numpy()
    need = ext.zcache_block_project_workspace_bytes(n, m, k)
    assert (need > 0) == (m >= 1024), "these shapes are short launches: all but the narrowest split"
    assert ext.zcache_block_workspace_bytes(n, m, k) >= need
    t = torch.full((n, k), 3.0, dtype=torch.float64, device=DEV)
    ext.hipZCacheBlockProject(zc, v, t, icpt)
    assert rel(t, ref) < 1e-13
    t2 = torch.full((n, k), -1.0, dtype=torch.float64, device=DEV)
    ws = torch.full((max(need, 16),), 0xFF, dtype=torch.uint8, device=DEV)          # a poisoned workspace: every partial is written before it is read
    ext.hipZCacheBlockProject(zc, v, t2, icpt, 0.0, ws)
    assert torch.equal(t, t2)
    # the unsplit kernel: no workspace at the C ABI
    t3 = torch.empty_like(t)
    lib = _lib.load()
    _lib.check(lib.xgpr_zcache_block_project_f32(C.c_void_p(zc.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(t3.data_ptr()),
                                                 n, m, k, int(icpt), 0.0, None, C.c_size_t(0), None))
    torch.cuda.synchronize()
    assert rel(t3, ref) < 1e-13 and rel(t3, t.cpu().numpy()) < 1e-14
    if need:
        small = torch.empty(need - 16, dtype=torch.uint8, device=DEV)
        with pytest.raises(RuntimeError):
            ext.hipZCacheBlockProject(zc, v, t2, icpt, 0.0, small)
    # the block matvec of a short shard runs the same split T kernel in front of the W kernel
    if k <= 32 and m % 4 == 0:
        w = torch.empty((m, k), dtype=torch.float64, device=DEV)
        bws = torch.empty(ext.zcache_block_workspace_bytes(n, m, k), dtype=torch.uint8, device=DEV)
        ext.hipZCacheBlockMatvec(zc, v, w, icpt, bws)
        refw = (z.T @ (z @ v)).cpu().numpy()
        assert rel(w, refw) < 1e-13
        w2 = torch.empty_like(w)
        ext.hipZCacheBlockMatvec(zc, v, w2, icpt, bws)
        assert torch.equal(w, w2)


@pytest.mark.parametrize("cache", [False, True])
def test_g11_classifier_vs_reference(cache):
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_classification_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.classification import NonlinearCGClassification, fit_classifier, predict_proba
    g = load_golden("g11_classifier.npz")
    x, y = g["xtrain"], g["ytrain"]
    ds = build_classification_dataset(x, y, chunk_size=2000, device=DEV)
    assert ds.get_n_classes() == 3 and ds.get_ndatapoints() == x.shape[0]
    kern = make_kernel("RBF", x.shape, 1024, 123, DEV, {"intercept": True})
    kern.set_hyperparams(g["hparam_log"], logspace=True)
    pre = RandNysPreconditioner(kern, ds, 256, False, 123, "srht", is_regression=False)
    assert pre.get_zty() is None
    op = NonlinearCGClassification(ds, kern, False, pre, cache_features=cache)
    grad, loss = op.cost_fun_classification(torch.from_numpy(g["w_probe"]).to(DEV))
    assert np.isclose(loss, float(g["loss_probe"]), rtol=1e-6)
    assert rel(grad, g["grad_probe"]) < 1e-5
    w, gamma, niter, losses = fit_classifier(kern, ds, pre, tol=1e-2, max_iter=500, cache_features=cache)
    assert niter == int(g["niter"]) and niter < 10        # the reference's acceptance bar
    assert np.allclose(losses, g["losses"], rtol=1e-5)
    assert rel(w, g["weights"]) < 1e-4                     # line-search steps amplify rounding ~10x
    probs = predict_proba(kern, w, gamma, torch.from_numpy(g["xtest"]).to(DEV))
    assert np.allclose(probs.cpu().numpy(), g["probs"], rtol=1e-4, atol=1e-6)
    assert (probs.argmax(dim=1).cpu().numpy() == g["ytest"]).mean() > 0.9


def test_classifier_conv_kernel_matches_oracle(oracle):
    """Sequence kernel + no preconditioner (the reference's aliasing of the search direction with the
    gradient included): cost function and a short fit against the oracle."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_classification_dataset
    from xgpr_amd.classification import NonlinearCGClassification
    rng = np.random.default_rng(31)
    n, L, C, m, ncls = 240, 16, 5, 64, 4
    x = rng.standard_normal((n, L, C)).astype(np.float32)
    sl = rng.integers(4, L + 1, size=n).astype(np.int32)
    y = rng.integers(0, ncls, size=n).astype(np.int64)
    y[:ncls] = np.arange(ncls)
    hp = np.array([0.8, 0.6])
    ds = build_classification_dataset(x, y, sl, chunk_size=100, device=DEV)
    kern = make_kernel("Conv1dRBF", x.shape, m, 123, DEV, {"conv_width": 3, "averaging": "sqrt"})
    kern.set_hyperparams(hp, logspace=False)
    ods = orc.OracleClassificationDataset(x.astype(np.float64), y, sl, chunk_size=100)
    okern = orc.OracleKernel("Conv1dRBF", m, x.shape, hp, 123, conv_width=3, averaging="sqrt", ops=oracle)
    w0 = 0.1 * rng.standard_normal((m, ncls))
    op = NonlinearCGClassification(ds, kern, False, None, cache_features=False)
    grad, loss = op.cost_fun_classification(torch.from_numpy(w0).to(DEV))
    gref, lref = orc.classification_cost(ods, okern, w0)
    assert np.isclose(loss, lref, rtol=1e-6) and rel(grad, gref) < 1e-5
    w, niter, losses = op.fit_model(max_iter=6, tol=1e-6)
    wref, nref, lossref = orc.fit_classifier(ods, okern, None, 6, 1e-6)
    assert niter == nref
    assert np.allclose(losses, lossref, rtol=1e-5)
    assert rel(w, wref) < 1e-4


def test_classifier_cost_with_more_than_32_classes(oracle):
    """More classes than one call of the block operators takes (32 columns): the cost function walks column
    groups; gradient and loss against the oracle."""
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_classification_dataset
    from xgpr_amd.classification import NonlinearCGClassification, predict_proba
    rng = np.random.default_rng(41)
    n, d, m, ncls = 900, 10, 128, 40
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    y = rng.integers(0, ncls, size=n).astype(np.int64)
    y[:ncls] = np.arange(ncls)
    hp = np.array([0.7, 0.9])
    ds = build_classification_dataset(x, y, chunk_size=256, device=DEV)
    kern = make_kernel("RBF", x.shape, m, 123, DEV, {})
    kern.set_hyperparams(hp, logspace=False)
    w0 = 0.2 * rng.standard_normal((m, ncls))
    for cache in (False, True):
        op = NonlinearCGClassification(ds, kern, False, None, cache_features=cache)
        grad, loss = op.cost_fun_classification(torch.from_numpy(w0).to(DEV))
        ods = orc.OracleClassificationDataset(x.astype(np.float64), y, chunk_size=256)
        okern = orc.OracleKernel("RBF", m, x.shape, hp, 123, ops=oracle)
        gref, lref = orc.classification_cost(ods, okern, w0)
        assert np.isclose(loss, lref, rtol=1e-6) and rel(grad, gref) < 1e-5
    probs = predict_proba(kern, torch.from_numpy(w0).to(DEV), torch.zeros(ncls, dtype=torch.float64, device=DEV),
                          torch.from_numpy(x[:50]).to(DEV))
    assert np.allclose(probs.cpu().numpy(), orc.predict_proba(okern, w0, x[:50].astype(np.float64)), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n,ncls", [(1000, 3), (257, 10), (1, 2), (4099, 40)])
def test_softmax_residual_kernel_equals_the_reference_chain(n, ncls):
    """hipSoftmaxResidual (one launch) against the chain of array operations of nonlinear_cg_toolkit.py:243-262: row
    maximum, 2.71828 ** (.), normalisation, clipped log loss, residual; incl. rows whose label probability underflows
    the 1e-16 clip."""
    import torch
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = torch.Generator(device=DEV).manual_seed(n + ncls)
    pred = torch.randn(n, ncls, dtype=torch.float64, device=DEV, generator=g) * 8.0
    labels = torch.randint(0, ncls, (n,), device=DEV, generator=g)
    pred[0, :] = 0.0
    pred[0, (int(labels[0]) + 1) % ncls] = 80.0          # the label's probability is ~ e^-80: clipped at 1e-16
    ref = pred.clone()
    ref -= ref.max(dim=1, keepdim=True).values
    ref = 2.71828 ** ref
    ref /= ref.sum(dim=1, keepdim=True)
    loss_ref = -torch.log(ref.clamp(min=1e-16)).gather(1, labels[:, None]).sum()
    ref.scatter_add_(1, labels[:, None], torch.full((n, 1), -1.0, dtype=torch.float64, device=DEV))
    out = pred.clone()
    loss = ext.hipSoftmaxResidual(out, labels)
    assert float((out - ref).abs().max()) < 1e-14
    assert abs(float(loss) - float(loss_ref)) <= 1e-12 * abs(float(loss_ref))
    out2 = pred.clone()
    assert torch.equal(ext.hipSoftmaxResidual(out2, labels), loss) and torch.equal(out2, out)      # reproducible
    # a label outside [0, classes) is not dropped silently: the loss turns NaN (the reference's gather raises)
    for badlab in (ncls, -1):
        bad = labels.clone()
        bad[n // 2] = badlab
        assert torch.isnan(ext.hipSoftmaxResidual(pred.clone(), bad)).all()
