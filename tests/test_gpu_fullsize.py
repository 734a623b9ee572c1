"""Parity at BASELINE.json's full sizes through size-independent properties (the CPU oracle
cannot finish these sizes in seconds): exact involution of the FHT on integer data, unit row
norms of the feature matrix, agreement of the fused kernel with the independent
(feature-gen operator + library GEMV) path, linearity and symmetry of the fused operator,
and the residual of a full CG solve."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cfg(name):
    return {"cfg2": ("RBF", 100_000, 256, 4096, {}),
            "cfg3": ("Matern", 1_000_000, 1024, 8192, {"matern_nu": 5 / 2}),
            "cfg5": ("RBF", 2_000_000, 512, 32768, {})}[name]


def _data(n, d, seed=0):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return torch.randn(n, d, generator=g, device=DEV) / np.sqrt(d)


@pytest.mark.parametrize("n,p", [(100_000, 256), (65_536, 1024), (4096, 8192), (512, 32768)])
def test_fht_involution_exact(n, p):
    """H(Hx) = P x exactly for small-integer x (every partial sum is an exactly representable integer)."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    x = torch.randint(-8, 9, (n, p), generator=g, device=DEV).float()
    y = x.clone()
    ext.hipFastHadamardTransform2D(y)
    ext.hipFastHadamardTransform2D(y)
    assert torch.equal(y, x * p)
    yd = x.double()
    ext.hipFastHadamardTransform2D(yd)
    ext.hipFastHadamardTransform2D(yd)
    assert torch.equal(yd, x.double() * p)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_feature_rows_have_unit_norm(cfg):
    """cos^2 + sin^2 = 1 per frequency => |z_i|^2 = F * scale^2 = 1 for every datapoint (no intercept)."""
    from xgpr_amd.kernels import make_kernel
    name, n, d, m, parms = _cfg(cfg)
    parms = dict(parms, intercept=False)
    k = make_kernel(name, (n, d), m, 123, DEV, parms)
    x = _data(n, d)
    chunk = 16384
    worst = 0.0
    for i in range(0, n, chunk):
        z = k.transform_x(x[i:i + chunk])
        worst = max(worst, float(((z * z).sum(dim=1) - 1.0).abs().max()))
    assert worst < 2e-6, worst


@pytest.mark.parametrize("cfg,rows", [("cfg2", 100_000), ("cfg3", 131_072), ("cfg5", 70_000)])
def test_fused_matvec_equals_chunked_path(cfg, rows):
    """Fused Z^T(Zv) == sum over chunks of Z.T @ (Z @ v) with Z from the stand-alone operator and
    the library GEMV (an independent code path), and is linear and symmetric."""
    from xgpr_amd.kernels import make_kernel, scale_input
    name, _, d, m, parms = _cfg(cfg)
    k = make_kernel(name, (rows, d), m, 123, DEV, parms)
    k.set_hyperparams(np.array([0.1, 1.3]), logspace=False)
    x = _data(rows, d, 5)
    xs = scale_input(x, k.hyperparams[1])
    g = torch.Generator(device=DEV)
    g.manual_seed(2)
    v1 = torch.randn(m, generator=g, device=DEV, dtype=torch.float64)
    v2 = torch.randn(m, generator=g, device=DEV, dtype=torch.float64)
    w1 = torch.zeros(m, dtype=torch.float64, device=DEV)
    w2 = torch.zeros_like(w1)
    w3 = torch.zeros_like(w1)
    k.ztz_matvec(xs, v1, w1)
    k.ztz_matvec(xs, v2, w2)
    ref = torch.zeros_like(w1)
    for i in range(0, rows, 8192):
        z = k.transform_x(x[i:i + 8192])
        ref += z.T @ (z @ v1)
    assert float((w1 - ref).abs().max() / ref.abs().max()) < 1e-9
    k.ztz_matvec(xs, 0.7 * v1 - 2.5 * v2, w3)
    lin = 0.7 * w1 - 2.5 * w2
    assert float((w3 - lin).abs().max() / lin.abs().max()) < 1e-11
    a, b = float(v2 @ w1), float(v1 @ w2)
    assert abs(a - b) <= 1e-10 * max(abs(a), abs(b))


def test_cfg2_cg_solve_residual():
    """BASELINE configs[1]: RBF, N = 1e5, d = 256, 4096 RFFs, one MI355X, CG fit: the returned weights
    satisfy the normal equations to the CG tolerance."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import cg_fit_lib_internal, calc_zty, ConjugateGrad
    n, d, m = 100_000, 256, 4096
    x = _data(n, d, 9)
    g = torch.Generator(device=DEV)
    g.manual_seed(4)
    y = torch.sin(x @ (3.0 * torch.randn(d, generator=g, device=DEV))).double() + \
        0.1 * torch.randn(n, generator=g, device=DEV, dtype=torch.float64)
    ds = build_regression_dataset(x, y, chunk_size=8192, device=DEV)
    k = make_kernel("RBF", (n, d), m, 123, DEV, {})
    k.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
    pre = RandNysPreconditioner(k, ds, 256, False, 123, "srht")
    w, niter, losses = cg_fit_lib_internal(k, ds, 1e-6, 500, pre, False)
    assert niter < 200
    zty, _ = calc_zty(ds, k)
    assert float((zty - pre.get_zty()).abs().max() / zty.abs().max()) < 1e-9     # fused z^T y == chunked
    aw = torch.zeros((m, 1), dtype=torch.float64, device=DEV)
    ConjugateGrad()._matvec(ds, k, w[:, None].contiguous(), aw)
    resid = float(torch.linalg.norm(aw[:, 0] - zty) / torch.linalg.norm(zty))
    assert resid < 1e-5, resid
