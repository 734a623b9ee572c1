"""Stream ordering at the boundary: every entry point takes the caller's stream (the Python surface passes torch's
current stream) and must enqueue ALL of its work there.  Each operator is called inside a side-stream context, directly
behind a slow producer of its input on that stream (a large matrix product): a launch that went to the null stream, or
to any other, would read the input before it exists.  Results must equal those of the default stream bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _slow_identity(t, big):
    """t, produced behind ~10 ms of work on the current stream (the product's result is folded in with weight 0)."""
    w = big @ big
    return t + 0.0 * w[0, 0].to(t.dtype)


def _run_all(slow):
    from oracle import oracle as orc
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import cg_fit_lib_internal
    g = torch.Generator(device=DEV).manual_seed(7)
    big = torch.randn(6144, 6144, device=DEV, generator=g)
    prod = (lambda t: _slow_identity(t, big)) if slow else (lambda t: t + 0.0)
    out = {}
    n, d, m = 3000, 256, 4096
    radem, chi = orc.draw_sorf_params(m, d, 123)
    rt, ct = torch.from_numpy(radem).to(DEV), torch.from_numpy(chi).to(DEV)
    x0 = torch.randn(n, d, device=DEV, generator=g) / d ** 0.5
    v0 = torch.randn(m, dtype=torch.float64, device=DEV, generator=g)
    x = prod(x0)
    z = torch.zeros((n, m), dtype=torch.float64, device=DEV)
    ext.hipRBFFeatureGen(x, z, rt, ct, True)
    out["features"] = z
    x = prod(x0)
    w = torch.zeros(m, dtype=torch.float64, device=DEV)
    ext.hipZtZMatvec(x, rt, ct, v0, w, True)
    out["fused matvec"] = w
    x = prod(x0)
    zc = torch.empty((n, m), dtype=torch.float32, device=DEV)
    ext.hipRBFFeatureCache(x, zc, rt, ct)
    out["cache rows"] = zc
    V = prod(torch.randn(m, 26, dtype=torch.float64, device=DEV, generator=g))
    W = torch.zeros_like(V)
    ws = torch.empty(ext.zcache_block_workspace_bytes(n, m, 26), dtype=torch.uint8, device=DEV)
    ext.hipZCacheBlockMatvec(zc, V, W, True, ws)
    out["block matvec"] = W
    u = torch.linalg.qr(torch.randn(m, 64, dtype=torch.float64, device=DEV, generator=g))[0].contiguous()
    ie = torch.rand(64, dtype=torch.float64, device=DEV, generator=g) + 0.5
    R = prod(torch.randn(m, 26, dtype=torch.float64, device=DEV, generator=g))
    Z = torch.empty_like(R)
    ext.hipPrecondApplyBlock(u, ie, 0.7, R, Z)
    out["block apply"] = Z
    f = prod(torch.randn(500, 4096, dtype=torch.float64, device=DEV, generator=g))
    ext.hipFastHadamardTransform2D(f)
    out["fht"] = f
    nseq, L, C, cw, m2 = 64, 40, 21, 9, 1024
    radem2, chi2 = orc.draw_sorf_params(m2, cw * C, 77, conv=True)
    xs = prod(torch.randn(nseq, L, C, device=DEV, generator=g))
    sl = np.random.default_rng(1).integers(cw, L + 1, size=nseq).astype(np.int32)
    oc = torch.zeros((nseq, m2), dtype=torch.float64, device=DEV)
    ext.hipConv1dFGen(xs, oc, torch.from_numpy(radem2).to(DEV), torch.from_numpy(chi2).to(DEV), sl, cw, 1)
    out["conv features"] = oc
    # a whole build + solve (its pinned-memory error polling and its fallback synchronise the CURRENT stream)
    y = torch.sin(x0 @ torch.randn(d, device=DEV, generator=g)).double()
    ds = build_regression_dataset(prod(x0), y, chunk_size=1000, device=DEV)
    kern = make_kernel("RBF", (n, d), 2048, 123, DEV, {})
    kern.set_hyperparams(np.array([0.3, 1.0]), logspace=False)
    pre = RandNysPreconditioner(kern, ds, 64, False, 123, "srht")
    wts, niter, _ = cg_fit_lib_internal(kern, ds, 1e-8, 200, pre, False, cache_features=False)
    out["cg weights"] = wts
    out["cg iterations"] = torch.tensor([niter])
    torch.cuda.current_stream().synchronize()
    return {k: v.clone() for k, v in out.items()}


def test_side_stream_results_equal_default_stream_results():
    ref = _run_all(slow=False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = _run_all(slow=True)
    side.synchronize()
    for key in ref:
        assert torch.equal(ref[key].cpu(), got[key].cpu()), key
