"""BASELINE configs[3] (cfg4: Conv1d sequence kernel, L <= 512, 21 channels, conv_width 9, 16384 RFFs) and
configs[4] (cfg5: randomized-Nystrom preconditioner build, d = 512, 32768 RFFs, rank 2048, srht / srht_2) AT THEIR
SHAPES: the HIP path against values the reference itself produced (tests/golden/g17_cfg4_conv.npz,
g18_cfg5_precond.npz; tests/golden/make_golden.py imports the reference to make them), and -- at the per-GPU share
of the full job -- through size-independent properties.

Reference paths: convolution_ops/rbf_convolution.cpp:84-136, kernels/convolution_kernels/conv_kernel_baseclass.py:116-147;
preconditioners/rand_nys_constructors.py:96-218, kernels/srht_compressor.py:87-97,
basic_ops/transform_functions.cpp:95-121."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ---- the seeded inputs of the two fixtures (same draws as tests/golden/make_golden.py:cfg4_inputs / cfg5_inputs;
# the .npz carries a checksum of them instead of the arrays)
def cfg4_inputs(nseq=4):
    rng = np.random.default_rng(123)
    L, C = 512, 21
    seqlen = np.array([512, 64, 301, 9, 130, 477][:nseq], dtype=np.int32)
    x = np.zeros((nseq, L, C), dtype=np.float32)
    for i in range(nseq):
        x[i, np.arange(seqlen[i]), rng.integers(0, C, size=seqlen[i])] = 1.0
        x[i, seqlen[i]:, :] = rng.standard_normal((L - seqlen[i], C)).astype(np.float32)
    return x, seqlen


def cfg5_inputs(n=4096, d=512):
    rng = np.random.default_rng(123)
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    a = rng.standard_normal(d) * 3.0
    y = np.sin(x.astype(np.float64) @ a) + 0.1 * rng.standard_normal(n)
    return x, y


def rel(a, b):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else a
    return np.linalg.norm(a - b) / np.linalg.norm(b)


# =============================================================================== cfg4
@pytest.mark.parametrize("averaging", ["none", "sqrt", "full"])
def test_cfg4_conv_features_vs_reference(averaging):
    """hipConv1dFGen through the kernel class at cfg4's shape vs the reference's Conv1dRBF.transform_x."""
    from xgpr_amd.kernels import make_kernel
    g = load_golden("g17_cfg4_conv.npz")
    x, seqlen = cfg4_inputs()
    assert np.isclose(np.abs(x.astype(np.float64)).sum(), float(g["x_checksum"]), rtol=1e-12)
    assert np.array_equal(seqlen, g["seqlen"])
    kern = make_kernel("Conv1dRBF", x.shape, int(g["num_rffs"]), 123, DEV,
                       {"conv_width": int(g["conv_width"]), "averaging": averaging})
    kern.set_hyperparams(g["hyperparams"], logspace=False)
    z = kern.transform_x(x, seqlen).cpu().numpy()
    ref = g[f"z_{averaging}"]
    # every entry is a sum over K k-mers of float32 cos/sin values, each within 4e-7 of the reference's, times
    # the row scaler sqrt(1/F) / {1, sqrt(K), K}
    nk = (seqlen - int(g["conv_width"]) + 1).astype(np.float64)
    scaler = np.sqrt(1.0 / (int(g["num_rffs"]) // 2)) / {"none": np.ones_like(nk), "sqrt": np.sqrt(nk), "full": nk}[averaging]
    bound = 4e-7 * nk * scaler
    err = np.abs(z - ref)
    err[:, 0] = 0.0                                     # intercept column: exactly 1 on both sides
    assert np.array_equal(z[:, 0], ref[:, 0])
    assert np.all(err.max(axis=1) <= bound), (err.max(axis=1), bound)
    # and the north-star bar on the feature matrix
    assert np.allclose(z, ref, rtol=1e-5, atol=1e-5 * scaler.max())


def test_cfg4_operator_direct_and_float64_overload(oracle):
    """The operator itself (cudaConv1dFGen's signature): accumulate semantics with float32 inputs vs the
    reference's values, and the float64 overload vs the oracle's float64 path."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    from xgpr_amd.kernels import make_kernel, scale_input
    g = load_golden("g17_cfg4_conv.npz")
    x, seqlen = cfg4_inputs()
    kern = make_kernel("Conv1dRBF", x.shape, 16384, 123, DEV, {"conv_width": 9, "averaging": "sqrt"})
    kern.set_hyperparams(g["hyperparams"], logspace=False)
    xs = scale_input(torch.from_numpy(x).to(DEV), kern.hyperparams[1])
    ref = g["z_sqrt"].copy()
    out = torch.full((4, 16384), 0.25, dtype=torch.float64, device=DEV)      # results are ADDED into outputArr
    ext.cudaConv1dFGen(xs, out, kern.radem_diag, kern.chi_arr, seqlen, 9, 1)
    got = out.cpu().numpy() - 0.25
    assert np.abs(got[:, 1:] - ref[:, 1:]).max() <= 4e-7 * np.sqrt(504) * np.sqrt(1 / 8192) * 1.01
    x64 = xs.double()
    chi64 = kern.chi_arr.double()
    out64 = torch.zeros((4, 16384), dtype=torch.float64, device=DEV)
    ext.cudaConv1dFGen(x64, out64, kern.radem_diag, chi64, seqlen, 9, 1)
    want = np.zeros((4, 16384))
    oracle.cpuConv1dFGen(x64.cpu().numpy(), want, kern.radem_diag.cpu().numpy(), chi64.cpu().numpy(), seqlen, 9, 1)
    assert np.abs(out64.cpu().numpy() - want).max() <= 1e-12 * np.sqrt(504) * np.sqrt(1 / 8192)


def test_cfg4_full_share_properties():
    """cfg4's per-GPU share (62 500 sequences of the 5e5, L = 512, C = 21, w = 9, M = 16384) through the path fit()
    takes for convolution kernels: features generated once into the resident float32 cache, CG matvec streamed
    from it.  Size-independent properties: cached matvec == chunked Z^T(Zv) from the stand-alone operator on a
    window of rows; additivity over a row split; symmetry; determinism; per-row norm (cos^2 + sin^2 = 1 per
    frequency and k-mer => with 'sqrt' averaging |z_i|^2 = |sum_k phi_k|^2 / (K F) lies in (0, K], K for K equal
    k-mers and about 1 for unrelated ones)."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    n, L, C, m = 62_500, 512, 21, 16384
    gen = torch.Generator(device=DEV).manual_seed(77)
    seqlen = torch.randint(64, L + 1, (n,), generator=gen, device=DEV, dtype=torch.int32)
    # one-hot residues, built chunk-wise (the whole [n, L, C] float32 array is 2.7 GB)
    x = torch.zeros((n, L, C), dtype=torch.float32, device=DEV)
    for lo in range(0, n, 8192):
        hi = min(n, lo + 8192)
        res = torch.randint(0, C, (hi - lo, L), generator=gen, device=DEV)
        x[lo:hi].scatter_(2, res[:, :, None], 1.0)
    sl_host = seqlen.cpu().numpy()
    y = torch.randn(n, generator=gen, device=DEV, dtype=torch.float64)
    ds = build_regression_dataset(x, y, sl_host, chunk_size=1024, device=DEV)
    kern = make_kernel("Conv1dRBF", (n, L, C), m, 123, DEV, {"conv_width": 9, "averaging": "sqrt"})
    kern.set_hyperparams(np.array([1.0, 0.8]), logspace=False)
    zc = ds.feature_cache(kern)                             # one convolution pass over the share
    assert zc.shape == (n, m) and zc.dtype == torch.float32
    norms = (zc[:, 1:].double() ** 2).sum(dim=1)            # without the intercept column
    nk = (seqlen - 8).double()
    assert bool((norms <= nk * (1 + 1e-5)).all()) and float(norms.min()) > 0.0
    assert 0.5 < float(norms.median()) < 8.0
    v = torch.randn(m, generator=gen, device=DEV, dtype=torch.float64)
    v2 = torch.randn(m, generator=gen, device=DEV, dtype=torch.float64)
    ws = torch.empty(kern.workspace_bytes(), dtype=torch.uint8, device=DEV)
    w_all = torch.zeros(m, dtype=torch.float64, device=DEV)
    kern.ztz_matvec_cached(zc, v, w_all, ws)
    # (1) a window of rows: the cache path == the reference formulation (operator output, float64 GEMVs)
    lo, hi = 30_000, 32_048
    w_win = torch.zeros_like(w_all)
    kern.ztz_matvec_cached(zc[lo:hi], v, w_win, ws)
    ref = torch.zeros_like(w_all)
    for c0 in range(lo, hi, 1024):
        z = kern.transform_x(x[c0:c0 + 1024], sl_host[c0:c0 + 1024])
        ref += z.T @ (z @ v)
    assert float((w_win - ref).abs().max() / ref.abs().max()) < 1e-6       # float32 rounding of the cache rows
    # (2) additivity over a ragged split of the share
    cut = 20_011
    wa, wb = torch.zeros_like(w_all), torch.zeros_like(w_all)
    kern.ztz_matvec_cached(zc[:cut], v, wa, ws)
    kern.ztz_matvec_cached(zc[cut:], v, wb, ws)
    assert float((wa + wb - w_all).abs().max() / w_all.abs().max()) < 1e-12
    # (3) symmetry of Z^T Z and (4) determinism
    w2 = torch.zeros_like(w_all)
    kern.ztz_matvec_cached(zc, v2, w2, ws)
    a, b = float(v2 @ w_all), float(v @ w2)
    assert abs(a - b) <= 1e-10 * max(abs(a), abs(b))
    w_again = torch.zeros_like(w_all)
    kern.ztz_matvec_cached(zc, v, w_again, ws)
    assert torch.equal(w_again, w_all)
    # (5) the operator is deterministic too: regenerate a chunk and compare with the cached rows bit for bit
    z = kern.transform_x(x[:1024], sl_host[:1024]).to(torch.float32)
    assert torch.equal(z, zc[:1024])
    # the block (matrix-core) path over the same cache agrees with the streaming kernel column by column
    vb = torch.stack([v, v2], dim=1).contiguous()
    wb2 = torch.empty_like(vb)
    bws = torch.empty(ext.zcache_block_workspace_bytes(n, m, 2), dtype=torch.uint8, device=DEV)
    kern.ztz_block_cached(zc, vb, wb2, bws)
    assert float((wb2[:, 0] - w_all).abs().max() / w_all.abs().max()) < 1e-12
    assert float((wb2[:, 1] - w2).abs().max() / w2.abs().max()) < 1e-12


# =============================================================================== cfg5
def test_cfg5_srht_operators_at_width_32768(oracle):
    """hipSRHT and the compressor (pad + SRHT + gather, srht_compressor.py:87-97) at padded width 32768 in float64
    -- a 256 KB row, beyond what a workgroup's LDS holds -- bit-exact against the reference's own output and the
    oracle's; with z^T y from the same pass."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    from xgpr_amd.kernels import SRHTCompressor
    g = load_golden("g18_cfg5_precond.npz")
    m, rank = int(g["num_rffs"]), int(g["rank"])
    comp = SRHTCompressor(rank, m, device=DEV, random_seed=123)
    assert np.array_equal(comp.radem.cpu().numpy(), g["srht_radem"])
    assert np.array_equal(comp.col_sampler.cpu().numpy(), g["srht_col_sampler"])
    z8 = torch.from_numpy(g["z_first8"]).to(DEV)
    got = comp.transform_x(z8).cpu().numpy()
    assert np.array_equal(got, g["z8_compressed"])              # bit-exact vs the reference
    # bare operator, in place, vs the oracle on more rows (ragged count)
    rng = np.random.default_rng(3)
    xr = rng.standard_normal((37, 32768))
    want = xr.copy()
    oracle.cpuSRHT(want, g["srht_radem"])
    xd = torch.from_numpy(xr).to(DEV)
    ext.hipSRHT(xd, comp.radem)
    assert np.array_equal(xd.cpu().numpy(), want)
    # fused pass: compressed rows + z^T y, vs the separate-operator formulation
    zin = torch.from_numpy(rng.standard_normal((300, m))).to(DEV)
    yv = torch.from_numpy(rng.standard_normal(300)).to(DEV)
    zty = torch.zeros(m, dtype=torch.float64, device=DEV)
    fused = comp.transform_x_zty(zin, yv, zty).clone()
    sep = zin.clone()
    ext.hipSRHT(sep, comp.radem)
    assert torch.equal(fused, sep[:, comp.truncated_sampler])
    assert float((zty - zin.T @ yv).abs().max()) < 1e-11 * float((zin.T @ yv).abs().max()) + 1e-12
    # float32 rows in (exactly representable in float64), float64 sketch out: same bits
    z32 = zin.to(torch.float32)
    fused32 = comp.transform_x(z32.double())
    if hasattr(comp, "transform_f32"):
        assert torch.equal(comp.transform_f32(z32), fused32)


@pytest.mark.parametrize("method,ptag", [("srht", "srht"), ("srht_2", "srht2")])
def test_cfg5_preconditioner_vs_reference(method, ptag):
    """RandNysPreconditioner at cfg5's shape (d = 512, M = 32768, rank 2048) on the fixture's 4096 rows vs the
    reference's: z^T y, eigenvalues, achieved ratio, the projector U U^T, and the preconditioned CG solve."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd.preconditioner import RandNysPreconditioner
    from xgpr_amd.cg import cg_fit_lib_internal
    g = load_golden("g18_cfg5_precond.npz")
    n, d, m, rank = int(g["n"]), int(g["d"]), int(g["num_rffs"]), int(g["rank"])
    x, y = cfg5_inputs(n, d)
    assert np.isclose(np.abs(x.astype(np.float64)).sum(), float(g["x_checksum"]), rtol=1e-12)
    assert np.isclose(np.abs(y).sum(), float(g["y_checksum"]), rtol=1e-12)
    ds = build_regression_dataset(x, y, chunk_size=int(g["chunk_size"]), device=DEV)
    assert np.isclose(ds.get_ymean(), float(g["y_mean"]), rtol=1e-12)
    kern = make_kernel("RBF", (n, d), m, 123, DEV, {})
    kern.set_hyperparams(g["hyperparams"], logspace=False)
    z8 = kern.transform_x(x[:8]).cpu().numpy()
    assert np.abs(z8 - g["z_first8"]).max() <= 4e-7 * np.sqrt(1.0 / (m // 2 - 0.5))
    pre = RandNysPreconditioner(kern, ds, rank, False, 123, method)
    assert rel(pre.get_zty(), g["zty"]) < 1e-6
    assert np.isclose(pre.get_yty(), float(g["yty"]), rtol=1e-10)
    eig_ref = g[f"{ptag}_eig"]
    eig = pre.eig.cpu().numpy()
    # leading eigenvalues to 1e-5; the trailing ones (1e-6 of the largest) to 1e-5 of the spectrum's scale
    assert np.allclose(eig, eig_ref, rtol=1e-5, atol=1e-9 * eig_ref.max()), np.abs(eig / eig_ref - 1).max()
    assert np.isclose(pre.achieved_ratio, float(g[f"{ptag}_ratio"]), rtol=1e-4)
    assert np.isclose(pre.prefactor, float(g[f"{ptag}_prefactor"]), rtol=1e-5)
    v = torch.from_numpy(np.linspace(-1, 1, m)).to(DEV)
    uutv = (pre.u_mat @ (pre.u_mat.T @ v)).cpu().numpy()
    assert np.abs(uutv - g[f"{ptag}_uutv"]).max() <= 1e-5 * np.abs(g[f"{ptag}_uutv"]).max()
    w, niter, losses = cg_fit_lib_internal(kern, ds, 1e-6, 500, pre, False)
    assert abs(niter - int(g[f"{ptag}_niter"])) <= 1, (niter, int(g[f"{ptag}_niter"]))
    assert rel(w, g[f"{ptag}_weights"]) < 1e-5
    nl = min(len(losses), len(g[f"{ptag}_losses"])) - 1
    assert np.allclose(losses[:nl], g[f"{ptag}_losses"][:nl], rtol=1e-3)


def test_cfg5_accumulated_sketch_vs_reference():
    """The accumulation pass itself, acc[rank, M] = sum_chunks SRHT(Z)[:, S]^T Z (rand_nys_constructors.py:96-123),
    against the reference's sketch through its Frobenius norm, a probe contraction from each side and sampled
    entries."""
    from xgpr_amd.kernels import make_kernel, SRHTCompressor
    from xgpr_amd.dataset import build_regression_dataset
    from xgpr_amd import preconditioner as pc
    g = load_golden("g18_cfg5_precond.npz")
    n, d, m, rank = int(g["n"]), int(g["d"]), int(g["num_rffs"]), int(g["rank"])
    x, y = cfg5_inputs(n, d)
    ds = build_regression_dataset(x, y, chunk_size=int(g["chunk_size"]), device=DEV)
    kern = make_kernel("RBF", (n, d), m, 123, DEV, {})
    kern.set_hyperparams(g["hyperparams"], logspace=False)
    acc, zty, yty, _ = pc._first_pass(ds, rank, kern, 123, False)
    accn = acc.cpu().numpy()
    assert np.isclose(np.linalg.norm(accn), float(g["acc_fro"]), rtol=1e-6)
    pr, pcv = np.cos(np.arange(rank) * 0.37), np.sin(np.arange(m) * 0.11)
    left = pr @ accn
    assert np.abs(left - g["acc_left"]).max() <= 1e-5 * np.abs(g["acc_left"]).max()
    assert np.isclose(left @ pcv, float(g["acc_probe"]), rtol=1e-4, atol=1e-6 * np.abs(g["acc_left"]).max() * np.sqrt(m))
    samples = accn[g["acc_sample_rows"], g["acc_sample_cols"]]
    assert np.abs(samples - g["acc_samples"]).max() <= 1e-5 * np.abs(accn).max()


def test_cfg5_build_properties_at_per_gpu_share_width():
    """A larger slice of cfg5's per-GPU share (32 768 of the 250 000 rows per GPU; d = 512, M = 32768, rank 2048):
    the accumulation pass is additive over a split of the rows, deterministic, and equals the separate-operator
    formulation (float64 Z chunk, pad + SRHT + gather, library GEMM) on the same rows."""
    from xgpr_amd.kernels import make_kernel, SRHTCompressor
    from xgpr_amd.dataset import DeviceDataset
    from xgpr_amd import preconditioner as pc
    n, d, m, rank = 32_768, 512, 32768, 2048
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, d, generator=gen, device=DEV) / np.sqrt(d)
    y = torch.randn(n, generator=gen, device=DEV, dtype=torch.float64)
    kern = make_kernel("RBF", (n, d), m, 123, DEV, {})
    kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)

    def first_pass(lo, hi):
        ds = DeviceDataset(x[lo:hi], y[lo:hi], None, 8192, 0.0, 1.0, hi - lo, DEV)
        acc, zty, yty, _ = pc._first_pass(ds, rank, kern, 123, False)
        return acc, zty, yty
    acc_all, zty_all, yty_all = first_pass(0, n)
    cut = 12_288
    acc_a, zty_a, yty_a = first_pass(0, cut)
    acc_b, zty_b, yty_b = first_pass(cut, n)
    scale = float(acc_all.abs().max())
    assert float((acc_a + acc_b - acc_all).abs().max()) <= 1e-12 * scale
    assert float((zty_a + zty_b - zty_all).abs().max()) <= 1e-12 * float(zty_all.abs().max())
    assert abs(yty_a + yty_b - yty_all) <= 1e-12 * yty_all
    acc_again, _, _ = first_pass(0, n)
    assert torch.equal(acc_again, acc_all)
    # separate-operator formulation on the first chunk
    comp = SRHTCompressor(rank, m, device=DEV, random_seed=123)
    z = kern.transform_x(x[:8192])
    zp = z.clone()
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    ext.hipSRHT(zp, comp.radem)
    ref = zp[:, comp.truncated_sampler].T @ z
    acc_c, _, _ = first_pass(0, 8192)
    assert float((acc_c - ref).abs().max()) <= 1e-11 * float(ref.abs().max())
