"""Edge cases the reference's tests exercise or imply: minimal shapes, ragged sequence lengths,
a single k-mer, widths that straddle the fast-path limits, large-argument cos/sin, empty input."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("n,d,rffs", [(1, 1, 2), (1, 2, 2), (3, 1, 10), (2, 5, 2), (1, 1024, 8192), (65537, 4, 8)])
def test_minimal_shapes(oracle, n, d, rffs):
    from oracle import oracle as orc
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(n + d)
    radem, chi = orc.draw_sorf_params(rffs, d, 3)
    x = rng.standard_normal((n, d)).astype(np.float32)
    ref = np.zeros((n, rffs))
    oracle.cpuRBFFeatureGen(x.copy(), ref, radem, chi, False)
    out = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
    ext.hipRBFFeatureGen(dev(x), out, dev(radem), dev(chi), False)
    assert np.abs(out.cpu().numpy() - ref).max() <= 4e-7 * np.sqrt(2.0 / rffs)


def test_empty_input_raises():
    from oracle import oracle as orc
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    radem, chi = orc.draw_sorf_params(64, 10, 3)
    with pytest.raises(RuntimeError):
        ext.hipRBFFeatureGen(torch.zeros(0, 10, device=DEV), torch.zeros(0, 64, dtype=torch.float64, device=DEV),
                             dev(radem), dev(chi), False)
    with pytest.raises(RuntimeError):
        ext.hipFastHadamardTransform2D(torch.zeros(0, 16, device=DEV))


def test_large_arguments_take_the_double_reduction(oracle):
    """|chi * x| far beyond 2^18: the reference evaluates glibc cosf/sinf; the device path switches to a
    double-precision reduction -- still within 4e-7 * scale."""
    from oracle import oracle as orc
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(0)
    radem, chi = orc.draw_sorf_params(2048, 256, 3)
    x = (rng.standard_normal((16, 256)) * 3.0e5).astype(np.float32)
    ref = np.zeros((16, 2048))
    oracle.cpuRBFFeatureGen(x.copy(), ref, radem, chi, False)
    out = torch.zeros((16, 2048), dtype=torch.float64, device=DEV)
    ext.hipRBFFeatureGen(dev(x), out, dev(radem), dev(chi), False)
    assert np.abs(out.cpu().numpy() - ref).max() <= 4e-7 * np.sqrt(1.0 / 1024)
    v = torch.randn(2048, dtype=torch.float64, device=DEV)
    w = torch.zeros(2048, dtype=torch.float64, device=DEV)
    ext.hipZtZMatvec(dev(x), dev(radem), dev(chi), v, w, False)
    wref = ref.T @ (ref @ v.cpu().numpy())
    assert np.abs(w.cpu().numpy() - wref).max() <= 1e-5 * np.abs(wref).max()


@pytest.mark.parametrize("amp", [30.0, 1.0e3, 3.0e4])
def test_mid_range_arguments_on_the_transcendental_unit(oracle, amp):
    """Arguments across the whole range the fast path serves (|chi * x| from tens to ~2^18): cos/sin come from
    v_sin_f32 / v_cos_f32 after a two-term reduction in revolutions (max abs error 3.7e-7 against double libm,
    tools/sincos_probe.hip); against the reference's glibc cosf/sinf on the bit-identical float32 argument the
    features stay within 5e-7 * scale (north-star bar: 1e-5 relative)."""
    from oracle import oracle as orc
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(int(amp))
    radem, chi = orc.draw_sorf_params(4096, 512, 5)
    x = (rng.standard_normal((64, 512)) * amp / 16.0).astype(np.float32)
    ref = np.zeros((64, 4096))
    oracle.cpuRBFFeatureGen(x.copy(), ref, radem, chi, False)
    out = torch.zeros((64, 4096), dtype=torch.float64, device=DEV)
    ext.hipRBFFeatureGen(dev(x), out, dev(radem), dev(chi), False)
    scale = np.sqrt(1.0 / 2048)
    err = np.abs(out.cpu().numpy() - ref).max()
    assert err <= 5e-7 * scale, err / scale
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize("L,C,cw", [(9, 21, 9), (12, 21, 9), (40, 1, 1), (30, 48, 21), (33, 100, 11)])
def test_conv_ragged_and_single_kmer(oracle, L, C, cw):
    """seqlen == conv_width (one k-mer), mixed lengths, graph kernels (conv_width 1), and windows on both
    sides of the 1024-wide fast path (21*48 = 1008 -> P = 1024; 11*100 = 1100 -> P = 2048)."""
    from oracle import oracle as orc
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(L * C)
    n, rffs = 6, 1024
    radem, chi = orc.draw_sorf_params(rffs, cw * C, 5, conv=True)
    x = rng.standard_normal((n, L, C)).astype(np.float32)
    sl = np.array([cw, L, cw, max(cw, L - 1), L, max(cw, L // 2)], np.int32)
    for sc in (0, 1, 2):
        ref = np.zeros((n, rffs))
        oracle.cpuConv1dFGen(x, ref, radem, chi, sl, cw, sc)
        out = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
        ext.hipConv1dFGen(dev(x), out, dev(radem), dev(chi), sl, cw, sc)
        kmax = int(sl.max()) - cw + 1
        assert np.abs(out.cpu().numpy() - ref).max() <= 4e-7 * np.sqrt(2.0 / rffs) * kmax
    F = rffs // 2
    P = orc.padded_dims(cw * C)
    reps = -(-F // P)
    radem_m = rng.choice(np.asarray([-1, 1], np.int8), size=(3, 1, reps * P))
    chi_m = np.ascontiguousarray(chi[:F])
    ref = np.zeros((n, F), np.float32)
    oracle.cpuConv1dMaxpool(x, ref, radem_m, chi_m, sl, cw)
    out = torch.zeros((n, F), dtype=torch.float32, device=DEV)
    ext.hipConv1dMaxpool(dev(x), out, dev(radem_m), dev(chi_m), sl, cw)
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("n,L", [(64, 40), (300, 64), (1500, 33)])
def test_conv_longest_first_order_changes_nothing(oracle, monkeypatch, n, L):
    """With room in the workspace (xgpr_conv_workspace_bytes) a convolution launch processes its longest sequences
    first; with the smaller workspace it runs in the caller's order.  Features, max-pool features and gradients must
    be bit-identical between the two, and equal to the oracle's."""
    from oracle import oracle as orc
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(n + L)
    C, cw, rffs = 5, 7, 2048
    radem, chi = orc.draw_sorf_params(rffs, cw * C, 11, conv=True)
    x = rng.standard_normal((n, L, C)).astype(np.float32)
    sl = rng.integers(cw, L + 1, size=n).astype(np.int32)
    sl[: n // 4] = L                                    # many ties, the longest first in the caller's order too
    sl[-3:] = cw

    def run_all():
        out = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
        ext.hipConv1dFGen(dev(x), out, dev(radem), dev(chi), sl, cw, 1)
        o2 = torch.zeros((n, rffs), dtype=torch.float64, device=DEV)
        g2 = torch.zeros((n, rffs, 1), dtype=torch.float64, device=DEV)
        ext.hipConvGrad(dev(x), o2, dev(radem), dev(chi), sl, g2, 0.7, cw, 2)
        F = rffs // 2
        P = orc.padded_dims(cw * C)
        radem_m = np.random.default_rng(3).choice(np.asarray([-1, 1], np.int8), size=(3, 1, -(-F // P) * P))
        mp = torch.zeros((n, F), dtype=torch.float32, device=DEV)
        ext.hipConv1dMaxpool(dev(x), mp, dev(radem_m), dev(np.ascontiguousarray(chi[:F])), sl, cw)
        return out, o2, g2, mp

    ordered = run_all()
    monkeypatch.setattr(ext, "_conv_ws", ext._sorf_ws)  # the smaller workspace: no room for the order
    plain = run_all()
    for a, b in zip(ordered, plain):
        assert torch.equal(a, b)
    ref = np.zeros((n, rffs))
    oracle.cpuConv1dFGen(x, ref, radem, chi, sl, cw, 1)
    assert np.abs(ordered[0].cpu().numpy() - ref).max() <= 4e-7 * np.sqrt(2.0 / rffs) * (L - cw + 1)


def test_fused_matvec_small_n_and_odd_tiles(oracle):
    """fewer datapoints than slots, F not a multiple of 1024 (partial last tile), nb = 3 and nb = 5."""
    from oracle import oracle as orc
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(1)
    for n, d, rffs in [(1, 7, 30), (3, 100, 5000), (17, 64, 6146), (5, 300, 9000), (2, 1024, 16384)]:
        radem, chi = orc.draw_sorf_params(rffs, d, 4)
        x = rng.standard_normal((n, d)).astype(np.float32)
        z = np.zeros((n, rffs))
        oracle.cpuRBFFeatureGen(x.copy(), z, radem, chi, True)
        z[:, 0] = 1.0
        v = rng.standard_normal(rffs)
        w = torch.zeros(rffs, dtype=torch.float64, device=DEV)
        ext.hipZtZMatvec(dev(x), dev(radem), dev(chi), dev(v), w, True)
        ref = z.T @ (z @ v)
        assert np.abs(w.cpu().numpy() - ref).max() <= 1e-6 * np.abs(ref).max(), (n, d, rffs)


@pytest.mark.parametrize("d,m", [(90, 4096), (1001, 8192), (130, 2050)])
def test_scaled_shard_rows_padded_to_16_bytes(d, m):
    """Input widths that are not a multiple of four: the shard's scaled copy gets zero columns up to the next multiple
    (what the transform's own zero padding would hold), so the three-wave fused matvec (16-byte-aligned rows) serves
    them too; Z^T (Z v) through it equals the product over the feature operator's float64 output on the unpadded rows."""
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.dataset import build_regression_dataset
    rng = np.random.default_rng(d)
    n = 3000
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    ds = build_regression_dataset(x, rng.standard_normal(n), chunk_size=1000, device=DEV)
    kern = make_kernel("RBF", (n, d), m, 123, DEV, {})
    kern.set_hyperparams(np.array([0.3, 0.8]), logspace=False)
    xs = ds.scaled_x(kern.hyperparams[1])
    assert xs.shape[1] % 4 == 0 and xs.shape[1] - d < 4 and float(xs[:, d:].abs().max()) == 0.0
    v = torch.randn(m, dtype=torch.float64, device=DEV)
    w = torch.empty_like(v)
    ws = torch.empty(kern.workspace_bytes(), dtype=torch.uint8, device=DEV)
    kern.ztz_matvec(xs, v, w, ws)
    z = kern.transform_x(torch.from_numpy(x).to(DEV))
    ref = z.T @ (z @ v)
    assert float((w - ref).abs().max()) <= 1e-7 * float(ref.abs().max())


def test_offline_dataset_on_device(tmp_path):
    """.npy chunk files -> pinned host buffers -> HBM shard (loader thread + copy stream); a fit on it equals
    the fit on the in-memory dataset."""
    from xgpr_amd.dataset import build_offline_np_dataset, build_regression_dataset
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd.cg import cg_fit_lib_internal
    rng = np.random.default_rng(4)
    xs, ys, xf, yf = [], [], [], []
    for i, n in enumerate((300, 211, 256, 40)):
        x, y = rng.uniform(-1, 1, size=(n, 24)), rng.standard_normal(n)
        np.save(tmp_path / f"x{i}.npy", x)
        np.save(tmp_path / f"y{i}.npy", y)
        xs.append(x), ys.append(y), xf.append(str(tmp_path / f"x{i}.npy")), yf.append(str(tmp_path / f"y{i}.npy"))
    off = build_offline_np_dataset(xf, yf, chunk_size=300, device="cuda")
    ref = build_regression_dataset(np.vstack(xs), np.concatenate(ys), chunk_size=300, device="cuda")
    assert off.get_ndatapoints() == ref.get_ndatapoints() and off.get_ymean() == ref.get_ymean()
    kern = make_kernel("RBF", ref.get_xdim(), 128, 123, "cuda", {})
    kern.set_hyperparams(np.array([0.5, 0.7]), logspace=False)
    w0, n0, _ = cg_fit_lib_internal(kern, off, 1e-8, 200, None, False, cache_features=False)
    w1, n1, _ = cg_fit_lib_internal(kern, ref, 1e-8, 200, None, False, cache_features=False)
    assert n0 == n1 and torch.equal(w0, w1)


@pytest.mark.parametrize("n,d,m,icpt", [(1, 1024, 8192, True), (5, 1024, 8192, True), (769, 1024, 8192, False),
                                        (3000, 128, 4096, True), (2500, 300, 6144, True), (2048, 512, 8192, False),
                                        (1000, 1000, 8192, True), (777, 256, 12288, True), (600, 130, 4096, True),
                                        (900, 20, 4096, True), (1500, 512, 16384, True), (40, 64, 2048, True)])
def test_fused_matvec_shapes_around_the_three_wave_kernel(n, d, m, icpt):
    """The fused CG matvec over the shapes that decide which kernel serves it -- the three-wave kernel (128 <= P <= 1024,
    2 / 3 / 4 / 6 tiles per datapoint, rows a multiple of 4 floats), the two-wave kernel (everything else: one tile,
    P < 128, d % 4 != 0, M = 12288 / 16384 without LDS room) -- incl. fewer datapoints than datapoint slots, zero-padded
    widths and odd log2 P: equal to Z^T (Z v) from the stand-alone operator's float64 Z, and deterministic."""
    from xgpr_amd.kernels import make_kernel, scale_input
    g = torch.Generator(device=DEV).manual_seed(n + d + m)
    x = torch.randn(n, d, generator=g, device=DEV) / np.sqrt(d)
    kern = make_kernel("RBF", (n, d), m, 123, DEV, {"intercept": icpt})
    kern.set_hyperparams(np.array([0.1, 1.1]), logspace=False)
    xs = scale_input(x, kern.hyperparams[1])
    v = torch.randn(m, generator=g, device=DEV, dtype=torch.float64)
    w = torch.zeros(m, dtype=torch.float64, device=DEV)
    kern.ztz_matvec(xs, v, w)
    z = kern.transform_x(x)
    ref = z.T @ (z @ v)
    assert float((w - ref).abs().max()) <= 1e-9 * float(ref.abs().max())
    w2 = torch.zeros_like(w)
    kern.ztz_matvec(xs, v, w2)
    assert torch.equal(w, w2)
