"""Pins the CPU oracle (oracle/) against the golden vectors the REFERENCE produced
(tests/golden/*.npz, see tests/golden/make_golden.py).  Bar: bit-exact for the native
ops (same IEEE operations in the same order); parameter draws bit-exact; CG iterates
1e-9 relative (numpy/BLAS on both sides, only thread-dependent summation order differs).
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc


def test_g1_fht(oracle):
    g = load_golden("g1_fht.npz")
    for P in [2, 4, 32, 1024, 4096]:
        x32 = g[f"x_{P}"].copy()
        x64 = x32.astype(np.float64)
        oracle.cpuFastHadamardTransform2D(x32)
        oracle.cpuFastHadamardTransform2D(x64)
        assert np.array_equal(x32, g[f"y32_{P}"])
        assert np.array_equal(x64, g[f"y64_{P}"])
    x = g["x3d"].copy()
    x64 = x.astype(np.float64)
    oracle.cpuFastHadamardTransform(x)
    oracle.cpuFastHadamardTransform(x64)
    assert np.array_equal(x, g["y3d32"])
    assert np.array_equal(x64, g["y3d64"])


def test_fht_matches_hadamard_matrix(oracle):
    """The reference's own known-answer test: FHT == scipy.linalg.hadamard @ x
    (reference tests/fht_operations_tests/test_basic_rfgen.py:98-138)."""
    from scipy.linalg import hadamard
    rng = np.random.default_rng(123)
    for P in [2, 8, 64, 512]:
        x = rng.uniform(-1, 1, size=(7, P))
        y = x.copy()
        oracle.cpuFastHadamardTransform2D(y)
        assert np.allclose(y, x @ hadamard(P).T.astype(np.float64))


def test_g2_rbf(oracle):
    g = load_golden("g2_rbf.npz")
    for si in range(int(g["n_settings"])):
        x32, radem, chi32 = g[f"x_{si}"], g[f"radem_{si}"], g[f"chi_{si}"]
        icpt = bool(g[f"intercept_{si}"])
        o32 = np.zeros_like(g[f"out32_{si}"])
        o64 = np.zeros_like(o32)
        oracle.cpuRBFFeatureGen(x32.copy(), o32, radem, chi32, icpt)
        oracle.cpuRBFFeatureGen(x32.astype(np.float64), o64, radem, chi32.astype(np.float64), icpt)
        assert np.array_equal(o32, g[f"out32_{si}"]), si
        assert np.array_equal(o64, g[f"out64_{si}"]), si
        if f"sigma_{si}" in g:
            sigma = float(g[f"sigma_{si}"])
            for tag, x, c in (("32", x32, chi32),
                              ("64", x32.astype(np.float64), chi32.astype(np.float64))):
                o = np.zeros_like(o32)
                gr = np.zeros(o32.shape + (1,))
                oracle.cpuRBFGrad(x.copy(), o, gr, radem, c, sigma, icpt)
                assert np.array_equal(o, g[f"gout{tag}_{si}"])
                assert np.array_equal(gr, g[f"grad{tag}_{si}"])


def test_g3_conv(oracle):
    g = load_golden("g3_conv.npz")
    for si in range(int(g["n_settings"])):
        x32, radem, chi32, sl = g[f"x_{si}"], g[f"radem_{si}"], g[f"chi_{si}"], g[f"seqlen_{si}"]
        cw, sc = int(g[f"conv_width_{si}"]), int(g[f"scaling_{si}"])
        o32 = np.zeros_like(g[f"out32_{si}"])
        o64 = np.zeros_like(o32)
        oracle.cpuConv1dFGen(x32.copy(), o32, radem, chi32, sl, cw, sc)
        oracle.cpuConv1dFGen(x32.astype(np.float64), o64, radem, chi32.astype(np.float64), sl, cw, sc)
        assert np.array_equal(o32, g[f"out32_{si}"]), si
        assert np.array_equal(o64, g[f"out64_{si}"]), si
    for tag, dt in (("32", np.float32), ("64", np.float64)):
        o = np.zeros_like(g["g_out32"])
        gr = np.zeros_like(g["g_grad32"])
        oracle.cpuConvGrad(g["g_x"].astype(dt), o, g["g_radem"], g["g_chi"].astype(dt),
                           g["g_seqlen"], gr, float(g["g_sigma"]), int(g["g_conv_width"]),
                           int(g["g_scaling"]))
        assert np.array_equal(o, g[f"g_out{tag}"])
        assert np.array_equal(gr, g[f"g_grad{tag}"])


def test_g4_maxpool(oracle):
    g = load_golden("g4_maxpool.npz")
    for si in range(int(g["n_settings"])):
        x32, radem, chi32, sl = g[f"x_{si}"], g[f"radem_{si}"], g[f"chi_{si}"], g[f"seqlen_{si}"]
        cw = int(g[f"conv_width_{si}"])
        o32 = np.zeros_like(g[f"out32_{si}"])
        o64 = np.zeros_like(o32)
        oracle.cpuConv1dMaxpool(x32.copy(), o32, radem, chi32, sl, cw)
        oracle.cpuConv1dMaxpool(x32.astype(np.float64), o64, radem, chi32.astype(np.float64), sl, cw)
        assert np.array_equal(o32, g[f"out32_{si}"]), si
        assert np.array_equal(o64, g[f"out64_{si}"]), si


def test_g5_srht(oracle):
    g = load_golden("g5_srht.npz")
    for P in [256, 512, 2048, 8192, 32768]:
        x32 = g[f"x_{P}"].copy()
        x64 = x32.astype(np.float64)
        oracle.cpuSRHT(x32, g[f"radem_{P}"])
        oracle.cpuSRHT(x64, g[f"radem_{P}"])
        assert np.array_equal(x32, g[f"y32_{P}"])
        assert np.array_equal(x64, g[f"y64_{P}"])


def test_g6_draws_bit_exact():
    """Rademacher / permutation / chi draws must be bit-exact (BASELINE.json north_star)."""
    g = load_golden("g6_draws.npz")
    cases = [("cfg1_RBF", 512, 32, None), ("cfg2_RBF", 4096, 256, None),
             ("cfg3_Matern", 8192, 1024, ("matern", 2.5)), ("cfg3_Cauchy", 8192, 1024, ("cauchy",)),
             ("cfg5_RBF", 32768, 512, None), ("fix_RBF", 4096, 84, None), ("small_RBF", 64, 3, None)]
    for tag, rffs, d, extra in cases:
        radem, chi = orc.draw_sorf_params(rffs, d, 123)
        if extra and extra[0] == "matern":
            orc.matern_rescale(chi, extra[1], 123)
        elif extra and extra[0] == "cauchy":
            orc.cauchy_rescale(chi, 123)
        assert radem.dtype == np.int8 and np.array_equal(radem, g[f"{tag}_radem"]), tag
        assert chi.dtype == np.float32 and np.array_equal(chi, g[f"{tag}_chi"]), tag
    for tag, rffs, width, nu in [("cfg4_Conv1dRBF", 16384, 9 * 21, None),
                                 ("graph_GraphRBF", 1024, 12, None),
                                 ("conv_Conv1dMatern", 2048, 5 * 21, 1.5)]:
        radem, chi = orc.draw_sorf_params(rffs, width, 123, conv=True)
        if nu:
            orc.matern_rescale(chi, nu, 123)
        assert np.array_equal(radem, g[f"{tag}_radem"]), tag
        assert np.array_equal(chi, g[f"{tag}_chi"]), tag
    for tag, rank, m in [("srht_256_4096", 256, 4096), ("srht_512_8192", 512, 8192),
                         ("srht_64_512", 64, 512), ("srht_100_1000", 100, 1000)]:
        radem, cs = orc.draw_srht_params(rank, m, 123)
        assert np.array_equal(radem, g[f"{tag}_radem"])
        assert np.array_equal(cs, g[f"{tag}_col_sampler"])


@pytest.mark.parametrize("kname", ["RBF", "Matern"])
def test_g7_cg(oracle, kname):
    g = load_golden("g7_cg.npz")
    x, y = g["x"].astype(np.float64), g["y"]
    ds = orc.OracleDataset(x, y, chunk_size=int(g["chunk_size"]))
    assert np.isclose(ds.y_mean, float(g["y_mean"]), rtol=1e-14)
    assert np.isclose(ds.y_std, float(g["y_std"]), rtol=1e-14)
    kern = orc.OracleKernel(kname, int(g["num_rffs"]), x.shape, g["hyperparams"], 123,
                            matern_nu=2.5, ops=oracle)
    assert np.array_equal(kern.transform_x(x[:8]), g[f"{kname}_z_first8"])
    zty, yty = orc.calc_zty(ds, kern)
    assert np.allclose(zty, g[f"{kname}_zty"], rtol=1e-10, atol=1e-10)
    assert np.isclose(yty, float(g[f"{kname}_yty"]), rtol=1e-12)
    for ptag, method in [("none", None), ("srht", "srht"), ("srht2", "srht_2")]:
        pre = None
        if method is not None:
            pre = orc.OracleRandNysPreconditioner(kern, ds, 64, 123, method)
            assert np.allclose(pre.eig, g[f"{kname}_{ptag}_eig"], rtol=1e-7)
            assert np.isclose(pre.achieved_ratio, float(g[f"{kname}_{ptag}_ratio"]), rtol=1e-6)
            assert np.allclose(pre.get_zty(), g[f"{kname}_{ptag}_zty"], rtol=1e-10, atol=1e-10)
            # U is defined up to the sign of each column: compare the projector
            u_ref = g[f"{kname}_{ptag}_u"]
            v = np.linspace(-1, 1, u_ref.shape[0])
            assert np.allclose(pre.u_mat @ (pre.u_mat.T @ v), u_ref @ (u_ref.T @ v), atol=1e-8)
        w, niter, losses, _ = orc.cg_fit_lib_internal(kern, ds, 1e-8, 500, pre)
        assert niter == int(g[f"{kname}_{ptag}_niter"])
        wref = g[f"{kname}_{ptag}_weights"]
        assert np.linalg.norm(w - wref) <= 1e-7 * np.linalg.norm(wref)
        trace = {}
        nit = g[f"{kname}_{ptag}_iterates"].shape[0]
        orc.cg_fit_lib_internal(kern, ds, 1e-30, nit, pre, trace=trace)
        n = ds.get_ndatapoints()
        for j in range(nit):
            ref = g[f"{kname}_{ptag}_iterates"][j]
            got = trace["x_k"][j][:, 0] * n
            assert np.linalg.norm(got - ref) <= 1e-9 * np.linalg.norm(ref), (ptag, j)
        assert np.allclose(losses, g[f"{kname}_{ptag}_losses"], rtol=1e-6)


def test_g8_reference_fixture(oracle):
    """Reference tests/fitting_tests/test_cg_fit.py:26-40: rank-256 SRHT preconditioner on the
    381x84 fixture with 4096 RFFs, tol 1e-6 => niter < 10 (reference got 7)."""
    g = load_golden("g8_e2e.npz")
    x, y = g["xtrain"], g["ytrain"]
    ds = orc.OracleDataset(x, y, chunk_size=2000)
    kern = orc.OracleKernel("RBF", 4096, x.shape, np.exp(g["hparam_log"]), 123, ops=oracle)
    pre = orc.OracleRandNysPreconditioner(kern, ds, 256, 123, "srht")
    assert np.isclose(pre.achieved_ratio, float(g["ratio"]), rtol=1e-6)
    w, niter, losses, conv = orc.cg_fit_lib_internal(kern, ds, 1e-6, 500, pre)
    assert conv and niter == int(g["niter"]) and niter < 10
    assert np.linalg.norm(w - g["weights"]) <= 1e-7 * np.linalg.norm(g["weights"])
    z = kern.transform_x(g["xtest"])
    preds = (z @ w) * ds.y_std + ds.y_mean
    assert np.allclose(preds, g["preds"], rtol=1e-7, atol=1e-9)


def test_oracle_validation_errors(oracle):
    """The ops raise RuntimeError where the reference throws (rbf_ops.cpp:49-62,
    rbf_convolution.cpp:49-82; reference tests/fht_operations_tests/
    test_variable_length_seq_handling.py:74-95)."""
    radem, chi = orc.draw_sorf_params(64, 10, 123)
    x = np.zeros((4, 10), np.float32)
    with pytest.raises(RuntimeError):
        oracle.cpuRBFFeatureGen(x, np.zeros((3, 64)), radem, chi, False)
    with pytest.raises(RuntimeError):
        oracle.cpuRBFFeatureGen(x, np.zeros((4, 62)), radem, chi, False)
    with pytest.raises(RuntimeError):
        oracle.cpuRBFFeatureGen(np.zeros((0, 10), np.float32), np.zeros((0, 64)), radem, chi, False)
    radem, chi = orc.draw_sorf_params(64, 3 * 4, 123, conv=True)
    xc = np.zeros((3, 10, 4), np.float32)
    good = np.array([10, 5, 3], np.int32)
    oracle.cpuConv1dFGen(xc, np.zeros((3, 64)), radem, chi, good, 3, 0)
    for bad in (np.array([11, 5, 3], np.int32), np.array([10, 5, 2], np.int32),
                np.array([10, 5], np.int32)):
        with pytest.raises(RuntimeError):
            oracle.cpuConv1dFGen(xc, np.zeros((3, 64)), radem, chi, bad, 3, 0)
    with pytest.raises(RuntimeError):
        oracle.cpuConv1dFGen(xc, np.zeros((3, 64)), radem, chi, good, 11, 0)
    with pytest.raises(RuntimeError):
        oracle.cpuFastHadamardTransform2D(np.zeros((3, 12)))
    with pytest.raises(RuntimeError):
        oracle.cpuSRHT(np.zeros((3, 16)), np.ones(8, np.int8))


@pytest.mark.parametrize("tag", ["easy", "hard"])
def test_g10_nmll_oracle_vs_reference(oracle, tag):
    """Exact NMLL, its gradient and the SLQ approximation (reference xgp_regression.py:152-367) on the
    reference fixture: the oracle's restatement against the reference's own run, including the
    intermediate probe draws and CG coefficients of the 26-column solve."""
    g8, g = load_golden("g8_e2e.npz"), load_golden("g10_nmll.npz")
    x, y = g8["xtrain"], g8["ytrain"]
    ds = orc.OracleDataset(x, y, chunk_size=2000)
    kern = orc.OracleKernel("RBF", 512, x.shape, np.exp(g[f"{tag}_hparam_log"]), 123, ops=oracle)
    assert np.isclose(orc.exact_nmll(kern, ds), float(g[f"{tag}_exact_nmll"]), rtol=1e-9)
    nll, grad = orc.exact_nmll_gradient(kern, ds)
    assert np.isclose(nll, float(g[f"{tag}_grad_nmll"]), rtol=1e-9)
    assert np.allclose(grad, g[f"{tag}_grad"], rtol=1e-6, atol=1e-8)
    pre = orc.OracleRandNysPreconditioner(kern, ds, 64, 123, "srht_2")
    assert np.isclose(pre.get_logdet(), float(g[f"{tag}_precond_logdet"]), rtol=1e-8)
    det = {}
    approx = orc.approximate_nmll(kern, ds, pre, 25, 500, 1e-6, 123, details=det)
    assert np.allclose(det["probes"], g[f"{tag}_probes"], rtol=1e-8, atol=1e-10)
    na = g[f"{tag}_alphas"].shape[0]
    assert abs(det["alphas"].shape[0] - na) <= 1
    nc = min(na, det["alphas"].shape[0], 10)
    assert np.allclose(det["alphas"][:nc], g[f"{tag}_alphas"][:nc], rtol=1e-6)
    assert np.allclose(det["betas"][:nc], g[f"{tag}_betas"][:nc], rtol=1e-6)
    assert np.isclose(det["logdet"], float(g[f"{tag}_logdet"]), rtol=1e-6)
    assert np.isclose(approx, float(g[f"{tag}_approx_nmll"]), rtol=1e-7)
    # the reference's own acceptance test: approximate within 1 % of exact (test_slq_nmll.py:73-79)
    assert 100 * abs(approx - float(g[f"{tag}_exact_nmll"])) / float(g[f"{tag}_exact_nmll"]) < 1.0


def test_g11_classifier_oracle_vs_reference(oracle):
    """xGPClassification on the wine data of the reference's own classifier test (reference
    tests/fitting_tests/test_cg_fit.py:76-91): one cost-function evaluation, the nonlinear-CG loss
    sequence, the fitted weights and the predicted class probabilities."""
    g = load_golden("g11_classifier.npz")
    x, y = g["xtrain"], g["ytrain"]
    ds = orc.OracleClassificationDataset(x, y, chunk_size=2000)
    assert ds.get_n_classes() == 3
    kern = orc.OracleKernel("RBF", 1024, x.shape, np.exp(g["hparam_log"]), 123, ops=oracle)
    grad, loss = orc.classification_cost(ds, kern, g["w_probe"])
    assert np.isclose(loss, float(g["loss_probe"]), rtol=1e-10)
    assert np.allclose(grad, g["grad_probe"], rtol=1e-8, atol=1e-10)
    pre = orc.OracleRandNysPreconditioner(kern, ds, 256, 123, "srht")
    w, niter, losses = orc.fit_classifier(ds, kern, pre, 500, 1e-2)
    assert niter == int(g["niter"]) and niter < 10
    assert np.allclose(losses, g["losses"], rtol=1e-6)
    assert np.linalg.norm(w - g["weights"]) <= 1e-5 * np.linalg.norm(g["weights"])
    probs = orc.predict_proba(kern, w, g["xtest"])
    assert np.allclose(probs, g["probs"], rtol=1e-5, atol=1e-7)
    assert (probs.argmax(axis=1) == g["ytest"]).mean() > 0.9


def test_g12_mini_ard_oracle_vs_reference_ground_truth(oracle):
    """cpuMiniARDGrad (reference rbf_ops/ard_ops.cpp:39-124) and the MiniARD host logic
    (kernels/ARD_kernels/mini_ard.py) against the reference's own ground truth, which reaches the same
    features and gradient without the gradient operator (its test_ARD_kernel_gradient.py:120-162)."""
    g = load_golden("g12_mini_ard.npz")
    for ci in range(int(g["ncases"])):
        x = g[f"c{ci}_x"]
        nf, icpt = int(g[f"c{ci}_num_freqs"]), bool(g[f"c{ci}_intercept"])
        for dp, rtol_f, atol_f, rtol_g, atol_g in ((True, 1e-5, 1e-8, 1e-5, 1e-8), (False, 1e-5, 1e-5, 1e-4, 1e-3)):
            kern = orc.OracleMiniARDKernel(2 * nf, x.shape, list(g[f"c{ci}_split_points"]), g[f"c{ci}_hyperparams"],
                                           123, double_precision=dp, fit_intercept=icpt, ops=oracle)
            kern.precompute_weights()
            if dp:
                assert np.allclose(kern.precomputed_weights, g[f"c{ci}_weights"], rtol=1e-12, atol=1e-14)
                assert np.allclose(kern.transform_x(x), g[f"c{ci}_transform_x"], rtol=1e-9, atol=1e-12)
            feats, grad = kern.gradient_x(x)
            ref_f, ref_g = g[f"c{ci}_features"].copy(), g[f"c{ci}_grad"].copy()
            if icpt:                       # as the reference's test does (:158-160)
                ref_g[:, 0, :] = 0
                ref_f[:, 0] = 1.0
            assert np.allclose(feats, ref_f, rtol=rtol_f, atol=atol_f)     # tolerances of the reference's test (:77-80)
            assert np.allclose(grad, ref_g, rtol=rtol_g, atol=atol_g)


def test_g13_rank_selection_oracle_vs_reference(oracle):
    """Sampled rank / ratio check and the rank autoselection (reference model_baseclass.py:376-480,
    rand_nys_constructors.py:60-93, :301-357) against the reference's own values on its fixture."""
    g8, g = load_golden("g8_e2e.npz"), load_golden("g13_rank_selection.npz")
    x, y = g8["xtrain"], g8["ytrain"]
    ds = orc.OracleDataset(x, y, chunk_size=int(g["chunk_size"]))
    kern = orc.OracleKernel("RBF", 512, x.shape, np.exp(g["hparam_log"]), 123, ops=oracle)
    for frac, rank, ratio in zip(g["sample_fracs"], g["ranks"], g["ratios"]):
        assert np.isclose(orc.check_rank_ratio(kern, ds, float(frac), int(rank), 123), float(ratio), rtol=1e-6)
    for tag, target in (("t30", 30.), ("t3", 3.)):
        rank, method = orc.autoselect_rank(kern, ds, 16, 200, 48, False, target, 123)
        assert rank == int(g[f"{tag}_rank"])
        pre = orc.OracleRandNysPreconditioner(kern, ds, rank, 123, method)
        assert np.isclose(pre.achieved_ratio, float(g[f"{tag}_achieved_ratio"]), rtol=1e-6)


def test_g14_two_layer_kernel_oracle_vs_reference(oracle):
    """Conv1dTwoLayer (max-pooled convolution features into an RBF map): the oracle's kernel against the
    reference's kernel class, features and sigma-gradient, ragged sequences."""
    g = load_golden("g14_two_layer.npz")
    k = orc.OracleTwoLayerKernel(int(g["num_rffs"]), g["x"].shape, g["hyperparams"], int(g["conv_width"]),
                                 int(g["init_rffs"]), 123, True, ops=oracle)
    assert np.array_equal(k.transform_x(g["x"], g["seqlen"]), g["features"])
    f, gr = k.gradient_x(g["x"], g["seqlen"])
    assert np.array_equal(f, g["grad_features"]) and np.array_equal(gr, g["grad"])


def test_g17_cfg4_conv_shape(oracle):
    """BASELINE configs[3]'s shape (L = 512, C = 21, conv_width 9, 16384 RFFs): the oracle's kernel restatement
    reproduces the reference's Conv1dRBF.transform_x bit for bit for the three averaging modes."""
    from test_gpu_cfg_shapes import cfg4_inputs
    g = load_golden("g17_cfg4_conv.npz")
    x, seqlen = cfg4_inputs()
    assert np.isclose(np.abs(x.astype(np.float64)).sum(), float(g["x_checksum"]), rtol=1e-12)
    for avg in ("none", "sqrt", "full"):
        k = orc.OracleKernel("Conv1dRBF", int(g["num_rffs"]), x.shape, g["hyperparams"], 123,
                             conv_width=int(g["conv_width"]), averaging=avg, ops=oracle)
        assert np.array_equal(k.transform_x(x.astype(np.float64), seqlen), g[f"z_{avg}"]), avg


def test_g18_cfg5_precond_shape(oracle):
    """BASELINE configs[4]'s shape (d = 512, 32768 RFFs, SRHT width 32768 in float64, rank 2048): features, the
    compressor, z^T y and the accumulated sketch (through sampled entries and a one-sided probe -- the dense
    rank x M product and the factorizations are left to the GPU test) against the reference's values."""
    from test_gpu_cfg_shapes import cfg5_inputs
    g = load_golden("g18_cfg5_precond.npz")
    n, d, m, rank = int(g["n"]), int(g["d"]), int(g["num_rffs"]), int(g["rank"])
    x, y = cfg5_inputs(n, d)
    assert np.isclose(np.abs(x.astype(np.float64)).sum(), float(g["x_checksum"]), rtol=1e-12)
    k = orc.OracleKernel("RBF", m, (n, d), g["hyperparams"], 123, ops=oracle)
    comp = orc.OracleSRHTCompressor(rank, m, random_seed=123, ops=oracle)
    assert np.array_equal(comp.radem, g["srht_radem"]) and np.array_equal(comp.col_sampler, g["srht_col_sampler"])
    z8 = k.transform_x(x[:8].astype(np.float64))
    assert np.array_equal(z8, g["z_first8"])
    assert np.array_equal(comp.transform_x(z8), g["z8_compressed"])
    ods = orc.OracleDataset(x.astype(np.float64), y, None, chunk_size=int(g["chunk_size"]))
    probe_r = np.cos(np.arange(rank) * 0.37)
    ri, ci = g["acc_sample_rows"], g["acc_sample_cols"]
    zty, left, samples, yty = np.zeros(m), np.zeros(m), np.zeros(len(ri)), 0.0
    for xin, yin, _ in ods.get_chunked_data():
        z = k.transform_x(xin)
        zty += z.T @ yin
        yty += float(yin @ yin)
        sz = comp.transform_x(z)
        left += (sz @ probe_r) @ z                      # probe_r^T (S(Z)^T Z)
        samples += np.einsum("ij,ij->j", sz[:, ri], z[:, ci])
    assert np.allclose(zty, g["zty"], rtol=1e-11, atol=1e-11 * np.abs(g["zty"]).max())
    assert np.isclose(yty, float(g["yty"]), rtol=1e-12)
    assert np.allclose(left, g["acc_left"], rtol=1e-9, atol=1e-11 * np.abs(g["acc_left"]).max())
    assert np.allclose(samples, g["acc_samples"], rtol=1e-9, atol=1e-12 * np.abs(g["acc_samples"]).max())
