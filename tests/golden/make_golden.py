#!/usr/bin/env python3
"""Generates tests/golden/*.npz FROM THE REFERENCE ITSELF (authoring container only).

What runs here is the reference, not this repository's code:

* the reference's compiled arithmetic core (oracle/_ref/libxgpr_ref.so, built
  by oracle/Makefile from the two nanobind-free files of
  /root/reference/src/xGPR/random_feature_generation/cpu_rf_gen/shared_fht_functions/),
* the reference's Python package, imported from /root/reference/src with that
  compiled core registered as the stand-in for its own (offline-unbuildable,
  nanobind-based) extension module ``xGPR.xgpr_cpu_rfgen_cpp_ext``.

The outputs are data only (inputs + expected outputs); the settings are the
ones the reference's tests use (file:line cited next to each block), trimmed
in row count so the fixtures stay small.  /root/reference does not travel to
the GPU box: tests read only the .npz files written here.

    python tests/golden/make_golden.py
"""
import os
import sys
import types
from math import ceil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/src"
REF_TESTDATA = "/root/reference/tests/test_data"

from oracle import oracle as orc  # noqa: E402

orc.build(ref=True)
REF = orc.RefCore()


def _standin_module():
    """The reference's extension-module surface (cpu_rf_gen/xgpr_cpu_rfgen_cpp_ext.cpp:24-146)
    served by the reference's own compiled core."""
    m = types.ModuleType("xGPR.xgpr_cpu_rfgen_cpp_ext")
    for name in ["cpuFastHadamardTransform", "cpuFastHadamardTransform2D", "cpuSRHT",
                 "cpuRBFFeatureGen", "cpuRBFGrad", "cpuConv1dMaxpool", "cpuConv1dFGen",
                 "cpuConvGrad"]:
        setattr(m, name, getattr(REF, name))

    def cpuMiniARDGrad(*a, **k):
        raise NotImplementedError("MiniARD is out of scope")
    m.cpuMiniARDGrad = cpuMiniARDGrad
    return m


def import_reference():
    sys.modules["xGPR.xgpr_cpu_rfgen_cpp_ext"] = _standin_module()
    sys.path.insert(0, REF_SRC)
    import xGPR  # noqa: F401
    return xGPR


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


# ---------------------------------------------------------------- G1: bare FHT
def g1_fht():
    """tests/fht_operations_tests/test_basic_rfgen.py:26-47, :98-138 (shapes there go up to
    (250,1,4096)); here 8 rows x P in {2,4,32,1024,4096} + one 3-D case."""
    rng = np.random.default_rng(123)
    out = {}
    for P in [2, 4, 32, 1024, 4096]:
        x = rng.uniform(-10, 10, size=(8, P))
        x32 = x.astype(np.float32)
        x64 = x32.astype(np.float64)   # one stored input serves both precisions
        y32, y64 = x32.copy(), x64.copy()
        REF.cpuFastHadamardTransform2D(y32)
        REF.cpuFastHadamardTransform2D(y64)
        out[f"x_{P}"] = x32
        out[f"y32_{P}"] = y32
        out[f"y64_{P}"] = y64
    x = rng.uniform(-10, 10, size=(5, 3, 64)).astype(np.float32)
    y32, y64 = x.copy(), x.astype(np.float64)
    REF.cpuFastHadamardTransform(y32)
    REF.cpuFastHadamardTransform(y64)
    out.update(x3d=x, y3d32=y32, y3d64=y64)
    save("g1_fht.npz", **out)


# ---------------------------------------------------------------- G2: SORF-RBF
def g2_rbf():
    """tests/fht_operations_tests/test_rbf_rfgen.py:26-68 (xdim/num_freqs list) with the
    set-up of :192-211 (seed 123, radem -> chi -> uniform[-10,10] inputs)."""
    from scipy.stats import chi
    settings = [((10, 50), 64, False, 6), ((10, 3), 64, False, 6), ((3, 2003), 2000, False, 3),
                ((11, 1076), 8192, False, 2), ((231, 856), 2000, True, 4),
                ((16, 32), 256, True, 8), ((16, 256), 2048, True, 4), ((8, 1024), 4096, True, 4),
                ((8, 512), 1000, False, 4)]
    out = {"n_settings": np.int64(len(settings))}
    for si, (xdim, num_freqs, intercept, rows) in enumerate(settings):
        pd = 2 ** ceil(np.log2(max(xdim[-1], 2)))
        nblocks = ceil(num_freqs / pd) if pd < num_freqs else 1
        rng = np.random.default_rng(123)
        radem = rng.choice(np.asarray([-1, 1], dtype=np.int8), size=(3, 1, nblocks * pd),
                           replace=True)
        chi_arr = chi.rvs(df=pd, size=num_freqs, random_state=123)
        x = rng.uniform(low=-10.0, high=10.0, size=(xdim[0], xdim[1]))[:rows]
        x32 = np.ascontiguousarray(x.astype(np.float32))
        x64 = x32.astype(np.float64)
        chi32 = chi_arr.astype(np.float32)
        chi64 = chi32.astype(np.float64)
        o32 = np.zeros((rows, 2 * num_freqs))
        o64 = np.zeros((rows, 2 * num_freqs))
        REF.cpuRBFFeatureGen(x32, o32, radem, chi32, intercept)
        REF.cpuRBFFeatureGen(x64, o64, radem, chi64, intercept)
        out[f"x_{si}"] = x32
        out[f"radem_{si}"] = radem
        out[f"chi_{si}"] = chi32
        out[f"intercept_{si}"] = np.bool_(intercept)
        out[f"out32_{si}"] = o32
        out[f"out64_{si}"] = o64
        if si in (0, 4):   # gradient (test_rbf_rfgen.py:51-68): sigma 0.7, unscaled input
            sigma = 0.7
            g32 = np.zeros((rows, 2 * num_freqs, 1))
            g64 = np.zeros((rows, 2 * num_freqs, 1))
            go32 = np.zeros((rows, 2 * num_freqs))
            go64 = np.zeros((rows, 2 * num_freqs))
            REF.cpuRBFGrad(x32, go32, g32, radem, chi32, sigma, intercept)
            REF.cpuRBFGrad(x64, go64, g64, radem, chi64, sigma, intercept)
            out[f"sigma_{si}"] = np.float64(sigma)
            out[f"gout32_{si}"] = go32
            out[f"grad32_{si}"] = g32
            out[f"gout64_{si}"] = go64
            out[f"grad64_{si}"] = g64
    save("g2_rbf.npz", **out)


def _conv_setup(ndatapoints, kernel_width, aa_dim, num_aas, num_freqs, maxpool):
    """tests/fht_operations_tests/conv_testing_functions.py:12-41."""
    dim2 = 2 ** ceil(np.log2(kernel_width * aa_dim))
    radem_size = ceil(num_freqs / dim2) * dim2
    rng = np.random.default_rng(123)
    xdata = rng.uniform(low=-10.0, high=10.0, size=(ndatapoints, num_aas, aa_dim))
    s_mat = rng.uniform(size=num_freqs)
    radem = rng.choice(np.asarray([-1, 1], dtype=np.int8), size=(3, 1, radem_size), replace=True)
    seqlen = rng.integers(low=kernel_width + 1, high=num_aas + 1,
                          size=ndatapoints).astype(np.int32)
    return xdata, s_mat, radem, seqlen


# ---------------------------------------------------------------- G3: conv SORF-RBF
def g3_conv():
    """tests/fht_operations_tests/test_conv1d_fht.py:25-33 (seven settings), :93-110
    (normalisation 1, 2), :49-57 (gradient).  Row counts trimmed; the (10,256,512,333)
    setting is shortened to 40 positions (still P = 8192 with > 1 k-mer)."""
    # kernel_width, num_aas, aa_dim, num_freqs, sigma, n, scaling
    settings = [(9, 23, 21, 1000, 0.5, 4, 0), (5, 56, 2, 62, 0.5, 6, 0), (9, 23, 21, 1000, 1, 4, 0),
                (7, 202, 105, 784, 1, 2, 0), (9, 10, 2000, 4096, 1, 2, 0),
                (10, 11, 200, 784, 1, 3, 0), (10, 40, 512, 333, 1, 2, 0),
                (7, 53, 105, 784, 1, 3, 1), (7, 53, 105, 784, 1, 3, 2),
                (9, 64, 21, 1024, 1, 4, 1)]
    out = {"n_settings": np.int64(len(settings))}
    for si, (kw, num_aas, aa_dim, nf, sigma, n, scaling) in enumerate(settings):
        xdata, s_mat, radem, seqlen = _conv_setup(n, kw, aa_dim, num_aas, nf, False)
        x32 = np.ascontiguousarray((xdata * sigma).astype(np.float32))
        x64 = x32.astype(np.float64)
        chi32 = s_mat.astype(np.float32)
        chi64 = chi32.astype(np.float64)
        o32 = np.zeros((n, 2 * nf))
        o64 = np.zeros((n, 2 * nf))
        REF.cpuConv1dFGen(x32, o32, radem, chi32, seqlen, kw, scaling)
        REF.cpuConv1dFGen(x64, o64, radem, chi64, seqlen, kw, scaling)
        out[f"x_{si}"] = x32
        out[f"radem_{si}"] = radem
        out[f"chi_{si}"] = chi32
        out[f"seqlen_{si}"] = seqlen
        out[f"conv_width_{si}"] = np.int64(kw)
        out[f"scaling_{si}"] = np.int64(scaling)
        out[f"out32_{si}"] = o32
        out[f"out64_{si}"] = o64
    # gradient, test_conv1d_fht.py:49-57: (9, 23, 21, 128), sigma 0.5
    kw, num_aas, aa_dim, nf, sigma, n = 9, 23, 21, 128, 0.5, 4
    xdata, s_mat, radem, seqlen = _conv_setup(n, kw, aa_dim, num_aas, nf, False)
    x32 = np.ascontiguousarray(xdata.astype(np.float32))
    chi32 = s_mat.astype(np.float32)
    for tag, x, c in (("32", x32, chi32), ("64", x32.astype(np.float64), chi32.astype(np.float64))):
        o = np.zeros((n, 2 * nf))
        g = np.zeros((n, 2 * nf, 1))
        REF.cpuConvGrad(x, o, radem, c, seqlen, g, sigma, kw, 1)
        out[f"g_out{tag}"] = o
        out[f"g_grad{tag}"] = g
    out.update(g_x=x32, g_radem=radem, g_chi=chi32, g_seqlen=seqlen, g_conv_width=np.int64(kw),
               g_sigma=np.float64(sigma), g_scaling=np.int64(1))
    save("g3_conv.npz", **out)


# ---------------------------------------------------------------- G4: conv max-pool
def g4_maxpool():
    """tests/fht_operations_tests/test_maxpool_rfgen.py:23-55; rows trimmed, the
    (5,512,1024,1024) setting shortened to 40 positions."""
    settings = [(9, 23, 21, 130, 4), (15, 23, 1060, 8194, 2), (5, 56, 2, 62, 6),
                (5, 56, 256, 500, 3), (5, 40, 1024, 1024, 2)]
    out = {"n_settings": np.int64(len(settings))}
    for si, (kw, num_aas, aa_dim, nf, n) in enumerate(settings):
        xdata, s_mat, radem, seqlen = _conv_setup(n, kw, aa_dim, num_aas, nf, True)
        x32 = np.ascontiguousarray(xdata.astype(np.float32))
        x64 = x32.astype(np.float64)
        chi32 = s_mat.astype(np.float32)
        o32 = np.zeros((n, nf), np.float32)
        o64 = np.zeros((n, nf), np.float32)
        REF.cpuConv1dMaxpool(x32, o32, radem, chi32, seqlen, kw)
        REF.cpuConv1dMaxpool(x64, o64, radem, chi32.astype(np.float64), seqlen, kw)
        out[f"x_{si}"] = x32
        out[f"radem_{si}"] = radem
        out[f"chi_{si}"] = chi32
        out[f"seqlen_{si}"] = seqlen
        out[f"conv_width_{si}"] = np.int64(kw)
        out[f"out32_{si}"] = o32
        out[f"out64_{si}"] = o64
    save("g4_maxpool.npz", **out)


# ---------------------------------------------------------------- G5: SRHT
def g5_srht():
    """tests/fht_operations_tests/test_basic_rfgen.py:49-56, :140-178 (shapes (150,256),
    (304,512), (5,2048)); rows trimmed to 6, plus the preconditioner widths 8192/32768."""
    rng = np.random.default_rng(123)
    out = {}
    for P, rows in [(256, 6), (512, 6), (2048, 5), (8192, 2), (32768, 2)]:
        x = rng.uniform(-10, 10, size=(rows, P)).astype(np.float32)
        radem = rng.choice(np.asarray([-1, 1], dtype=np.int8), size=(P), replace=True)
        y32, y64 = x.copy(), x.astype(np.float64)
        REF.cpuSRHT(y32, radem)
        REF.cpuSRHT(y64, radem)
        out[f"x_{P}"] = x
        out[f"radem_{P}"] = radem
        out[f"y32_{P}"] = y32
        out[f"y64_{P}"] = y64
    save("g5_srht.npz", **out)


# ---------------------------------------------------------------- G6: parameter draws
def g6_draws(xgpr):
    """Kernel parameter draws by the reference's own kernel classes:
    kernels/basic_kernels/sorf_kernel_baseclass.py:71-84, matern.py:50-54, cauchy.py:39-41,
    convolution_kernels/conv_kernel_baseclass.py:85-99, kernels/srht_compressor.py:61-65."""
    from xGPR.kernels import KERNEL_NAME_TO_CLASS
    from xGPR.kernels.srht_compressor import SRHTCompressor
    out = {}
    cases = [("cfg1_RBF", "RBF", (1, 32), 512, {}),
             ("cfg2_RBF", "RBF", (1, 256), 4096, {}),
             ("cfg3_Matern", "Matern", (1, 1024), 8192, {"matern_nu": 5 / 2}),
             ("cfg3_Cauchy", "Cauchy", (1, 1024), 8192, {}),
             ("cfg5_RBF", "RBF", (1, 512), 32768, {}),
             ("fix_RBF", "RBF", (1, 84), 4096, {}),
             ("small_RBF", "RBF", (1, 3), 64, {}),
             ("cfg4_Conv1dRBF", "Conv1dRBF", (1, 512, 21), 16384, {"conv_width": 9}),
             ("graph_GraphRBF", "GraphRBF", (1, 30, 12), 1024, {}),
             ("conv_Conv1dMatern", "Conv1dMatern", (1, 60, 21), 2048,
              {"conv_width": 5, "matern_nu": 3 / 2})]
    for tag, kname, xdim, rffs, parms in cases:
        cls = KERNEL_NAME_TO_CLASS[kname]
        k = cls(xdim, rffs, random_seed=123, device="cpu", kernel_spec_parms=parms)
        out[f"{tag}_radem"] = np.asarray(k.radem_diag)
        out[f"{tag}_chi"] = np.asarray(k.chi_arr)
    for tag, rank, m in [("srht_256_4096", 256, 4096), ("srht_512_8192", 512, 8192),
                         ("srht_64_512", 64, 512), ("srht_100_1000", 100, 1000)]:
        c = SRHTCompressor(rank, m, random_seed=123, device="cpu")
        out[f"{tag}_radem"] = c.radem
        out[f"{tag}_col_sampler"] = c.col_sampler
    save("g6_draws.npz", **out)


# ---------------------------------------------------------------- G7: CG iterates
def g7_cg(xgpr):
    """BASELINE cfg1-sized synthetic problem (N=2000, d=32, M=512, chunk 500, seed 123,
    (lambda, sigma) = (0.277, 0.358)) pushed through the reference's own
    build_regression_dataset, RBF kernel, RandNysPreconditioner and CPU_ConjugateGrad
    (fitting_toolkit/cg_tools.py:203-302, cg_fitting_toolkit.py:18-70).  Per-iteration
    iterates come from re-running the (deterministic, x0 = 0) solver with max_iter = j."""
    import warnings
    from xGPR.data_handling.dataset_builder import build_regression_dataset
    from xGPR.kernels import KERNEL_NAME_TO_CLASS
    from xGPR.preconditioners.rand_nys_preconditioners import RandNysPreconditioner
    from xGPR.fitting_toolkit.cg_fitting_toolkit import cg_fit_lib_internal
    from xGPR.scoring_toolkit.exact_nmll_calcs import calc_zty

    rng = np.random.default_rng(123)
    n, d, m = 2000, 32, 512
    x = rng.uniform(-1, 1, size=(n, d))
    a = rng.standard_normal(d)
    y = np.sin(x @ a) + 0.1 * rng.standard_normal(n)
    x = x.astype(np.float32).astype(np.float64)
    hyper = np.array([0.277, 0.358])
    ds = build_regression_dataset(x, y, chunk_size=500)
    out = dict(x=x.astype(np.float32), y=y, hyperparams=hyper, chunk_size=np.int64(500),
               num_rffs=np.int64(m), y_mean=np.float64(ds.get_ymean()),
               y_std=np.float64(ds.get_ystd()))

    for kname, parms in [("RBF", {}), ("Matern", {"matern_nu": 5 / 2})]:
        kern = KERNEL_NAME_TO_CLASS[kname]((n, d), m, random_seed=123, device="cpu",
                                           kernel_spec_parms=parms)
        kern.set_hyperparams(hyper, logspace=False)
        z_first = kern.transform_x(x[:8])
        out[f"{kname}_z_first8"] = z_first
        zty, yty = calc_zty(ds, kern)
        out[f"{kname}_zty"] = zty
        out[f"{kname}_yty"] = np.float64(yty)
        for ptag, method, rank in [("none", None, 0), ("srht", "srht", 64), ("srht2", "srht_2", 64)]:
            if method is None:
                pre = None
            else:
                pre = RandNysPreconditioner(kern, ds, rank, False, 123, method)
                out[f"{kname}_{ptag}_u"] = pre.u_mat
                out[f"{kname}_{ptag}_eig"] = pre.eig
                out[f"{kname}_{ptag}_inv_eig"] = pre.inv_eig
                out[f"{kname}_{ptag}_ratio"] = np.float64(pre.achieved_ratio)
                out[f"{kname}_{ptag}_prefactor"] = np.float64(pre.prefactor)
                out[f"{kname}_{ptag}_zty"] = pre.get_zty()
                out[f"{kname}_{ptag}_yty"] = np.float64(pre.get_yty())
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                w, niter, losses = cg_fit_lib_internal(kern, ds, 1e-8, 500, pre, False)
                iters = []
                for j in range(1, min(niter, 12) + 1):
                    wj, nj, _ = cg_fit_lib_internal(kern, ds, 1e-30, j, pre, False)
                    assert nj == j
                    iters.append(wj)
            out[f"{kname}_{ptag}_weights"] = w
            out[f"{kname}_{ptag}_niter"] = np.int64(niter)
            out[f"{kname}_{ptag}_losses"] = np.asarray(losses)
            out[f"{kname}_{ptag}_iterates"] = np.stack(iters)
            print(f"  G7 {kname} {ptag}: niter={niter}")
    save("g7_cg.npz", **out)


# ---------------------------------------------------------------- G8: reference fixture, end to end
def g8_e2e(xgpr):
    """tests/fitting_tests/test_cg_fit.py:26-40 on the reference's own 381x84 fixture
    (tests/test_data/0_block_train{x,y}values.npy): RBF, 4096 RFFs, rank-256 SRHT
    preconditioner, tol 1e-6 => niter < 10.  The fixture arrays are data files of the
    reference's test-suite and are stored alongside the expected outputs."""
    from xGPR import xGPRegression
    from xGPR.data_handling.dataset_builder import build_regression_dataset
    xtr = np.load(os.path.join(REF_TESTDATA, "0_block_trainxvalues.npy"))
    ytr = np.load(os.path.join(REF_TESTDATA, "0_block_trainyvalues.npy"))
    xte = np.load(os.path.join(REF_TESTDATA, "4_block_testxvalues.npy"))[:64]
    ds = build_regression_dataset(xtr, ytr, chunk_size=2000)
    hparam = np.array([np.log(np.sqrt(0.0767)), np.log(0.358)])
    mod = xGPRegression(num_rffs=4096, kernel_choice="RBF", variance_rffs=12, random_seed=123,
                        device="cpu", kernel_settings={"intercept": True})
    mod.set_hyperparams(hparam, ds)
    pre, ratio = mod.build_preconditioner(ds, max_rank=256, method="srht")
    niter, losses = mod.fit(ds, preconditioner=pre, max_iter=500, run_diagnostics=True,
                            tol=1e-6, mode="cg")
    preds = mod.predict(xte, get_var=False)
    print(f"  G8: niter={niter} ratio={ratio:.4f}")
    save("g8_e2e.npz", xtrain=xtr, ytrain=ytr, xtest=xte, hparam_log=hparam,
         niter=np.int64(niter), losses=np.asarray(losses), weights=np.asarray(mod.weights),
         ratio=np.float64(ratio), preds=preds, zty=pre.get_zty(), yty=np.float64(pre.get_yty()))


# ---------------------------------------------------------------- G9: exact fit + variance (next rows)
def g9_exact(xgpr):
    """mode="exact" on the reference fixture (tests/fitting_tests/test_exact_fit.py:29-37 settings:
    RBF, 512 RFFs here to keep the file small): weights, the variance matrix (variance_rffs = 12),
    predictions.  fitting_toolkit/exact_fitting_toolkit.py:16-72."""
    from xGPR import xGPRegression
    from xGPR.data_handling.dataset_builder import build_regression_dataset
    xtr = np.load(os.path.join(REF_TESTDATA, "0_block_trainxvalues.npy"))
    ytr = np.load(os.path.join(REF_TESTDATA, "0_block_trainyvalues.npy"))
    xte = np.load(os.path.join(REF_TESTDATA, "4_block_testxvalues.npy"))[:32]
    ds = build_regression_dataset(xtr, ytr, chunk_size=2000)
    hparam = np.array([np.log(np.sqrt(0.0767)), np.log(0.358)])
    mod = xGPRegression(num_rffs=512, kernel_choice="RBF", variance_rffs=12, random_seed=123,
                        device="cpu", kernel_settings={"intercept": True})
    mod.set_hyperparams(hparam, ds)
    mod.fit(ds, mode="exact")
    preds, var = mod.predict(xte, get_var=True)
    save("g9_exact.npz", hparam_log=hparam, weights=np.asarray(mod.weights), var=np.asarray(mod.var),
         xtest=xte, preds=preds, pred_var=var)


def g10_nmll(xgpr):
    """Exact NMLL, its gradient and the approximate (SLQ) NMLL on the reference fixture
    (tests/approximate_nmll_tests/test_slq_nmll.py:18-21 hyperparameters, 512 RFFs here):
    xgp_regression.py:152-260 (exact), :264-367 (approximate) with the intermediate CG
    coefficients of scoring_toolkit/approximate_nmll_calcs.py:12-50 recorded."""
    from xGPR import xGPRegression
    from xGPR.data_handling.dataset_builder import build_regression_dataset
    from xGPR.preconditioners.rand_nys_preconditioners import RandNysPreconditioner
    from xGPR.fitting_toolkit.cg_tools import CPU_ConjugateGrad
    from xGPR.scoring_toolkit.probe_generators import generate_normal_probes_cpu
    from xGPR.scoring_toolkit.approximate_nmll_calcs import estimate_logdet
    xtr = np.load(os.path.join(REF_TESTDATA, "0_block_trainxvalues.npy"))
    ytr = np.load(os.path.join(REF_TESTDATA, "0_block_trainyvalues.npy"))
    ds = build_regression_dataset(xtr, ytr, chunk_size=2000)
    out = {}
    settings = {"max_rank": 64, "preconditioner_mode": "srht_2", "nsamples": 25, "nmll_iter": 500,
                "nmll_tol": 1e-6}
    for tag, hparam in (("easy", np.array([0., 1.0])), ("hard", np.array([np.log(1e-3), 1.0]))):
        mod = xGPRegression(num_rffs=512, kernel_choice="RBF", variance_rffs=12, random_seed=123,
                            device="cpu", kernel_settings={"intercept": True})
        out[f"{tag}_hparam_log"] = hparam
        out[f"{tag}_exact_nmll"] = np.float64(mod.exact_nmll(hparam, ds))
        nll, grad = mod.exact_nmll_gradient(hparam, ds)
        out[f"{tag}_grad_nmll"] = np.float64(nll)
        out[f"{tag}_grad"] = np.asarray(grad)
        out[f"{tag}_approx_nmll"] = np.float64(mod.approximate_nmll(hparam, ds, manual_settings=dict(settings)))
        # the same computation step by step, to record what the test double must reproduce
        kernel = mod.kernel
        kernel.set_hyperparams(hparam, logspace=True)
        pre = RandNysPreconditioner(kernel, ds, settings["max_rank"], False, 123, settings["preconditioner_mode"])
        probes = generate_normal_probes_cpu(settings["nsamples"], kernel.get_num_rffs(), 123, pre)
        resid = np.zeros((kernel.get_num_rffs(), 2, settings["nsamples"] + 1))
        resid[:, 0, 0] = pre.get_zty() / ds.get_ndatapoints()
        resid[:, 0, 1:] = probes
        x_k, alphas, betas = CPU_ConjugateGrad().fit(ds, kernel, pre, resid, settings["nmll_iter"],
                                                     settings["nmll_tol"], verbose=False, nmll_settings=True)
        out[f"{tag}_probes"] = probes
        out[f"{tag}_alphas"] = alphas
        out[f"{tag}_betas"] = betas
        out[f"{tag}_xk0"] = x_k[:, 0] * ds.get_ndatapoints()
        out[f"{tag}_precond_logdet"] = np.float64(pre.get_logdet())
        out[f"{tag}_logdet"] = np.float64(estimate_logdet(alphas, betas, kernel.get_num_rffs(), pre, "cpu"))
    save("g10_nmll.npz", **out)


def g11_classifier(xgpr):
    """xGPClassification on the wine data of the reference's own classifier test
    (tests/utils/build_classification_dataset.py:15-44, tests/fitting_tests/test_cg_fit.py:76-91:
    RBF, DISCRIM_HPARAM = (-1, -0.75), rank-256 "srht" preconditioner, tol 1e-2 => niter < 10; 1024
    RFFs here to keep the file small): the first cost-function evaluation, the nonlinear-CG losses,
    the weights and the class probabilities (fitting_toolkit/nonlinear_cg_toolkit.py:72-275,
    xgp_classification.py:59-109).  The standardised inputs are stored with the outputs."""
    import sklearn.datasets
    from sklearn.preprocessing import StandardScaler
    from xGPR import xGPClassification
    from xGPR.data_handling.dataset_builder import build_classification_dataset
    from xGPR.fitting_toolkit.nonlinear_cg_toolkit import nonlinear_CG_classification
    xvalues, yvalues = sklearn.datasets.load_wine(return_X_y=True)
    xvalues = StandardScaler().fit_transform(xvalues)
    rng = np.random.default_rng(123)
    idx = rng.permutation(xvalues.shape[0])
    xvalues, yvalues = xvalues[idx, :], yvalues[idx]
    cutoff = int(0.75 * idx.shape[0])
    xtr, ytr, xte, yte = xvalues[:cutoff], yvalues[:cutoff], xvalues[cutoff:], yvalues[cutoff:]
    ds = build_classification_dataset(xtr, ytr, chunk_size=2000)
    hparam = np.array([-1, -0.75])
    mod = xGPClassification(num_rffs=1024, kernel_choice="RBF", random_seed=123, device="cpu",
                            kernel_settings={"intercept": True}, verbose=False)
    mod.set_hyperparams(hparam, ds)
    pre, ratio = mod.build_preconditioner(ds, max_rank=256, method="srht")
    # one cost-function evaluation at a fixed, non-trivial weight matrix
    wrng = np.random.default_rng(7)
    w_probe = 0.05 * wrng.standard_normal((1024, int(ds.get_n_classes())))
    op = nonlinear_CG_classification(ds, mod.kernel, "cpu", False, pre)
    grad_probe, loss_probe = op.cost_fun_classification(w_probe)
    niter, losses = mod.fit(ds, preconditioner=pre, max_iter=500, run_diagnostics=True, tol=1e-2)
    probs = mod.predict(xte)
    print(f"  G11: niter={niter} losses={losses[:3]}... test acc={(probs.argmax(axis=1) == yte).mean():.3f}")
    save("g11_classifier.npz", xtrain=xtr, ytrain=ytr.astype(np.int64), xtest=xte, ytest=yte.astype(np.int64),
         hparam_log=hparam.astype(np.float64), ratio=np.float64(ratio), w_probe=w_probe, grad_probe=grad_probe,
         loss_probe=np.float64(loss_probe), niter=np.int64(niter), losses=np.asarray(losses),
         weights=np.asarray(mod.weights), probs=probs)


def g12_mini_ard(xgpr):
    """MiniARD: the reference's OWN ground truth for cpu/cudaMiniARDGrad, built as in its
    tests/fht_operations_tests/test_ARD_kernel_gradient.py:120-162 from two routes that do not involve the
    gradient operator: features from MiniARD.transform_x (per-feature scaling + the SORF operator on the
    compiled reference core) and the gradient from an einsum with MiniARD.precompute_weights() (three FHT
    rounds on the identity).  Settings are the test's (:22-38), trimmed to a few rows / frequencies."""
    from xGPR.kernels.ARD_kernels.mini_ard import MiniARD
    out = {}
    cases = [((6, 50), 128, [25], False), ((6, 50), 128, [25], True), ((5, 232), 96, [100, 200], False),
             ((3, 2049), 48, [30, 450], True)]
    for ci, (xdim, num_freqs, split_points, icpt) in enumerate(cases):
        rng = np.random.default_rng(123)
        x = rng.uniform(low=-10.0, high=10.0, size=xdim)
        hp = np.array([1.0] + [0.02 + 0.01 * i for i in range(len(split_points) + 1)])   # lambda, inverse lengthscales

        def build(intercept):
            k = MiniARD(xdim, 2 * num_freqs, 123, device="cpu", double_precision=True,
                        kernel_spec_parms={"split_points": split_points, "intercept": intercept})
            k.set_hyperparams(hp, logspace=False)
            k.precompute_weights()
            return k
        kernel, nik = build(icpt), build(False)
        xtrans = kernel.transform_x(x)
        pw = kernel.precomputed_weights.copy()
        if icpt:
            xtrans[:, 0] = nik.transform_x(x)[:, 0]
            xtrans[:, 0] /= np.sqrt(2 / (pw.shape[0]))
            xtrans[:, 0] *= np.sqrt(2 / (pw.shape[0] - 0.5))
        grad = np.zeros((xtrans.shape[0], xtrans.shape[1], len(split_points) + 1))
        ks = kernel.split_pts
        for i in range(ks.shape[0] - 1):
            tmp = np.einsum("ij,kj->ki", pw[:, ks[i]:ks[i + 1]], x[:, ks[i]:ks[i + 1]])
            for j in range(pw.shape[0]):
                grad[:, 2 * j, i] = -tmp[:, j] * xtrans[:, 2 * j + 1]
                grad[:, 2 * j + 1, i] = tmp[:, j] * xtrans[:, 2 * j]
        out[f"c{ci}_x"] = x
        out[f"c{ci}_num_freqs"] = np.int64(num_freqs)
        out[f"c{ci}_split_points"] = np.asarray(split_points, dtype=np.int64)
        out[f"c{ci}_intercept"] = np.bool_(icpt)
        out[f"c{ci}_hyperparams"] = hp
        out[f"c{ci}_features"] = xtrans          # column 0 is the un-overwritten cos feature (see the test's note)
        out[f"c{ci}_grad"] = grad
        out[f"c{ci}_weights"] = pw
        out[f"c{ci}_transform_x"] = kernel.transform_x(x)
    out["ncases"] = np.int64(len(cases))
    save("g12_mini_ard.npz", **out)


def g13_rank_selection(xgpr):
    """Preconditioner rank selection on the reference fixture (model_baseclass.py:376-480,
    rand_nys_constructors.py:60-93, :301-357): the sampled ratio for a few (sample_frac, rank) pairs and the
    rank / achieved ratio the autoselection ends with for two ratio targets."""
    from xGPR import xGPRegression
    from xGPR.data_handling.dataset_builder import build_regression_dataset
    xtr = np.load(os.path.join(REF_TESTDATA, "0_block_trainxvalues.npy"))
    ytr = np.load(os.path.join(REF_TESTDATA, "0_block_trainyvalues.npy"))
    ds = build_regression_dataset(xtr, ytr, chunk_size=100)
    hparam = np.array([np.log(np.sqrt(0.0767)), np.log(0.358)])
    mod = xGPRegression(num_rffs=512, kernel_choice="RBF", variance_rffs=12, random_seed=123, device="cpu",
                        kernel_settings={"intercept": True}, verbose=False)
    mod.set_hyperparams(hparam, ds)
    out = {"hparam_log": hparam, "chunk_size": np.int64(100)}
    fr, rk, ratios = [], [], []
    for sample_frac, rank in ((1.0, 32), (0.5, 32), (0.25, 64), (1.0, 128)):
        fr.append(sample_frac), rk.append(rank)
        ratios.append(mod._check_rank_ratio(ds, sample_frac=sample_frac, max_rank=rank))
    out["sample_fracs"], out["ranks"], out["ratios"] = np.asarray(fr), np.asarray(rk), np.asarray(ratios)
    for tag, target in (("t30", 30.), ("t3", 3.)):
        pre = mod._autoselect_preconditioner(ds, min_rank=16, max_rank=200, increment_size=48, ratio_target=target)
        out[f"{tag}_rank"] = np.int64(pre.u_mat.shape[1])
        out[f"{tag}_achieved_ratio"] = np.float64(pre.achieved_ratio)
    save("g13_rank_selection.npz", **out)


def g14_two_layer(xgpr):
    """Conv1dTwoLayer (kernels/convolution_kernels/l2_conv1d.py): features and sigma-gradient from the
    reference's kernel class on the compiled reference core, for ragged sequences."""
    from xGPR.kernels.convolution_kernels.l2_conv1d import Conv1dTwoLayer
    rng = np.random.default_rng(123)
    n, L, C = 6, 24, 7
    x = rng.uniform(-1, 1, size=(n, L, C))
    sl = np.array([24, 9, 17, 5, 24, 12], dtype=np.int32)
    hp = np.array([0.7, 0.35])
    k = Conv1dTwoLayer((n, L, C), 128, 123, device="cpu", kernel_spec_parms={"conv_width": 5, "init_rffs": 96,
                                                                              "intercept": True})
    k.set_hyperparams(hp, logspace=False)
    feats = k.transform_x(x, sl)
    gfeats, grad = k.gradient_x(x, sl)
    save("g14_two_layer.npz", x=x, seqlen=sl, hyperparams=hp, features=feats, grad_features=gfeats, grad=grad,
         conv_width=np.int64(5), init_rffs=np.int64(96), num_rffs=np.int64(128))


def g15_crude_tuning(xgpr):
    """tune_hyperparams_crude on the reference fixture (xgp_regression.py:497-561, lb_optimizer.py, bayes_grid.py):
    the lambda search at a few fixed sigmas and the result of the whole Bayesian loop."""
    from xGPR import xGPRegression
    from xGPR.data_handling.dataset_builder import build_regression_dataset
    from xGPR.scoring_toolkit.lb_optimizer import shared_hparam_search
    xtr = np.load(os.path.join(REF_TESTDATA, "0_block_trainxvalues.npy"))
    ytr = np.load(os.path.join(REF_TESTDATA, "0_block_trainyvalues.npy"))
    ds = build_regression_dataset(xtr, ytr, chunk_size=2000)
    mod = xGPRegression(num_rffs=512, kernel_choice="RBF", variance_rffs=12, random_seed=123, device="cpu",
                        kernel_settings={"intercept": True}, verbose=False)
    mod.set_hyperparams(np.array([0., 0.]), ds)
    bounds = mod.kernel.get_bounds()
    sig, sc, lb = [], [], []
    for s in (-3.0, -1.0272223, 0.5):
        score, best_lb = shared_hparam_search(np.array([s]), mod.kernel, ds, bounds[:1, :])
        sig.append(s), sc.append(score), lb.append(best_lb[0])
    hp, nfev, best = mod.tune_hyperparams_crude(ds)
    save("g15_crude_tuning.npz", bounds=bounds, sigmas=np.asarray(sig), scores=np.asarray(sc), best_lbs=np.asarray(lb),
         crude_hparams=np.asarray(hp), crude_nfev=np.int64(nfev), crude_best=np.float64(best))


def g16_aux(xgpr):
    """KernelFGen and FastConv1d (kernel_fgen.py, static_layers/fast_conv.py): feature arrays from the reference's
    classes for small inputs."""
    from xGPR import KernelFGen, FastConv1d
    rng = np.random.default_rng(123)
    x2 = rng.uniform(-1, 1, size=(7, 19))
    fg = KernelFGen(num_rffs=64, hyperparams=np.array([np.log(0.6)]), num_features=19, kernel_choice="Matern",
                    device="cpu", kernel_settings={"matern_nu": 1.5}, random_seed=123, verbose=False)
    x3 = rng.uniform(-1, 1, size=(5, 30, 4))
    sl = np.array([30, 9, 12, 30, 21], dtype=np.int32)
    fc = FastConv1d(seq_width=4, device="cpu", random_seed=123, conv_width=9, num_features=70)
    save("g16_aux.npz", x2=x2, fgen=fg.predict(x2), x3=x3, seqlen=sl, fastconv=fc.predict(x3, sl))


# ---------------------------------------------------------------- G17: BASELINE configs[3] shape (cfg4), conv kernel
def cfg4_inputs(nseq=4):
    """The cfg4-shaped input both this script and the GPU tests build (kept out of the .npz: it is a seeded draw):
    one-hot protein-like sequences, L = 512, 21 channels, lengths spanning conv_width .. 512."""
    rng = np.random.default_rng(123)
    L, C = 512, 21
    seqlen = np.array([512, 64, 301, 9, 130, 477][:nseq], dtype=np.int32)
    x = np.zeros((nseq, L, C), dtype=np.float32)
    for i in range(nseq):
        x[i, np.arange(seqlen[i]), rng.integers(0, C, size=seqlen[i])] = 1.0
        # positions past seqlen hold junk the operator must never read into a k-mer
        x[i, seqlen[i]:, :] = rng.standard_normal((L - seqlen[i], C)).astype(np.float32)
    return x, seqlen


def g17_cfg4_conv(xgpr):
    """The reference's Conv1dRBF kernel class at BASELINE configs[3]'s shape -- L = 512, C = 21, conv_width 9
    (padded window 256), 16384 RFFs = 8 wave tiles, 32 SORF repeats -- through its own transform_x
    (kernels/convolution_kernels/conv_kernel_baseclass.py:116-147 -> rbf_convolution.cpp:84-136) for the three
    averaging modes; sequence lengths from one k-mer (9) to the full 504 k-mers."""
    from xGPR.kernels import KERNEL_NAME_TO_CLASS
    x, seqlen = cfg4_inputs()
    hyper = np.array([1.0, 0.8])
    out = dict(hyperparams=hyper, seqlen=seqlen, num_rffs=np.int64(16384), conv_width=np.int64(9),
               x_checksum=np.float64(np.abs(x.astype(np.float64)).sum()))
    for avg in ("none", "sqrt", "full"):
        kern = KERNEL_NAME_TO_CLASS["Conv1dRBF"](x.shape, 16384, random_seed=123, device="cpu",
                                                 kernel_spec_parms={"conv_width": 9, "averaging": avg})
        kern.set_hyperparams(hyper, logspace=False)
        out[f"z_{avg}"] = kern.transform_x(x.astype(np.float64), seqlen)
    save("g17_cfg4_conv.npz", **out)


# ---------------------------------------------------------------- G18: BASELINE configs[4] shape (cfg5), preconditioner build
def cfg5_inputs(n=4096, d=512):
    """Seeded cfg5-shaped inputs (X ~ N(0,1)/sqrt(d), y = sin(Xa) + 0.1 eps), float32-representable."""
    rng = np.random.default_rng(123)
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    a = rng.standard_normal(d) * 3.0
    y = np.sin(x.astype(np.float64) @ a) + 0.1 * rng.standard_normal(n)
    return x, y


def g18_cfg5_precond(xgpr):
    """The reference's randomized-Nystrom preconditioner at BASELINE configs[4]'s shape -- d = 512, 32768 RFFs
    (SRHT width 32768 in float64), rank 2048, methods srht and srht_2 -- on 4096 datapoints: its own
    SRHTCompressor + single_pass_srht_zty for the accumulated sketch (rand_nys_constructors.py:96-123,
    srht_compressor.py:87-97), RandNysPreconditioner for eigenvalues / ratio / U (:127-296,
    rand_nys_preconditioners.py:18-72), and the preconditioned CG solve that follows."""
    import warnings
    from xGPR.data_handling.dataset_builder import build_regression_dataset
    from xGPR.kernels import KERNEL_NAME_TO_CLASS
    from xGPR.kernels.srht_compressor import SRHTCompressor
    from xGPR.preconditioners.rand_nys_constructors import single_pass_srht_zty
    from xGPR.preconditioners.rand_nys_preconditioners import RandNysPreconditioner
    from xGPR.fitting_toolkit.cg_fitting_toolkit import cg_fit_lib_internal
    n, d, m, rank = 4096, 512, 32768, 2048
    x, y = cfg5_inputs(n, d)
    hyper = np.array([0.1, 1.0])
    ds = build_regression_dataset(x.astype(np.float64), y, chunk_size=2048)
    kern = KERNEL_NAME_TO_CLASS["RBF"]((n, d), m, random_seed=123, device="cpu", kernel_spec_parms={})
    kern.set_hyperparams(hyper, logspace=False)
    out = dict(hyperparams=hyper, n=np.int64(n), d=np.int64(d), num_rffs=np.int64(m), rank=np.int64(rank),
               chunk_size=np.int64(2048), x_checksum=np.float64(np.abs(x.astype(np.float64)).sum()),
               y_checksum=np.float64(np.abs(y).sum()), y_mean=np.float64(ds.get_ymean()), y_std=np.float64(ds.get_ystd()))
    # the accumulated sketch itself, summarised: Frobenius norm, a fixed probe contraction, sampled entries
    comp = SRHTCompressor(rank, m, device="cpu", random_seed=123)
    acc = np.zeros((rank, m))
    zty = np.zeros(m)
    yty = single_pass_srht_zty(ds, kern, comp, acc, zty, False)
    probe_r, probe_c = np.cos(np.arange(rank) * 0.37), np.sin(np.arange(m) * 0.11)
    ri, ci = (np.arange(64) * 31) % rank, (np.arange(64) * 509) % m
    out.update(acc_fro=np.float64(np.linalg.norm(acc)), acc_probe=np.float64(probe_r @ acc @ probe_c),
               acc_left=probe_r @ acc, acc_samples=acc[ri, ci], acc_sample_rows=ri, acc_sample_cols=ci,
               zty=zty, yty=np.float64(yty), srht_radem=comp.radem, srht_col_sampler=comp.col_sampler)
    # one SRHT'd + sampled feature row block for the operator-level check (8 rows)
    z8 = kern.transform_x(x[:8].astype(np.float64))
    out["z_first8"] = z8
    out["z8_compressed"] = comp.transform_x(z8)
    del acc
    v = np.linspace(-1, 1, m)
    for ptag, method in (("srht", "srht"), ("srht2", "srht_2")):
        pre = RandNysPreconditioner(kern, ds, rank, False, 123, method)
        out[f"{ptag}_eig"] = pre.eig
        out[f"{ptag}_ratio"] = np.float64(pre.achieved_ratio)
        out[f"{ptag}_prefactor"] = np.float64(pre.prefactor)
        out[f"{ptag}_uutv"] = pre.u_mat @ (pre.u_mat.T @ v)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            w, niter, losses = cg_fit_lib_internal(kern, ds, 1e-6, 500, pre, False)
        out[f"{ptag}_weights"] = w
        out[f"{ptag}_niter"] = np.int64(niter)
        out[f"{ptag}_losses"] = np.asarray(losses)
        print(f"  G18 {ptag}: ratio {pre.achieved_ratio:.4g}, niter {niter}")
        del pre
    save("g18_cfg5_precond.npz", **out)


if __name__ == "__main__":
    # python make_golden.py            -> everything;   python make_golden.py g17 g18  -> only the named fixtures
    plain = [g1_fht, g2_rbf, g3_conv, g4_maxpool, g5_srht]
    with_ref = [g6_draws, g7_cg, g8_e2e, g9_exact, g10_nmll, g11_classifier, g12_mini_ard, g13_rank_selection,
                g14_two_layer, g15_crude_tuning, g16_aux, g17_cfg4_conv, g18_cfg5_precond]
    wanted = set(sys.argv[1:])

    def selected(fn):
        return not wanted or fn.__name__.split("_")[0] in wanted
    for fn in plain:
        if selected(fn):
            fn()
    if any(selected(fn) for fn in with_ref):
        xgpr = import_reference()
        for fn in with_ref:
            if selected(fn):
                fn(xgpr)
