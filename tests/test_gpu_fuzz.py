"""Seeded random-shape sweep of the operators against the oracle: odd widths, feature counts that are not
multiples of the tile, single rows, sequence lengths at both ends.  Small sizes, one process, fixed seeds
(XGPR_FUZZ_SEED selects another sweep)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_fixed_vector_operator_sweep(oracle):
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel, scale_input, block_workspace_bytes
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(int(os.environ.get("XGPR_FUZZ_SEED", 2026)))
    for case in range(24):
        n = int(rng.integers(1, 300))
        d = int(rng.choice([1, 2, 3, 7, 31, 33, 64, 100, 129, 255, 512, 700, 1024, 1500]))
        m = 2 * int(rng.integers(1, 700))
        icpt = bool(rng.integers(0, 2))
        kind = str(rng.choice(["RBF", "Matern", "Cauchy"]))
        parms = {"intercept": icpt, "matern_nu": 1.5}
        x = rng.uniform(-2, 2, size=(n, d)).astype(np.float32)
        hp = np.array([0.5, float(rng.uniform(0.05, 1.5))])
        kern = make_kernel(kind, x.shape, m, 123, DEV, parms)
        kern.set_hyperparams(hp, logspace=False)
        okern = orc.OracleKernel(kind, m, x.shape, hp, 123, matern_nu=1.5, fit_intercept=icpt, ops=oracle)
        z = kern.transform_x(x).cpu().numpy()
        zref = okern.transform_x(x.astype(np.float64))
        scale = np.sqrt(1.0 / (m // 2 - 0.5 if icpt else m // 2))
        assert np.abs(z - zref).max() <= 1e-6 * scale, (case, n, d, m, kind)
        v = rng.standard_normal(m)
        if kern.fused_ok():
            xs = scale_input(torch.from_numpy(x).to(DEV), hp[1])
            out = torch.empty(m, dtype=torch.float64, device=DEV)
            ws = torch.empty(kern.workspace_bytes(), dtype=torch.uint8, device=DEV)
            kern.ztz_matvec(xs, torch.from_numpy(v).to(DEV), out, ws)
            assert _rel(out.cpu().numpy(), zref.T @ (zref @ v)) < 2e-6, (case, n, d, m, kind)
            y = rng.standard_normal(n)
            kern.zty(xs, torch.from_numpy(y).to(DEV), out, ws)
            assert _rel(out.cpu().numpy(), zref.T @ y) < 2e-6, (case, n, d, m, kind)
            if kern.block_ok():
                k = int(rng.integers(1, 40))
                vv = rng.standard_normal((m, k))
                zc = torch.empty((n, m), dtype=torch.float32, device=DEV)
                kern.fill_feature_cache(xs, zc)
                ob = torch.empty((m, k), dtype=torch.float64, device=DEV)
                kern.ztz_block_cached(zc, torch.from_numpy(vv).to(DEV), ob,
                                      torch.empty(block_workspace_bytes(n, m, k), dtype=torch.uint8, device=DEV))
                assert _rel(ob.cpu().numpy(), zref.T @ (zref @ vv)) < 2e-6, (case, n, d, m, k)
        xg = torch.from_numpy(x).to(DEV)
        og = torch.zeros((n, m), dtype=torch.float64, device=DEV)
        gg = torch.zeros((n, m, 1), dtype=torch.float64, device=DEV)
        ext.hipRBFGrad(xg, og, gg, kern.radem_diag, kern.chi_arr, float(hp[1]), icpt)
        zr, gr = okern.gradient_x(x.astype(np.float64))
        if icpt:
            og[:, 0] = 1.0
            gg[:, 0, :] = 0.0
        assert np.abs(og.cpu().numpy() - zr).max() <= 1e-6 * scale
        assert np.allclose(gg.cpu().numpy(), gr, rtol=1e-4, atol=2e-5 * scale * max(1.0, np.abs(x).max() * d ** 0.5))


def test_sequence_operator_sweep(oracle):
    from oracle import oracle as orc
    from xgpr_amd.kernels import make_kernel
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(int(os.environ.get("XGPR_FUZZ_SEED", 2026)) + 1)
    for case in range(14):
        n = int(rng.integers(1, 40))
        L = int(rng.integers(3, 60))
        C = int(rng.choice([1, 2, 5, 21, 33]))
        w = int(rng.integers(1, min(L, 12) + 1))
        m = 2 * int(rng.integers(1, 300))
        avg = str(rng.choice(["none", "sqrt", "full"]))
        x = rng.standard_normal((n, L, C)).astype(np.float32)
        sl = rng.integers(w, L + 1, size=n).astype(np.int32)
        sl[0] = w                                   # a sequence with a single k-mer
        sl[-1] = L
        hp = np.array([0.5, float(rng.uniform(0.1, 1.0))])
        kern = make_kernel("Conv1dRBF", x.shape, m, 123, DEV, {"conv_width": w, "averaging": avg})
        kern.set_hyperparams(hp, logspace=False)
        okern = orc.OracleKernel("Conv1dRBF", m, x.shape, hp, 123, conv_width=w, averaging=avg, ops=oracle)
        z = kern.transform_x(x, sl).cpu().numpy()
        zref = okern.transform_x(x.astype(np.float64), sl)
        assert np.abs(z - zref).max() <= 2e-6 * max(1.0, np.abs(zref).max()), (case, n, L, C, w, m, avg)
        # max-pool layer: bit-exact
        nf = 2 * int(rng.integers(1, 100))            # the reference requires an even output width (:57-60)
        pd = 2 ** int(np.ceil(np.log2(max(w * C, 2))))
        radem = rng.choice(np.asarray([-1, 1], dtype=np.int8), size=(3, 1, int(np.ceil(nf / pd)) * pd))
        chi = rng.uniform(0.5, 2.0, size=nf).astype(np.float32)
        oref = np.zeros((n, nf), np.float32)
        oracle.cpuConv1dMaxpool(x, oref, np.ascontiguousarray(radem), chi, sl, w)
        og = torch.zeros((n, nf), dtype=torch.float32, device=DEV)
        ext.hipConv1dMaxpool(torch.from_numpy(x).to(DEV), og, torch.from_numpy(np.ascontiguousarray(radem)).to(DEV),
                             torch.from_numpy(chi).to(DEV), sl, w)
        assert np.array_equal(og.cpu().numpy(), oref), (case, n, L, C, w, nf)
