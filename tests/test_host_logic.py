"""CPU-only tests of the host-side mirror of the reference's kernel objects: the random
draws (made on the host with the reference's numpy / scipy calls) must be bit-exact against
the draws the reference's own kernel classes made (tests/golden/g6_draws.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def test_kernel_draws_bit_exact():
    from xgpr_amd.kernels import make_kernel, SRHTCompressor
    g = load_golden("g6_draws.npz")
    cases = [("cfg1_RBF", "RBF", (1, 32), 512, {}), ("cfg2_RBF", "RBF", (1, 256), 4096, {}),
             ("cfg3_Matern", "Matern", (1, 1024), 8192, {"matern_nu": 5 / 2}),
             ("cfg3_Cauchy", "Cauchy", (1, 1024), 8192, {}), ("cfg5_RBF", "RBF", (1, 512), 32768, {}),
             ("fix_RBF", "RBF", (1, 84), 4096, {}), ("small_RBF", "RBF", (1, 3), 64, {}),
             ("cfg4_Conv1dRBF", "Conv1dRBF", (1, 512, 21), 16384, {"conv_width": 9}),
             ("graph_GraphRBF", "GraphRBF", (1, 30, 12), 1024, {}),
             ("conv_Conv1dMatern", "Conv1dMatern", (1, 60, 21), 2048, {"conv_width": 5, "matern_nu": 3 / 2})]
    for tag, name, xdim, rffs, parms in cases:
        k = make_kernel(name, xdim, rffs, random_seed=123, device="cpu", kernel_spec_parms=parms)
        assert k.radem_diag.dtype == torch.int8 and k.chi_arr.dtype == torch.float32
        assert np.array_equal(k.radem_diag.numpy(), g[f"{tag}_radem"]), tag
        assert np.array_equal(k.chi_arr.numpy(), g[f"{tag}_chi"]), tag
    for tag, rank, m in [("srht_256_4096", 256, 4096), ("srht_512_8192", 512, 8192),
                         ("srht_64_512", 64, 512), ("srht_100_1000", 100, 1000)]:
        c = SRHTCompressor(rank, m, device="cpu", random_seed=123)
        assert np.array_equal(c.radem.numpy(), g[f"{tag}_radem"])
        assert np.array_equal(c.col_sampler.numpy(), g[f"{tag}_col_sampler"])


def test_draw_anchors_from_survey():
    """Anchors recorded in SURVEY.md section 8c (seed 123)."""
    import hashlib
    from xgpr_amd.kernels import make_kernel
    k = make_kernel("Matern", (1, 1024), 8192, device="cpu", kernel_spec_parms={"matern_nu": 2.5})
    assert hashlib.sha256(k.radem_diag.numpy().tobytes()).hexdigest()[:16] == "de51cbb1014e6be3"
    assert np.allclose(k.chi_arr[:3].numpy(), [49.065594, 23.925774, 25.538355], rtol=1e-7)
    k = make_kernel("RBF", (1, 32), 512, device="cpu")
    assert hashlib.sha256(k.radem_diag.numpy().tobytes()).hexdigest()[:16] == "61db7b68b904dece"
    assert k.radem_diag.numpy().ravel()[:8].tolist() == [-1, 1, 1, -1, 1, -1, -1, -1]


def test_kernel_argument_errors():
    from xgpr_amd.kernels import make_kernel
    with pytest.raises(RuntimeError):
        make_kernel("RBF", (1, 10), 31, device="cpu")
    with pytest.raises(ValueError):
        make_kernel("RBF", (1, 10, 3), 32, device="cpu")
    with pytest.raises(ValueError):
        make_kernel("Matern", (1, 10), 32, device="cpu")
    with pytest.raises(ValueError):
        make_kernel("Conv1dRBF", (1, 10, 3), 32, device="cpu")
    with pytest.raises(RuntimeError):
        make_kernel("Polynomial", (1, 10), 32, device="cpu")        # not a kernel of the reference's registry
    with pytest.raises(ValueError):
        make_kernel("MiniARD", (1, 10), 32, device="cpu")           # split_points missing
    with pytest.raises(ValueError):
        make_kernel("Conv1dTwoLayer", (1, 10, 3), 32, device="cpu", kernel_spec_parms={"conv_width": 3})
    lin = make_kernel("Linear", (1, 10), 32, device="cpu")
    assert lin.get_num_rffs() == 11 and make_kernel("Linear", (1, 10), 32, device="cpu",
                                                    kernel_spec_parms={"intercept": False}).get_num_rffs() == 10


def test_scale_input_matches_numpy_inplace():
    """``input_x *= hyperparams[1]`` on a float32 array with an np.float64 scalar (numpy 2)."""
    from xgpr_amd.kernels import scale_input
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 7)).astype(np.float32)
    hp = np.array([0.3, 0.358])
    ref = x.copy()
    ref *= hp[1]
    got = scale_input(torch.from_numpy(x), hp[1]).numpy()
    assert np.array_equal(got, ref)


def test_shard_bounds_cover_rows():
    from xgpr_amd.dist import Comm
    for n in [1, 7, 8, 1000003]:
        for ws in [1, 2, 3, 8]:
            spans = [Comm(r, ws).shard_bounds(n) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_offline_dataset_streams_npy_chunk_files(tmp_path):
    """build_offline_np_dataset (reference data_handling/dataset_builder.py:193-340): a list of .npy chunk
    files ends up as the same shard, y statistics and chunks as the in-memory builder, and the reference's
    validation errors are raised."""
    from xgpr_amd.dataset import build_offline_np_dataset, build_regression_dataset
    rng = np.random.default_rng(3)
    xs, ys, xf, yf = [], [], [], []
    for i, n in enumerate((50, 17, 64)):
        x, y = rng.standard_normal((n, 9)), rng.standard_normal(n) * 3 + 1
        np.save(tmp_path / f"x{i}.npy", x)
        np.save(tmp_path / f"y{i}.npy", y)
        xs.append(x), ys.append(y), xf.append(str(tmp_path / f"x{i}.npy")), yf.append(str(tmp_path / f"y{i}.npy"))
    off = build_offline_np_dataset(xf, yf, chunk_size=64, device="cpu")
    ref = build_regression_dataset(np.vstack(xs), np.concatenate(ys), chunk_size=64, device="cpu")
    assert off.get_ndatapoints() == 131 and off.get_xdim() == ref.get_xdim()
    assert off.get_ymean() == ref.get_ymean() and off.get_ystd() == ref.get_ystd()
    for (xa, ya, _), (xb, yb, _) in zip(off.get_chunked_data(), ref.get_chunked_data()):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
    with pytest.raises(RuntimeError):
        build_offline_np_dataset(xf, yf[:2], device="cpu")
    with pytest.raises(RuntimeError):
        build_offline_np_dataset(xf, yf, chunk_size=32, device="cpu")        # a file exceeds chunk_size
    bad = rng.standard_normal((5, 9))
    bad[2, 3] = np.nan
    np.save(tmp_path / "xbad.npy", bad)
    np.save(tmp_path / "ybad.npy", np.zeros(5))
    with pytest.raises(RuntimeError):
        build_offline_np_dataset(xf + [str(tmp_path / "xbad.npy")], yf + [str(tmp_path / "ybad.npy")],
                                 chunk_size=64, device="cpu")
    # classification labels
    yi = [rng.integers(0, 3, size=x.shape[0]) for x in xs]
    yi[0][:3] = [0, 1, 2]
    yif = []
    for i, y in enumerate(yi):
        np.save(tmp_path / f"yi{i}.npy", y)
        yif.append(str(tmp_path / f"yi{i}.npy"))
    cls = build_offline_np_dataset(xf, yif, chunk_size=64, device="cpu", task_type="classification")
    assert cls.get_n_classes() == 3 and cls.get_ndatapoints() == 131


def test_inv_sqrt_apply_gates_the_cholesky_route_on_a_real_condition_estimate():
    """preconditioner._inv_sqrt_apply: acc @ C^(-1/2) up to an orthogonal factor, so the thin SVD of the result must not
    depend on the route.  A sketch with a ROTATED spectrum of condition 1e9 has an innocuous-looking Cholesky diagonal
    (ratio^2 far below 1e6, what the gate used to test) and must NOT take the triangular route; a well-conditioned one
    may.  Both are compared with the SVD form the reference calls (rand_nys_constructors.py:275-285)."""
    import torch
    from xgpr_amd import preconditioner as pc
    gen = torch.Generator().manual_seed(5)
    n, m = 96, 400
    q, _ = torch.linalg.qr(torch.randn(n, n, generator=gen, dtype=torch.float64))
    acc_t = torch.randn(m, n, generator=gen, dtype=torch.float64)

    def reference(c_mat):
        _, s1, v1 = torch.linalg.svd(c_mat, full_matrices=False)
        return acc_t @ v1.T @ ((1 / torch.sqrt(s1))[:, None] * v1)

    for cond, expect_chol in ((1e3, True), (1e9, False)):
        lam = torch.logspace(0, -np.log10(cond), n, dtype=torch.float64)
        c_mat = (q * lam[None, :]) @ q.T
        c_mat = 0.5 * (c_mat + c_mat.T)
        chol = torch.linalg.cholesky(c_mat)
        est = pc._chol_cond_estimate(chol)
        assert cond / 4 <= est <= cond * 1.001, (cond, est)              # a lower bound, within the safety factor
        diag = chol.diagonal()
        if not expect_chol:
            assert float((diag.max() / diag.min()) ** 2) < cond / 100       # the diagonal test would have been fooled
        assert (4.0 * est < pc.CHOL_COND_LIMIT) == expect_chol
        got, ref = pc._inv_sqrt_apply(acc_t, c_mat), reference(c_mat)
        s_got, s_ref = torch.linalg.svdvals(got), torch.linalg.svdvals(ref)
        assert float(((s_got - s_ref).abs() / s_ref).max()) < (1e-9 if expect_chol else 1e-6)
        u_got, u_ref = torch.linalg.svd(got, full_matrices=False)[0], torch.linalg.svd(ref, full_matrices=False)[0]
        lead = 8                                                           # well-separated leading directions
        assert float((u_got[:, :lead].T @ u_ref[:, :lead]).abs().diagonal().min()) > 1 - 1e-6


def test_row_windows_keep_16_byte_alignment_for_any_even_feature_count():
    from xgpr_amd import preconditioner as pc
    for m in (1026, 2050, 4098, 8192, 30):
        step = max(8192, pc.ROW_WINDOW_BYTES // (4 * m)) // 4 * 4
        assert step % 4 == 0 and (step * m * 4) % 16 == 0
