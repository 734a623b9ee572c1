"""The matrix-core contractions of the preconditioner passes over float32 feature rows (xgpr_sketch_gemm_f64) and the
compressor step fed by float32 rows (xgpr_srht_sample_rows_f32) against float64 torch / the separate operators
(reference formulation: preconditioners/rand_nys_constructors.py:34, :54, :113-119; kernels/srht_compressor.py:87-97)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _z64(zc, scale, icpt):
    z = zc.double() * scale
    if icpt:
        z[:, 0] = 1.0
    return z


@pytest.mark.parametrize("n,m,r,icpt,scale,pad", [
    (4096, 1024, 64, True, 0.031, 64), (1000, 516, 37, False, 1.0, 64), (5003, 2048, 200, True, 0.02, 64),
    (16384, 8192, 512, True, 0.0156, 64), (70, 128, 5, True, 0.5, 64),
    # the LDS-staged kernel (features and padded rank multiples of 128): one chunk, an odd chunk count, a tail of
    # K % 16 rows, ragged row tiles, with and without the intercept column, many contraction ranges
    (16, 256, 128, True, 0.07, 128), (48, 256, 128, False, 0.07, 128), (1031, 384, 129, True, 0.05, 128),
    (8192 + 16, 1024, 300, False, 0.03, 128), (100000, 256, 70, True, 0.088, 128)])
def test_sketch_gemm_contract_datapoints(n, m, r, icpt, scale, pad):
    """C[r, M] (+)= S^T Z and its transposed store, ragged shapes, several contraction ranges."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = torch.Generator(device=DEV).manual_seed(n + m)
    zc = torch.rand(n, m, generator=g, device=DEV) * 2 - 1
    lda = (r + pad - 1) // pad * pad
    s = torch.zeros(n, lda, dtype=torch.float64, device=DEV)
    s[:, :r] = torch.randn(n, r, generator=g, device=DEV, dtype=torch.float64)
    ref = s[:, :r].T @ _z64(zc, scale, icpt)
    out = torch.full((r, m), 7.0, dtype=torch.float64, device=DEV)
    ext.hipSketchGemm(s, zc, out, r, False, False, icpt, scale)
    tol = 1e-13 * float(ref.abs().max()) * np.sqrt(n)
    assert float((out - ref).abs().max()) <= tol
    ext.hipSketchGemm(s, zc, out, r, False, False, icpt, scale, accumulate=True)
    assert float((out - 2 * ref).abs().max()) <= 2 * tol
    out_t = torch.zeros((m, r), dtype=torch.float64, device=DEV)
    ext.hipSketchGemm(s, zc, out_t, r, False, True, icpt, scale)
    assert float((out_t - ref.T).abs().max()) <= tol
    again = torch.zeros_like(out_t)
    ext.hipSketchGemm(s, zc, again, r, False, True, icpt, scale)
    assert torch.equal(again, out_t)                    # deterministic


@pytest.mark.parametrize("n,m,r,icpt,scale,pad", [
    (4096, 1024, 64, True, 0.031, 64), (1001, 516, 37, False, 1.0, 64), (3000, 8192, 512, True, 0.0156, 64),
    (65, 128, 5, True, 0.5, 64),
    # the LDS-staged kernel (features a multiple of 16, padded rank a multiple of 128): one chunk, an odd chunk count,
    # ragged datapoint tiles (clamped rows), ragged row tiles, many contraction ranges (intercept added by range 0 only)
    (16, 16, 128, True, 0.3, 128), (130, 48, 128, False, 0.2, 128), (130, 4096, 128, True, 0.02, 128),
    (5000, 1024, 300, True, 0.03, 128), (1, 256, 129, True, 0.07, 128)])
def test_sketch_gemm_contract_features(n, m, r, icpt, scale, pad):
    """T[n, r] = Z Q (stored through the transposed epilogue) and its untransposed form [r, n]: the first product of
    single_pass_gauss."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = torch.Generator(device=DEV).manual_seed(n * 3 + m)
    zc = torch.rand(n, m, generator=g, device=DEV) * 2 - 1
    lda = (r + pad - 1) // pad * pad
    q = torch.zeros(m, lda, dtype=torch.float64, device=DEV)
    q[:, :r] = torch.randn(m, r, generator=g, device=DEV, dtype=torch.float64)
    ref = _z64(zc, scale, icpt) @ q[:, :r]
    tol = 1e-13 * float(ref.abs().max()) * np.sqrt(m)
    t = torch.zeros((n, lda), dtype=torch.float64, device=DEV)
    ext.hipSketchGemm(q, zc, t, r, True, True, icpt, scale)
    assert float((t[:, :r] - ref).abs().max()) <= tol
    assert float(t[:, r:].abs().max()) == 0.0 if lda > r else True      # the padding columns stay zero
    ldc = (n + 1) // 2 * 2
    u = torch.zeros((r, ldc), dtype=torch.float64, device=DEV)
    ext.hipSketchGemm(q, zc, u, r, True, False, icpt, scale)
    assert float((u[:, :n] - ref.T).abs().max()) <= tol
    ext.hipSketchGemm(q, zc, u, r, True, False, icpt, scale, accumulate=True)
    assert float((u[:, :n] - 2 * ref.T).abs().max()) <= 2 * tol


@pytest.mark.parametrize("n,m,rank,icpt", [(300, 4100, 256, True), (257, 8192, 512, True), (100, 1000, 100, False),
                                           (64, 16384, 300, True), (40, 32768, 2048, True), (33, 20000, 1000, False)])
def test_srht_sample_rows_equals_separate_operators(n, m, rank, icpt):
    """Compressed rows and z^T y from float32 rows == float64 Z -> pad -> hipSRHT -> gather (bit for bit) and
    Z^T y -- including padded widths beyond the LDS capacity (16384 x 8 B = 128 KiB fits; 32768 does not)."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    from xgpr_amd.kernels import SRHTCompressor
    g = torch.Generator(device=DEV).manual_seed(m + rank)
    scale = float(np.float32(np.sqrt(1.0 / (m // 2 - (0.5 if icpt else 0.0)))))
    zc = torch.rand(n, m, generator=g, device=DEV) * 2 - 1
    y = torch.randn(n, generator=g, device=DEV, dtype=torch.float64)
    comp = SRHTCompressor(rank, m, device=DEV, random_seed=123)
    z64 = _z64(zc, scale, icpt)
    padded = torch.zeros((n, comp.padded_dims), dtype=torch.float64, device=DEV)
    padded[:, :m] = z64
    ext.hipSRHT(padded, comp.radem)
    want = padded[:, comp.truncated_sampler]
    ldo = (rank + 63) // 64 * 64
    out = torch.full((n, ldo), 3.0, dtype=torch.float64, device=DEV)
    zty = torch.zeros(m, dtype=torch.float64, device=DEV)
    ext.hipSRHTSampleRows(zc, comp.radem, comp.truncated_sampler, out, rank, icpt, scale, y, zty)
    assert torch.equal(out[:, :rank], want)
    assert float(out[:, rank:].abs().max()) == 0.0 if ldo > rank else True
    ref_zty = z64.T @ y
    assert float((zty - ref_zty).abs().max()) <= 1e-12 * float(ref_zty.abs().max()) + 1e-14


@pytest.mark.parametrize("n,m,msub,icpt,scale", [
    (4096, 1024, 1024, True, 0.031),        # whole chunks; 36 tiles x 256 chunks over 512 workgroups: every tile is split
    (1000, 512, 256, False, 1.0),           # leading block of the features (the variance step), a tail of n % 16 rows
    (37, 256, 128, True, 0.5),              # two chunks and a tail
    (9, 128, 128, True, 0.25),              # fewer rows than one chunk: the tail kernel alone
    (20000, 1024, 1024, False, 0.02),       # many workgroups start inside a tile (stream-K spills)
    (16384, 2048, 2048, True, 0.0156),
    # one tile, 1 .. 11 chunks of 16 rows: every tail length of the chunk loop unrolled by 6 (and its pipeline prologue)
    (16, 128, 128, True, 0.5), (32, 128, 128, False, 0.5), (48, 128, 128, True, 0.5), (64, 128, 128, True, 0.5),
    (80, 128, 128, False, 0.5), (96, 128, 128, True, 0.5), (112, 128, 128, True, 0.5), (133, 128, 128, False, 0.5),
    (176, 256, 256, True, 0.5), (208, 256, 128, True, 0.5)])
def test_gram_from_float32_rows(n, m, msub, icpt, scale):
    """C[msub, msub] (+)= Z[:, :msub]^T Z[:, :msub] with both operands the float32 feature rows (xgpr_ztz_gram_f64:
    exact_nmll_calcs.py:42-78, :116-139) against the float64 product over the same rows; both triangles, accumulation,
    a padded row pitch, determinism."""
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = torch.Generator(device=DEV).manual_seed(n + m)
    zc = torch.rand(n, m, generator=g, device=DEV) * 2 - 1
    z = _z64(zc, scale, icpt)[:, :msub]
    ref = z.T @ z
    tol = 1e-13 * float(ref.abs().max()) * np.sqrt(n)
    out = torch.full((msub, msub + 6), 7.0, dtype=torch.float64, device=DEV)      # ldc > msub: the padding stays untouched
    ws = ext.hipZtZGram(zc, out, icpt, scale)
    assert float((out[:, :msub] - ref).abs().max()) <= tol
    assert torch.equal(out[:, :msub], out[:, :msub].T.contiguous())                # mirrored exactly
    assert bool((out[:, msub:] == 7.0).all())
    ext.hipZtZGram(zc, out, icpt, scale, accumulate=True, workspace=ws)
    assert float((out[:, :msub] - 2 * ref).abs().max()) <= 2 * tol
    again = torch.zeros((msub, msub), dtype=torch.float64, device=DEV)
    first = torch.zeros_like(again)
    ext.hipZtZGram(zc, first, icpt, scale, workspace=ws)
    ext.hipZtZGram(zc, again, icpt, scale, workspace=ws)
    assert torch.equal(first, again)                                               # deterministic
