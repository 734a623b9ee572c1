"""Pins the CPU oracle bit-for-bit against the reference's own compiled arithmetic core
(oracle/_ref/libxgpr_ref.so) on seeded random inputs beyond the committed fixtures.
Runs only where oracle/Makefile could build the core (the authoring container); on the
GPU box the prebuilt .so travels with the snapshot, otherwise the test is skipped."""
import numpy as np
import pytest

from oracle import oracle as orc


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fht_and_srht(oracle, refcore, dtype):
    rng = np.random.default_rng(7)
    for P in [2, 8, 64, 128, 2048, 16384]:
        x = rng.standard_normal((5, P)).astype(dtype)
        a, b = x.copy(), x.copy()
        oracle.cpuFastHadamardTransform2D(a)
        refcore.cpuFastHadamardTransform2D(b)
        assert np.array_equal(a, b)
        radem = rng.choice(np.asarray([-1, 1], np.int8), size=P)
        a, b = x.copy(), x.copy()
        oracle.cpuSRHT(a, radem)
        refcore.cpuSRHT(b, radem)
        assert np.array_equal(a, b)
    x = rng.standard_normal((4, 6, 32)).astype(dtype)
    a, b = x.copy(), x.copy()
    oracle.cpuFastHadamardTransform(a)
    refcore.cpuFastHadamardTransform(b)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,rffs,icpt", [(5, 16, False), (32, 512, True), (100, 300, False),
                                         (513, 4096, True), (1024, 2048, False), (3000, 1024, True)])
def test_rbf(oracle, refcore, dtype, d, rffs, icpt):
    rng = np.random.default_rng(11)
    radem, chi = orc.draw_sorf_params(rffs, d, 5, double_precision=(dtype == np.float64))
    x = (rng.standard_normal((7, d)) * 3).astype(dtype)
    a, b = np.zeros((7, rffs)), np.zeros((7, rffs))
    oracle.cpuRBFFeatureGen(x.copy(), a, radem, chi, icpt)
    refcore.cpuRBFFeatureGen(x.copy(), b, radem, chi, icpt)
    assert np.array_equal(a, b)
    a, b = np.zeros((7, rffs)), np.zeros((7, rffs))
    ga, gb = np.zeros((7, rffs, 1)), np.zeros((7, rffs, 1))
    oracle.cpuRBFGrad(x.copy(), a, ga, radem, chi, 1.3, icpt)
    refcore.cpuRBFGrad(x.copy(), b, gb, radem, chi, 1.3, icpt)
    assert np.array_equal(a, b) and np.array_equal(ga, gb)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("L,C,cw,rffs,sc", [(30, 21, 9, 1024, 0), (17, 4, 1, 64, 1), (40, 21, 5, 600, 2),
                                            (12, 300, 4, 512, 1)])
def test_conv(oracle, refcore, dtype, L, C, cw, rffs, sc):
    rng = np.random.default_rng(13)
    n = 5
    radem, chi = orc.draw_sorf_params(rffs, cw * C, 9, double_precision=(dtype == np.float64),
                                      conv=True)
    x = rng.standard_normal((n, L, C)).astype(dtype)
    sl = rng.integers(cw, L + 1, size=n).astype(np.int32)
    a, b = np.zeros((n, rffs)), np.zeros((n, rffs))
    oracle.cpuConv1dFGen(x, a, radem, chi, sl, cw, sc)
    refcore.cpuConv1dFGen(x, b, radem, chi, sl, cw, sc)
    assert np.array_equal(a, b)
    a, b = np.zeros((n, rffs)), np.zeros((n, rffs))
    ga, gb = np.zeros((n, rffs, 1)), np.zeros((n, rffs, 1))
    oracle.cpuConvGrad(x, a, radem, chi, sl, ga, 0.8, cw, sc)
    refcore.cpuConvGrad(x, b, radem, chi, sl, gb, 0.8, cw, sc)
    assert np.array_equal(a, b) and np.array_equal(ga, gb)
    # max-pool: F == num_rffs, radem length exactly reps * P
    P = orc.padded_dims(cw * C)
    F = rffs // 2
    reps = -(-F // P)
    radem_m = radem[:, :, :reps * P].copy() if radem.shape[2] >= reps * P else \
        rng.choice(np.asarray([-1, 1], np.int8), size=(3, 1, reps * P))
    chi_m = np.ascontiguousarray(chi[:F])
    a, b = np.zeros((n, F), np.float32), np.zeros((n, F), np.float32)
    oracle.cpuConv1dMaxpool(x, a, radem_m, chi_m, sl, cw)
    refcore.cpuConv1dMaxpool(x, b, radem_m, chi_m, sl, cw)
    assert np.array_equal(a, b)
