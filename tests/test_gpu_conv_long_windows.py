"""The convolution feature operator on LONG windows (padded width 512 and 1024: conv_width x channels above 256) under
load: many sequences in one launch, so that a window's loads are still in flight when the previous k-mer's transform ends.
Rounds 2-3 prefetched such windows with loads issued from inline assembly, and the compiler copied their destination
registers in front of the wait that covered them -- a stale window now and then under memory load (never in the small
cases the other tests run; caught by tools/stress_parity.py).  Each launch here is compared with the CPU oracle
(rbf_convolution.cpp:23-140 restated), and twice with itself."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


# (64 x 21 = 1344 and 100 x 21 = 2100: padded windows of 2048 / 4096 elements -- two / four waves of a workgroup per transform, meeting at
#  barriers inside the k-mer loop, every workgroup running to its longest sequence: csrc/wave_tile.inc; float64: the same kernels at every width)
@pytest.mark.timeout(300)
@pytest.mark.parametrize("conv_width,nseq,dtype", [(24, 768, np.float32), (48, 512, np.float32), (13, 768, np.float32), (64, 384, np.float32),
                                                   (100, 256, np.float32), (24, 384, np.float64), (64, 256, np.float64), (100, 128, np.float64)])
def test_long_window_features_match_the_oracle_under_load(oracle, conv_width, nseq, dtype):
    from oracle import oracle as orc
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    rng = np.random.default_rng(conv_width)
    L, C, m = 160, 21, 2048
    x = np.zeros((nseq, L, C), dtype=dtype)
    x[np.arange(nseq)[:, None], np.arange(L)[None, :], rng.integers(0, C, (nseq, L))] = 1.0
    x += 0.01 * rng.standard_normal(x.shape).astype(dtype)
    sl = rng.integers(conv_width, L + 1, size=nseq).astype(np.int32)
    radem, chi = orc.draw_sorf_params(m, conv_width * C, 77, conv=True, double_precision=dtype == np.float64)
    ref = np.zeros((nseq, m))
    oracle.cpuConv1dFGen(x, ref, radem, chi, sl, conv_width, 1)
    xt, rt, ct = (torch.from_numpy(a).to(DEV) for a in (x, radem, chi))
    outs = []
    for _ in range(3):
        out = torch.zeros((nseq, m), dtype=torch.float64, device=DEV)
        ext.hipConv1dFGen(xt, out, rt, ct, sl, conv_width, 1)
        outs.append(out)
    kmax = int(sl.max()) - conv_width + 1
    bar = (4e-7 if dtype == np.float32 else 1e-13) * np.sqrt(2.0 / m) * np.sqrt(kmax)
    err = np.abs(outs[0].cpu().numpy() - ref)
    assert err.max() <= bar, (float(err.max()), bar, np.argwhere(err > bar)[:5].tolist())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
