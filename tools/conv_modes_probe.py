"""Max-pool and gradient convolution operators at short and cfg4-sized windows (development aid):
    python tools/conv_modes_probe.py        (XGPR_HIP_LIB selects the build)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import oracle as orc      # (parameter draws only: development tooling)
from xgpr_amd import xgpr_hip_rfgen_ext as ext
dev = "cuda"
n, L, C = 2048, 512, 21
g = torch.Generator(device=dev).manual_seed(3)
x = torch.nn.functional.one_hot(torch.randint(0, C, (n, L), device=dev, generator=g), C).to(torch.float32)
sl = torch.randint(64, L + 1, (n,), generator=torch.Generator().manual_seed(5)).numpy().astype(np.int32)


def timed(fn, reps=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for w in (1, 3, 9):
    m = 4096
    radem, chi = orc.draw_sorf_params(m, w * C, 77, conv=True)
    rd, ch = torch.from_numpy(radem).to(dev), torch.from_numpy(chi).to(dev)
    om = torch.zeros((n, m), dtype=torch.float32, device=dev)             # max-pool: one float32 output per frequency, chi of length m
    rm, cm = orc.draw_sorf_params(2 * m, w * C, 77, conv=True)
    rdm, chm = torch.from_numpy(rm).to(dev), torch.from_numpy(cm).to(dev)
    t_mp = timed(lambda: ext.hipConv1dMaxpool(x, om, rdm, chm, sl, w))
    o = torch.zeros((n, m), dtype=torch.float64, device=dev)
    gr = torch.zeros((n, m, 1), dtype=torch.float64, device=dev)
    t_g = timed(lambda: ext.hipConvGrad(x, o, rd, ch, sl, gr, 0.8, w, 1))
    o2 = torch.zeros((n, m), dtype=torch.float64, device=dev)
    t_f = timed(lambda: ext.hipConv1dFGen(x, o2, rd, ch, sl, w, 1))
    print(f"conv_width {w}: feature operator {t_f:.2f} ms, gradient operator {t_g:.2f} ms, max-pool {t_mp:.2f} ms per {n} sequences x {m}; checksums {float(o2.sum()):.8e} {float(gr.sum()):.8e} {float(om.sum()):.8e}")
