"""Per-wave timeline of one datapoint of the fused matvec (workgroup 0, iteration 200): needs a development build with
-DXGPR_ZTZ_STAMPS (tools/ablate_build.sh stamps "-DXGPR_ZTZ_STAMPS"):
    XGPR_HIP_LIB=tools/ablate/lib_stamps.so python tools/wave_timeline.py [rows d M]
Stamps (s_memtime): 0 loop top, 1 after the prefetch issue, 2 x in registers, then per round s: 2+4s start, 3+4s after flips+strides 1-8,
4+4s after the R->C exchange, 5+4s end of round; 14 before cos/sin, 15 after, 16 after the dot, 17 after the wave sum, 18 after the
barrier, 19 after the rank-1 update."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import _lib
n, d, m = (int(sys.argv[i]) if len(sys.argv) > i else dflt for i, dflt in ((1, 400000), (2, 1024), (3, 8192)))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
xs = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
kern = make_kernel("Matern", (n, d), m, 123, dev, {"matern_nu": 2.5})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
v = torch.randn(m, dtype=torch.float64, device=dev, generator=g)
w = torch.empty_like(v)
ws = torch.zeros(kern.workspace_bytes(), dtype=torch.uint8, device=dev)
for _ in range(5):
    kern.ztz_matvec(xs, v, w, ws)
torch.cuda.synchronize()
mb = int(_lib.load().xgpr_rbf_workspace_bytes(kern.radem_diag.shape[2]))
F = m // 2
off = mb + 1001 * 2 * F * 8
st = ws[off:off + 12 * 32 * 8].view(torch.int64).reshape(12, 32).cpu().numpy()
t0 = st[:, 0].min()
names = {0: "top", 1: "dma", 2: "x", 14: "chi", 15: "sincos", 16: "dot", 17: "sum", 18: "barrier", 19: "update"}
for s in range(3):
    names.update({2 + 4 * s: f"r{s}:in", 3 + 4 * s: f"r{s}:rows", 4 + 4 * s: f"r{s}:xchg", 5 + 4 * s: f"r{s}:cols"})
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19]
print("ticks since the first wave's loop top; columns = waves 0..11 (wave w = slot w / nb, tile w % nb; waves w, w+4, w+8 share a SIMD)")
print("%-10s" % "stamp" + "".join("%7d" % w for w in range(12)))
for k in order:
    print("%-10s" % names[k] + "".join("%7d" % (st[w, k] - t0) for w in range(12)))
print("%-10s" % "segment" + "  (duration per wave)")
for a, b in zip(order[:-1], order[1:]):
    print("%-10s" % names[b] + "".join("%7d" % (st[w, b] - st[w, a]) for w in range(12)))
