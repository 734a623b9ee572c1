"""Per-workgroup timing of the fused matvec (which XCD finishes when, and at what clock): needs a development build of the
library with -DXGPR_ZTZ_TIMING,
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DXGPR_ZTZ_TIMING xgpr_amd/csrc/xgpr_hip.hip -o tools/ablate/libxgpr_timing.so
    XGPR_HIP_LIB=tools/ablate/libxgpr_timing.so python tools/wg_times.py [rows]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
d, m = 1024, 8192
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
R = kern.radem_diag.shape[2]
mb = int(_lib.load().xgpr_rbf_workspace_bytes(R))
F = m // 2
off = mb + 1000 * 2 * F * 8
dbg = ws[off:off + 256 * 4 * 8].view(torch.float64).reshape(256, 4).cpu().numpy()
t0, t1, cyc, smid = dbg[:, 0], dbg[:, 1], dbg[:, 2], dbg[:, 3]
print("rows", n, "wall_clock ticks: start spread", t0.max() - t0.min(), "end spread", t1.max() - t1.min(), "durations min/med/max", (t1 - t0).min(), np.median(t1 - t0), (t1 - t0).max())
dur = (t1 - t0)
# wall_clock64 ticks at 100 MHz -> 10 ns
print("durations us: min %.1f med %.1f max %.1f; end times rel to first end (us): p50 %.1f p90 %.1f max %.1f" % (dur.min() / 100, np.median(dur) / 100, dur.max() / 100, np.percentile(t1 - t1.min(), 50) / 100, np.percentile(t1 - t1.min(), 90) / 100, (t1.max() - t1.min()) / 100))
xcd = np.arange(256) % 8
for x in range(8):
    print("  XCD %d: mean duration %.1f us  mean clock %.3f GHz" % (x, dur[xcd == x].mean() / 100, (cyc[xcd == x] / (dur[xcd == x] * 10)).mean()))
