"""The matrix-core contraction of the preconditioner pass alone, at a window of cfg3's or cfg5's shape:
    python tools/bench_sketch_gemm.py [rows] [num_rffs] [rank] [bt]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
m = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
r = int(sys.argv[3]) if len(sys.argv) > 3 else 512
bt = len(sys.argv) > 4 and sys.argv[4] == "1"
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
zc = torch.rand(n, m, device=dev, generator=g) * 2 - 1
lda = (r + 63) // 64 * 64
if bt:
    a = torch.randn(m, lda, dtype=torch.float64, device=dev, generator=g)
    out = torch.zeros(n, lda, dtype=torch.float64, device=dev)
    ws = torch.empty(ext.sketch_gemm_workspace_bytes(r, n, m, lda, True), dtype=torch.uint8, device=dev)
    run = lambda: ext.hipSketchGemm(a, zc, out, r, True, True, True, 0.0, workspace=ws)
else:
    a = torch.randn(n, lda, dtype=torch.float64, device=dev, generator=g)
    out = torch.zeros(r, m, dtype=torch.float64, device=dev)
    ws = torch.empty(ext.sketch_gemm_workspace_bytes(r, m, n, m, False), dtype=torch.uint8, device=dev)
    run = lambda: ext.hipSketchGemm(a, zc, out, r, False, False, True, 0.0, accumulate=True, workspace=ws)
for _ in range(2):
    run()
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"sketch gemm bt={int(bt)} rows={n} M={m} rank={r}: {dt*1e3:.2f} ms  {2.0*n*m*r/dt/1e12:.1f} TFLOP/s")
