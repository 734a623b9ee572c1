"""Z^T Z from float32 feature rows on the matrix cores (xgpr_ztz_gram_f64), one window:
    python tools/bench_gram.py [rows] [num_rffs]
Reports the time and the EXECUTED flop rate (tiles on or above the diagonal: rows x M x (M + 128) flop) against the FP64
matrix peak, next to the library GEMM on a float64 copy of the same rows (the formulation this replaces)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
m = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
zc = torch.rand(n, m, device=dev, generator=g) * 2 - 1
out = torch.zeros(m, m, dtype=torch.float64, device=dev)
ws = ext.hipZtZGram(zc, out, True, 0.0)
for _ in range(2):
    ext.hipZtZGram(zc, out, True, 0.0, accumulate=True, workspace=ws)
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    ext.hipZtZGram(zc, out, True, 0.0, accumulate=True, workspace=ws)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
executed = float(n) * m * (m + 128)          # 2 * n * 128 * 128 flop per tile, M / 128 * (M / 128 + 1) / 2 tiles
print(f"gram rows={n} M={m}: {dt*1e3:.2f} ms  executed {executed/dt/1e12:.1f} TFLOP/s = {executed/dt/1e12/78.6:.3f} of the FP64 matrix peak"
      f"  (full-product equivalent {2.0*n*m*m/dt/1e12:.1f} TFLOP/s)")
rows64 = min(n, 32768)
z = zc[:rows64].double()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    out.addmm_(z.T, z)
torch.cuda.synchronize(); dl = (time.perf_counter() - t0) / reps
print(f"library GEMM on float64 Z ({rows64} rows, Z already converted and in memory): {dl*1e3:.2f} ms  {2.0*rows64*m*m/dl/1e12:.1f} TFLOP/s")
