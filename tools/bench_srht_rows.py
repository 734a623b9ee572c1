"""SRHT + sample + z^T y from float32 feature rows:  python tools/bench_srht_rows.py [rows] [num_rffs] [rank]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
from xgpr_amd.kernels import SRHTCompressor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
m = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
r = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
zc = torch.rand(n, m, device=dev, generator=g) * 2 - 1
y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
comp = SRHTCompressor(r, m, device=dev, random_seed=123)
ldo = (r + 63) // 64 * 64
out = torch.empty(n, ldo, dtype=torch.float64, device=dev)
zty = torch.empty(m, dtype=torch.float64, device=dev)
ws = torch.empty(ext.srht_sample_workspace_bytes(m), dtype=torch.uint8, device=dev)
run = lambda: ext.hipSRHTSampleRows(zc, comp.radem, comp.truncated_sampler, out, r, True, 0.0, y, zty, ws)
for _ in range(2):
    run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"srht rows: n={n} M={m} rank={r}: {dt*1e3:.3f} ms ({dt/n*1e9:.1f} ns/row, input {n*m*4/dt/1e12:.2f} TB/s) checksum {float(out.sum()):.10e} {float(zty.sum()):.10e}")
