"""Block (k right-hand sides) matvec over the resident feature cache: check against a float64 torch
contraction and time it.  Usage: python tools/bench_block.py [rows] [num_rffs] [k]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
k = int(sys.argv[3]) if len(sys.argv) > 3 else 26
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
zc = torch.rand(rows, M, dtype=torch.float32, device=dev, generator=g) * 2 - 1
V = torch.randn(M, k, dtype=torch.float64, device=dev, generator=g)
W = torch.empty_like(V)
ws = torch.empty(ext.zcache_block_workspace_bytes(rows, M, k), dtype=torch.uint8, device=dev)
for icpt in (False, True):
    ext.hipZCacheBlockMatvec(zc, V, W, icpt, ws)
    torch.cuda.synchronize()
    F = M // 2
    s = (1.0 / (F - 0.5 if icpt else F)) ** 0.5
    nchk = min(rows, 8192)
    if nchk == rows:
        Z = zc.double()
        if icpt:
            Z[:, 0] = 1.0 / s
        ref = (Z.T @ (Z @ V)) * s * s
        err = ((W - ref).abs().max() / ref.abs().max()).item()
        print(f"intercept={icpt} max rel err vs torch f64: {err:.3e}")
if rows > 8192:
    # linearity check at full size: rows split in two halves must add up
    h = rows // 2
    W1 = torch.empty_like(V); W2 = torch.empty_like(V)
    ext.hipZCacheBlockMatvec(zc[:h], V, W1, True, ws)
    ext.hipZCacheBlockMatvec(zc[h:], V, W1, True, ws, accumulate=True)
    ext.hipZCacheBlockMatvec(zc, V, W2, True, ws)
    print("split-sum rel err:", ((W1 - W2).abs().max() / W2.abs().max()).item())
    # against k single-RHS cached matvecs
    ws1 = torch.empty(ext.ztz_workspace_bytes(M, M // 2), dtype=torch.uint8, device=dev)
    w1 = torch.empty(M, dtype=torch.float64, device=dev)
    ext.hipZCacheMatvec(zc, V[:, 3].contiguous(), w1, True, ws1)
    print("vs single-RHS kernel col 3:", ((W2[:, 3] - w1).abs().max() / w1.abs().max()).item())
for _ in range(2):
    ext.hipZCacheBlockMatvec(zc, V, W, True, ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    ext.hipZCacheBlockMatvec(zc, V, W, True, ws)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
kp = 16 if k <= 16 else 32
print(f"rows={rows} M={M} k={k}: {dt*1e3:.3f} ms  useful {4*rows*M*k/dt/1e12:.1f} TFLOP/s  issued {4*rows*M*kp/dt/1e12:.1f} TFLOP/s "
      f"cache read {2*rows*M*4/dt/1e12:.2f} TB/s")
