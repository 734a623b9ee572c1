"""Same-process A/B of the 26-right-hand-side block matvec with / without per-wave private staging (XGPR_ZB_PRIV is read
once per process, so the two forms come from two builds or two env settings: here two LIBRARIES):
    python tools/ab_block.py libA.so libB.so ...      ("current" = xgpr_amd/libxgpr_hip.so)"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xgpr_amd import _lib
n, m, k = 262144, 8192, int(os.environ.get("K", "26"))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
zc = torch.rand(n, m, device=dev, generator=g) * 2 - 1
v = torch.randn(m, k, dtype=torch.float64, device=dev, generator=g)
ws = torch.empty(int(_lib.load().xgpr_zcache_block_workspace_bytes(n, m, k)), dtype=torch.uint8, device=dev)
vp, l, i, d, sz = C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_size_t
libs = []
for p in sys.argv[1:]:
    path = _lib.LIB_PATH if p == "current" else p if os.path.exists(p) else f"tools/ablate/lib_{p}.so"
    fn = C.CDLL(path).xgpr_zcache_block_matvec_f32
    fn.argtypes = [vp, vp, vp, l, l, l, i, d, i, vp, sz, vp]; fn.restype = C.c_int
    libs.append((p, fn, torch.empty_like(v)))
def call(fn, w):
    assert fn(zc.data_ptr(), v.data_ptr(), w.data_ptr(), n, m, k, 1, 0.0, 0, ws.data_ptr(), ws.numel(), 0) == 0
for _, fn, w in libs:
    for _ in range(2): call(fn, w)
times = {p: [] for p, _, _ in libs}
for r in range(7):
    for p, fn, w in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): call(fn, w)
        e1.record(); e1.synchronize()
        times[p].append(e0.elapsed_time(e1) / 3)
ref = libs[0][2]
for p, fn, w in libs:
    t = statistics.median(times[p])
    print(f"{p:12s} median {t:.3f} ms  useful {4.0*n*m*k/t/1e9:.1f} TFLOP/s = {4.0*n*m*k/t/1e9/78.6:.3f} of peak   max rel diff vs first {float((w-ref).abs().max()/ref.abs().max()):.2e}")
