"""Same-process, interleaved A/B of the fused CG matvec across builds of the library (development aid):
    python tools/ab_inproc.py "rows d M" libA.so libB.so ...        ("current" = xgpr_amd/libxgpr_hip.so)
Every library is loaded with ctypes into ONE process and called through the C ABI in alternation (ROUNDS rounds of
CALLS launches each, HIP events on the current stream); prints min / median ms per launch and a checksum per library.
Separate processes on this pool differ by 2-3 % for the same binary (clock state), interleaving removes that."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import _lib

ROUNDS, CALLS = 7, 5
ZTY = "--zty" in sys.argv            # time xgpr_zty_f32 (z^T y) instead of the matvec
if ZTY: sys.argv.remove("--zty")
n, d, m = (int(t) for t in sys.argv[1].split())
paths = [(_lib.LIB_PATH if p == "current" else p if os.path.exists(p) else f"tools/ablate/lib_{p}.so") for p in sys.argv[2:]]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
xs = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
kern = make_kernel("Matern", (n, d), m, 123, dev, {"matern_nu": 2.5})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
v = torch.randn(n if ZTY else m, dtype=torch.float64, device=dev, generator=g)
ws = torch.empty(kern.workspace_bytes(), dtype=torch.uint8, device=dev)
radem, chi = kern.radem_diag, kern.chi_arr
vp, l, i, sz = C.c_void_p, C.c_long, C.c_int, C.c_size_t
libs = []
for p in paths:
    lib = C.CDLL(p)
    fn = lib.xgpr_zty_f32 if ZTY else lib.xgpr_ztz_matvec_f32
    fn.argtypes = [vp, vp, vp, vp, vp, l, l, l, l, l, i, vp, sz, vp]; fn.restype = C.c_int
    libs.append((p, fn, torch.empty(m, dtype=torch.float64, device=dev)))
def call(fn, w):
    rc = fn(xs.data_ptr(), radem.data_ptr(), chi.data_ptr(), v.data_ptr(), w.data_ptr(), n, d, m, kern.num_freqs,
            radem.shape[2], int(kern.fit_intercept), ws.data_ptr(), ws.numel(), 0)
    assert rc == 0, rc
stream0 = torch.cuda.current_stream()
assert stream0.cuda_stream == 0, "library calls go to the null stream here"
for _, fn, w in libs:
    for _ in range(3): call(fn, w)
times = {p: [] for p, _, _ in libs}
for r in range(ROUNDS):
    for p, fn, w in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(CALLS): call(fn, w)
        e1.record(); e1.synchronize()
        times[p].append(e0.elapsed_time(e1) / CALLS)
for p, fn, w in libs:
    t = times[p]
    print(f"{os.path.basename(p):28s} min {min(t):.3f}  median {statistics.median(t):.3f} ms   checksum {float(w.sum()):.12e}")
