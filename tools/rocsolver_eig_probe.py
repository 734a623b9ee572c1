"""rocSOLVER's symmetric eigensolvers called directly (ctypes) next to torch.linalg.eigh: syevd (what torch calls), syevdj
(divide and conquer with Jacobi base cases), syevj (Jacobi), at the preconditioner's sizes.   python tools/rocsolver_eig_probe.py"""
import ctypes as C, os, time
import torch
lib_dir = os.path.join(os.path.dirname(torch.__file__), "lib")
rs = C.CDLL(os.path.join(lib_dir, "librocsolver.so")) if os.path.exists(os.path.join(lib_dir, "librocsolver.so")) else C.CDLL("librocsolver.so")
rb = C.CDLL(os.path.join(lib_dir, "librocblas.so")) if os.path.exists(os.path.join(lib_dir, "librocblas.so")) else C.CDLL("librocblas.so")
handle = C.c_void_p()
assert rb.rocblas_create_handle(C.byref(handle)) == 0
EV, NONE, LOWER, ASC = 211, 213, 122, 252
vp, i, d = C.c_void_p, C.c_int, C.c_double
rs.rocsolver_dsyevd.argtypes = [vp, i, i, i, vp, i, vp, vp, vp]
rs.rocsolver_dsyevdj.argtypes = [vp, i, i, i, vp, i, vp, vp]
rs.rocsolver_dsyevj.argtypes = [vp, i, i, i, i, vp, i, d, vp, i, vp, vp, vp]
dev = "cuda"
def T(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps
for n in (512, 1024, 2048):
    g = torch.Generator(device=dev).manual_seed(0)
    b = torch.randn(4 * n, n, dtype=torch.float64, device=dev, generator=g) * torch.logspace(0, -3, n, dtype=torch.float64, device=dev)
    a0 = b.T @ b
    ev_ref, _ = torch.linalg.eigh(a0)
    w = torch.empty(n, dtype=torch.float64, device=dev); e = torch.empty(n, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev); resid = torch.zeros(1, dtype=torch.float64, device=dev); nsw = torch.zeros(1, dtype=torch.int32, device=dev)
    res = {"torch eigh": T(lambda: torch.linalg.eigh(a0))}
    def run(fn):
        a = a0.clone()
        rc = fn(a)
        return a
    def syevd(a): return rs.rocsolver_dsyevd(handle, EV, LOWER, n, a.data_ptr(), n, w.data_ptr(), e.data_ptr(), info.data_ptr())
    def syevdj(a): return rs.rocsolver_dsyevdj(handle, EV, LOWER, n, a.data_ptr(), n, w.data_ptr(), info.data_ptr())
    def syevj(a): return rs.rocsolver_dsyevj(handle, ASC, EV, LOWER, n, a.data_ptr(), n, 1e-15, resid.data_ptr(), 30, nsw.data_ptr(), w.data_ptr(), info.data_ptr())
    for name, fn in (("syevd", syevd), ("syevdj", syevdj), ("syevj", syevj)):
        try:
            ms = T(lambda: run(fn))
            vec = run(fn); torch.cuda.synchronize()
            err = float((w - ev_ref).abs().max() / ev_ref.abs().max())
            # column-major eigenvectors: a (row-major view) holds V^T
            v = vec.T
            rr = float((a0 @ v - v * w[None, :]).abs().max() / ev_ref.abs().max())
            res[name] = f"{ms:.1f} ms (eigenvalue err {err:.1e}, residual {rr:.1e}, info {int(info.item())})"
        except Exception as ex:
            res[name] = "failed: " + str(ex)[:80]
    print(n, res, flush=True)
