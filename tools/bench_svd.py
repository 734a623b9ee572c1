"""Timing of the small factorizations of the preconditioner build on the device (rocSOLVER through torch)."""
import time, torch
dev = "cuda"
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device=dev).manual_seed(0)
for (m, r) in [(8192, 512), (32768, 2048)]:
    a = torch.randn(m, r, dtype=torch.float64, device=dev, generator=g)
    c = torch.randn(r, r, dtype=torch.float64, device=dev, generator=g)
    print(f"M={m} rank={r}: svd[r,r] {t(lambda: torch.linalg.svd(c, full_matrices=False)):.1f} ms | "
          f"svd[M,r] {t(lambda: torch.linalg.svd(a, full_matrices=False)):.1f} ms | qr[M,r] {t(lambda: torch.linalg.qr(a)):.1f} ms | "
          f"eigh[r,r] {t(lambda: torch.linalg.eigh(c @ c.T)):.1f} ms | chol[r,r] {t(lambda: torch.linalg.cholesky(c @ c.T + r * torch.eye(r, dtype=torch.float64, device=dev))):.1f} ms")
    ac = a.cpu()
    t0 = time.perf_counter(); torch.linalg.svd(ac, full_matrices=False); print(f"   cpu svd[M,r] {(time.perf_counter()-t0)*1e3:.0f} ms")
    if r <= 512:
        for drv in ("gesvd", "gesvdj", "gesvda"):
            try:
                print(f"   driver {drv}: svd[M,r] {t(lambda: torch.linalg.svd(a, full_matrices=False, driver=drv)):.1f} ms")
            except Exception as e:
                print("   driver", drv, "unavailable:", str(e)[:80])
