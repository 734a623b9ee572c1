import time, torch
dev = "cuda"
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n in (512, 2048, 3000):
    g = torch.Generator(device=dev).manual_seed(0)
    b = torch.randn(8192, n, dtype=torch.float64, device=dev, generator=g) * torch.logspace(0, -3, n, dtype=torch.float64, device=dev)
    a = b.T @ b
    for lib in ("default", "cusolver", "magma"):
        try:
            torch.backends.cuda.preferred_linalg_library(lib)
            ms = t(lambda: torch.linalg.eigh(a))
            ev, evec = torch.linalg.eigh(a)
            res = float((a @ evec - evec * ev[None, :]).abs().max() / ev.abs().max())
            print(f"n={n} {lib}: eigh {ms:.1f} ms residual {res:.1e};  eigvalsh {t(lambda: torch.linalg.eigvalsh(a)):.1f} ms")
        except Exception as e:
            print(f"n={n} {lib}: failed {type(e).__name__} {str(e)[:100]}")
    ac = a.cpu()
    t0 = time.perf_counter(); torch.linalg.eigh(ac); print(f"n={n} cpu eigh {1e3*(time.perf_counter()-t0):.1f} ms")
