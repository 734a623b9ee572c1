#!/usr/bin/env python3
"""Development probe: torch.linalg.eigh of a rank x rank Gram matrix eager vs replayed from a HIP graph."""
import sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
b = torch.randn(8192, n, dtype=torch.float64, device=dev, generator=g) * torch.logspace(0, -3, n, dtype=torch.float64, device=dev)
a = b.T @ b


def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"eager eigh {n}: {t(lambda: torch.linalg.eigh(a)):.2f} ms")
print(f"cholesky_ex + trsm [8192 rhs]: {t(lambda: torch.linalg.solve_triangular(torch.linalg.cholesky_ex(a + 1e-3 * torch.eye(n, dtype=torch.float64, device=dev))[0], b.T, upper=False)):.2f} ms")
try:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.linalg.eigh(a)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        ev, evec = torch.linalg.eigh(a)
    torch.cuda.synchronize()
    print(f"graph replay eigh {n}: {t(gr.replay):.2f} ms")
    ev0, evec0 = torch.linalg.eigh(a)
    print("eigenvalue diff", float((ev - ev0).abs().max()), "ok")
except Exception as e:      # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:300])
