import time, os, numpy as np, scipy.linalg as sl
import torch
print("cpus", os.cpu_count(), "torch threads", torch.get_num_threads())
for n in (512, 1024, 2048):
    rng = np.random.default_rng(0)
    b = rng.standard_normal((4 * n, n)) * np.logspace(0, -3, n)
    a = b.T @ b
    def T(fn, reps=3):
        fn(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        return 1e3 * (time.perf_counter() - t0) / reps
    print(n, "numpy eigh %.1f ms" % T(lambda: np.linalg.eigh(a)), "| scipy evd %.1f" % T(lambda: sl.eigh(a, driver="evd")),
          "| scipy evr %.1f" % T(lambda: sl.eigh(a, driver="evr")), "| eigvalsh %.1f" % T(lambda: np.linalg.eigvalsh(a)))
    at = torch.from_numpy(a)
    for th in (1, 4, 16):
        torch.set_num_threads(th)
        print("   torch cpu eigh threads=%d: %.1f ms" % (th, T(lambda: torch.linalg.eigh(at))))
    ag = at.cuda()
    def g():
        r = torch.linalg.eigh(ag); torch.cuda.synchronize(); return r
    print("   torch gpu eigh: %.1f ms;  D2H+H2D of the matrix: %.2f ms" % (T(g), T(lambda: (ag.cpu().cuda(), torch.cuda.synchronize()))))
