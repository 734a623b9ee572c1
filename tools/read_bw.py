import torch, time
x = torch.empty(8 * 1024**3 // 4, dtype=torch.float32, device="cuda").normal_()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for name, fn in (("sum f32", lambda: x.sum()), ("max f32", lambda: x.max()), ("sum as f64 view", lambda: x.view(torch.float64).sum()), ("copy (read+write)", lambda: x[: x.numel() // 2].copy_(x[x.numel() // 2:]))):
    dt = t(fn)
    nbytes = x.numel() * 4 if "copy" not in name else x.numel() * 4
    print(f"{name}: {dt*1e3:.3f} ms  {nbytes/dt/1e12:.2f} TB/s")
