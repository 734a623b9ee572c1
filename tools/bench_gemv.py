import time, torch
dev="cuda"
n,m=16384,8192
z=torch.randn(n,m,dtype=torch.float64,device=dev); y=torch.randn(n,dtype=torch.float64,device=dev)
def t(fn,reps=10):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
print("z.T @ y      ", t(lambda: z.T @ y))
print("y @ z        ", t(lambda: y @ z))
print("mv(z.T, y)   ", t(lambda: torch.mv(z.T, y)))
print("(y[None]@z)  ", t(lambda: (y[None,:] @ z)))
print("(z*y[:,None]).sum(0)", t(lambda: (z*y[:,None]).sum(0)))
y2=torch.randn(n,2,dtype=torch.float64,device=dev)
print("z.T @ y2 (2 cols)", t(lambda: z.T @ y2))
y8=torch.randn(n,8,dtype=torch.float64,device=dev)
print("z.T @ y8 (8 cols)", t(lambda: z.T @ y8))
