"""What the launches around the block matvec cost in the 26-column solve (round 4): hipCGStep1Block / hipCGStep2Block, the
two products of the preconditioner apply, queued behind a long kernel (tools/cg_tail_probe.py's method).
    python tools/block_tail_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
dev = "cuda"
m, rank, k = 8192, 512, 26
g = torch.Generator(device=dev).manual_seed(1)
f64 = dict(dtype=torch.float64, device=dev)
u = torch.linalg.qr(torch.randn(m, rank, generator=g, **f64))[0].contiguous()
us = (u * torch.rand(rank, generator=g, **f64)[None, :]).contiguous()
ut = u.T
w, p, x, r, rn, z, zn, pn = (torch.randn(m, k, generator=g, **f64) for _ in range(8))
rz, al, be, nrm = (torch.ones(k, **f64) for _ in range(4))
err = torch.zeros(128, k, dtype=torch.float64).pin_memory()
utr = torch.empty(rank, k, **f64)
big_a = torch.randn(8192, 8192, generator=g, **f64); big_c = torch.empty_like(big_a)
REPS = 100
cws = torch.empty(ext.cg_block_workspace_bytes(m, k), dtype=torch.uint8, device=dev)
def timed(fn):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(big_a, big_a, out=big_c)
    e0.record()
    for i in range(REPS): fn(i)
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / REPS
res = {
 "step1_block": timed(lambda i: ext.hipCGStep1Block(w, p, x, r, rn, z, rz, al, err[i], nrm, 0.01, cws)),
 "step2_block": timed(lambda i: ext.hipCGStep2Block(rn, zn, p, pn, rz, be, cws)),
 "mm_Ut_r": timed(lambda i: torch.mm(ut, rn, out=utr)),
 "addmm_r_plus_Us_t": timed(lambda i: torch.addmm(rn, us, utr, out=zn)),
 "zero_": timed(lambda i: w.zero_()),
}
uws = torch.empty(ext.precond_utr_block_workspace_bytes(m, rank, k), dtype=torch.uint8, device=dev)
res["hipPrecondUtRBlock (2 launches)"] = timed(lambda i: ext.hipPrecondUtRBlock(u, rn, utr, uws))
inv_eig = torch.rand(rank, generator=g, **f64) + 0.5
pws = torch.empty(ext.precond_apply_block_workspace_bytes(m, rank, k), dtype=torch.uint8, device=dev)
res["hipPrecondApplyBlock (3 launches)"] = timed(lambda i: ext.hipPrecondApplyBlock(u, inv_eig, 1.5, rn, zn, pws))
def chain(i):
    ext.hipCGStep1Block(w, p, x, r, rn, z, rz, al, err[i], nrm, 0.01, cws)
    ext.hipPrecondApplyBlock(u, inv_eig, 1.5, rn, zn, pws)
    ext.hipCGStep2Block(rn, zn, p, pn, rz, be, cws)
res["chain"] = timed(chain)
for kk, v in res.items(): print(f"{kk:22s} {v:8.2f} us")
