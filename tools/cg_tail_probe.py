"""What the small launches behind the fused matvec cost (round 4, item 4): cg_step1, the three launches of the
preconditioner apply, cg_step2 at M = 8192, rank = 512 -- each alone, back to back on one stream (its duration + the
dispatch of the next one), and as the chain an iteration runs.
    python tools/cg_tail_probe.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
dev = "cuda"
m, rank = 8192, 512
g = torch.Generator(device=dev).manual_seed(1)
f64 = dict(dtype=torch.float64, device=dev)
u = torch.linalg.qr(torch.randn(m, rank, generator=g, **f64))[0].contiguous()
inv_eig = torch.rand(rank, generator=g, **f64) + 0.5
w, p, x, r, rn, z, zn, pn = (torch.randn(m, generator=g, **f64) for _ in range(8))
scal = torch.zeros(8, **f64)
pws = torch.empty(ext.precond_workspace_bytes(rank), dtype=torch.uint8, device=dev)
err_host = torch.zeros(4096, dtype=torch.float64).pin_memory()
err_dev = torch.zeros(1, **f64)
REPS = 100
big_a = torch.randn(8192, 8192, generator=g, **f64)
big_c = torch.empty_like(big_a)

def timed(fn):
    """REPS calls queued BEHIND a ~15 ms kernel (a float64 matrix product): the host has enqueued all of them before
    the device gets to the first one, so the time is what the device needs -- durations + dependent dispatches -- as
    inside a CG iteration, where the host runs ahead of the ~0.6 ms matvec."""
    for i in range(10): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(big_a, big_a, out=big_c)
    e0.record()
    for i in range(REPS): fn(i)
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / REPS

out = {}
out["step1_no_err_out"] = timed(lambda i: ext.hipCGStep1(w, p, x, r, rn, z, scal, 0.01, 1.0))
out["step1_err_out_device"] = timed(lambda i: ext.hipCGStep1(w, p, x, r, rn, z, scal, 0.01, 1.0, 0.0, err_dev))
out["step1_err_out_pinned_host"] = timed(lambda i: ext.hipCGStep1(w, p, x, r, rn, z, scal, 0.01, 1.0, 0.0, err_host[i:i + 1]))
out["precond_apply_3_launches"] = timed(lambda i: ext.hipPrecondApply(u, inv_eig, 1.5, rn, zn, pws))
out["step2"] = timed(lambda i: ext.hipCGStep2(rn, zn, p, pn, scal))
def chain(i):
    ext.hipCGStep1(w, p, x, r, rn, z, scal, 0.01, 1.0, 0.0, err_host[i:i + 1])
    ext.hipPrecondApply(u, inv_eig, 1.5, rn, zn, pws)
    ext.hipCGStep2(rn, zn, p, pn, scal)
out["chain_step1_precond_step2"] = timed(chain)
small, small2 = torch.zeros(64, **f64), torch.zeros(64, **f64)
out["step2_M64_dispatch_floor"] = timed(lambda i: ext.hipCGStep2(small, small, small, small2, scal))
for k2, v in out.items(): print(f"{k2:34s} {v:7.2f} us")
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump({"what": __doc__.split(chr(10))[0], "M": m, "rank": rank, "unit": "us per call, %d calls queued behind a 15 ms kernel" % REPS, "results": out}, open(sys.argv[1], "w"), indent=1)
