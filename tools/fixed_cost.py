"""Where the fused matvec's fixed cost per launch goes (the ~70-96 us a shard pays whatever its size).

Needs a development build of the library with -DXGPR_ZTZ_TIMING (per-workgroup stamps on the 100 MHz wall clock at kernel
entry / end of the prologue / end of the datapoint loop / after the slab stores landed, core-clock cycles, XCC id):
    tools/ablate_build.sh timing "-DXGPR_ZTZ_TIMING"
    XGPR_HIP_LIB=tools/ablate/lib_timing.so python tools/fixed_cost.py [out.json] [rows ...]
Per row count: the launch's duration by HIP events (same stream, 20 launches back to back, mean) and the phase table of
the LAST launch.  `outside` = event time of one launch - (last workgroup's end - first workgroup's begin): dispatch + drain
+ the slab reduction kernel when it is part of the timed launch pair."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import _lib

args = sys.argv[1:]
out_file = args[0] if args and args[0].endswith(".json") else None
rows_list = [int(a) for a in args if a.isdigit()] or [31250, 62500, 125000, 250000, 500000, 1000000]
d, m = 1024, 8192
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
nmax = max(rows_list)
xall = torch.randn(nmax, d, device=dev, generator=g) / d ** 0.5
res = {"shape": {"d": d, "num_rffs": m}, "wall_clock_hz": 1e8, "rows": {}}
for n in rows_list:
    xs = xall[:n]
    kern = make_kernel("Matern", (n, d), m, 123, dev, {"matern_nu": 2.5})
    kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
    v = torch.randn(m, dtype=torch.float64, device=dev, generator=g)
    w = torch.empty_like(v)
    ws = torch.zeros(kern.workspace_bytes(), dtype=torch.uint8, device=dev)
    for _ in range(10):
        kern.ztz_matvec(xs, v, w, ws)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        kern.ztz_matvec(xs, v, w, ws)
    e1.record(); torch.cuda.synchronize()
    ev_us = e0.elapsed_time(e1) * 1e3 / reps
    R = kern.radem_diag.shape[2]
    mb = int(_lib.load().xgpr_rbf_workspace_bytes(R))
    F = m // 2
    off = mb + 1000 * 2 * F * 8
    dbg = ws[off:off + 256 * 8 * 8].view(torch.float64).reshape(256, 8).cpu().numpy()
    t0, tp, tl, t1, cyc, smid, xcc, lcyc = (dbg[:, i] for i in range(8))
    us = lambda ticks: ticks / 100.0
    first, last = t0.min(), t1.max()
    iters = -(-n // 768)
    ent = {
        "event_us_per_launch(matvec+reduce)": round(ev_us, 2),
        "iterations_per_slot": iters,
        "span_us(first begin..last end)": round(us(last - first), 2),
        "outside_us": round(ev_us - us(last - first), 2),
        "start_skew_us": {"p50": round(us(np.percentile(t0 - first, 50)), 2), "max": round(us((t0 - first).max()), 2)},
        "prologue_us": {"min": round(us((tp - t0).min()), 2), "p50": round(us(np.median(tp - t0)), 2), "max": round(us((tp - t0).max()), 2)},
        "loop_us": {"min": round(us((tl - tp).min()), 2), "p50": round(us(np.median(tl - tp)), 2), "max": round(us((tl - tp).max()), 2)},
        "loop_us_per_iteration_p50": round(us(np.median(tl - tp)) / iters, 4),
        "epilogue_us(slot sum + slab store landed)": {"min": round(us((t1 - tl).min()), 2), "p50": round(us(np.median(t1 - tl)), 2), "max": round(us((t1 - tl).max()), 2)},
        "end_spread_us": {"p10": round(us(np.percentile(last - t1, 90)), 2), "p50": round(us(np.percentile(last - t1, 50)), 2), "max": round(us((last - t1).max()), 2)},
        "loop_clock_ghz": {"min": round(float((lcyc / ((tl - tp) * 10)).min()), 3), "p50": round(float(np.median(lcyc / ((tl - tp) * 10))), 3), "max": round(float((lcyc / ((tl - tp) * 10)).max()), 3)},
        "per_xcc": {},
    }
    for x in sorted(set(xcc.astype(int))):
        sel = xcc.astype(int) == x
        ent["per_xcc"][str(x)] = {"workgroups": int(sel.sum()), "begin_us": round(us((t0[sel] - first).mean()), 2),
                                  "loop_us_mean": round(us((tl - tp)[sel].mean()), 2), "loop_us_max": round(us((tl - tp)[sel].max()), 2),
                                  "end_us_max": round(us((t1[sel] - first).max()), 2),
                                  "clock_ghz": round(float((lcyc[sel] / ((tl - tp)[sel] * 10)).mean()), 3)}
    res["rows"][str(n)] = ent
    print(n, json.dumps({k: ent[k] for k in ent if k != "per_xcc"}))
    print("   per XCC:", json.dumps(ent["per_xcc"]))
if out_file:
    with open(out_file, "w") as f:
        json.dump(res, f, indent=1)
