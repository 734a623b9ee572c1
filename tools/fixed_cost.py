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
    # clock against time inside the launch (workgroup 0: samples every 8 datapoints)
    ns = (iters + 7) // 8
    smp = ws[off + 2 * F * 8: off + 2 * F * 8 + 8 * 512 * 8].view(torch.float64).reshape(8, 256, 2).cpu().numpy()[:, :ns, :]
    curve = []
    for j in range(1, ns):
        dt_us = (smp[0, j, 0] - smp[0, j - 1, 0]) / 100.0
        curve.append({"t_us": round((smp[0, j, 0] - t0[0]) / 100.0, 1), "us_per_datapoint": round(dt_us / 8, 3),
                      "clock_ghz": round(float((smp[0, j, 1] - smp[0, j - 1, 1]) / (dt_us * 1e3)), 3)})
    ent["workgroup0_clock_curve"] = curve[:24] + (curve[24::8] if len(curve) > 24 else [])
    res["rows"][str(n)] = ent
    print(n, json.dumps({k: ent[k] for k in ent if k not in ("per_xcc", "workgroup0_clock_curve")}))
    print("   per XCC:", json.dumps(ent["per_xcc"]))
    print("   clock curve:", " ".join("%g:%.2f/%.2f" % (c["t_us"], c["clock_ghz"], c["us_per_datapoint"]) for c in ent["workgroup0_clock_curve"]))
# does the launch slow down when the device idles in front of it?  One matvec timed by its own pair of events behind a gap
# in which one wave spins (torch.cuda._sleep) and the other 255 CUs idle
n = 125000 if 125000 in rows_list else rows_list[0]
xs = xall[:n]
kern = make_kernel("Matern", (n, d), m, 123, dev, {"matern_nu": 2.5})
kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
v = torch.randn(m, dtype=torch.float64, device=dev, generator=g)
w = torch.empty_like(v)
ws = torch.zeros(kern.workspace_bytes(), dtype=torch.uint8, device=dev)
gaps = {}
for gap_us in (0, 20, 50, 100, 200, 500, 2000):
    for _ in range(5):
        kern.ztz_matvec(xs, v, w, ws)
    ts = []
    for _ in range(20):
        eg = torch.cuda.Event(enable_timing=True); eg.record()
        if gap_us:
            torch.cuda._sleep(int(gap_us * 2100))          # ~cycles at 2.1 GHz (the measured gap is reported)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); kern.ztz_matvec(xs, v, w, ws); e1.record()
        ts.append((eg, e0, e1))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) * 1e3 for _, a, b in ts])
    gm = float(np.median([a.elapsed_time(b) * 1e3 for a, b, _ in ts]))
    gaps[str(gap_us)] = {"measured_gap_us": round(gm, 1), "matvec+reduce_us_p50": round(float(np.median(t)), 1), "min": round(float(t.min()), 1)}
    print("idle gap %5d us requested, %.1f measured: launch %.1f us (min %.1f)" % (gap_us, gm, np.median(t), t.min()))
res["idle_gap_in_front_of_a_125000_row_launch"] = gaps
if out_file:
    with open(out_file, "w") as f:
        json.dump(res, f, indent=1)
