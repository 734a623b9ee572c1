#!/usr/bin/env python3
"""Clock and matrix-core rate of the MFMA kernels from a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace` pass (tools/collect_profiles.sh step 6/7) -> profiles/r2_mfma_clock.json.
GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md): clock = counter / 8 / kernel duration.  The peak at
that clock is 1024 SIMDs x 32 flop/cycle (one v_mfma_f64_16x16x4_f64 = 2048 flop per 64 cycles)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
out = {}
# issued flop per launch: the contraction 2 * rows * M * rank; a block kernel 2 * rows * M * (16 + 12) columns at k = 26
SPEC = {"sketch_gemm_lds_kernel": ("prof_mfma_gemm", 2.0 * 131072 * 8192 * 512, 2.0 * 131072 * 8192 * 512),
        "zblock_t_kernel": ("prof_mfma_block", 2.0 * 262144 * 8192 * 28, 2.0 * 262144 * 8192 * 26),
        "zblock_w_kernel": ("prof_mfma_block", 2.0 * 262144 * 8192 * 28, 2.0 * 262144 * 8192 * 26)}
for kname, (d, issued, useful) in SPEC.items():
    f = glob.glob(os.path.join(G, d, "*counter_collection.csv"))
    if not f:
        print("missing", d)
        continue
    rows = [r for r in csv.DictReader(open(max(f, key=os.path.getmtime))) if kname in r["Kernel_Name"]]
    per = {}
    for r in rows:
        per.setdefault(r["Dispatch_Id"], {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})[r["Counter_Name"]] = float(r["Counter_Value"])
    sel = [v for v in per.values() if v["dur"] > 0.5 * max(x["dur"] for x in per.values())]
    dur = sum(v["dur"] for v in sel) / len(sel) * 1e-9
    gui = sum(v["GRBM_GUI_ACTIVE"] for v in sel) / len(sel)
    clock = gui / 8 / dur
    peak = 1024 * 32 * clock / 1e12
    out[kname] = {"launches": len(sel), "duration_ms": dur * 1e3, "clock_GHz": clock / 1e9, "peak_at_clock_TFLOPs": peak,
                  "issued_TFLOPs": issued / dur / 1e12, "useful_TFLOPs": useful / dur / 1e12,
                  "issued_over_peak_at_clock": issued / dur / 1e12 / peak, "useful_over_nominal_78.6": useful / dur / 1e12 / 78.6,
                  "SQ_VALU_MFMA_BUSY_CYCLES": sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for v in sel) / len(sel),
                  "SQ_BUSY_CU_CYCLES": sum(v.get("SQ_BUSY_CU_CYCLES", 0) for v in sel) / len(sel)}
    print(kname, json.dumps(out[kname]))
out["note"] = ("durations are under the profiler (a few % longer than the un-profiled launches); SQ_VALU_MFMA_BUSY_CYCLES "
               "saturates at 2^35 on launches this long and is recorded only for completeness")
json.dump(out, open(os.path.join(ROOT, "profiles", os.environ.get("XGPR_ROUND", "r3") + "_mfma_clock.json"), "w"), indent=1)
