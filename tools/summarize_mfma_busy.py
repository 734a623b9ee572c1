#!/usr/bin/env python3
"""profiles/r6_mfma_clock.json from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace`
pass of tools/mfma_busy_probe.py (gpurun_out/prof_mfma_busy).  Per kernel and launch size:
  clock            = GRBM_GUI_ACTIVE / 8 / duration           (the counter is summed over the 8 XCDs)
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
                     = the fraction of SIMD-cycles the matrix pipe was executing (the counter counts cycles per SIMD:
                     checked against the issued instructions -- one v_mfma_f64_16x16x4_f64 = 2048 flop holds the pipe 64 cycles,
                     so issued flop / 32 is the number of busy cycles the instruction stream implies; `busy_over_issued` ~ 1)
  issued_over_peak_at_clock = issued flop / duration / (1024 x 32 flop/cycle x clock)
A value equal to 2^35 or 0xE0000000, or the same integer on every launch of different sizes, is a stopped counter: flagged."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_mfma_busy")
outf = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r6_mfma_clock.json")
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
rows = list(csv.DictReader(open(max(f, key=os.path.getmtime))))
M, R = 8192, 512
# issued flop per launch by (kernel, launch order): the probe's sizes, in order
SIZES = {"sketch_gemm_lds_kernel": [(8192, 2.0 * 8192 * M * R), (4096, 2.0 * 4096 * M * R)],
         "zblock_t_kernel": [(65536, 2.0 * 65536 * M * 28), (32768, 2.0 * 32768 * M * 28)],     # 26 columns run as 16 + 12
         "zblock_w_kernel": [(65536, 2.0 * 65536 * M * 28), (32768, 2.0 * 32768 * M * 28)],
         "gram_lds_kernel": [(1024, 1024.0 * M * (M + 128))]}
per = {}
for r in rows:
    k = next((n for n in SIZES if n in r["Kernel_Name"]), None)
    if k is None:
        continue
    e = per.setdefault((k, int(r["Dispatch_Id"])), {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9,
                                                    "grid": int(r["Grid_Size"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
out = {}
for k, sizes in SIZES.items():
    disp = sorted((did, e) for (kk, did), e in per.items() if kk == k)
    if not disp:
        continue
    reps = len(disp) // len(sizes)
    for si, (nrows, flop) in enumerate(sizes):
        sel = [e for _, e in disp[si * reps:(si + 1) * reps]][1:]            # (first launch of a size: warm-up)
        dur = sum(e["dur"] for e in sel) / len(sel)
        gui = sum(e["GRBM_GUI_ACTIVE"] for e in sel) / len(sel)
        busy = [e["SQ_VALU_MFMA_BUSY_CYCLES"] for e in sel]
        cyc = gui / 8
        ent = {"rows": nrows, "launches": len(sel), "duration_ms": dur * 1e3, "clock_GHz": cyc / dur / 1e9,
               "SQ_VALU_MFMA_BUSY_CYCLES": sum(busy) / len(busy), "SQ_VALU_MFMA_BUSY_CYCLES_min_max": [min(busy), max(busy)],
               "SQ_BUSY_CU_CYCLES": sum(e.get("SQ_BUSY_CU_CYCLES", 0.0) for e in sel) / len(sel), "GRBM_GUI_ACTIVE": gui,
               "mfma_busy_frac_of_simd_cycles": sum(busy) / len(busy) / (1024 * cyc),
               "issued_TFLOPs": flop / dur / 1e12, "issued_over_peak_at_clock": flop / dur / (1024 * 32 * cyc / dur),
               "busy_over_issued_cycles": sum(busy) / len(busy) / (flop / 32.0),
               "stopped_counter": any(b in (2.0 ** 35, float(0xE0000000)) for b in busy)}
        out.setdefault(k, []).append(ent)
        print(k, json.dumps(ent))
out["note"] = ("one rocprofv3 --pmc pass of tools/mfma_busy_probe.py (short launches: the summed per-XCD counters stop on launches of ~2 ms and "
               "more); durations are under the profiler; SQ_BUSY_CU_CYCLES is in units of 4 cycles summed over the CUs")
json.dump(out, open(outf, "w"), indent=1)
