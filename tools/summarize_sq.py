#!/usr/bin/env python3
"""profiles/r1_fused_pmc_sq.json from the two SQ passes of tools/collect_profiles.sh (gpurun_out/prof_sq1, prof_sq2:
rocprofv3 --pmc ... -- python tools/pmc_probe.py, the fused CG matvec at cfg3 shape on 262144 rows).  gfx950: SQ_*
cycle counters are in units of 4 cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
ROWS, TILES_PER_ROW, SIMDS, CUS = 262144, 4, 1024, 256
# second argument: substring of the kernel name to summarise (default: the fused matvec); third: wave tiles of the launch
KSEL = sys.argv[2] if len(sys.argv) > 2 else "ztz3_kernel"
NTILES = float(sys.argv[3]) if len(sys.argv) > 3 else ROWS * TILES_PER_ROW
PROBE = sys.argv[4] if len(sys.argv) > 4 else "prof"
out = {}
for tag in ("sq1", "sq2"):
    f = glob.glob(os.path.join(G, f"{PROBE}_{tag}/*/*counter_collection.csv"))
    if not f:
        sys.exit(f"missing prof_{tag}")
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):      # latest merge wins
        if KSEL not in r["Kernel_Name"]:
            continue
        out["kernel_name"] = r["Kernel_Name"][:80]
        out[r["Counter_Name"] if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or tag == "sq1" else "GRBM_GUI_ACTIVE_pass2"] = float(r["Counter_Value"])
        out[f"duration_us_{tag}"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out["vgpr_count"] = int(r["VGPR_Count"])
dur = out["duration_us_sq1"] * 1e-6
clock = out["GRBM_GUI_ACTIVE"] / 8 / dur
cyc1 = clock * dur
cyc2 = out["GRBM_GUI_ACTIVE_pass2"] / 8
out["derived"] = {
    "clock_GHz": round(clock / 1e9, 3),
    "valu_insts_per_tile": out["SQ_INSTS_VALU"] / NTILES,
    "waves_resident_per_simd": round(out["SQ_WAVE_CYCLES"] * 4 / (SIMDS * cyc1), 2),
    "valu_active_frac_of_simd_cycles": round(out["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * cyc1), 3),
    "lds_active_frac": round(out["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc2), 3),
    "lds_bank_conflict_frac_of_lds_active": round(out["SQ_LDS_BANK_CONFLICT"] / out["SQ_LDS_IDX_ACTIVE"], 3),
    "wait_any_frac_per_wave": round(out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"], 3),
    "wait_inst_any_frac_per_wave": round(out["SQ_WAIT_INST_ANY"] / out["SQ_WAVE_CYCLES"], 3),
}
out["_note"] = ("tools/collect_profiles.sh + tools/summarize_sq.py: fused CG matvec at cfg3 shape, 262144 rows; two rocprofv3 "
                "--pmc passes of tools/pmc_probe.py; SQ_* cycle counters are in units of 4 cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs")
OUT = sys.argv[1] if len(sys.argv) > 1 else "r3_fused_pmc_sq.json"      # file name under profiles/
json.dump({"wave_ztz_kernel" if KSEL == "ztz3_kernel" else KSEL: out}, open(os.path.join(P, OUT), "w"), indent=1)
print(json.dumps(out["derived"], indent=1), out["duration_us_sq1"])
