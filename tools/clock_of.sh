#!/bin/bash
# clock under load (GRBM_GUI_ACTIVE / 8 XCDs / duration) and VALU instruction count of the fused matvec for the
# current build and for each named ablation library:   tools/clock_of.sh "rows d M" name1 name2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
shape=$1; shift
for n in current "$@"; do
  rm -rf gpurun_out/clk_$n
  if [ $n = current ]; then unset XGPR_HIP_LIB; else export XGPR_HIP_LIB=tools/ablate/lib_$n.so; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/clk_$n -- python tools/bench_fused.py $shape > gpurun_out/clk_$n.log 2>&1
  python - "$n" <<'PY'
import csv, glob, sys, collections
n = sys.argv[1]
f = glob.glob(f"gpurun_out/clk_{n}/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "ztz3_" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc["dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
k = len(acc["GRBM_GUI_ACTIVE"])
dur = sum(acc["dur"]) / len(acc["dur"])
gui = sum(acc["GRBM_GUI_ACTIVE"]) / k
print(f"{n:12s} launches {k:3d}  {dur:9.1f} us  clock {gui / 8 / dur / 1e3:.3f} GHz  VALU insts {sum(acc['SQ_INSTS_VALU']) / k:.4g}  wave-cycles(x4) {sum(acc['SQ_WAVE_CYCLES']) / k:.4g}")
PY
done
