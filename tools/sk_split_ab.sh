#!/bin/bash
# contraction at cfg3's window for several numbers of contraction ranges (XGPR_SK_SPLIT): time and L2 <-> fabric bytes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for sp in 2 4 8 16 32; do
  export XGPR_SK_SPLIT=$sp
  echo "split $sp: $(timeout -k 10 120 python tools/bench_sketch_gemm.py 2>/dev/null | tail -1)"
  rm -rf gpurun_out/sksplit_$sp
  timeout -k 10 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/sksplit_$sp -- python tools/bench_sketch_gemm.py > gpurun_out/sksplit_$sp.log 2>&1 || echo "pass failed"
  python - "$sp" <<'PY'
import csv, glob, sys
sp = sys.argv[1]
v = [float(r["Counter_Value"]) for f in glob.glob(f"gpurun_out/sksplit_{sp}/*/*counter_collection.csv") for r in csv.DictReader(open(f))
     if "sketch_gemm_lds_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print(f"  split {sp}: fetched {2 * sum(v) / len(v) * 1024 / 1e9:.2f} GB per launch vs algorithmic 4.90 GB")
PY
done
