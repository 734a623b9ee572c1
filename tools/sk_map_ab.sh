#!/bin/bash
# A/B of the contraction's workgroup -> tile map (XGPR_SK_MAP): time and L2 <-> fabric bytes per launch at cfg3's window.
# FETCH_SIZE takes 3 of the 4 TCC slots, so hits / misses are a second pass; every profiled run is under a timeout.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 1; do
  export XGPR_SK_MAP=$m
  timeout -k 10 120 python tools/bench_sketch_gemm.py 2>/dev/null | tail -1
  for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=${pass%% *}
    rm -rf gpurun_out/skmap_${m}_$tag
    timeout -k 10 180 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/skmap_${m}_$tag -- python tools/bench_sketch_gemm.py > gpurun_out/skmap_${m}_$tag.log 2>&1 || echo "pass $tag failed"
  done
  python - "$m" <<'PY'
import csv, glob, sys, collections
m = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/skmap_{m}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sketch_gemm_lds_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  map {m}: {k} mean {sum(v)/len(v):.4g} over {len(v)} launches")
if acc["FETCH_SIZE"]:
    fs = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])
    print(f"  map {m}: fetched {2 * fs * 1024 / 1e9:.2f} GB per launch (x2-corrected) vs algorithmic 4.90 GB")
PY
done
