#!/bin/bash
# padded widths 2 .. 16 on the three-wave kernel (tile stays in the rows layout: no exchange) against the two-wave kernel
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cg.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r5/gputests_tinyp.log 2>&1; rc=$?; tail -3 gpurun_out/r5/gputests_tinyp.log
[ $rc -eq 0 ] || { grep -E "Error|assert" gpurun_out/r5/gputests_tinyp.log | head -20; exit 1; }
{
for shape in "16 8192" "16 4096" "8 8192" "8 4096" "5 4096" "3 4096" "2 4096" "12 8192" "32 8192"; do
  set -- $shape
  for waves in 3 2; do
    echo "== d=$1 M=$2 XGPR_ZTZ_WAVES=$waves"
    XGPR_ZTZ_WAVES=$waves python tools/bench_fused.py 262144 $1 $2
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/tinyp_ab.log
