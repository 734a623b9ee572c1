#!/bin/bash
# padded widths 16 .. 64 (d = 9 .. 64: most tabular data) on the three-wave kernel against the two-wave kernel's register-only transform
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cg.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r5/gputests_smallp.log 2>&1; rc=$?; tail -3 gpurun_out/r5/gputests_smallp.log
[ $rc -eq 0 ] || { grep -E "Error|assert" gpurun_out/r5/gputests_smallp.log | head -20; exit 1; }
{
for shape in "64 8192" "64 4096" "32 8192" "32 4096" "16 4096" "20 8192" "50 4096" "33 6000" "64 10240"; do
  set -- $shape
  for waves in 3 2; do
    echo "== d=$1 M=$2 XGPR_ZTZ_WAVES=$waves"
    XGPR_ZTZ_WAVES=$waves python tools/bench_fused.py 262144 $1 $2
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/smallp_ab.log
