#!/bin/bash
# five tiles per datapoint (e.g. 10 000 RFFs) on the three-wave kernel (two slots on ten waves, two spare) against the two-wave kernel
# (XGPR_ZTZ_WAVES=2) and the cache stream; tests of the fused matvec
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
for d in 1024 512 256; do
  for m in 10240 10000 8194; do
    for waves in 3 2; do
      echo "== d=$d M=$m XGPR_ZTZ_WAVES=$waves"
      XGPR_ZTZ_WAVES=$waves python tools/bench_fused.py 131072 $d $m
    done
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/nb5_ab.log
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r5/gputests_nb5.log 2>&1; tail -3 gpurun_out/r5/gputests_nb5.log
