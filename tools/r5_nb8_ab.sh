#!/bin/bash
# round 5: eight tiles per datapoint (16384 RFFs) -- one pass on the two-wave kernel (XGPR_ZTZ_NB8=1, what rounds 1-4 ran) against two
# passes of the three-wave kernel in tile groups of four (the new default)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
for d in 1024 512 256 128; do
  for keep in 1 0; do
    echo "== d=$d XGPR_ZTZ_NB8=$keep"
    XGPR_ZTZ_NB8=$keep python tools/bench_fused.py 131072 $d 16384
  done
done
XGPR_ZTZ_NB8=0 python tools/bench_fused.py 131072 1024 15000
XGPR_ZTZ_NB8=1 python tools/bench_fused.py 131072 1024 15000
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/nb8_two_pass_ab.log
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cg.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r5/gputests_nb8.log 2>&1; tail -3 gpurun_out/r5/gputests_nb8.log
