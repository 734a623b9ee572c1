#!/bin/bash
# five and seven tiles per datapoint: one pass on the two-wave kernel (default) against two passes of the three-wave kernel
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
for d in 1024 512 256; do
  for m in 10240 14336; do
    for above in 7168 4096; do
      echo "== d=$d M=$m two passes above $above frequencies"
      XGPR_ZTZ_TWO_PASS_ABOVE=$above python tools/bench_fused.py 131072 $d $m
    done
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/nb57_two_pass_ab.log
