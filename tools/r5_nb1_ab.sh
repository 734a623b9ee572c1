#!/bin/bash
# one tile per datapoint (M <= 2048): the three-wave kernel with twelve (ten at P = 1024) one-wave slots (XGPR_ZTZ3_ONE_TILE=1) against the
# two-wave kernel it has been left to since round 2
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
for shape in "1024 2048" "512 2048" "256 2048" "128 2048" "256 1024" "1000 2000"; do
  set -- $shape
  for one in 1 0; do
    echo "== d=$1 M=$2 XGPR_ZTZ3_ONE_TILE=$one"
    XGPR_ZTZ3_ONE_TILE=$one python tools/bench_fused.py 262144 $1 $2
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/nb1_ab.log
