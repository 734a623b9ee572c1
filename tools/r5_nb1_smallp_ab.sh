#!/bin/bash
# one tile per datapoint (M <= 2048, most small problems) at padded widths below 128: three-wave kernel (XGPR_ZTZ3_ONE_TILE=1) against the
# two-wave kernel's register-only transform
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
for shape in "64 2048" "32 2048" "32 512" "16 2048" "8 1024" "20 2000" "64 1024" "100 2048"; do
  set -- $shape
  for one in 1 0; do
    echo "== d=$1 M=$2 XGPR_ZTZ3_ONE_TILE=$one"
    XGPR_ZTZ3_ONE_TILE=$one python tools/bench_fused.py 262144 $1 $2
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/nb1_smallp_ab.log
