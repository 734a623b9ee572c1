#!/bin/bash
# per-kernel times of the block matvec (zblock_t_kernel / zblock_w_kernel) for the current build with private staging on / off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pv in 1 0; do
  export XGPR_ZB_PRIV=$pv
  rm -rf gpurun_out/zbk_$pv
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/zbk_$pv -- python tools/ab_block.py current > gpurun_out/zbk_$pv.log 2>&1
  echo "XGPR_ZB_PRIV=$pv"; grep -h "zblock" gpurun_out/zbk_$pv/*/*kernel_stats.csv | awk -F, '{print "   ", $1, "calls", $2, "avg ns", $4}'
done
