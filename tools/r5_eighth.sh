#!/bin/bash
# round 5: what the split block projection buys end to end -- the approximate NMLL on a 32 768-row shard, the classifier's cost
# function on 32 768 rows, predict on 2000-row chunks; each with the unsplit kernel beside it (XGPR_ZB_SPLIT=1)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
for split in "" 1; do
  echo "== XGPR_ZB_SPLIT=${split:-default}"
  XGPR_ZB_SPLIT=$split python tools/bench_nmll_e2e.py 32768
  XGPR_ZB_SPLIT=$split python tools/classifier_probe.py 32768 10
  XGPR_ZB_SPLIT=$split python tools/predict_probe.py
done
} > gpurun_out/r5/split_e2e.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/split_e2e.log
