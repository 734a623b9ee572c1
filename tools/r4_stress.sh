#!/bin/bash
# round 4: long randomised parity sweep (tools/stress_parity.py), ten fresh seeds x 40 cases each
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
: > gpurun_out/r4/stress.log
for seed in ${SEEDS:-101 102 103 104 105 106 107 108 109 110}; do
  echo "=== seed $seed" >> gpurun_out/r4/stress.log
  timeout -k 10 280 python tools/stress_parity.py 40 $seed >> gpurun_out/r4/stress.log 2>&1 || { echo "FAILED seed $seed"; tail -5 gpurun_out/r4/stress.log; exit 1; }
  tail -1 gpurun_out/r4/stress.log
done
