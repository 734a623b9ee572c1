#!/bin/bash
# round 4: randomised parity sweep of the remaining operators (tools/stress_parity2.py)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
: > gpurun_out/r4/stress2.log
for seed in ${SEEDS:-1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16}; do
  echo "=== seed $seed" >> gpurun_out/r4/stress2.log
  timeout -k 10 280 python tools/stress_parity2.py 16 $seed >> gpurun_out/r4/stress2.log 2>&1 || { echo "FAILED seed $seed"; tail -5 gpurun_out/r4/stress2.log; exit 1; }
  tail -1 gpurun_out/r4/stress2.log | cut -c1-200
done
