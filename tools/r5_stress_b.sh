#!/bin/bash
# round 5: twelve more seeds of the first randomised sweep on the final tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
: > gpurun_out/r5/stress_${TAG:-b}.log
for seed in ${SEEDS:-511 512 513 514 515 516 517 518 519 520 521 522}; do
  echo "=== seed $seed" >> gpurun_out/r5/stress_${TAG:-b}.log
  timeout -k 10 280 python tools/stress_parity.py 40 $seed >> gpurun_out/r5/stress_${TAG:-b}.log 2>&1 || { echo "FAILED seed $seed"; tail -5 gpurun_out/r5/stress_${TAG:-b}.log; exit 1; }
  tail -1 gpurun_out/r5/stress_${TAG:-b}.log | cut -c1-220
done
