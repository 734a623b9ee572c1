#!/bin/bash
# round 5: the randomised parity sweeps on the round's tree (buffer-resource window loads, transposed-columns layout, SGPR sign
# flips, split block projection, table-driven large-argument reduction: amplitudes up to 1e15 are in the first sweep's draw)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
: > gpurun_out/r5/stress.log
for seed in ${SEEDS:-501 502 503 504 505 506 507 508}; do
  echo "=== seed $seed" >> gpurun_out/r5/stress.log
  timeout -k 10 280 python tools/stress_parity.py 40 $seed >> gpurun_out/r5/stress.log 2>&1 || { echo "FAILED seed $seed"; tail -5 gpurun_out/r5/stress.log; exit 1; }
  tail -1 gpurun_out/r5/stress.log
done
: > gpurun_out/r5/stress2.log
for seed in ${SEEDS2:-601 602 603 604}; do
  echo "=== seed $seed" >> gpurun_out/r5/stress2.log
  timeout -k 10 280 python tools/stress_parity2.py 32 $seed >> gpurun_out/r5/stress2.log 2>&1 || { echo "FAILED sweep 2 seed $seed"; tail -5 gpurun_out/r5/stress2.log; exit 1; }
  tail -1 gpurun_out/r5/stress2.log
done
