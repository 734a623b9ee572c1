#!/bin/bash
# round 5: eight more seeds of the second randomised sweep (FHT / SRHT bit for bit, gradients, max-pool, sketch GEMM, Gram, preconditioner apply,
# CG steps) on the final tree -- the convolution gradient and max-pool operators moved to the new layouts late in the round
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
: > gpurun_out/r5/stress2_b.log
for seed in 811 812 813 814 815 816 817 818; do
  echo "=== seed $seed" >> gpurun_out/r5/stress2_b.log
  timeout -k 10 280 python tools/stress_parity2.py 32 $seed >> gpurun_out/r5/stress2_b.log 2>&1 || { echo "FAILED seed $seed"; tail -5 gpurun_out/r5/stress2_b.log; exit 1; }
  tail -1 gpurun_out/r5/stress2_b.log | cut -c1-160
done
