#!/bin/bash
# Randomised parity sweeps of round 6 (shapes now include padded widths 2048 / 4096): tools/stress_parity.py x 8 seeds x 40 cases, tools/stress_parity2.py x 2
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
: > gpurun_out/r6/stress_parity.txt
for s in ${STRESS_SEEDS:-601 602 603 604 605 606 607 608}; do
  echo "== stress_parity.py 40 $s" >> gpurun_out/r6/stress_parity.txt
  timeout -k 10 400 python tools/stress_parity.py 40 $s 2>&1 | grep -v "amdgpu.ids\|starts:" | tail -4 >> gpurun_out/r6/stress_parity.txt || exit 1
done
for s in ${STRESS_SEEDS2:-71 72}; do
  echo "== stress_parity2.py 64 $s" >> gpurun_out/r6/stress_parity.txt
  timeout -k 10 400 python tools/stress_parity2.py 64 $s 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r6/stress_parity.txt || exit 1
done
grep -c "ok$" gpurun_out/r6/stress_parity.txt; grep -i "error\|Traceback" gpurun_out/r6/stress_parity.txt | head
