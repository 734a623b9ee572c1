#!/bin/bash
# round 4: the fused matvec with a slot-local rendezvous instead of the workgroup barrier -- same-process A/B over builds
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
L="current sync0 slot_noprio slot_sleep0 slot_rise slot_p2 slot_tail3 sync0_noprio"
for shape in "1000000 1024 8192" "125000 1024 8192" "100000 256 4096" "250000 512 8192"; do
  echo "=== $shape"; timeout -k 10 300 python tools/ab_inproc.py "$shape" $L || exit 1
done > gpurun_out/r4/slotsync_ab.log 2>&1
cat gpurun_out/r4/slotsync_ab.log
timeout -k 10 600 python -m pytest tests/test_gpu_cg.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -x -q > gpurun_out/r4/slotsync_tests.log 2>&1; tail -3 gpurun_out/r4/slotsync_tests.log
