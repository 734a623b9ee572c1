#!/bin/bash
# round 5, third GPU call: the split block projection -- tests, then time against rows with the unsplit kernel beside it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_classifier.py tests/test_gpu_cg.py tests/test_gpu_nmll.py -m gpu -x -q > gpurun_out/r5/gputests_3.log 2>&1; rc=$?; tail -5 gpurun_out/r5/gputests_3.log
[ $rc -eq 0 ] || exit $rc
XGPR_ZB_SPLIT=1 timeout -k 10 300 python tools/zblock_small_probe.py gpurun_out/r5/zblock_small_unsplit.json > gpurun_out/r5/zblock_small_unsplit.log 2>&1; tail -30 gpurun_out/r5/zblock_small_unsplit.log
timeout -k 10 300 python tools/zblock_small_probe.py gpurun_out/r5/zblock_small.json > gpurun_out/r5/zblock_small.log 2>&1; tail -30 gpurun_out/r5/zblock_small.log
