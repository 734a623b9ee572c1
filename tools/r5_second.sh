#!/bin/bash
# round 5, second GPU call: clock-against-time inside a launch + the idle-gap experiment, the cache-rule shapes, the tests touched
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
XGPR_HIP_LIB=tools/ablate/lib_timing.so timeout -k 10 300 python tools/fixed_cost.py gpurun_out/r5/fixed_cost.json 31250 125000 1000000 > gpurun_out/r5/fixed_cost.log 2>&1; grep -v "per XCC" gpurun_out/r5/fixed_cost.log | cut -c1-1500 | tail -16
timeout -k 10 300 python tools/cache_rule_probe.py gpurun_out/r5/cache_rule.json > gpurun_out/r5/cache_rule.log 2>&1; tail -14 gpurun_out/r5/cache_rule.log
timeout -k 10 600 python -m pytest tests/test_gpu_classifier.py tests/test_gpu_cg.py -m gpu -x -q > gpurun_out/r5/gputests_2.log 2>&1; tail -5 gpurun_out/r5/gputests_2.log
