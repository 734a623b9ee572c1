#!/bin/bash
# round 5, first GPU call: the GPU tests on the new tree (table-driven large-argument reduction, buffer-resource window
# loads in the convolution kernels), the bench line, the long-window convolution timings and the fixed-cost phase table.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_1.log 2>&1; rc=$?; tail -5 gpurun_out/r5/gputests_1.log
[ $rc -eq 0 ] || exit $rc
for w in 9 24 48; do python tools/bench_conv.py 2048 $w; done > gpurun_out/r5/conv_windows.log 2>&1; cat gpurun_out/r5/conv_windows.log
XGPR_HIP_LIB=tools/ablate/lib_timing.so timeout -k 10 300 python tools/fixed_cost.py gpurun_out/r5/fixed_cost_before.json > gpurun_out/r5/fixed_cost_before.log 2>&1; tail -12 gpurun_out/r5/fixed_cost_before.log
python bench.py > gpurun_out/r5/bench_n1.json 2> gpurun_out/r5/bench_n1.err || { tail -20 gpurun_out/r5/bench_n1.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r5/bench_n1.json').read().strip().splitlines()[-1])
print('N=1 ms/step %.3f kernel %.3f loss %r check %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['final_loss'], d['final_loss_check']))
"
