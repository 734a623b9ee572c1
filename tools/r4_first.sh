#!/bin/bash
# round 4, first GPU call: the bench line at N = 1 and the 2- / 4-rank rehearsal on ONE device (gloo stages the sums through
# the host; what it shows is that the N-rank job completes and reports the 1-rank loss), then the GPU tests.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python bench.py > gpurun_out/r4/bench_n1.json 2> gpurun_out/r4/bench_n1.err || { tail -20 gpurun_out/r4/bench_n1.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r4/bench_n1.json').read().strip().splitlines()[-1])
print('N=1 ms/step %.3f kernel %.3f loss %r check %s fit_to_tol %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['final_loss'], d['final_loss_check'], d['fit_to_tol']))
print(json.dumps(d['configs'], indent=1))
"
for n in 2 4; do
  XGPR_DIST_BACKEND=gloo XGPR_LOCAL_DEVICE=0 timeout -k 10 400 python bench.py --gpus $n --no-cpu-baseline > gpurun_out/r4/bench_gloo_n$n.json 2> gpurun_out/r4/bench_gloo_n$n.err || { tail -30 gpurun_out/r4/bench_gloo_n$n.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r4/bench_gloo_n$n.json').read().strip().splitlines()[-1])
print('N=$n (gloo, one device) ms/step %.3f loss %r check %s tol %s ranks %s' % (d['ms_per_step'], d['final_loss'], d['final_loss_check'], d['fit_to_tol']['iterations'], d['distributed']['n_ranks_seen']))
"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputests_1.log 2>&1; tail -5 gpurun_out/r4/gputests_1.log
