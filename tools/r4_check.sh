#!/bin/bash
# round 4: GPU tests + the default bench line + the shard sizes of an 8-GPU run, one call
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputests.log 2>&1; rc=$?; tail -3 gpurun_out/r4/gputests.log; [ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/r4/bench_n1.json 2> gpurun_out/r4/bench_n1.err || { tail -20 gpurun_out/r4/bench_n1.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r4/bench_n1.json').read().strip().splitlines()[-1])
print('N=1 ms/step %.3f kernel %.3f loss %r check %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['final_loss'], d['final_loss_check']))
print('fit_to_tol', d['fit_to_tol'])
print('block k26', d['cached_z_mode']['block_matvec_k26']['ms_per_matvec'], 'precond', d['precond_build']['seconds'], 'valu', d['roofline']['vector_pipe']['valu_only'])
"
for rows in 500000 250000 125000; do
  python bench.py --rows $rows --no-cpu-baseline --no-configs > gpurun_out/r4/shard_$rows.json 2> gpurun_out/r4/shard_$rows.err || { tail -20 gpurun_out/r4/shard_$rows.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r4/shard_$rows.json').read().strip().splitlines()[-1])
print($rows, 'ms/step %.3f kernel %.3f build %.3f s  fit_to_tol %d it %.3f s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['precond_build']['seconds'], d['fit_to_tol']['iterations'], d['fit_to_tol']['cg_seconds']))
"
done
