#!/bin/bash
# round 5, final state: GPU tests, smoke, the default bench line (kept as profiles/r5_bench_n1_line_final.json), and the 2- / 4-rank
# rehearsal on ONE device over gloo (each followed by its watched direct-path child job)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_final.log 2>&1; rc=$?; tail -3 gpurun_out/r5/gputests_final.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
python bench.py > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err || { tail -20 gpurun_out/r5/bench_final.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r5/bench_final.json').read().strip().splitlines()[-1])
print('N=1 ms/step %.3f kernel %.3f loss check %s build_id %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['final_loss_check'], d['build_id'][:16]))
print('fit_to_tol', {k: d['fit_to_tol'][k] for k in ('iterations','cg_seconds','seconds')}, d['fit_to_tol']['product_default']['cg_seconds'])
print('nmll', d['configs']['nmll_k26']['ms_per_iteration'], 'block k26', d['cached_z_mode']['block_matvec_k26']['ms_per_matvec'], 'precond', d['precond_build']['seconds'], 'featgen', d['featgen_op']['ms'], 'conv', d['conv_featgen'].get('sequences_per_s'))
print('vector_pipe', json.dumps(d['roofline']['vector_pipe'])[:600])
for c in ('cfg2','cfg4','cfg5'): print(c, d['configs'][c]['precond_build_s'], [(f['iterations'], round(f['ms_per_iteration'],3)) for f in d['configs'][c]['fits']])
"
for n in 2 4; do
  XGPR_DIST_BACKEND=gloo XGPR_LOCAL_DEVICE=0 XGPR_BENCH_CHILD_FILE=gpurun_out/r5/bench_gloo_n${n}_child.json timeout -k 10 500 python bench.py --gpus $n --no-cpu-baseline --no-configs > gpurun_out/r5/bench_gloo_n$n.json 2> gpurun_out/r5/bench_gloo_n$n.err || { tail -30 gpurun_out/r5/bench_gloo_n$n.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r5/bench_gloo_n$n.json').read().strip().splitlines()[-1])
import os; c=json.load(open('gpurun_out/r5/bench_gloo_n${n}_child.json')) if os.path.exists('gpurun_out/r5/bench_gloo_n${n}_child.json') else {'status': 'skipped (ranks share one device)'}
print('N=$n (gloo, one device) ms/step %.3f loss %r check %s tol %s ranks %s | child: %s, loss check %s' % (d['ms_per_step'], d['final_loss'], d['final_loss_check'], d['fit_to_tol']['iterations'], d['distributed']['n_ranks_seen'], c['status'], (c.get('line') or {}).get('final_loss_check')))
"
done
