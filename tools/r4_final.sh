#!/bin/bash
# round 4, final state: GPU tests, the default bench line (kept as profiles/r4_bench_n1_line.json), smoke
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputests_final.log 2>&1; rc=$?; tail -3 gpurun_out/r4/gputests_final.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
python bench.py > gpurun_out/r4/bench_final.json 2> gpurun_out/r4/bench_final.err || { tail -20 gpurun_out/r4/bench_final.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r4/bench_final.json').read().strip().splitlines()[-1])
print('N=1 ms/step %.3f kernel %.3f loss check %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['final_loss_check']))
print('fit_to_tol', {k: d['fit_to_tol'][k] for k in ('iterations','cg_seconds','seconds')}, d['fit_to_tol']['product_default']['cg_seconds'])
print('nmll', d['configs']['nmll_k26'])
print('block k26', d['cached_z_mode']['block_matvec_k26']['ms_per_matvec'], 'precond', d['precond_build']['seconds'], 'featgen', d['featgen_op']['ms'])
for c in ('cfg2','cfg4','cfg5'): print(c, d['configs'][c]['precond_build_s'], [(f['iterations'], round(f['ms_per_iteration'],3)) for f in d['configs'][c]['fits']])
"
