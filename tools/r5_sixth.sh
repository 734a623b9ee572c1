#!/bin/bash
# round 5, sixth GPU call: SGPR-mask flips shipped (fused matvec: third round; convolution operator at P = 256: third round) -- tests,
# the convolution and cfg2 timings, the bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_6.log 2>&1; rc=$?; tail -4 gpurun_out/r5/gputests_6.log
[ $rc -eq 0 ] || exit $rc
{ for rep in 1 2; do python tools/bench_conv.py 8192 9; python tools/bench_fused.py 100000 256 4096; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/flip_conv.log
python bench.py > gpurun_out/r5/bench_n1_b.json 2> gpurun_out/r5/bench_n1_b.err || { tail -20 gpurun_out/r5/bench_n1_b.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r5/bench_n1_b.json').read().strip().splitlines()[-1])
print('N=1 ms/step %.3f kernel %.3f loss check %s build_id %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['final_loss_check'], d['build_id'][:16]))
print('fit_to_tol', {k: d['fit_to_tol'][k] for k in ('iterations','cg_seconds','seconds')})
print('nmll', d['configs']['nmll_k26'])
print('block k26', d['cached_z_mode']['block_matvec_k26']['ms_per_matvec'], 'precond', d['precond_build']['seconds'], 'featgen', d['featgen_op']['ms'], 'conv', d['conv_featgen'].get('sequences_per_s'))
for c in ('cfg2','cfg4','cfg5'): print(c, d['configs'][c]['precond_build_s'], [(f['iterations'], round(f['ms_per_iteration'],3)) for f in d['configs'][c]['fits']])
"
