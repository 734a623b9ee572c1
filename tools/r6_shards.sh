#!/bin/bash
# Single-GPU shard times at the world sizes the driver runs and the 2- / 4-rank rehearsal of bench.py over gloo
# on ONE device, round 6 tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
: > gpurun_out/r6/shards.txt
for rows in 1000000 500000 250000 125000; do
  python bench.py --rows $rows --no-cpu-baseline --no-configs > gpurun_out/r6/shard_$rows.json 2> gpurun_out/r6/shard_$rows.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r6/shard_$rows.json').read().strip().splitlines()[-1])
c=d['cached_z_mode']
print($rows, 'ms/step %.3f kernel %.3f | cached ms/step %.3f kernel %.3f | build %.3f s' % (d['ms_per_step'], d['roofline']['kernel_ms'], c['ms_per_step'], c['roofline']['kernel_ms'], d['precond_build']['seconds']))
" | tee -a gpurun_out/r6/shards.txt
done
for n in 2 4; do
  XGPR_DIST_BACKEND=gloo XGPR_LOCAL_DEVICE=0 XGPR_BENCH_CHILD_FILE=gpurun_out/r6/bench_gloo_n${n}_child.json timeout -k 10 500 python bench.py --gpus $n --no-cpu-baseline --no-configs > gpurun_out/r6/bench_gloo_n$n.json 2> gpurun_out/r6/bench_gloo_n$n.err || { tail -30 gpurun_out/r6/bench_gloo_n$n.err; exit 1; }
  python3 -c "
import json, os
d=json.loads(open('gpurun_out/r6/bench_gloo_n$n.json').read().strip().splitlines()[-1])
c=json.load(open('gpurun_out/r6/bench_gloo_n${n}_child.json')) if os.path.exists('gpurun_out/r6/bench_gloo_n${n}_child.json') else {'status': 'skipped (ranks share one device)'}
print('N=$n (gloo, one device) ms/step %.3f loss %r check %s tol %s ranks %s | child: %s, loss check %s' % (d['ms_per_step'], d['final_loss'], d['final_loss_check'], d['fit_to_tol']['iterations'], d['distributed']['n_ranks_seen'], c['status'], (c.get('line') or {}).get('final_loss_check')))
" | tee -a gpurun_out/r6/shards.txt
done
