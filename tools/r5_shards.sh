#!/bin/bash
# Single-GPU shard times at the world sizes the driver runs, round 5 tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
for rows in 1000000 500000 250000 125000; do
  python bench.py --rows $rows --no-cpu-baseline --no-configs > gpurun_out/r5/shard_$rows.json 2> gpurun_out/r5/shard_$rows.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r5/shard_$rows.json').read().strip().splitlines()[-1])
c=d['cached_z_mode']
print($rows, 'ms/step %.3f kernel %.3f | cached ms/step %.3f kernel %.3f | build %.3f s' % (d['ms_per_step'], d['roofline']['kernel_ms'], c['ms_per_step'], c['roofline']['kernel_ms'], d['precond_build']['seconds']))
"
done
