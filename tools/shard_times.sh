#!/bin/bash
# Single-GPU shard times at the world sizes the driver runs, and the RCCL path with one rank.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
for rows in 1000000 500000 250000 125000; do
  python bench.py --rows $rows --no-cpu-baseline > gpurun_out/r3/shard_$rows.json 2> gpurun_out/r3/shard_$rows.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r3/shard_$rows.json').read().strip().splitlines()[-1])
print($rows, 'ms/step %.3f kernel %.3f build %.3f s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['precond_build']['seconds']))
"
done
XGPR_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3/nccl1.json 2> gpurun_out/r3/nccl1.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r3/nccl1.json').read().strip().splitlines()[-1])
print('torchrun 1 rank:', d['ms_per_step'], d['distributed'])
"
