#!/bin/bash
# Duration of the fused matvec's kernels against the number of rows (rocprofv3 kernel trace): the intercept is the fixed
# cost a shard pays per CG iteration whatever its size.
cd /tmp && export TMPDIR=/tmp
for n in 31250 62500 125000 250000 500000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/ff_$n -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_fused.py $n > $GRAFT_REPO_ROOT/gpurun_out/r2/ff_$n.log 2>&1
  echo "rows $n"; grep "ztz3\|reduce_slabs\|pack_radem" $GRAFT_REPO_ROOT/gpurun_out/r2/ff_$n/run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
