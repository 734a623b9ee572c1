#!/bin/bash
# Collects the rocprofv3 evidence bench.py quotes (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the default bench command        -> gpurun_out/prof_stats
#   2. --pmc FETCH_SIZE  (own pass, kernel trace only)            -> gpurun_out/prof_fetch
#   3. --pmc WRITE_SIZE  (own pass)                                -> gpurun_out/prof_write
# then tools/summarize_pmc.py turns them into profiles/r1_*.  The program itself follows `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py > gpurun_out/prof_stats.json 2> gpurun_out/prof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_fetch.json 2> gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_write.json 2> gpurun_out/prof_write.err
tail -c 600 gpurun_out/prof_stats.json
