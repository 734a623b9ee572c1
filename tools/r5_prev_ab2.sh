#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
python tools/ab_inproc.py "1000000 1024 8192" current prev current prev
python tools/ab_inproc.py "125000 1024 8192" current prev
python tools/ab_inproc.py "100000 256 4096" current prev
python tools/bench_fused.py 131072 1022 8192
python tools/bench_fused.py 131072 1024 10240
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/prev_ab2.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_tmpl.log 2>&1; tail -3 gpurun_out/r5/gputests_tmpl.log
