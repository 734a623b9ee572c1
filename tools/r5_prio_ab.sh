#!/bin/bash
# round 5: the wave-priority plan of the three-wave kernel at cfg2's shape (P = 256, two tiles per datapoint, six slots per workgroup;
# the plan was tuned at cfg3's) -- same process, interleaved
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
python tools/ab_inproc.py "100000 256 4096" current prio_none prio_b prio_c prio_d current 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/prio_ab.log
python tools/ab_inproc.py "1000000 1024 8192" current prio_b prio_c prio_d 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5/prio_ab.log
python tools/ab_inproc.py "250000 512 8192" current prio_none prio_b prio_c prio_d 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5/prio_ab.log
