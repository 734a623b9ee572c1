#!/bin/bash
# round 5: the Rademacher flips of ONE round of the fused matvec through SGPR lane masks (v_cndmask with a negated source) against
# the shipped per-lane sign words (v_add_u32 + v_bitop3): same process, interleaved (tools/ab_inproc.py)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
python tools/ab_inproc.py "1000000 1024 8192" current flip0 flip1 flip2 current
python tools/ab_inproc.py "125000 1024 8192" current flip0 flip1 flip2
python tools/ab_inproc.py "100000 256 4096" current flip0 flip1 flip2
} > gpurun_out/r5/signflip_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/signflip_ab.log
