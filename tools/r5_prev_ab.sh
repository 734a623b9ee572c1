#!/bin/bash
# does the shape generality added to the three-wave kernel (spare waves, float-by-float row fetch: two SGPRs spilled outside the loop)
# cost the shapes it already served?  same process: this tree against the commit before
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
python tools/ab_inproc.py "1000000 1024 8192" current prev current prev
python tools/ab_inproc.py "125000 1024 8192" current prev
python tools/ab_inproc.py "100000 256 4096" current prev
python tools/ab_inproc.py "250000 512 8192" current prev
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/prev_ab.log
