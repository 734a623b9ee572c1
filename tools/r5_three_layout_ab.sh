#!/bin/bash
# round 5: would a THIRD register layout pay at P = 1024?  It would remove the two quad_perm DPP stages of every round (timing build
# nodpp: results wrong) at the price of one more exchange per round = 144 LDS cycles per tile (timing builds xtra1 / xtra2: 112 / 224
# extra LDS cycles per tile as exchange round trips that compose to the identity)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
python tools/ab_inproc.py "1000000 1024 8192" current nodpp xtra1 nodpp_xtra1 nodpp_xtra2 current 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/three_layout_ab.log
python tools/ab_inproc.py "125000 1024 8192" current nodpp xtra1 nodpp_xtra1 nodpp_xtra2 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5/three_layout_ab.log
