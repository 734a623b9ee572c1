#!/bin/bash
# padded widths 32 / 64: rows-only layout with one / two quad_perm DPP stages per round (shipped) against the transposed-columns layout with two
# exchanges per round (-DXGPR_ABL_NORD); tests first
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cg.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_feat_plans.py -m gpu -x -q > gpurun_out/r5/gputests_rd.log 2>&1; rc=$?; tail -3 gpurun_out/r5/gputests_rd.log
[ $rc -eq 0 ] || { grep -E "Error|assert" gpurun_out/r5/gputests_rd.log | head; exit 1; }
{
python tools/ab_inproc.py "262144 64 8192" current nord current nord
python tools/ab_inproc.py "262144 32 8192" current nord
python tools/ab_inproc.py "262144 50 4096" current nord
python tools/ab_inproc.py "262144 33 6000" current nord
python tools/ab_inproc.py "131072 64 32768" current nord
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/rd_ab.log
