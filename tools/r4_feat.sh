#!/bin/bash
# round 4: the feature operator / cache rows on the three-wave persistent plan (ztz3_kernel Z3_FEAT64 / Z3_FEAT32) vs wave_rbf_kernel
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_cfg_shapes.py tests/test_gpu_stress.py -x -q > gpurun_out/r4/feat_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4/feat_tests.log; [ $rc -eq 0 ] || exit $rc
{ echo "=== three-wave persistent plan, consecutive rows per slot"; python tools/bench_featgen.py; echo "=== wave_rbf_kernel (XGPR_FEAT_PLAN=wave)"; XGPR_FEAT_PLAN=wave python tools/bench_featgen.py;
  echo "=== three-wave persistent plan, strided rows"; XGPR_HIP_LIB=tools/ablate/lib_feat_strided.so python tools/bench_featgen.py;
  echo "=== three-wave persistent plan, consecutive rows (again)"; python tools/bench_featgen.py; } > gpurun_out/r4/feat_ab.log 2>&1
grep -v "amdgpu.ids\|fill_\|widen" gpurun_out/r4/feat_ab.log
