#!/bin/bash
# round 5, fifth GPU call: the transposed-columns layout (P <= 256: no cross-lane stage) -- GPU tests, then same-box A/B
# against the build with the old layout at every width (-DXGPR_ABL_NOC2): cfg2's fused matvec, cfg4's convolution features
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_5.log 2>&1; rc=$?; tail -6 gpurun_out/r5/gputests_5.log
[ $rc -eq 0 ] || exit $rc
{
for rep in 1 2; do
  for lib in "" tools/ablate/lib_noc2.so; do
    echo "== lib: ${lib:-shipped}"
    XGPR_HIP_LIB=${lib:-xgpr_amd/libxgpr_hip.so} python tools/bench_fused.py 100000 256 4096
    XGPR_HIP_LIB=${lib:-xgpr_amd/libxgpr_hip.so} python tools/bench_fused.py 262144 128 4096
    XGPR_HIP_LIB=${lib:-xgpr_amd/libxgpr_hip.so} python tools/bench_conv.py 8192 9

  done
done
} > gpurun_out/r5/c2_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/c2_ab.log
