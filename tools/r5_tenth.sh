#!/bin/bash
# round 5: the preconditioner pass with two windows in flight (side stream: feature rows + SRHT of the next window under the
# contraction of this one) against the one-stream pass, same box; then the preconditioner / CG / NMLL / fullsize tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
for rep in 1 2; do
  for pipe in 1 0; do
    XGPR_PRECOND_PIPELINE=$pipe python tools/bench_precond_build.py 1000000 1024 8192 512 srht
    XGPR_PRECOND_PIPELINE=$pipe python tools/bench_precond_build.py 250000 512 32768 2048 srht_2 8192
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/precond_pipeline_ab.log
timeout -k 10 900 python -m pytest tests/test_gpu_cg.py tests/test_gpu_sketch.py tests/test_gpu_nmll.py tests/test_gpu_cfg_shapes.py tests/test_gpu_classifier.py tests/test_gpu_models.py -m gpu -x -q > gpurun_out/r5/gputests_10.log 2>&1; tail -4 gpurun_out/r5/gputests_10.log
