#!/bin/bash
# rows that are not whole aligned 16-byte groups on the three-wave kernel (float-by-float LDS-DMA) against the two-wave kernel; then
# the GPU tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
{
for shape in "1022 8192" "1023 8192" "617 4096" "130 4096" "250 10240" "1000 16384"; do
  set -- $shape
  for waves in 3 2; do
    echo "== d=$1 M=$2 XGPR_ZTZ_WAVES=$waves"
    XGPR_ZTZ_WAVES=$waves python tools/bench_fused.py 131072 $1 $2
  done
done
echo "== aligned reference: d=1024 M=8192"; python tools/bench_fused.py 131072 1024 8192
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/dma1_ab.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_dma1.log 2>&1; tail -3 gpurun_out/r5/gputests_dma1.log
