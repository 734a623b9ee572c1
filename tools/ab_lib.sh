#!/bin/bash
# A/B of two builds of the library on the fused CG matvec (same box, same process layout):
#   tools/ab_lib.sh <other libxgpr_hip.so> [shape ...]     (default shapes: cfg3 window, cfg2, 125000-row shard)
# prints time and a checksum of w for each build (checksums must agree to the last digit).
cd "$GRAFT_REPO_ROOT"
OTHER=$1; shift
SHAPES=("$@")
[ ${#SHAPES[@]} -eq 0 ] && SHAPES=("262144 1024 8192" "262144 256 4096" "131072 512 8192" "125000 1024 8192" "1000000 1024 8192")
for shape in "${SHAPES[@]}"; do
  echo "--- other: $OTHER"; XGPR_HIP_LIB=$OTHER python tools/bench_fused.py $shape
  echo "--- current";        python tools/bench_fused.py $shape
done
