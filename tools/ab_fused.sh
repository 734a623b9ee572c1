#!/bin/bash
# A/B of the fused CG matvec: two-wave kernel (XGPR_ZTZ_WAVES=2) vs the three-wave kernel, same process layout,
# cfg3 / cfg2 / cfg5-width shapes; prints time and a checksum of w (bit-identical features => equal to ~1e-15).
cd "$GRAFT_REPO_ROOT"
for shape in "262144 1024 8192" "262144 256 4096" "131072 512 8192" "100000 1000 8192" "50000 128 2048"; do
  XGPR_ZTZ_WAVES=2 python tools/bench_fused.py $shape
  python tools/bench_fused.py $shape
done
