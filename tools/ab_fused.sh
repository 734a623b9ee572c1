#!/bin/bash
# A/B of the fused CG matvec: two-wave kernel (XGPR_ZTZ_WAVES=2) vs the three-wave kernel, same process layout,
# cfg3 / cfg2 / cfg5-width shapes and small shards; prints time and a checksum of w.
cd "$GRAFT_REPO_ROOT"
for shape in "262144 1024 8192" "262144 256 4096" "131072 512 8192" "100000 1000 8192" "50000 128 2048" "1000000 128 2048" "2000 32 512" "20000 256 4096" "125000 1024 8192" "30000 200 6144" "30000 300 12288"; do
  XGPR_ZTZ_WAVES=2 python tools/bench_fused.py $shape
  python tools/bench_fused.py $shape
done
