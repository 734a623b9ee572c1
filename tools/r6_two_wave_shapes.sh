#!/bin/bash
# The shapes xgpr_ztz_matvec_f32 still sends to the two-wave kernel (wave_ztz_kernel): seven tiles per datapoint
# (12288 < num_rffs <= 14336) and one tile per datapoint at padded width >= 128 (num_rffs <= 2048) -- each against the
# three-wave kernel's alternative plan on the same box (tools/bench_fused.py, 131072 rows):
#   seven tiles: two passes in tile groups of four (XGPR_ZTZ_TWO_PASS_ABOVE=6000)
#   one tile:    twelve one-wave slots on the three-wave kernel (XGPR_ZTZ3_ONE_TILE=1)
cd "$GRAFT_REPO_ROOT"
echo "# rows=131072; shipped plan first, alternative second; identical checksums required"
for d in 128 256 512 1024; do
  echo "## seven tiles, d=$d M=14336: two-wave single pass (shipped) | three-wave two passes, groups of 4"
  python tools/bench_fused.py 131072 $d 14336 2>/dev/null
  XGPR_ZTZ_TWO_PASS_ABOVE=6000 python tools/bench_fused.py 131072 $d 14336 2>/dev/null
done
for d in 128 256 512 1024; do
  echo "## one tile, d=$d M=2048: two-wave (shipped) | three-wave, twelve one-wave slots"
  python tools/bench_fused.py 131072 $d 2048 2>/dev/null
  XGPR_ZTZ3_ONE_TILE=1 python tools/bench_fused.py 131072 $d 2048 2>/dev/null
done
