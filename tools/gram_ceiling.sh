#!/bin/bash
# What bounds gram_lds_kernel: the shipped build next to timing-only ablations (wrong results) built by
#   tools/ablate_build.sh gr_noconv -DXGPR_ABL_GR_NOCONV gr_nobar -DXGPR_ABL_GR_NOBAR gr_neither "-DXGPR_ABL_GR_NOCONV -DXGPR_ABL_GR_NOBAR"
cd "$GRAFT_REPO_ROOT"
for l in xgpr_amd/libxgpr_hip.so tools/ablate/lib_gr_noconv.so tools/ablate/lib_gr_nobar.so tools/ablate/lib_gr_neither.so xgpr_amd/libxgpr_hip.so; do
  echo "--- $l"; XGPR_HIP_LIB=$l timeout -k 10 120 python tools/bench_gram.py 131072 8192 2>/dev/null | head -1
done
