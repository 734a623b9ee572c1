#!/bin/bash
# Development aid: builds variants of the library with extra -D flags into tools/ablate/ (git-ignored; they travel to
# the GPU box with the snapshot) for same-box A/B timing through XGPR_HIP_LIB (tools/ab_lib.sh, tools/ab_many.sh).
#   tools/ablate_build.sh name1 "-DFLAG1 -DFLAG2" [name2 "-D..."] ...
cd "$(dirname "$0")/.."
mkdir -p tools/ablate
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 $flags xgpr_amd/csrc/xgpr_hip.hip -o tools/ablate/lib_$name.so 2>/dev/null && echo "built lib_$name.so ($flags)" || echo "FAILED $name" ) &
done
wait
