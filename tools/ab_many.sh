#!/bin/bash
# times the fused matvec for every library in tools/ablate/ matching the given names plus the current build:
#   tools/ab_many.sh "rows d M" name1 name2 ...
cd "$GRAFT_REPO_ROOT"
shape=$1; shift
echo "=== current"; python tools/bench_fused.py $shape 2>/dev/null
for n in "$@"; do echo "=== $n"; XGPR_HIP_LIB=tools/ablate/lib_$n.so python tools/bench_fused.py $shape 2>/dev/null; done
echo "=== current (again)"; python tools/bench_fused.py $shape 2>/dev/null
