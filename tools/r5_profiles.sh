#!/bin/bash
# round 5: the rocprofv3 evidence of the final tree (tools/collect_profiles.sh), then the GPU tests and smoke
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
bash tools/collect_profiles.sh > gpurun_out/r5/collect_profiles.log 2>&1; rc=$?; tail -5 gpurun_out/r5/collect_profiles.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
