#!/bin/bash
# round 5, the final tree: rocprofv3 evidence (tools/collect_profiles.sh), then tools/r5_final.sh (GPU tests, smoke, bench line, gloo rehearsals)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
bash tools/collect_profiles.sh > gpurun_out/r5/collect_profiles.log 2>&1; rc=$?; tail -3 gpurun_out/r5/collect_profiles.log
[ $rc -eq 0 ] || exit $rc
bash tools/r5_final.sh
