#!/bin/bash
# round 5, fourth GPU call: the whole GPU suite on the tree with the split projection, the child job and the cache rule
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests_4.log 2>&1; rc=$?; tail -6 gpurun_out/r5/gputests_4.log
exit $rc
