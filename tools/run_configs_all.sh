#!/bin/bash
# cfg2 / cfg4 / cfg5 end to end at their per-GPU share, each twice in one process-pair (second = libraries warm)
cd "$GRAFT_REPO_ROOT"
for c in cfg2 cfg4 cfg5; do python tools/run_configs.py $c 2>&1 | grep -v amdgpu; done
