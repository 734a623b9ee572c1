#!/bin/bash
# MFMA-busy counter pass on short launches (tools/mfma_busy_probe.py) -> gpurun_out/prof_mfma_busy -> profiles/r6_mfma_clock.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_mfma_busy
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_mfma_busy -- python tools/mfma_busy_probe.py > gpurun_out/prof_mfma_busy.log 2>&1
tail -2 gpurun_out/prof_mfma_busy.log
python tools/summarize_mfma_busy.py
