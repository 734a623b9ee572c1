#!/bin/bash
# Two rocprofv3 --pmc passes (SQ counters) of tools/pmc_probe.py -> gpurun_out/prof_sq1, prof_sq2; summarise with
# tools/summarize_sq.py <name under profiles/>.  The program itself follows `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_sq1 gpurun_out/prof_sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_sq1 -- python tools/pmc_probe.py > gpurun_out/prof_sq1.log 2>&1 && \
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_sq2 -- python tools/pmc_probe.py > gpurun_out/prof_sq2.log 2>&1
ls gpurun_out/prof_sq1/*/ gpurun_out/prof_sq2/*/
