#!/bin/bash
# SQ counters of the fused matvec on the wide transforms (padded width 2048: two wave tiles per transform, 4096: four) next to
# padded width 1024 on the same box -- two rocprofv3 --pmc passes each (as tools/r5_sq_shapes.sh); summaries:
#   python tools/summarize_sq.py r6_fused_pmc_sq_p2048.json ztz3_kernel 524288 sq_p2048
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in "131072 2048 8192 p2048" "131072 4096 8192 p4096" "131072 1024 8192 p1024"; do
  set -- $shape
  rm -rf gpurun_out/sq_$4_sq1 gpurun_out/sq_$4_sq2
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/sq_$4_sq1 -- python tools/pmc_probe_shapes.py $1 $2 $3 > gpurun_out/sq_$4_1.log 2>&1 && \
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/sq_$4_sq2 -- python tools/pmc_probe_shapes.py $1 $2 $3 > gpurun_out/sq_$4_2.log 2>&1 && \
  python tools/summarize_sq.py r6_fused_pmc_sq_$4.json ztz3_kernel 524288 sq_$4
done
