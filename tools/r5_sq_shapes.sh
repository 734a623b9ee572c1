#!/bin/bash
# SQ counters of the fused matvec on the three register layouts: cfg2's shape (P = 256: R <-> C2), d = 32 (P = 32: R <-> C2, two column stages fewer),
# d = 16 (P = 16: rows only, no exchange) -- two passes each, as tools/collect_profiles.sh does for cfg3's shape
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in "262144 256 4096 p256" "262144 32 8192 p32" "262144 16 8192 p16"; do
  set -- $shape
  rm -rf gpurun_out/sq_$4_1 gpurun_out/sq_$4_2
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/sq_$4_sq1 -- python tools/pmc_probe_shapes.py $1 $2 $3 > gpurun_out/sq_$4_1.log 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/sq_$4_sq2 -- python tools/pmc_probe_shapes.py $1 $2 $3 > gpurun_out/sq_$4_2.log 2>&1
  ls gpurun_out/sq_$4_sq1/*/ | head -3
done
