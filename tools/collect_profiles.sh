#!/bin/bash
# Collects the rocprofv3 evidence bench.py quotes (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the default bench command        -> gpurun_out/prof_stats
#   2. --pmc FETCH_SIZE  (own pass, kernel trace only)            -> gpurun_out/prof_fetch
#   3. --pmc WRITE_SIZE  (own pass)                                -> gpurun_out/prof_write
#   4./5. two SQ passes of tools/pmc_probe.py (fused matvec at cfg3 shape, 262144 rows) -> gpurun_out/prof_sq1, prof_sq2
#   6./7. GRBM_GUI_ACTIVE (clock under load) + MFMA counters of the contraction and of the block kernels -> gpurun_out/prof_mfma_gemm, prof_mfma_block
# then tools/summarize_pmc.py / tools/summarize_sq.py / tools/summarize_mfma.py turn them into profiles/r3_* (XGPR_ROUND).  The program itself follows `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_mfma_gemm gpurun_out/prof_mfma_block gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq1 gpurun_out/prof_sq2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py > gpurun_out/prof_stats.json 2> gpurun_out/prof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > gpurun_out/prof_fetch.json 2> gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > gpurun_out/prof_write.json 2> gpurun_out/prof_write.err
tail -c 400 gpurun_out/prof_stats.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_sq1 -- python tools/pmc_probe.py > gpurun_out/prof_sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_sq2 -- python tools/pmc_probe.py > gpurun_out/prof_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/prof_mfma_gemm -o run -- python3 tools/bench_sketch_gemm.py > gpurun_out/prof_mfma_gemm.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/prof_mfma_block -o run -- python3 tools/bench_nmll.py > gpurun_out/prof_mfma_block.log 2>&1
# 8./9. the same two SQ passes for the convolution feature operator at cfg4 shape (2048 sequences)
rm -rf gpurun_out/conv_sq1 gpurun_out/conv_sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/conv_sq1 -- python tools/pmc_probe_conv.py > gpurun_out/conv_sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/conv_sq2 -- python tools/pmc_probe_conv.py > gpurun_out/conv_sq2.log 2>&1
ls gpurun_out/prof_sq1/*/ gpurun_out/prof_sq2/*/ gpurun_out/prof_mfma_gemm gpurun_out/prof_mfma_block
