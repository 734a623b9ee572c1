#!/bin/bash
# round 6: the rocprofv3 evidence of the final tree, collected and summarised on the GPU box; the summaries (profiles/r6_*) are copied
# to gpurun_out/r6/profiles/ so that they come back with the call.
#   1. --kernel-trace --stats of the default bench command                  -> profiles/r6_bench_n1_kernel_stats.csv
#   2./3. --pmc FETCH_SIZE / WRITE_SIZE passes of the bench command          -> r6_pmc_traffic*.json (+ build_id), r6_bench_n1_pmc_*_size.csv
#   4./5. two SQ passes of the fused matvec at cfg3 shape                    -> r6_fused_pmc_sq.json
#   6. SQ passes at padded widths 2048 / 4096 / 1024 (tools/r6_sq_wide.sh)   -> r6_fused_pmc_sq_p*.json
#   7. MFMA-busy pass on short launches (tools/r6_mfma_busy.sh)              -> r6_mfma_clock.json
#   8./9. SQ passes of the convolution feature operator                      -> r6_conv_pmc_sq.json
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export XGPR_ROUND=r6
mkdir -p gpurun_out/r6/profiles
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq1 gpurun_out/prof_sq2 gpurun_out/conv_sq1 gpurun_out/conv_sq2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py > gpurun_out/prof_stats.json 2> gpurun_out/prof_stats.err
echo "1 done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > gpurun_out/prof_fetch.json 2> gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > gpurun_out/prof_write.json 2> gpurun_out/prof_write.err
echo "2/3 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_sq1 -- python tools/pmc_probe.py > gpurun_out/prof_sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_sq2 -- python tools/pmc_probe.py > gpurun_out/prof_sq2.log 2>&1
echo "4/5 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/conv_sq1 -- python tools/pmc_probe_conv.py > gpurun_out/conv_sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/conv_sq2 -- python tools/pmc_probe_conv.py > gpurun_out/conv_sq2.log 2>&1
echo "8/9 done"
python tools/summarize_pmc.py
python tools/summarize_sq.py r6_fused_pmc_sq.json
python tools/summarize_sq.py r6_conv_pmc_sq.json wave_conv_kernel 4571224 conv || echo "conv summary failed"
bash tools/r6_sq_wide.sh | tail -3
bash tools/r6_mfma_busy.sh | tail -3
tail -c 300 gpurun_out/prof_stats.json > /dev/null
python - <<'PY'
import json
l = [x for x in open('gpurun_out/prof_stats.json').read().strip().splitlines() if x.startswith('{')][-1]
json.dump(json.loads(l), open('profiles/r6_bench_n1_line_profiled.json', 'w'), indent=1)
PY
# the line of a plain run AFTER the traffic JSONs of this build exist (roofline.traffic quoted): the one a later `python bench.py` reproduces
python bench.py > gpurun_out/r6/bench_final.json 2> gpurun_out/r6/bench_final.err
python - <<'PY'
import json
l = [x for x in open('gpurun_out/r6/bench_final.json').read().strip().splitlines() if x.startswith('{')][-1]
json.dump(json.loads(l), open('profiles/r6_bench_n1_line.json', 'w'), indent=1)
PY
python tools/wide_path_probe.py --rows 131072 --tag final --out profiles/r6_generic_path.json > gpurun_out/r6/generic_path_final.log 2>&1 || tail -5 gpurun_out/r6/generic_path_final.log
cp profiles/r6_* gpurun_out/r6/profiles/
ls gpurun_out/r6/profiles | wc -l
