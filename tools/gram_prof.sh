cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
python tools/bench_gram.py 131072 8192
rm -rf gpurun_out/gramk
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gramk -- python tools/bench_gram.py 131072 8192 > gpurun_out/gramk.log 2>&1
grep -h "gram\|Cijk\|convert" gpurun_out/gramk/*/*kernel_stats.csv | cut -c1-200
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/gramp -- python tools/bench_gram.py 131072 8192 > gpurun_out/gramp.log 2>&1
ls gpurun_out/gramp/*/
