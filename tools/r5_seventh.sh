#!/bin/bash
# is the 5.56 ms of the sixth call the box or the build?  same box: interleaved A/B and two bench lines (flips on / off)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
python tools/ab_inproc.py "1000000 1024 8192" current noflip current noflip 2>&1 | grep -v amdgpu.ids
for lib in xgpr_amd/libxgpr_hip.so tools/ablate/lib_noflip.so xgpr_amd/libxgpr_hip.so; do
  XGPR_HIP_LIB=$lib python bench.py --no-configs --no-cpu-baseline > gpurun_out/r5/bench_ab.json 2> gpurun_out/r5/bench_ab.err || { tail -5 gpurun_out/r5/bench_ab.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r5/bench_ab.json').read().strip().splitlines()[-1])
print('$lib', 'ms/step %.3f kernel %.3f valu_only %s cached kernel %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['vector_pipe']['valu_only']['ms'], d['cached_z_mode']['roofline']['kernel_ms']))
"
done
