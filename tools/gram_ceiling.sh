#!/bin/bash
# What bounds gram_lds_kernel: the shipped build next to timing-only ablations (wrong results) built by
#   tools/ablate_build.sh gr_noconv -DXGPR_ABL_GR_NOCONV gr_nobar -DXGPR_ABL_GR_NOBAR gr_neither "-DXGPR_ABL_GR_NOCONV -DXGPR_ABL_GR_NOBAR"
# (and, when present, tools/ablate/lib_head.so = a build of an earlier commit) -> profiles/r3_gram_ceiling.json
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
: > gpurun_out/r3/gram_ceiling.log
for l in xgpr_amd/libxgpr_hip.so tools/ablate/lib_gr_noconv.so tools/ablate/lib_gr_nobar.so tools/ablate/lib_gr_neither.so tools/ablate/lib_head.so xgpr_amd/libxgpr_hip.so; do
  [ -f $l ] || continue
  echo "--- $l" | tee -a gpurun_out/r3/gram_ceiling.log
  XGPR_HIP_LIB=$l timeout -k 10 120 python tools/bench_gram.py 131072 8192 2>/dev/null | head -1 | tee -a gpurun_out/r3/gram_ceiling.log
done
python3 - <<'PY'
import json, re
rows, cur = [], None
for line in open("gpurun_out/r3/gram_ceiling.log"):
    if line.startswith("--- "): cur = line[4:].strip()
    m = re.search(r"rows=(\d+) M=(\d+): ([\d.]+) ms\s+executed ([\d.]+) TFLOP/s = ([\d.]+)", line)
    if m: rows.append({"library": cur, "rows": int(m.group(1)), "M": int(m.group(2)), "ms": float(m.group(3)), "executed_TFLOPs": float(m.group(4)), "frac_of_78.6": float(m.group(5))})
names = {"xgpr_amd/libxgpr_hip.so": "shipped", "tools/ablate/lib_gr_noconv.so": "no conversion of the row operand (timing only)",
         "tools/ablate/lib_gr_nobar.so": "no workgroup barrier per chunk (timing only)", "tools/ablate/lib_gr_neither.so": "neither (timing only)",
         "tools/ablate/lib_head.so": "the kernel before the chunk loop was unrolled by 6 (rolled loop: 20 vector instructions per 32 MFMAs)"}
for r in rows: r["what"] = names.get(r["library"], r["library"])
out = {"kernel": "gram_lds_kernel (xgpr_ztz_gram_f64): Z^T Z from float32 feature rows on v_mfma_f64_16x16x4_f64, tiles on or above the diagonal",
       "how": "tools/gram_ceiling.sh: tools/bench_gram.py 131072 8192 per library through XGPR_HIP_LIB, one gpurun call, same box",
       "runs": rows,
       "reading": "every vector instruction between float64 MFMAs costs matrix-pipe time on this part: the rolled loop carried 8 conversions + 12 address / select instructions per 32 MFMAs; unrolled by 6 (compile-time ring positions) only the 8 conversions are left. An integer widening of the float32 operand (5 full-rate instructions instead of one v_cvt_f64_f32) measured 2.7 % slower than the conversion it replaced (gpurun_out/r3/gram_int.log: 130.9 vs 127.4 ms on the rolled loop)."}
json.dump(out, open("profiles/r3_gram_ceiling.json", "w"), indent=1)
print("wrote profiles/r3_gram_ceiling.json")
PY
cp profiles/r3_gram_ceiling.json gpurun_out/r3/
