#!/usr/bin/env python3
"""Short launches of the float64 matrix-core kernels for a NON-saturated SQ_VALU_MFMA_BUSY_CYCLES reading (the counter is a
sum of per-XCD counters that stop near 2^29..2^32: launches of 2 ms and more read 0xE0000000 / 2^35 -- profiles/r5_mfma_clock.json):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- python tools/mfma_busy_probe.py
the preconditioner contraction (sketch_gemm_lds_kernel) on 8192 and 4096 rows x 8192 x rank 512, the block matvec's two
contractions (zblock_t_kernel / zblock_w_kernel, k = 26) on 65536 and 32768 rows, one Gram window (gram_lds_kernel) of 1024
rows.  tools/summarize_mfma_busy.py turns the pass into profiles/r6_mfma_clock.json."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
M, R, K = 8192, 512, 26
zc = torch.rand(65536, M, device=dev, generator=g) * 2 - 1
REPS = 6
for n in (8192, 4096):
    a = torch.randn(n, R, dtype=torch.float64, device=dev, generator=g)
    out = torch.zeros(R, M, dtype=torch.float64, device=dev)
    ws = torch.empty(ext.sketch_gemm_workspace_bytes(R, M, n, M, False), dtype=torch.uint8, device=dev)
    for _ in range(REPS):
        ext.hipSketchGemm(a, zc[:n], out, R, False, False, True, 0.0, accumulate=True, workspace=ws)
    torch.cuda.synchronize()
V = torch.randn(M, K, dtype=torch.float64, device=dev, generator=g)
W = torch.empty_like(V)
for n in (65536, 32768):
    ws = torch.empty(ext.zcache_block_workspace_bytes(n, M, K), dtype=torch.uint8, device=dev)
    for _ in range(REPS):
        ext.hipZCacheBlockMatvec(zc[:n], V, W, True, ws)
    torch.cuda.synchronize()
out = torch.zeros(M, M, dtype=torch.float64, device=dev)
wsg = ext.hipZtZGram(zc[:1024], out, True, 0.0)
for _ in range(REPS):
    ext.hipZtZGram(zc[:1024], out, True, 0.0, accumulate=True, workspace=wsg)
torch.cuda.synchronize()
print("done")
