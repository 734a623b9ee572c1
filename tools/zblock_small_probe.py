"""Short launches of the block projection T = Zc V (zblock_t_kernel) and of the block matvec: time against rows.
    python tools/zblock_small_probe.py [out.json]       (XGPR_ZB_SPLIT=1 in the environment: the unsplit kernel; =n: n splits)
M = 8192 features, k = 26 right-hand sides (the approximate NMLL's block) and k = 10 (a classifier's classes)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext

dev = "cuda"
m = 8192
g = torch.Generator(device=dev).manual_seed(2)
res = {"num_rffs": m, "forced_split": os.environ.get("XGPR_ZB_SPLIT"), "project": [], "block_matvec": []}


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for k in (26, 10):
    v = torch.randn((m, k), dtype=torch.float64, device=dev, generator=g)
    for n in (2000, 2048, 4096, 8192, 16384, 32768, 65536, 131072):
        zc = torch.rand((n, m), dtype=torch.float32, device=dev, generator=g) * 2 - 1
        t = torch.empty((n, k), dtype=torch.float64, device=dev)
        need = ext.zcache_block_project_workspace_bytes(n, m, k)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        us = timed(lambda: ext.hipZCacheBlockProject(zc, v, t, True, 0.0, ws))
        flops = 2.0 * n * m * k
        ent = {"rows": n, "k": k, "us": round(us, 1), "useful_tflops": round(flops / us / 1e6, 2), "workspace_bytes": need}
        res["project"].append(ent)
        print("project", json.dumps(ent))
        if k == 26:
            w = torch.empty((m, k), dtype=torch.float64, device=dev)
            bws = torch.empty(ext.zcache_block_workspace_bytes(n, m, k), dtype=torch.uint8, device=dev)
            us = timed(lambda: ext.hipZCacheBlockMatvec(zc, v, w, True, bws))
            ent = {"rows": n, "k": k, "us": round(us, 1), "useful_tflops": round(2 * flops / us / 1e6, 2)}
            res["block_matvec"].append(ent)
            print("block_matvec", json.dumps(ent))
        del zc
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
