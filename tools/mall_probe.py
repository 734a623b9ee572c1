"""Does the 256 MiB Infinity Cache pay for the block matvec's second read of the feature cache?  (round 4, item 2)
The pair T = Zc V (contract features), W = Zc^T T (contract datapoints) reads Zc twice.  Here the two contractions run as
separate calls (xgpr_zcache_block_project_f32 / _backproject_f32 = zblock_t_kernel / zblock_w_kernel + reduce) on row
windows of a 65 536-row cache (2 GiB, far beyond the Infinity Cache):
   cold: every call on a window nobody touched for 2 GiB of traffic
   warm: W on the window T has just read (what a row-blocked schedule T(b), W(b), T(b+1), ... would see)
    python tools/mall_probe.py [k]"""
import os, statistics, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xgpr_amd import xgpr_hip_rfgen_ext as ext
k = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n, m = 65536, 8192
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
zc = torch.rand(n, m, device=dev, generator=g) * 2 - 1
v = torch.randn(m, k, dtype=torch.float64, device=dev, generator=g)
out = {}
for rows in (2048, 4096, 6144, 8192, 16384):
    nw = n // rows
    t = torch.empty(rows, k, dtype=torch.float64, device=dev)
    w = torch.empty(m, k, dtype=torch.float64, device=dev)
    ws = torch.empty(ext.zcache_block_workspace_bytes(rows, m, k), dtype=torch.uint8, device=dev)
    def T(i): ext.hipZCacheBlockProject(zc[i * rows:(i + 1) * rows], v, t, True, 0.0)
    def W(i): ext.hipZCacheBlockBackproject(zc[i * rows:(i + 1) * rows], t, w, True, ws, 0.0)
    def timed(fn, reps):
        fn(0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps): fn(r)
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps
    reps = 2 * nw
    res = {}
    for rnd in range(5):
        res.setdefault("T_cold", []).append(timed(lambda r: T(r % nw), reps))
        res.setdefault("T_warm", []).append(timed(lambda r: T(0), reps))
        res.setdefault("W_cold", []).append(timed(lambda r: W(r % nw), reps))
        res.setdefault("W_warm", []).append(timed(lambda r: W(0), reps))
        res.setdefault("TW_blocked", []).append(timed(lambda r: (T(r % nw), W(r % nw)), reps))        # W reads what T just read
        res.setdefault("TW_apart", []).append(timed(lambda r: (T(r % nw), W((r + nw // 2) % nw)), reps))  # W reads a cold window
    med = {key: statistics.median(val) for key, val in res.items()}
    flop = 2.0 * rows * m * k
    out[rows] = {"window_MB": rows * m * 4 / 1e6, **{key: round(val * 1e3, 1) for key, val in med.items()},
                 "unit": "us per call (pair for TW_*)", "W_warm_over_cold": round(med["W_warm"] / med["W_cold"], 3),
                 "T_warm_over_cold": round(med["T_warm"] / med["T_cold"], 3),
                 "pair_blocked_over_apart": round(med["TW_blocked"] / med["TW_apart"], 3),
                 "pair_blocked_useful_TFLOPs": round(2 * flop / med["TW_blocked"] / 1e9, 1)}
    print(rows, out[rows], flush=True)
json.dump({"what": __doc__.split("\n")[0], "k": k, "cache": [n, m], "results": out}, open(sys.argv[2] if len(sys.argv) > 2 else "/dev/null", "w"), indent=1)
