"""k = 1 matvec: regenerating the features against streaming the resident float32 cache, at shapes each plan of
xgpr_ztz_matvec_plan serves -- the numbers behind SORFKernel.cache_pays() (cache_features="auto").
    python tools/cache_rule_probe.py [out.json] [--rows=N]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from xgpr_amd.kernels import make_kernel
from xgpr_amd import xgpr_hip_rfgen_ext as ext

dev = "cuda"
shapes = [(1024, 8192), (256, 4096),                 # plan 1 (cfg3, cfg2)
          (256, 2048), (1024, 2048),                 # one tile per datapoint
          (512, 10240), (512, 14336), (1024, 16384), # 5, 7, 8 tiles
          (64, 4096), (20, 2048), (32, 8192),        # padded width < 128
          (1022, 8192),                              # d % 4 != 0
          (512, 32768),                              # two passes
          (64, 8192), (16, 8192),                    # tabular widths on the three-wave kernel (round 5)
          (2003, 4000), (1076, 8192), (4000, 8192)]  # wide transforms (round 6)
args = [a for a in sys.argv[1:] if not a.startswith("--rows=")]
n = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--rows=")), 131072)
sys.argv = sys.argv[:1] + args
res = []
g = torch.Generator(device=dev).manual_seed(1)
for d, m in shapes:
    x = torch.randn(n, d, device=dev, generator=g) / d ** 0.5
    kern = make_kernel("RBF", (n, d), m, 123, dev, {})
    kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
    v = torch.randn(m, dtype=torch.float64, device=dev, generator=g)
    w = torch.empty_like(v)
    ws = torch.zeros(kern.workspace_bytes(), dtype=torch.uint8, device=dev)
    zc = torch.empty((n, m), dtype=torch.float32, device=dev)
    ext.hipRBFFeatureCache(x, zc, kern.radem_diag, kern.chi_arr)

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_gen = timed(lambda: kern.ztz_matvec(x, v, w, ws))
    wg = w.clone()
    t_cache = timed(lambda: kern.ztz_matvec_cached(zc, v, w, ws))
    rel = float((w - wg).abs().max() / wg.abs().max())
    tiles = n * ((m // 2 + 1023) // 1024)
    ent = {"d": d, "num_rffs": m, "plan": ext.ztz_matvec_plan(d, m // 2), "cache_pays": bool(kern.cache_pays()),
           "regenerate_ms": round(t_gen, 4), "cached_ms": round(t_cache, 4), "regenerate_ns_per_tile": round(t_gen * 1e6 / tiles, 3),
           "cached_ns_per_tile": round(t_cache * 1e6 / tiles, 3), "faster": "cache" if t_cache < t_gen else "regenerate", "rel_diff": rel}
    res.append(ent)
    print(json.dumps(ent))
    del zc, x
if len(sys.argv) > 1:
    json.dump({"rows": n, "shapes": res}, open(sys.argv[1], "w"), indent=1)
