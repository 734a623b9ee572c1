#!/usr/bin/env python3
"""What padded widths beyond 1024 (and the float64 overloads) cost: the feature operator, one CG iteration's matvec
(whatever route ConjugateGrad._matvec takes for the shape), z^T y and a rank-512 preconditioner build, at the
reference's own test shapes (tests/fht_operations_tests/test_rbf_rfgen.py:37,41: d = 2003 / M = 4000, d = 1076 /
M = 8192), at d = 4000 and -- for comparison -- at d = 1024, on `rows` datapoints.

    python tools/wide_path_probe.py [--rows 131072] [--out profiles/r6_generic_path.json] [--tag before]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xgpr_amd.kernels import make_kernel   # noqa: E402
from xgpr_amd.dataset import build_regression_dataset   # noqa: E402
from xgpr_amd.cg import ConjugateGrad, calc_zty   # noqa: E402
from xgpr_amd.preconditioner import RandNysPreconditioner   # noqa: E402
from xgpr_amd import xgpr_hip_rfgen_ext as ext   # noqa: E402
from xgpr_amd import _lib   # noqa: E402


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=131072)
    ap.add_argument("--out", default="")
    ap.add_argument("--tag", default="")
    ap.add_argument("--shapes", default="2003x4000,1076x8192,4000x8192,1024x8192")
    ap.add_argument("--no-build", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = args.rows
    res = {"rows": n, "tag": args.tag, "build_id": _lib.build_id(), "shapes": []}
    for shp in args.shapes.split(","):
        d, m = (int(t) for t in shp.split("x"))
        g = torch.Generator(device=dev).manual_seed(7)
        x = torch.randn(n, d, device=dev, generator=g) / np.sqrt(d)
        y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
        kern = make_kernel("RBF", (n, d), m, 123, dev, {})
        kern.set_hyperparams(np.array([0.1, 1.0]), logspace=False)
        ds = build_regression_dataset(x, y, chunk_size=16384, device=dev)
        row = {"d": d, "num_rffs": m, "padded_width": int(2 ** int(np.ceil(np.log2(max(d, 2))))),
               "tiles": int(n) * ((m // 2 + 1023) // 1024),
               "fused_ok": bool(kern.fused_ok()), "cache_ok": bool(kern.cache_ok()), "block_ok": bool(kern.block_ok()),
               "matvec_plan": ext.ztz_matvec_plan(d, m // 2)}
        # feature operator, float32 input -> float64 rows (hipRBFFeatureGen), in windows of 32768 rows
        z = torch.empty(32768, m, dtype=torch.float64, device=dev)
        xs = ds.scaled_x(kern.hyperparams[1])

        def featgen():
            for lo in range(0, n, 32768):
                ext.hipRBFFeatureGen(xs[lo:lo + 32768], z[:min(32768, n - lo)], kern.radem_diag, kern.chi_arr, True)
        row["featgen_f32_ms"] = timed(featgen)
        row["featgen_f32_checksum"] = float(z[:min(32768, n)].sum())
        del z
        # float64 input -> float64 rows (the double_precision=True operator, kernel_baseclass.py:278-285)
        nd = min(n, 32768)
        xd = xs[:nd].to(torch.float64)
        chid = kern.chi_arr.to(torch.float64)
        zd = torch.empty(nd, m, dtype=torch.float64, device=dev)
        row["featgen_f64_rows"] = nd
        row["featgen_f64_ms"] = timed(lambda: ext.hipRBFFeatureGen(xd, zd, kern.radem_diag, chid, True), reps=3)
        del xd, zd
        # one CG matvec (k = 1): the route _matvec takes
        cg = ConjugateGrad()
        vec = torch.randn(m, 1, dtype=torch.float64, device=dev, generator=g)
        out = torch.zeros_like(vec)
        row["cg_matvec_ms"] = timed(lambda: cg._matvec(ds, kern, vec, out), reps=3)
        row["cg_matvec_checksum"] = float(out.sum())
        row["cg_matvec_ns_per_tile"] = row["cg_matvec_ms"] * 1e6 / row["tiles"]
        # z^T y
        row["zty_ms"] = timed(lambda: calc_zty(ds, kern), reps=3)
        # k = 26 block matvec
        vecb = torch.randn(m, 26, dtype=torch.float64, device=dev, generator=g)
        outb = torch.zeros_like(vecb)
        row["cg_matvec_k26_ms"] = timed(lambda: cg._matvec(ds, kern, vecb, outb), reps=2)
        del cg
        if not args.no_build:
            def build():
                return RandNysPreconditioner(kern, ds, 512, False, 123, "srht")
            row["precond_build_rank512_ms"] = timed(build, reps=2)
            pre = build()
            row["precond_ratio"] = float(pre.achieved_ratio)
            del pre
        print(json.dumps(row), flush=True)
        res["shapes"].append(row)
        del x, y, ds, kern, xs
        torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
