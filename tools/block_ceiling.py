#!/usr/bin/env python3
"""The 26-right-hand-side block matvec (zblock_t_kernel + zblock_w_kernel) against its own ceiling, by script:
    python tools/block_ceiling.py --build      (here: hipcc builds the two timing-only variants into tools/ablate/)
    python tools/block_ceiling.py              (on the GPU box: times the three builds in ONE process, interleaved, and
                                                writes profiles/r3_block_ceiling.json)
Ablations (results meaningless, timing only): no steady-state workgroup barrier (-DXGPR_ABL_ZB_NOBAR); the streamed
operand's HBM loads replaced by register constants (-DXGPR_ABL_ZB_NOLOAD).  The ceiling of the formulation: per element
of the float32 operand one v_mfma_f64_16x16x4_f64 (64 matrix-pipe cycles) + three v_mfma_f64_4x4x4_4b_f64 (17 each)
issue 28 columns of which 26 are useful, and its f32 -> f64 conversion costs the matrix pipe ~10 cycles
(tools/mfma_probe2.hip); at the clock the part holds under these kernels (profiles/r3_mfma_clock.json) that is
26/28 x 115/125 x clock/2.4 of the 78.6 TFLOP/s FP64 matrix peak."""
import ctypes as C, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"nobar": "-DXGPR_ABL_ZB_NOBAR", "noload": "-DXGPR_ABL_ZB_NOLOAD"}
if "--build" in sys.argv:
    os.makedirs(os.path.join(ROOT, "tools", "ablate"), exist_ok=True)
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", flag,
                               os.path.join(ROOT, "xgpr_amd/csrc/xgpr_hip.hip"), "-o", os.path.join(ROOT, "tools/ablate", f"lib_zb_{n}.so")],
                              stderr=subprocess.DEVNULL) for n, flag in VARIANTS.items()]
    sys.exit(max(p.wait() for p in procs))
sys.path.insert(0, ROOT)
import torch
from xgpr_amd import _lib
n, m, k = 262144, 8192, 26
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
zc = torch.rand(n, m, device=dev, generator=g) * 2 - 1
v = torch.randn(m, k, dtype=torch.float64, device=dev, generator=g)
w = torch.empty_like(v)
lib0 = _lib.load()
ws = torch.empty(int(lib0.xgpr_zcache_block_workspace_bytes(n, m, k)), dtype=torch.uint8, device=dev)
paths = {"shipped": _lib.LIB_PATH, **{name: os.path.join(ROOT, "tools/ablate", f"lib_zb_{name}.so") for name in VARIANTS}}
fns = {}
vp, l, i, d, sz = C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_size_t
for name, p in paths.items():
    if not os.path.exists(p):
        sys.exit(f"{p} missing: run tools/block_ceiling.py --build first")
    fn = C.CDLL(p).xgpr_zcache_block_matvec_f32
    fn.argtypes = [vp, vp, vp, l, l, l, i, d, i, vp, sz, vp]; fn.restype = C.c_int
    fns[name] = fn
def call(fn):
    rc = fn(zc.data_ptr(), v.data_ptr(), w.data_ptr(), n, m, k, 1, 0.0, 0, ws.data_ptr(), ws.numel(), 0)
    assert rc == 0
times = {name: [] for name in fns}
for fn in fns.values():
    for _ in range(2): call(fn)
for rnd in range(7):
    for name, fn in fns.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): call(fn)
        e1.record(); e1.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
ms = {name: statistics.median(t) for name, t in times.items()}
useful = 4.0 * n * m * k
clk = json.load(open(os.path.join(ROOT, "profiles", "r3_mfma_clock.json")))
clock = 0.5 * (clk["zblock_t_kernel"]["clock_GHz"] + clk["zblock_w_kernel"]["clock_GHz"])
ceiling = 26 / 28 * 115 / 125 * clock / 2.4
out = {"what": "block matvec W = Zc^T (Zc V), k = 26 right-hand sides, 262144 x 8192 float32 rows (cfg3 shape); median of 7 interleaved rounds",
       "ms": ms, "useful_TFLOPs": {name: useful / (t * 1e-3) / 1e12 for name, t in ms.items()},
       "useful_over_fp64_matrix_peak_78.6": {name: useful / (t * 1e-3) / 1e12 / 78.6 for name, t in ms.items()},
       "clock_under_load_GHz": clock, "clock_source": "profiles/r3_mfma_clock.json (GRBM_GUI_ACTIVE / 8 XCDs / duration, mean of the two kernels)",
       "ceiling_of_the_formulation": {"issued_columns": 28, "useful_columns": 26, "matrix_cycles_per_operand_element": 115,
                                      "conversion_cycles_per_operand_element": 10, "useful_over_peak_at_2.4GHz": 26 / 28 * 115 / 125,
                                      "useful_over_peak_at_clock_under_load": ceiling},
       "shipped_over_ceiling": useful / (ms["shipped"] * 1e-3) / 1e12 / 78.6 / ceiling,
       "reading": "no barrier: what the steady-state workgroup barrier costs; no load: what is left when the streamed operand never leaves registers "
                  "(matrix pipe + conversions + LDS operand reads only)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r3_block_ceiling.json"), "w"), indent=1)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)      # (the GPU box returns only gpurun_out/: copy it to profiles/ from there)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r3", "r3_block_ceiling.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
