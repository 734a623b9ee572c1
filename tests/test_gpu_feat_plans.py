"""The two plans behind the feature rows -- wave_rbf_kernel (one wave per datapoint tile) and the three-wave persistent
plan (ztz3_kernel Z3_FEAT64 / Z3_FEAT32: row by LDS-DMA one datapoint ahead, stores issued through the next datapoint's
transform) -- produce the same bits: both evaluate the reference's butterflies in the reference's order
(shared_rfgen_ops.cpp:51-114) and the same cos/sin.  The plan is chosen once per process (XGPR_FEAT_PLAN, a timing aid),
so each plan runs in a child process over the same seeded shapes and the parent compares what they wrote: the float64
operator output (hipRBFFeatureGen), the float32 cache rows (hipRBFFeatureCache), and both against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (rows, d, num_rffs): whole tiles, a ragged last tile, tile counts that do not divide 12 (5, 7: tile groups with idle
# waves), more than 8 tiles (groups on blockIdx.y), one row, fewer rows than slots, every padded width 128 .. 1024
SHAPES = [(700, 1024, 8192), (513, 256, 4096), (37, 128, 2048 + 2 * 300), (1, 512, 4096), (260, 300, 2 * 5000),
          (129, 700, 2 * 7168), (65, 512, 32768), (9, 1000, 2 * 9000), (300, 132, 6144),
          # padded widths 32 and 64 on the three-wave plan (round 5), rows that are not multiples of four floats
          (400, 64, 8192), (333, 32, 4096), (150, 50, 2 * 5000), (77, 33, 6144), (210, 1022, 8192)]

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from xgpr_amd import xgpr_hip_rfgen_ext as ext
from oracle import oracle as orc
shapes = %(shapes)r
out = {}
for i, (n, d, m) in enumerate(shapes):
    rng = np.random.default_rng(100 + i)
    x = (rng.standard_normal((n, d)) / np.sqrt(d) * (30.0 if i %% 3 == 2 else 1.0)).astype(np.float32)
    radem, chi = orc.draw_sorf_params(m, d, 7 + i)
    xt, rt, ct = (torch.from_numpy(a).cuda() for a in (x, radem, chi))
    z = torch.zeros((n, m), dtype=torch.float64, device="cuda")
    ext.hipRBFFeatureGen(xt, z, rt, ct, bool(i %% 2))
    zc = torch.zeros((n, m), dtype=torch.float32, device="cuda")
    ext.hipRBFFeatureCache(xt, zc, rt, ct)
    out["z%%d" %% i] = z.cpu().numpy()
    out["c%%d" %% i] = zc.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def _run(plan, path):
    env = dict(os.environ)
    env["XGPR_FEAT_PLAN"] = plan
    code = CHILD % {"root": ROOT, "shapes": SHAPES}
    res = subprocess.run([sys.executable, "-c", code, path], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return np.load(path)


def test_both_plans_write_the_same_bits_and_match_the_oracle(tmp_path, oracle):
    from oracle import oracle as orc
    wave = _run("wave", str(tmp_path / "wave.npz"))
    z3 = _run("z3", str(tmp_path / "z3.npz"))
    for i, (n, d, m) in enumerate(SHAPES):
        assert np.array_equal(wave[f"z{i}"], z3[f"z{i}"]), (i, n, d, m)
        assert np.array_equal(wave[f"c{i}"], z3[f"c{i}"]), (i, n, d, m)
        rng = np.random.default_rng(100 + i)
        x = (rng.standard_normal((n, d)) / np.sqrt(d) * (30.0 if i % 3 == 2 else 1.0)).astype(np.float32)
        radem, chi = orc.draw_sorf_params(m, d, 7 + i)
        icpt = bool(i % 2)
        ref = np.zeros((n, m))
        oracle.cpuRBFFeatureGen(x.copy(), ref, radem, chi, icpt)
        freqs = m // 2
        scale = np.sqrt(1.0 / (freqs - 0.5 if icpt else freqs))
        assert np.abs(z3[f"z{i}"] - ref).max() <= 4e-7 * scale, (i, n, d, m)
        # the cache rows are the float32 (cos, sin) before scaling: the operator's output is their widening times the constant
        assert np.array_equal(z3[f"c{i}"].astype(np.float64) * float(np.float32(scale)), z3[f"z{i}"]), (i, n, d, m)
