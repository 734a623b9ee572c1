"""Pins, on the CPU oracle, the near-breakdown of the un-preconditioned CG solve of the g7 problem that
tests/test_gpu_cg.py relaxes its iterate bar for (reference loop: fitting_toolkit/cg_tools.py:255-287)."""
import numpy as np
import pytest

from conftest import load_golden
import cg_sensitivity as cs


@pytest.mark.parametrize("kname", ["RBF", "Matern"])
def test_unpreconditioned_iterates_near_breakdown(kname):
    g = load_golden("g7_cg.npz")
    z, y_chunks, lam, chunk = cs.oracle_problem(g, kname)
    # the oracle itself reproduces the reference's iterates (same float64 operations in the same order)
    errs, _ = cs.iterate_errors(g, kname, z, y_chunks, lam, chunk)
    assert errs.max() <= 1e-12, errs
    # a perturbation BELOW float32 rounding of the features (3e-8 relative) ...
    outside = [j for j in range(len(errs)) if j not in cs.WINDOW]
    worst_in = []
    for seed in range(5):
        rng = np.random.default_rng(seed)
        zp = z * (1.0 + 3e-8 * rng.uniform(-1.0, 1.0, size=z.shape))
        e, _ = cs.iterate_errors(g, kname, zp, y_chunks, lam, chunk)
        # ... leaves every iterate outside the window at the perturbation's own size
        assert e[outside].max() <= 3e-7, (seed, e)
        # ... moves the edges of the window by at most 1e-5
        assert max(e[cs.WINDOW[0]], e[cs.WINDOW[-1]]) <= 1e-5, (seed, e)
        # ... and moves the iterates inside it (8 and 9, 1-based) four orders of magnitude further
        assert e[7] >= 1e-4, (seed, e)
        assert e[list(cs.WINDOW)].max() <= 5e-2, (seed, e)
        worst_in.append(e[7])
    # the amplification is not proportional to the perturbation (a breakdown, not conditioning):
    # a 13x larger perturbation lands in the same range
    env, counts = cs.envelope(g, kname, 4e-7, seeds=range(3))
    assert env[outside].max() <= 4e-6
    assert 1e-4 <= env[7] <= 5e-2
    assert 0.05 < env[7] / max(worst_in) < 200
    # the full solve's iteration count moves by a few iterations under the same perturbations
    ref_n = int(g[f"{kname}_none_niter"])
    assert counts[0] - 3 <= ref_n <= counts[1] + 3, (counts, ref_n)
