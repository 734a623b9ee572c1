"""Randomised shapes through the operators, the fused matvec, z^T y, the block kernels and the convolution operator
against the oracle (tools/stress_parity.py, fixed seeds): the edge cases a hand-picked shape list does not think of --
ragged last tiles, one-row inputs, widths below 64, right-hand-side counts between the tile sizes."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_against_the_oracle(seed, monkeypatch):
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_parity.py")
    monkeypatch.setattr(sys, "argv", [tool, "10", str(seed)])
    runpy.run_path(tool, run_name="__main__")          # asserts inside; raises on the first failing case


@pytest.mark.parametrize("seed", [21, 22])
def test_random_shapes_of_the_remaining_operators(seed, monkeypatch):
    """tools/stress_parity2.py: FHT / SRHT (bit for bit), the gradient and max-pool operators, SRHT-free dense products on
    float32 rows (sketch GEMM both ways, Gram), the preconditioner apply and the CG step kernels (one column and block),
    random shapes, every fourth case a chip-filling launch."""
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_parity2.py")
    monkeypatch.setattr(sys, "argv", [tool, "6", str(seed)])
    runpy.run_path(tool, run_name="__main__")
