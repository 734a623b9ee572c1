"""pytest configuration: registers the ``gpu`` marker and shared fixtures.

``-m "not gpu"`` covers the oracle (against the golden vectors and, where the
reference's compiled core is present, against that), the host logic, and that
the C-ABI library loads and exports every symbol include/xgpr_hip.h declares.
``-m gpu`` tests are the parity tests proper: HIP path (through the C-ABI) vs
oracle / golden vectors.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build(ref=os.path.isdir("/root/reference"))
    return orc.Oracle()


@pytest.fixture(scope="session")
def refcore():
    from oracle import oracle as orc
    if os.path.isdir("/root/reference"):
        orc.build(ref=True)
    if not orc.RefCore.available():
        pytest.skip("reference core not built (no /root/reference on this box)")
    return orc.RefCore()


# Development aid: XGPR_POISON_EMPTY=1 fills every tensor that torch.empty hands out on the GPU with 0xFF bytes (NaN as
# float32 / float64, -1 as integers) -- workspaces and output buffers alike -- so that a kernel that reads memory nobody
# wrote, or leaves part of its output unwritten, turns a test red instead of passing on whatever the allocator returned.
if os.environ.get("XGPR_POISON_EMPTY") == "1":
    import torch as _torch
    _orig_empty = _torch.empty

    def _poisoned_empty(*args, **kwargs):
        t = _orig_empty(*args, **kwargs)
        if t.is_cuda and t.numel() > 0 and t.dtype in (_torch.uint8, _torch.float32, _torch.float64, _torch.int32, _torch.int64):
            t.view(_torch.uint8).fill_(255) if t.is_contiguous() else None
        return t
    _torch.empty = _poisoned_empty
