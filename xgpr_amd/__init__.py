"""xgpr_amd -- MI355X-native hot path of xGPR (SORF random features + preconditioned CG).

Importing the package loads the HIP extension (libxgpr_hip.so); there is no CPU fallback.
"""
from . import _lib

_lib.load()

__all__ = ["_lib"]
