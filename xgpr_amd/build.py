"""Builds libxgpr_hip.so (the C-ABI library with the gfx950 kernels) in-tree with hipcc.

    python xgpr_amd/build.py [--force]      (run as a script: importing the package needs the built library)

hipcc cross-compiles for gfx950 without a GPU, so this runs in the authoring container;
the built .so travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "xgpr_hip.hip")
HDR = os.path.join(HERE, "..", "include", "xgpr_hip.h")
LIB = os.path.join(HERE, "libxgpr_hip.so")
# Roofline probe, NOT a product library: the same translation unit with -DXGPR_ABL_VALUONLY, in which the fused CG
# matvec keeps its vector instruction stream but has its LDS traffic, workgroup barrier and prefetch DMA compiled
# out (its results are meaningless).  bench.py times it beside the real kernel: the ratio is how much of the kernel's
# time its own vector instructions need at the kernel's occupancy.  It is built into tools/ (not into the product
# package); nothing under xgpr_amd/ loads it.
PROBE_LIB = os.path.join(HERE, "..", "tools", "libxgpr_hip_valuonly_probe.so")

# -ffp-contract=off: the butterflies / Rademacher multiplies must round like the reference's
# scalar code (see the header comment of csrc/xgpr_hip.hip).
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libxgpr_hip.so")


def sources():
    """xgpr_hip.hip is one translation unit that includes the .inc files next to it."""
    csrc = os.path.dirname(SRC)
    return [SRC, HDR] + sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".inc"))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in sources())


def build_probe(force=False):
    """Compile the VALU-only timing probe (see PROBE_LIB) if missing or older than its sources."""
    if not force and os.path.exists(PROBE_LIB) and all(os.path.getmtime(p) <= os.path.getmtime(PROBE_LIB) for p in sources()):
        return PROBE_LIB
    res = subprocess.run([hipcc_path()] + FLAGS + ["-DXGPR_ABL_VALUONLY", SRC, "-o", PROBE_LIB], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed (probe):\n" + res.stdout + res.stderr)
    return PROBE_LIB


def build_extension(force=False, verbose=False):
    """Compile csrc/xgpr_hip.hip -> libxgpr_hip.so if missing or older than its sources."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path()] + FLAGS + [SRC, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return LIB


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
    if "--probe" in sys.argv:
        print(build_probe(force="--force" in sys.argv))
