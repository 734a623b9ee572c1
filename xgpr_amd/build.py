"""Builds libxgpr_hip.so (the C-ABI library with the gfx950 kernels) in-tree with hipcc.

    python xgpr_amd/build.py [--force]      (run as a script: importing the package needs the built library)

hipcc cross-compiles for gfx950 without a GPU, so this runs in the authoring container;
the built .so travels to the GPU box with the repository snapshot.
"""
import hashlib
import os
import re
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


def source_id(extra_flags=()):
    """sha256 over what determines the binary: every file of the translation unit (csrc/*, include/xgpr_hip.h: name and
    contents, in sorted order) and the flag list.  Baked into the library at compile time (-DXGPR_BUILD_ID) and
    returned by xgpr_build_id(): a .so states which tree it was built from, and build() rebuilds on a mismatch --
    modification times are not consulted (a checkout of older sources leaves the .so NEWER than them)."""
    h = hashlib.sha256()
    for p in sorted(sources(), key=os.path.basename):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(FLAGS + list(extra_flags)).encode())
    return h.hexdigest()


def built_id(lib_path):
    """The id a built library carries (read from the file's bytes: loading it here would bind the system HIP runtime
    before torch's), or None."""
    if not os.path.exists(lib_path):
        return None
    with open(lib_path, "rb") as f:
        m = re.search(rb"xgpr-build-id:([0-9a-f]{64})", f.read())
    return m.group(1).decode() if m else None


def is_stale():
    return built_id(LIB) != source_id()


def _compile(out, extra_flags, verbose=False):
    sid = source_id(extra_flags)
    cmd = [hipcc_path()] + FLAGS + list(extra_flags) + ['-DXGPR_BUILD_ID="%s"' % sid, SRC, "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(out + ".tmp", out)          # a process that has the old file mapped keeps its copy
    return out


PROBE_FLAGS = ("-DXGPR_ABL_VALUONLY",)


def build_probe(force=False):
    """Compile the VALU-only timing probe (see PROBE_LIB) if missing or built from other sources."""
    if not force and built_id(PROBE_LIB) == source_id(PROBE_FLAGS):
        return PROBE_LIB
    return _compile(PROBE_LIB, PROBE_FLAGS)


def build_extension(force=False, verbose=False):
    """Compile csrc/xgpr_hip.hip -> libxgpr_hip.so if missing or built from other sources / flags."""
    if not force and not is_stale():
        return LIB
    return _compile(LIB, (), verbose)


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
    if "--probe" in sys.argv:
        print(build_probe(force="--force" in sys.argv))
