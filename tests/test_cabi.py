"""CPU-only checks of the drop-in boundary: libxgpr_hip.so builds for gfx950, loads, and
exports every symbol include/xgpr_hip.h declares (no compute calls -- there is no GPU here);
the ctypes table in xgpr_amd/_lib.py covers the same set; the product never imports oracle/."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "xgpr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(xgpr_[a-z0-9_]+)\s*\(", hdr)))


def test_library_builds_and_exports_header_symbols():
    subprocess.run([sys.executable, os.path.join(ROOT, "xgpr_amd", "build.py")], check=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "xgpr_amd", "libxgpr_hip.so"))
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/xgpr_hip.h but not exported"


def test_ctypes_table_matches_header():
    from xgpr_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.SIZE_FUNCS) | set(_lib.STRING_FUNCS)
    assert bound == set(_declared())
    lib = _lib.load()
    assert lib.xgpr_build_arch() == b"gfx950"
    assert lib.xgpr_rbf_workspace_bytes(4096) >= 3 * 4096 // 8


def test_gfx950_code_object_present():
    """The shared library embeds a gfx950 code object (hipcc --offload-arch=gfx950)."""
    blob = open(os.path.join(ROOT, "xgpr_amd", "libxgpr_hip.so"), "rb").read()
    assert b"gfx950" in blob


def test_product_does_not_reach_into_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "xgpr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("test oracle", ""), f"{f} mentions the oracle"


def test_every_entry_point_is_mapped_in_integration_md():
    """INTEGRATION.md maps each exported function (by name, or the *_workspace_bytes family by wildcard) to the
    reference interface it replaces."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "xgpr_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(xgpr_[a-z0-9_]+)\s*\(", header)))
    assert len(names) > 30
    for name in names:
        base = re.sub(r"_f(32|64)$", "", name)
        if base.endswith("_workspace_bytes"):
            assert "xgpr_*_workspace_bytes" in doc or base in doc
        else:
            assert base in doc, f"{name} is not mentioned in INTEGRATION.md"
