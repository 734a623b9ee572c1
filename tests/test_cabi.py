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


def _build_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("xgpr_amd_build_t", os.path.join(ROOT, "xgpr_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_library_identifies_its_sources():
    """xgpr_build_id() == sha256 of csrc/*, include/xgpr_hip.h and the flag list of THIS tree (build.py source_id)."""
    from xgpr_amd import _lib
    bm = _build_module()
    bm.build_extension()
    assert bm.built_id(bm.LIB) == bm.source_id()
    assert _lib.build_id() == bm.source_id() and len(_lib.build_id()) == 64


def test_build_rebuilds_on_source_mismatch_not_on_mtime(tmp_path, monkeypatch):
    """A library built from OTHER sources is rebuilt even when it is newer than every source file (a checkout of older
    sources); one built from these sources is kept even when a source file's mtime is newer.  hipcc is replaced by a
    recorder: the decision is what is under test."""
    bm = _build_module()
    calls = []

    def fake_compile(out, extra_flags, verbose=False):
        calls.append(out)
        return out
    monkeypatch.setattr(bm, "_compile", fake_compile)
    lib = tmp_path / "libxgpr_hip.so"
    monkeypatch.setattr(bm, "LIB", str(lib))
    lib.write_bytes(b"\x7fELF....xgpr-build-id:" + b"0" * 64 + b"\0")          # newer than all sources, other id
    bm.build_extension()
    assert calls == [str(lib)]
    lib.write_bytes(b"\x7fELF....xgpr-build-id:" + bm.source_id().encode() + b"\0")
    os.utime(lib, (1, 1))                                                        # older than all sources, same id
    bm.build_extension()
    assert calls == [str(lib)]
    lib.write_bytes(b"\x7fELF....no marker")
    bm.build_extension()
    assert len(calls) == 2


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


def test_host_side_plans_and_workspaces_need_no_gpu():
    """The launcher's shape predicates are host code: which plan the fused matvec takes by padded width and frequency count (round 6:
    padded widths 2048 / 4096 on the wave-tile kernels), and workspace sizes that cover every plan of a shape (the two-pass row-window
    buffer wherever ANY padded width could take two passes)."""
    from xgpr_amd import _lib
    lib = _lib.load()
    plan = lambda d, f: int(lib.xgpr_ztz_matvec_plan(d, f))
    # one pass on the three-wave kernel / two-wave kernel / two passes / unsupported
    assert [plan(1024, 4096), plan(256, 1024), plan(1024, 8192), plan(512, 16384)] == [1, 2, 3, 3]
    assert [plan(2000, 2000), plan(1025, 100), plan(4000, 4096), plan(2048, 4096), plan(3000, 1)] == [1, 1, 1, 1, 1]
    assert [plan(2000, 4097), plan(4000, 8192), plan(4096, 65536)] == [3, 3, 3]
    assert [plan(5000, 4096), plan(4097, 10), plan(100, 65537), plan(100, 0)] == [0, 0, 0, 0]
    ws = lambda m, r: int(lib.xgpr_ztz_matvec_workspace_bytes(m, r))
    slabs = lambda m: 2048 * m * 8
    # up to 4096 frequencies no padded width takes two passes; beyond it the wide transforms do, beyond 7168 every width does
    assert ws(8192, 4096) < slabs(8192) + (1 << 20)
    assert ws(8194, 8192) >= slabs(8194) + 131072 * 5 * 8
    assert ws(16384, 8192) >= slabs(16384) + 131072 * 8 * 8
    # monotone in the feature count
    sizes = [ws(m, 65536) for m in (2, 4096, 8192, 8194, 16384, 32768, 131072)]
    assert sizes == sorted(sizes)
