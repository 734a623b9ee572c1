"""CPU sanitizer job for the oracle's C restatement: oracle/xgpr_oracle.c is built with
``-fsanitize=address,undefined -fno-sanitize-recover=all`` (``make -C oracle asan``) and every golden-vector
check of tests/test_oracle_golden.py is replayed against that build in a child interpreter that preloads the
ASan runtime.  Any out-of-bounds access, misaligned access, signed overflow, ... in the restatement aborts the
child with a sanitizer report.  (GPU AddressSanitizer is not available on this pool: CPU build only.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _asan_runtime():
    gcc = shutil.which("gcc")
    if not gcc:
        return None
    path = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_c_under_asan_ubsan():
    runtime = _asan_runtime()
    if runtime is None:
        pytest.skip("gcc / libasan not available")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    lib = os.path.join(ROOT, "oracle", "_asan", "liboracle_asan.so")
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": runtime,
        # leak detection off: the interpreter itself "leaks" by design; everything else aborts the child
        "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1",
        "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
        "XGPR_ORACLE_LIB": lib,
        "OMP_NUM_THREADS": "4",
    })
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                          "-x", "-q", "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True,
                         timeout=900)
    report = res.stdout[-3000:] + res.stderr[-3000:]
    assert "AddressSanitizer" not in report and "runtime error" not in report, report
    assert res.returncode == 0, report
    assert " passed" in res.stdout, report
