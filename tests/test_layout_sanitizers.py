"""AddressSanitizer + UBSan over the host-side layout builder (CPU build only: GPU ASan is not
available on the pool).  Compiles tests/native/layout_sanitize.cpp with g++ and runs it."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_layout_builder_under_asan_ubsan(tmp_path):
    exe = tmp_path / "layout_sanitize"
    src = os.path.join(ROOT, "tests", "native", "layout_sanitize.cpp")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=all", "-pthread", "-o", str(exe), src],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "layout sanitize: ok" in run.stdout
