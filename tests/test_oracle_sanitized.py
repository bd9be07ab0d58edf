"""The oracle's known-answer cases again, built with AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: the GPU
pool has no sanitizer runs).  The checker the parity tests lean on must not itself read out of bounds or overflow."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_kat_cases_are_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "kat_tests_san")
    build = subprocess.run(["g++", "-O0", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                            "-o", exe, os.path.join(ROOT, "oracle", "kat_tests.cpp")], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, cwd=os.path.join(ROOT, "oracle"))
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert "0 failed" in run.stdout.splitlines()[-1]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
