"""The C ABI driven from plain C99 (-m gpu): tests/c/abi_check.c is compiled with gcc -std=c99 against
include/rivulus_gpu.h, linked with librivulus_gpu.so and run -- one rv_filter_project through the boundary the way
a foreign (Rust / C) caller makes it, checked against the generator's arithmetic restated in C."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c99_caller_runs_a_query_through_the_abi(tmp_path):
    exe = str(tmp_path / "abi_check")
    lib = os.path.join(ROOT, "rivulus_amd", "csrc")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "c", "abi_check.c"),
                        "-L" + lib, "-lrivulus_gpu", "-Wl,-rpath," + lib, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok abi_check"), r.stdout + r.stderr
