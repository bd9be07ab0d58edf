"""C++ host layer (rivulus_amd/host/rivulus_host.hpp): the mirror of the reference's RecordBatch /
DataStream / planner / eager-plan interfaces over the C ABI.  The cases live in
tests/cpp/host_tests.cpp (they re-express the reference's unit tests); every case is one pytest item."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "rivulus_amd", "host")
BIN = os.path.join(HOST, "host_tests")
SRC = open(os.path.join(ROOT, "tests", "cpp", "host_tests.cpp")).read()
CPU_CASES = re.findall(r"^CPU_TEST\((\w+)\)", SRC, re.M)
GPU_CASES = re.findall(r"^GPU_TEST\((\w+)\)", SRC, re.M)
_cache = {}


def _run(cpu_only: bool):
    if cpu_only not in _cache:
        subprocess.run(["make", "-C", os.path.join(ROOT, "rivulus_amd", "csrc"), "-j8"], check=True, stdout=subprocess.DEVNULL)
        subprocess.run(["make", "-C", HOST], check=True, stdout=subprocess.DEVNULL)
        _cache[cpu_only] = subprocess.run([BIN] + (["--cpu"] if cpu_only else []), capture_output=True, text=True, timeout=600)
    return _cache[cpu_only]


def _assert_case(result, case):
    for line in result.stdout.splitlines():
        if line.split()[1:2] == [case] or line.startswith(f"FAIL {case}:"):
            assert line.startswith("ok "), line
            return
    pytest.fail(f"case {case} produced no line; stderr: {result.stderr[-500:]}")


@pytest.mark.parametrize("case", CPU_CASES)
def test_host_logic(case):
    _assert_case(_run(True), case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES)
def test_host_layer_on_device(case):
    _assert_case(_run(False), case)
