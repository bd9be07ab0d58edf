"""The CPU oracle against the reference's own known-answer vectors.

oracle/kat_tests.cpp re-expresses the reference's inline unit tests; every case is one
pytest item here so a regression names the reference test it breaks.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


def _run():
    subprocess.run(["make", "-C", ORACLE, "-j4"], check=True, stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ORACLE, "kat_tests")], capture_output=True, text=True)
    return out


_RESULT = None


def _result():
    global _RESULT
    if _RESULT is None:
        _RESULT = _run()
    return _RESULT


def _cases():
    lines = [l for l in _result().stdout.splitlines() if l.startswith(("ok ", "FAIL "))]
    return [l.split()[1].rstrip(":") for l in lines]


def test_kat_binary_passes():
    r = _result()
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 failed" in r.stdout
    assert len(_cases()) >= 70


@pytest.mark.parametrize("case", _cases())
def test_reference_known_answer(case):
    for line in _result().stdout.splitlines():
        if line.split()[1:2] == [case] or line.startswith(f"FAIL {case}:"):
            assert line.startswith("ok "), line
            return
    pytest.fail(f"case {case} produced no line")
