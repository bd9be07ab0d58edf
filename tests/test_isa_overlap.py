"""Static guard for the prefetch overlap of the BASELINE kernels (CPU: hipcc cross-compiles to gfx950 assembly, no GPU).

Round 2 found that the tile loop's flush could not run under the next tile's prefetch: spilled loop-invariant addresses were
reloaded with `s_waitcnt vmcnt(0)` right after the prefetch had been issued (profiles/README.md).  tools/isa_scan.py looks for
that pattern in the ISA; the kernels of BASELINE configs 2 and 3 must stay free of it."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def scan():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_scan.py"), "fused_lean1", "fused_multi"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = {}
    for line in r.stdout.splitlines():
        m = re.match(r"^(\w+)\s+<([\d,]+)\s*>.*behind the prefetch: (\d+) spill reloads, (\d+) full waits", line)
        if m:
            rows[m.group(2).rstrip(",")] = (int(m.group(3)), int(m.group(4)))
    return rows


@pytest.mark.parametrize("kernel", ["1,16,2,16,32", "1,16,2,16,64", "1,16,2,16,40", "2,16,1,16,385", "2,8,1,16,385", "2,8,2,16,0"])
def test_nothing_waits_behind_the_prefetch(scan, kernel):
    assert kernel in scan, f"instantiation <{kernel}> not found in the assembly (have: {sorted(scan)[:8]} ...)"
    assert scan[kernel] == (0, 0), f"<{kernel}>: {scan[kernel][0]} spill reloads / {scan[kernel][1]} full vmcnt waits behind the prefetch"
