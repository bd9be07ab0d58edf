"""Static check of the fused kernels' tile loops (no GPU needed): compiles the instantiation units to gfx950 assembly and
reports, per kernel, what the round-2 overlap bugs looked like in the ISA:
  * scratch reloads inside the tile loop, away from the out-of-line look-back call (a spilled loop-invariant value: its
    reload is followed by `s_waitcnt vmcnt(0)`, i.e. by a wait for the whole prefetch of the next tile);
  * `s_waitcnt vmcnt(0)` between the last prefetch load and the look-back call / the end of the loop.
Usage:  python tools/isa_scan.py [unit ...]        (default: every fused_*.hip unit; ~1 min)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rivulus_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
units = sys.argv[1:] or ["fused_lean1", "fused_valid1", "fused_multi", "fused_bool", "fused_full", "fused_expr"]
flagged = 0
with tempfile.TemporaryDirectory() as tmp:
    procs = []
    for u in units:
        out = os.path.join(tmp, u + ".s")
        procs.append((u, out, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                                                 "-o", out, os.path.join(CSRC, u + ".hip")], stderr=subprocess.DEVNULL)))
    for u, out, pr in procs:
        pr.wait()
        lines = open(out).read().split("\n")
        for st in [i for i, l in enumerate(lines) if re.match(r"^_ZN3rvk20fused_filter_compact[^ ]*:", l)]:
            name = re.search(r"compactI(.*?)EEEvNS", lines[st]).group(1).replace("Li", "").replace("E", ",")
            end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
            body = lines[st:end]
            try:
                lp = next(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l)
                ep = next(i for i, l in enumerate(body) if "s_endpgm" in l and i > lp)
            except StopIteration:
                continue
            loop = body[lp:ep]
            calls = [i for i, l in enumerate(loop) if "s_swappc_b64" in l and "lookback" in "".join(loop[max(0, i - 12):i])]
            loads = [i for i, l in enumerate(loop) if re.search(r"buffer_load_dwordx[24] ", l)]
            last_load = max(loads) if loads else 0
            limit = min([c for c in calls if c > last_load], default=len(loop)) - 80
            reloads = [i for i, l in enumerate(loop) if "scratch_load" in l and i > last_load and all(abs(i - c) > 120 for c in calls)]
            waits = [i for i, l in enumerate(loop) if "s_waitcnt vmcnt(0)" in l and last_load < i < limit]
            scratch = sum("scratch_" in l for l in body)
            mark = "  <-- look" if reloads or waits else ""
            flagged += bool(reloads or waits)
            print(f"{u:13s} <{name:18s}> loop {len(loop):5d} lines, scratch ops {scratch:3d}; behind the prefetch: {len(reloads)} spill reloads, "
                  f"{len(waits)} full waits{mark}")
print(f"{flagged} kernels to look at")
