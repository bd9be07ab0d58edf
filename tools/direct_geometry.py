"""The direct kernel over ONE loaded column, x > t -> [x] over 1e9 rows at 10-95 % kept: 12 against 16 rows per lane (6144- against
8192-row tiles), the geometry the rule picks (thresholds.hpp, kDirectTallBelow) and the launch the library chooses by itself.
    python3 tools/direct_geometry.py          -> profiles/r05d_direct_geometry.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec
n = 1_000_000_000
ctx = capi.Context(0)
ctx.set_option("segments", -1)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
for lit in (899, 699, 499, 399, 299, 199, 159, 49):
    pred = Predicate([Term(0, ">", lit)])
    for label, opts in (("R12x8", dict(direct=1, direct_r=12, direct_waves=8)), ("R16x8", dict(direct=1, direct_r=16, direct_waves=8)), ("direct, rule", dict(direct=1)), ("auto", {})):
        for k, v in opts.items():
            ctx.set_option(k, v)
        for _ in range(2):
            outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
        ctx.synchronize()
        print(f"x > {lit} {label:13s}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms {ctx.last_kernel()}", flush=True)
        for k in opts:
            ctx.set_option(k, 0)
