"""The skew rule (thresholds.hpp: kRedoMsPerShare, kDirectPenalty*) against both of its alternatives: x > t -> [x] over 1e9 rows at
10-30 % kept, independent rows and runs of 1e3 / 1e5 rows -- the staged pass + redo kernel (skew = -1), the direct kernel (direct = 1) and
what the library picks.      python3 tools/skew_rule_check.py          -> profiles/r05d_skew_rule_check.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec
n = 1_000_000_000
ctx = capi.Context(0)
for name, kw in [("iid", {}), ("runs_1e3", dict(pattern="clustered", run_rows=1000)), ("runs_1e5", dict(pattern="clustered", run_rows=100_000))]:
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, **kw))
    for lit in (899, 849, 799, 749, 699):
        pred = Predicate([Term(0, ">", lit)])
        out = []
        for label, opts in (("staged+redo", dict(skew=-1, direct=-1)), ("direct", dict(direct=1)), ("auto", {})):
            for k, v in opts.items():
                ctx.set_option(k, v)
            for _ in range(2):
                outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
            ctx.synchronize()
            out.append(f"{label} {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms {ctx.last_kernel()[6:30]} redo {ctx.get_option('last_redo_ppm')}")
            for k in opts:
                ctx.set_option(k, 0)
        print(f"{name:9s} keep {rows / n:.3f}: " + " | ".join(out), flush=True)
    x.free()
