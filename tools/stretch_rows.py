"""x > t -> [x] over SORTED tables of 4e7 .. 5e8 rows: filtered stretch by stretch (option segments = 1) against one pass (-1) and
against independent rows -- where the ~55 us a stretch more costs are earned back (thresholds.hpp, kStretchFromRows).
    python3 tools/stretch_rows.py          -> profiles/r05d_stretches_by_rows.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec
ctx = capi.Context(0)
for n in (40_000_000, 70_000_000, 130_000_000, 270_000_000, 500_000_000):
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, pattern="sorted"))
    y = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    for lit in (899, 499, 159):
        pred = Predicate([Term(0, ">", lit)])
        res = {}
        for label, col, seg in (("iid", y, 0), ("sorted stretches", x, 1), ("sorted one pass", x, -1)):
            ctx.set_option("segments", seg)
            for _ in range(3):
                outs, rows, _ = ctx.filter_project([col], pred, [0]); [o.free() for o in outs]
            ctx.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(10):
                    outs, rows, _ = ctx.filter_project([col], pred, [0]); [o.free() for o in outs]
                ctx.synchronize()
                best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
            res[label] = (best, ctx.last_kernel()[:24])
        ctx.set_option("segments", 0)
        print(f"n {n:>11d} x > {lit}: " + " | ".join(f"{k} {v[0]:.3f} ms ({v[1]})" for k, v in res.items()), flush=True)
    x.free(); y.free()
