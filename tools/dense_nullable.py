"""Dense selections whose projected columns keep nulls: x > t -> [x, y, fn] (fn nullable), 5e8 rows, staged geometries against the
direct kernel's FF_OUTVALID instantiations (validity bits compacted with the rows)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

ctx = capi.Context(0)
n = 500_000_000
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
y = ctx.generate(synth_spec(RV_INT64, seed=46, length=n))
fn = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
xn = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
for label, cols, proj, mk in (("x > t -> [x, y, fn]", [x, y, fn], [0, 1, 2], lambda lit: Predicate([Term(0, ">", lit)])),
                              ("x > t -> [x, fn]", [x, fn], [0, 1], lambda lit: Predicate([Term(0, ">", lit)])),
                              ("xn > t (nulls least) -> [xn, fn]", [xn, fn], [0, 1], lambda lit: Predicate([Term(0, ">", lit)], "least"))):
    for lit in (899, 699, 499, 99):
        pred = mk(lit)
        line = f"{label:34s}"
        for direct in (-1, 0):
            ctx.set_option("direct", direct)
            best = 1e9
            for rep in range(4):
                ctx.synchronize()
                t0 = time.perf_counter()
                outs, rows, _ = ctx.filter_project(cols, pred, proj)
                ctx.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e3)
                [o.free() for o in outs]
            k = ctx.last_kernel()
            line += f" | {'staged' if direct < 0 else 'auto  '} {best:6.3f} ms {k[k.index('<'):]:22s}"
        print(f"{line}  sel {rows / n:.2f}", flush=True)
