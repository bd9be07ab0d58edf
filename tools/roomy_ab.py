"""Dense selections on a multi-column shape: the default geometry + redo kernel against 'roomy' geometries whose LDS slots hold
every row of a wave (option "roomy" = 1: 144 KiB, two stages).  Wall time per call, 5e8 rows, x > lit -> [x, y, fn]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

n = 500_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
fn = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
y = ctx.generate(synth_spec(RV_INT64, seed=46, length=n))
geoms = [("auto", 0, 0), ("R=12 W=16", 12 | 16 << 8, 0), ("R=4 W=16 roomy", 4 | 16 << 8, 1), ("R=4 W=8 roomy", 4 | 8 << 8, 1)]
for lit in (899, 799, 699, 499, 99):
    pred = Predicate([Term(0, ">", lit)])
    for name, rpl, roomy in geoms:
        ctx.set_option("rows_per_lane", rpl)
        ctx.set_option("roomy", roomy)
        best = 1e9
        for rep in range(4):  # "auto" meets a new selectivity unprepared on its first call and adapts on the next
            ctx.synchronize()
            t0 = time.perf_counter()
            outs, rows, s = ctx.filter_project([x, y, fn], pred, [0, 1, 2])
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
            [o.free() for o in outs]
        print(f"sel {rows / n:4.2f}  {name:16s} {ctx.last_kernel():38s} call {best:7.3f} ms   redone {ctx.get_option('last_redo_ppm') / 1e4:5.1f} %", flush=True)

# the same walk over the other multi-column shapes, automatic geometry only
xn = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
ctx.set_option("rows_per_lane", 0)
ctx.set_option("roomy", 0)
for label, cols, mk, proj in [
    ("fn > t AND xn < 950 -> [fn, xn]", [fn, xn], lambda t: [Term(0, ">", t), Term(1, "<", 950)], [0, 1]),
    ("x > t -> [y]", [x, y], lambda t: [Term(0, ">", int(1000 * t) - 1)], [1]),
    ("x > t -> [x, y, z]", [x, y, x], lambda t: [Term(0, ">", int(1000 * t) - 1)], [0, 1, 2]),
    ("x > t -> [x, fn]", [x, fn], lambda t: [Term(0, ">", int(1000 * t) - 1)], [0, 1]),
    ("x > t -> [xn, fn]", [x, xn, fn], lambda t: [Term(0, ">", int(1000 * t) - 1)], [1, 2]),
    ("x > t -> [x, y, fn, xn]", [x, y, fn, xn], lambda t: [Term(0, ">", int(1000 * t) - 1)], [0, 1, 2, 3]),
]:
    for t in (0.9, 0.85, 0.8, 0.7, 0.5, 0.1):
        pred = Predicate(mk(t))
        ctx.set_option("direct", -1)  # the staged geometries only
        times = []
        for rep in range(4):
            ctx.synchronize()
            t0 = time.perf_counter()
            outs, rows, s = ctx.filter_project(cols, pred, proj)
            ctx.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
            [o.free() for o in outs]
        ctx.set_option("direct", 1)  # the unstaged kernel, where the launch is eligible
        forced = []
        for rep in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            outs, rows, s = ctx.filter_project(cols, pred, proj)
            ctx.synchronize()
            forced.append((time.perf_counter() - t0) * 1e3)
            [o.free() for o in outs]
        dk = ctx.last_kernel()
        ctx.set_option("direct", 0)
        print(f"[direct {min(forced):6.3f} {dk[:22]}] ", end="")
        print(f"{label:34s} sel {rows / n:4.2f}  first call {times[0]:7.3f} ms, then {min(times[1:]):7.3f}  {ctx.last_kernel():38s} redone {ctx.get_option('last_redo_ppm') / 1e4:5.1f} %", flush=True)
