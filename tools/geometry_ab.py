"""Same-box A/B of launch geometries (rows per lane, load width) for the multi-column shapes: every (shape, geometry) pair is
timed in turn, three rounds, in ONE process on one device (box to box the same kernel differs by +-4 %).
    python3 tools/geometry_ab.py [crowd]     (crowd: only the selectivity walk at the end) prints kernel ms (HIP events inside the library) and the instantiation launched"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

n = 500_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
xn = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n))
fn = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
y = ctx.generate(synth_spec(RV_INT64, seed=46, length=n))
ctx.set_option("profile_kernels", 1)
z = ctx.generate(synth_spec(RV_INT64, seed=49, length=n))
shapes = [
    ("x > 899 -> [x, fn]", [x, fn], [Term(0, ">", 899)], [0, 1], 16.125),
    ("x > 899 -> [x, y, fn]", [x, y, fn], [Term(0, ">", 899)], [0, 1, 2], 24.125),
    ("x > 899 -> [x, y, fn, xn]", [x, y, fn, xn], [Term(0, ">", 899)], [0, 1, 2, 3], 32.25),
    ("x > 899 -> [xn, fn] (x not projected)", [x, xn, fn], [Term(0, ">", 899)], [1, 2], 24.25),
]
geometries = [(0, 0), (8, 1), (12, 1), (16, 1)]  # (rows per lane, load width); 0 = the library's default
for label, cols, terms, proj, bpr in ([] if sys.argv[1:] == ["crowd"] else shapes):
    pred = Predicate(terms)
    best = {}
    for rnd in range(3):
        for r, v in geometries:
            ctx.set_option("rows_per_lane", r)
            ctx.set_option("vec", v)
            outs, rows, s = ctx.filter_project(cols, pred, proj)
            [o.free() for o in outs]
            ctx.kernel_stats(reset=True)
            for rep in range(3):
                outs, rows, s = ctx.filter_project(cols, pred, proj)
                [o.free() for o in outs]
            ctx.synchronize()
            ms, k = ctx.kernel_stats()
            key = (r, v, ctx.last_kernel())
            best.setdefault(key, []).append(ms / 3)
    for (r, v, kern), t in best.items():
        print(f"{label:55s} R={r:2d} vec={v}  {kern:42s} " + " ".join(f"{q:6.3f}" for q in t) + f" ms   best {bpr * n / min(t) / 1e6 / 80:4.1f} %", flush=True)

# a selectivity that crowds the default geometry's LDS slots: the first call meets it unprepared (dense tiles go to the redo
# kernel), the following ones take the geometry with the fewest rows per lane
ctx.set_option("rows_per_lane", 0)
ctx.set_option("vec", 0)
for lit in (899, 799, 699, 499, 99, 899):
    pred = Predicate([Term(0, ">", lit)])
    for call in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        outs, rows, s = ctx.filter_project([x, y, fn], pred, [0, 1, 2])
        ctx.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        [o.free() for o in outs]
        print(f"x > {lit} -> [x, y, fn]  call {call}: {ctx.last_kernel():40s} call {wall:6.3f} ms, selectivity {rows / n:5.3f}, "
              f"tiles redone {ctx.get_option('last_redo_ppm') / 1e4:5.1f} %", flush=True)
