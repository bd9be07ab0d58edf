"""Multi-column shapes over tables whose PREDICATE column is sorted / clustered, against the same shapes over independent rows (same
process, steady-state call time): config 3's AND of two compares over nullable columns, three plain columns, the nine-column frame.
    python3 tools/skew_shapes.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
ctx = capi.Context(0)


def timed(cols, pred, proj):
    for _ in range(2):
        outs, rows, _ = ctx.filter_project(cols, pred, proj)
        [o.free() for o in outs]
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        outs, rows, _ = ctx.filter_project(cols, pred, proj)
        [o.free() for o in outs]
    ctx.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3, rows, ctx.last_kernel(), ctx.get_option("last_redo_ppm")


y = ctx.generate(synth_spec(RV_INT64, seed=46, length=n))
z = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n))
xv = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
nullable = [ctx.generate(synth_spec(RV_INT64 if j % 2 else RV_FLOAT64, seed=50 + j, length=n // 2, validity_seed=70 + j)) for j in range(1, 5)]
base = {}
for name, kw in (("iid", {}), ("sorted", dict(pattern="sorted")), ("runs_1e5", dict(pattern="clustered", run_rows=100_000)), ("runs_1e3", dict(pattern="clustered", run_rows=1_000))):
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, **kw))
    f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44, **kw))
    shapes = [("config3: f > 0.5 and x < 200 -> [f, x]", [f, xv], Predicate([Term(0, ">", 0.5), Term(1, "<", 200)]), [0, 1]),
              ("x > 899 -> [x, y, z]", [x, y, z], Predicate([Term(0, ">", 899)]), [0, 1, 2]),
              ("x > 499 -> [x, y, z]", [x, y, z], Predicate([Term(0, ">", 499)]), [0, 1, 2]),
              ("x > 899 -> [x, 4 nullable] (half the rows)", [x.slice(0, n // 2)] + nullable, Predicate([Term(0, ">", 899)]), [0, 1, 2, 3, 4])]
    for label, cols, pred, proj in shapes:
        ms, rows, kern, redo = timed(cols, pred, proj)
        if name == "iid":
            base[label] = ms
        print(f"{name:9s} {label:48s} {ms:7.3f} ms = {ms / base[label]:5.2f} x iid  {kern:38s} redo {redo:7d} ppm kept {rows / cols[0].length:.3f}", flush=True)
    x.free(), f.free()
