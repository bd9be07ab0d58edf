"""Kernel time of other query shapes on 5e8 synthetic rows (regression / coverage view; not bench lines)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_BOOLEAN, RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec

n = 500_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
xn = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n))
fn = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
y = ctx.generate(synth_spec(RV_INT64, seed=46, length=n))
b = ctx.generate(synth_spec(RV_BOOLEAN, seed=47, length=n, true_percent=10))
bn = ctx.generate(synth_spec(RV_BOOLEAN, seed=47, length=n, true_percent=10, validity_seed=48))
ctx.set_option("profile_kernels", 1)
shapes = [
    ("x > 899 -> [x]                      (config 2)", [x], [Term(0, ">", 899)], [0], 8.0),
    ("f <= 0.1 -> [f]                     (Float64)", [f], [Term(0, "<=", 0.1)], [0], 8.0),
    ("x > 899 -> [x], nullable x", [xn], [Term(0, ">", 899)], [0], 8.125),
    ("x > 899 AND x < 950 -> [x]          (two terms)", [x], [Term(0, ">", 899), Term(0, "<", 950)], [0], 8.0),
    ("x > 899 -> [y]                      (predicate column not projected)", [x, y], [Term(0, ">", 899)], [1], 16.0),
    ("x > 899 -> [x, y, f]                (three columns)", [x, y, f], [Term(0, ">", 899)], [0, 1, 2], 24.0),
    ("f > 0.5 AND x < 200 -> [f, x], nullable (config 3)", [fn, xn], [Term(0, ">", 0.5), Term(1, "<", 200)], [0, 1], 16.25),
    ("b is true -> [x]                    (Boolean predicate column)", [b, x], [Term(0, "is_true")], [1], 8.125),
    ("b is true -> [x, b], nullable b", [bn, x], [Term(0, "is_true")], [1, 0], 8.25),
    ("x > 899 -> [x] + selection bitmap", [x], [Term(0, ">", 899)], [0], 8.0),
]
if len(sys.argv) > 1:  # geometry override for experiments: R | waves << 8
    ctx.set_option("rows_per_lane", int(sys.argv[1], 0))
if len(sys.argv) > 2:
    ctx.set_option("vec", int(sys.argv[2]))
for label, cols, terms, proj, bpr in shapes:
    pred = Predicate(terms)
    sel = label.endswith("selection bitmap")
    for rep in range(2):
        outs, rows, s = ctx.filter_project(cols, pred, proj, sel); [o.free() for o in outs]; s and s.free()
    ctx.kernel_stats(reset=True)
    ctx.synchronize()
    import time
    t0 = time.perf_counter()
    for rep in range(3):
        outs, rows, s = ctx.filter_project(cols, pred, proj, sel); [o.free() for o in outs]; s and s.free()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / 3 * 1e3
    ms, k = ctx.kernel_stats()
    ms /= 3
    print(f"{label:62s} {ms:7.3f} ms  sel {rows/n:5.3f}  read {bpr*n/ms/1e6:7.0f} GB/s = {bpr*n/ms/1e6/80:4.1f} %  (call {wall:6.3f} ms)", flush=True)
