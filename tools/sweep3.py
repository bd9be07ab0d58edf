"""Geometry sweep of BASELINE config 3: (f > 0.5) AND (x < 200) over nullable Float64 + Int64 (tuning aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, RV_FLOAT64, Predicate, Term, synth_spec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
ctx.set_option("profile_kernels", 1)
for (r, w, vec, depth) in [(12, 16, 2, 0), (16, 16, 2, 0), (8, 16, 2, 0), (16, 16, 1, 0), (12, 16, 1, 0), (8, 16, 1, 0), (16, 16, 1, 1)]:
    ctx.set_option("rows_per_lane", r | (w << 8)); ctx.set_option("vec", vec); ctx.set_option("depth", depth)
    for dbg in (0,):
        ctx.set_option("debug", dbg)
        for rep in range(2):
            outs, rows, _ = ctx.filter_project([f, x], pred, [0, 1]); [o.free() for o in outs]
        ctx.kernel_stats(reset=True)
        for rep in range(3):
            outs, rows, _ = ctx.filter_project([f, x], pred, [0, 1]); [o.free() for o in outs]
        ms, k = ctx.kernel_stats()
        ms /= 3
        print(f"R={r:2d} W={w:2d} VEC={vec} depth={depth} debug={dbg}: {ms:8.3f} ms ({k//3} kernels) rows={rows} read {16.25*n/ms/1e6:8.1f} GB/s frac {16.25*n/ms/1e6/8000:.3f}", flush=True)
ctx.set_option("debug", 0)
