"""Geometry sweep of the one-column fused kernel (tuning aid, not part of the product)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
pred = Predicate([Term(0, ">", 899)])
ctx.set_option("profile_kernels", 1)
res = []
for (r, w, vec, depth) in [(32, 8, 2, 0), (16, 16, 2, 0), (16, 16, 2, 1), (32, 16, 2, 0), (32, 16, 1, 0), (24, 16, 2, 0), (16, 12, 2, 0), (16, 16, 2, 0)]:
    if True:
        ctx.set_option("rows_per_lane", r | (w << 8))
        ctx.set_option("vec", vec)
        ctx.set_option("depth", depth)
        for rep in range(2):
            outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
        ctx.kernel_stats(reset=True)
        for rep in range(5):
            outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
        ms, k = ctx.kernel_stats()
        ms /= k
        res.append((r, w, vec, ms, 8.0 * n / ms / 1e6))
        print(f"R={r:2d} W={w:2d} VEC={vec} depth={depth}: {ms:8.3f} ms  read {8.0*n/ms/1e6:8.1f} GB/s  frac {8.0*n/ms/1e6/8000:.3f}", flush=True)

# read-only reference point: filter + SUM/COUNT (same scan front end, no look-back, no writes)
for vec in [1, 2]:
    ctx.set_option("vec", vec)
    for rep in range(2):
        ctx.filter_agg([x], pred, 0)
    ctx.kernel_stats(reset=True)
    for rep in range(5):
        ctx.filter_agg([x], pred, 0)
    ms, k = ctx.kernel_stats()
    ms /= k
    print(f"agg VEC={vec}: {ms:8.3f} ms  read {8.0*n/ms/1e6:8.1f} GB/s  frac {8.0*n/ms/1e6/8000:.3f}", flush=True)

# selectivity sweep on the default geometry
ctx.set_option("rows_per_lane", 0); ctx.set_option("vec", 0); ctx.set_option("depth", 0)
for lit in [999, 989, 949, 899, 799, 499, 99, -1]:
    pred = Predicate([Term(0, ">", lit)])
    for rep in range(2):
        outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
    ctx.kernel_stats(reset=True)
    for rep in range(4):
        outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
    ms, k = ctx.kernel_stats()
    print(f"selectivity {rows/n:.3f}: {ms/4:8.3f} ms/step ({k//4} kernels/step)  total bytes {8.0*(n+rows)/(ms/4)/1e6:8.1f} GB/s", flush=True)
