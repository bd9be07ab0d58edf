"""Ablation of BASELINE config 3 with the diagnostic (FF_STAMP) instantiation: option "debug" bit 0 = no output stores,
bit 1 = no output-offset lookup (results are wrong, timing shares only); the aggregate over the same columns for scale."""
import sys, os
sys.path.insert(0, os.getcwd())
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, RV_FLOAT64, Predicate, Term, synth_spec
n = 500_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
ctx.set_option("profile_kernels", 1)
ctx.set_option("rows_per_lane", 8 | (16 << 8)); ctx.set_option("vec", 1)  # the diagnostic instantiation's geometry
for stamp, dbg in [(0, 0), (1, 0), (0, 1), (0, 2), (0, 3), (0, 0)]:
    ctx.set_option("stamp", stamp); ctx.set_option("debug", dbg)
    for rep in range(2):
        outs, rows, _ = ctx.filter_project([f, x], pred, [0, 1]); [o.free() for o in outs]
    ctx.kernel_stats(reset=True)
    for rep in range(5):
        outs, rows, _ = ctx.filter_project([f, x], pred, [0, 1]); [o.free() for o in outs]
    ms, k = ctx.kernel_stats()
    print(f"stamp={stamp} debug={dbg}: {ms/5:8.3f} ms", flush=True)
for g in (-1, 0):
    ctx.set_option("agg_grid", g)
    si = ctx.filter_agg([f, x], pred, 1)
    ctx.kernel_stats(reset=True)
    for rep in range(5): ctx.filter_agg([f, x], pred, 1)
    ms, k = ctx.kernel_stats(); print(f"agg (agg_grid={g}): {ms/5:8.3f} ms")
