"""Diagnostic: per-phase cycle shares for `b is true -> [x, b]` with a nullable Boolean column (FF_STAMP build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_BOOLEAN, RV_INT64, Predicate, Term, synth_spec
n = 500_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
bn = ctx.generate(synth_spec(RV_BOOLEAN, seed=47, length=n, true_percent=10, validity_seed=48))
pred = Predicate([Term(0, "is_true")])
ctx.set_option("profile_kernels", 1)
for stamp in (0, 1):
    ctx.set_option("stamp", stamp)
    for rep in range(2):
        outs, rows, _ = ctx.filter_project([bn, x], pred, [1, 0]); [o.free() for o in outs]
    ctx.kernel_stats(reset=True)
    outs, rows, _ = ctx.filter_project([bn, x], pred, [1, 0]); [o.free() for o in outs]
    ms, k = ctx.kernel_stats()
    print(f"stamp={stamp}: {ms/k:.3f} ms rows={rows}", flush=True)
