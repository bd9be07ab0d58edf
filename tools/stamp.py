"""Diagnostic: per-phase cycle shares of the fused kernel (FF_STAMP build; shares only, never quote its run time)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec
n = 1_000_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
pred = Predicate([Term(0, ">", 899)])
ctx.set_option("profile_kernels", 1)
for (r, w) in [(16, 16), (32, 8)]:
    for wg in [0]:
        ctx.set_option("rows_per_lane", r | (w << 8)); ctx.set_option("vec", 2); ctx.set_option("wgs_per_cu", wg)
        ctx.set_option("stamp", 0)
        for rep in range(3):
            outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
        ctx.kernel_stats(reset=True)
        for rep in range(3):
            outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
        ms, k = ctx.kernel_stats(); print(f"R={r} W={w} wgs_per_cu={wg}: {ms/k:.3f} ms", flush=True)
        ctx.set_option("stamp", 1)
        outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
        sys.stderr.flush()
