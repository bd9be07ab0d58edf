"""Diagnostic ablations of the fused kernel (results are WRONG under debug != 0; timing shares only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec
n = 1_000_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
ctx.set_option("profile_kernels", 1)
for (r, w, vec) in [(16, 16, 2)]:
    ctx.set_option("rows_per_lane", r | (w << 8)); ctx.set_option("vec", vec)
    for lit, label in [(899, "10%"), (999, "0%"), (989, "1%")]:
        pred = Predicate([Term(0, ">", lit)])
        for dbg in [0, 4, 1, 2, 3]:
            ctx.set_option("debug", dbg)
            for rep in range(2):
                outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
            ctx.kernel_stats(reset=True)
            for rep in range(4):
                outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
            ms, k = ctx.kernel_stats()
            print(f"R={r} W={w} V={vec} sel={label} debug={dbg} (1=no stores,2=no lookback): {ms/k:.3f} ms", flush=True)
ctx.set_option("debug", 0)
