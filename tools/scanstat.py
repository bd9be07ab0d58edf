"""Diagnostic: scanner / fallback look-back counters of the fused kernel (debug bit 4; results stay correct)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec
n = 1_000_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
ctx.set_option("profile_kernels", 1)
ctx.set_option("rows_per_lane", 16 | (16 << 8)); ctx.set_option("vec", 2)
for lit, depth, dbg in [(899, 1, 0), (899, 2, 0), (899, 2, 4), (899, 2, 2), (999, 2, 0), (999, 1, 0)]:
    pred = Predicate([Term(0, ">", lit)])
    ctx.set_option("depth", depth)
    if True:
        ctx.set_option("debug", dbg)
        for rep in range(2):
            outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
        ctx.kernel_stats(reset=True)
        for rep in range(3):
            outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
        ms, k = ctx.kernel_stats()
        print(f"lit={lit} depth={depth} debug={dbg} rows={rows}: {ms/k:.3f} ms", flush=True)
