"""Diagnostic: scanner / fallback counters on BASELINE config 3 (debug bit 4; results stay correct)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, RV_FLOAT64, Predicate, Term, synth_spec
n = 500_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
ctx.set_option("profile_kernels", 1)
for dbg in (0, 4, 2, 1, 3):
    ctx.set_option("debug", dbg)
    for rep in range(2):
        outs, rows, _ = ctx.filter_project([f, x], pred, [0, 1]); [o.free() for o in outs]
    ctx.kernel_stats(reset=True)
    for rep in range(3):
        outs, rows, _ = ctx.filter_project([f, x], pred, [0, 1]); [o.free() for o in outs]
    ms, k = ctx.kernel_stats()
    print(f"debug={dbg}: {ms/3:.3f} ms rows={rows} frac {16.25*n/(ms/3)/1e6/8000:.3f}", flush=True)
