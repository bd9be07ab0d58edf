import os, sys, time
sys.path.insert(0, os.getcwd())
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec
ctx = capi.Context(0)
n = 1 << 28
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
pred = Predicate([Term(0, ">", 899)])
b = 1024
bs = [[x.slice(i * b, b)] for i in range(n // b)]
h = ctx.batch_handles(bs)
for rep in range(4):
    t0 = time.perf_counter()
    outs, rows, _, tot = ctx.filter_project_batches(None, pred, [0], want_nulls=False, handles=h)
    ctx.synchronize()
    print("call ms", (time.perf_counter() - t0) * 1e3, flush=True)
    [o.free() for o in outs]
