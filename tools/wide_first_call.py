import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rivulus_amd import capi
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec
ctx = capi.Context(0)
n = 200_000_000
cols = [ctx.generate(synth_spec(RV_INT64, seed=42, length=n))]
cols += [ctx.generate(synth_spec(RV_INT64 if j % 2 else RV_FLOAT64, seed=50 + j, length=n)) for j in range(1, 9)]
outs, rows, _ = ctx.filter_project(cols, Predicate([Term(0, ">", 5)]), list(range(9))); [o.free() for o in outs]
for lit in (899, 898, 897, 799, 159, 158):
    ctx.synchronize(); t0 = time.perf_counter()
    outs, rows, _ = ctx.filter_project(cols, Predicate([Term(0, ">", lit)]), list(range(9)))
    ctx.synchronize(); t1 = time.perf_counter()
    [o.free() for o in outs]
    t2 = time.perf_counter()
    outs, rows, _ = ctx.filter_project(cols, Predicate([Term(0, ">", lit)]), list(range(9)))
    ctx.synchronize(); t3 = time.perf_counter()
    [o.free() for o in outs]
    print(f"x > {lit}: first call {(t1 - t0) * 1e3:.3f} ms, second {(t3 - t2) * 1e3:.3f} ms, {ctx.last_kernel()}", flush=True)
