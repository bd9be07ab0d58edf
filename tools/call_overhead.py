import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec
ctx = capi.Context(0)
for n in (65536, 1 << 22, 1 << 26):
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    pred = Predicate([Term(0, ">", 899)])
    for _ in range(20):
        outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
    ctx.synchronize(); t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        outs, rows, _ = ctx.filter_project([x], pred, [0]); [o.free() for o in outs]
    ctx.synchronize()
    print(n, "call us", round((time.perf_counter() - t0) / K * 1e6, 1), flush=True)
    # python-only part: time of the free loop + arg marshalling is inside; measure an empty-ish C call
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.get_option("sample")
    print("   get_option us", round((time.perf_counter() - t0) / K * 1e6, 2))
