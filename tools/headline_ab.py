"""The headline pass (x > 899 -> [x], 1e9 rows) under option variants, alternating on ONE box (boxes differ by +- 3 %): kernel time by
HIP events, best and median of several rounds.   python3 tools/headline_ab.py [rounds]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=1_000_000_000))
run = ctx.prepared_filter_project([x], Predicate([Term(0, ">", 899)]), [0])
variants = [("default", {}), ("wgs_per_cu=1", {"wgs_per_cu": 1}), ("wgs_per_cu=2", {"wgs_per_cu": 2}), ("depth=1", {"depth": 1}), ("depth=2", {"depth": 2}),
            ("rows_per_lane=8", {"rows_per_lane": 8}), ("rows_per_lane=32", {"rows_per_lane": 32}), ("vec=1", {"vec": 1})]
res = {name: [] for name, _ in variants}
for r in range(rounds):
    for name, opts in variants:
        for k, v in opts.items():
            ctx.set_option(k, v)
        try:
            for _ in range(3):
                run()
            ctx.set_option("profile_kernels", 1)
            ctx.kernel_stats(reset=True)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                run()
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / 20 * 1e3
            ms, _ = ctx.kernel_stats()
            ctx.set_option("profile_kernels", 0)
            res[name].append((ms / 20, wall, ctx.last_kernel()))
        except Exception as e:  # an option value this shape has no geometry for
            res[name].append((float("nan"), float("nan"), str(e)[:60]))
        finally:
            for k in opts:
                ctx.set_option(k, 0)
for name, _ in variants:
    ks = [a for a, _, _ in res[name]]
    ws = [b for _, b, _ in res[name]]
    print(f"{name:18s} kernel best {min(ks):.4f} median {statistics.median(ks):.4f} ms   call median {statistics.median(ws):.4f} ms   {res[name][0][2]}", flush=True)
