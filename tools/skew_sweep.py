"""x > t -> [x] over 1e9 rows, independent rows against sorted / clustered tables at the same global selectivity, same box, same
process: call time (wall over 5 calls, after a first call that samples), kernel, wave ranges left to the redo kernel.
    python3 tools/skew_sweep.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = capi.Context(0)
PATTERNS = [("iid", {}), ("sorted", dict(pattern="sorted")), ("sorted_desc", dict(pattern="sorted_desc")), ("runs_1e3", dict(pattern="clustered", run_rows=1_000)),
            ("runs_1e4", dict(pattern="clustered", run_rows=10_000)), ("runs_1e5", dict(pattern="clustered", run_rows=100_000)), ("runs_1e7", dict(pattern="clustered", run_rows=10_000_000))]
base = {}


def short(kernel):
    """fused_filter_compact<1, 16, 2, 16, 32> -> filter<1,16,2,16,32>; a table run stretch by stretch lists every stretch's kernel"""
    parts = kernel.replace("stretches: ", "").split(" + ")
    names = [p.split("<")[0].replace("fused_", "").replace("_compact", "") + "<" + p.split("<")[1].replace(" ", "") if "<" in p else p for p in parts]
    return ("stretches " if kernel.startswith("stretches") else "") + " + ".join(names)


for name, kw in PATTERNS:
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, **kw))
    for pct, lit in ((10, 899), (50, 499), (84, 159)):
        pred = Predicate([Term(0, ">", lit)])
        t0 = time.perf_counter()
        outs, rows, _ = ctx.filter_project([x], pred, [0])
        ctx.synchronize()
        first = (time.perf_counter() - t0) * 1e3
        first_kernel, first_redo = ctx.last_kernel(), ctx.get_option("last_redo_ppm")
        [o.free() for o in outs]
        outs, rows, _ = ctx.filter_project([x], pred, [0])
        [o.free() for o in outs]
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            outs, rows, _ = ctx.filter_project([x], pred, [0])
            [o.free() for o in outs]
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        if name == "iid":
            base[pct] = ms
        kern = ctx.last_kernel()
        print(f"{name:12s} {pct:3d} % | first call {first:7.3f} ms {short(first_kernel):44s} redo {first_redo:7d} ppm | steady {ms:7.3f} ms = {ms / base[pct]:5.2f} x iid  "
              f"{short(kern):44s} redo {ctx.get_option('last_redo_ppm'):7d} ppm | kept {rows / n:.4f} reruns {ctx.get_option('overflow_reruns')}", flush=True)
    x.free()
