"""One column, x > t -> [x], 1e9 rows: wall time per call against the selectivity for the geometry the library picks from the
context's last selectivity ("auto": first call at the new selectivity / following calls) and for forced geometries
(rows per lane, waves, dense sizing = option "roomy")."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec  # noqa: E402

n = 1_000_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
geoms = [("auto", 0, 0), ("direct", -1, 0), ("16 x16", 16 | 16 << 8, 0), ("16 x16 dense", 16 | 16 << 8, 1), ("8 x16", 8 | 16 << 8, 0), ("4 x16", 4 | 16 << 8, 0), ("8 x16 dense", 8 | 16 << 8, 1)]
for sel in (10, 20, 30, 35, 45, 55, 70, 90, 100):
    pred = Predicate([Term(0, ">", 999 - 10 * sel)])
    line = f"sel {sel:3d} %"
    for name, rpl, roomy in geoms:
        ctx.set_option("rows_per_lane", max(rpl, 0))
        ctx.set_option("roomy", roomy)
        ctx.set_option("direct", 1 if rpl < 0 else (0 if name == "auto" else -1))
        times = []
        for rep in range(4):
            ctx.synchronize()
            t0 = time.perf_counter()
            outs, rows, s = ctx.filter_project([x], pred, [0])
            ctx.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
            [o.free() for o in outs]
        k = ctx.last_kernel()
        geo = k[k.index("<") + 1:k.index(">")].split(",")
        geo = "direct" if k.startswith("fused_direct") else f"<{geo[1]},{geo[3]}>"
        line += f" | {name} {min(times[1:]):6.3f}" + (f" (first {times[0]:6.3f}, {geo})" if name == "auto" else "") + ("*" if ctx.get_option("last_redo_ppm") else "")
    print(line, flush=True)
