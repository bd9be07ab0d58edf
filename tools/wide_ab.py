"""x > t -> [x, c1 .. c8] over 2e8 rows (the eager Filter's shape): later column groups compacted at the first pass's wave offsets
(compact_ranges_kernel, option groups_by_ranges = 0) against passes of their own (-1), alternating on one box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

ctx = capi.Context(0)
n = 200_000_000
cols = [ctx.generate(synth_spec(RV_INT64, seed=42, length=n))]
nullable = "nullable" in sys.argv  # c1 .. c8 with null bitmaps (a CSV scan's columns all carry one)
cols += [ctx.generate(synth_spec(RV_INT64 if j % 2 else RV_FLOAT64, seed=50 + j, length=n, validity_seed=(70 + j) if nullable else None)) for j in range(1, 9)]
for k in (2, 3, 5, 9):
    for lit in ((899, 799, 699, 599, 499, 159) if "sweep" in sys.argv else (899, 499, 159)):
        pred, proj = Predicate([Term(0, ">", lit)]), list(range(k))
        out = {}
        for mode in (0, -1, 0, -1):
            ctx.set_option("groups_by_ranges", mode)
            for _ in range(3):
                outs, rows, _ = ctx.filter_project(cols[:k], pred, proj)
                [o.free() for o in outs]
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                outs, rows, _ = ctx.filter_project(cols[:k], pred, proj)
                [o.free() for o in outs]
            ctx.synchronize()
            out.setdefault(mode, []).append((time.perf_counter() - t0) / 6 * 1e3)
        gb = (8.0 + 0.125 * nullable) * k * (n + rows) / 1e9
        print(f"{k} columns, keep {rows / n:4.2f}: at the wave offsets {min(out[0]):6.3f} ms = {gb / min(out[0]) / 8 * 100:4.1f} %   own passes {min(out[-1]):6.3f} ms = {gb / min(out[-1]) / 8 * 100:4.1f} %", flush=True)
ctx.set_option("groups_by_ranges", 0)
