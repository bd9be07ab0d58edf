"""Dense selections through the direct (register-staged) kernel: kernel time (HIP events) per geometry against the staged
pass, per shape and selectivity.  5e8 rows (one column: 1e9).
    python3 tools/dense_sweep.py [one|three|two|four|all] [quick]
Prints: shape, selectivity, then per geometry  kernel ms | TB/s counting reads + writes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "all"
quick = len(sys.argv) > 2
stats = os.environ.get("DENSE_STATS") == "1"
ctx = capi.Context(0)


def timed(cols, pred, proj, reps=4):
    outs, rows, _ = ctx.filter_project(cols, pred, proj)
    [o.free() for o in outs]
    ctx.set_option("profile_kernels", 1)
    ctx.kernel_stats(reset=True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        outs, rows, _ = ctx.filter_project(cols, pred, proj)
        [o.free() for o in outs]
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ms, k = ctx.kernel_stats()
    ctx.set_option("profile_kernels", 0)
    return ms / max(1, k), wall, rows, ctx.last_kernel(), ctx.get_option("last_redo_ppm")


def sweep(label, n, cols, mk, proj, geoms, sels):
    for sel in sels:
        pred = Predicate(mk(sel))
        line = f"{label:22s} sel {sel:3d} %"
        for name, opts in geoms:
            for k in ("direct", "direct_r", "direct_waves", "rows_per_lane", "roomy", "wgs_per_cu", "debug"):
                ctx.set_option(k, opts.get(k, 0))
            ms, wall, rows, kern, redo = timed(cols, pred, proj)
            if stats and name != "staged" and kern.startswith("fused_direct"):  # scanner / fallback counters on stderr ([scan] ...)
                sys.stderr.write(f"{label} sel {sel} {name} {kern}: ")
                sys.stderr.flush()
                ctx.set_option("debug", 4 | opts.get("debug", 0))
                outs, _, _ = ctx.filter_project(cols, pred, proj)
                [o.free() for o in outs]
                ctx.set_option("debug", opts.get("debug", 0))
            if name != "staged" and not kern.startswith("fused_direct"):  # no such instantiation
                line += f" | {name} n/a"
                continue
            traffic = 8.0 * n * len(cols) + 8.0 * rows * len(proj)
            short = kern[kern.index("<"):]
            line += f" | {name} {ms:6.3f} ms {traffic / ms / 1e9:5.2f} TB/s {short if name in ('staged', 'direct') else ''}{'*' if redo else ''}"
        print(line, flush=True)


sels = (50, 90) if quick else (10, 30, 50, 70, 90, 100)
if which in ("one", "all"):
    n = 1_000_000_000
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    geoms = [("staged", {"direct": -1}), ("direct", {"direct": 1}), ("r8w8", {"direct": 1, "direct_r": 8, "direct_waves": 8}),
             ("r12w8", {"direct": 1, "direct_r": 12, "direct_waves": 8}), ("r16w16", {"direct": 1, "direct_r": 16, "direct_waves": 16}),
             ("r16w4", {"direct": 1, "direct_r": 16, "direct_waves": 4}), ("r16w8x1", {"direct": 1, "wgs_per_cu": 1})]
    sweep("x > t -> [x] 1e9", n, [x], lambda s: [Term(0, ">", 999 - 10 * s)], [0], geoms, sels)
    x.free()
n = 500_000_000
if which in ("three", "two", "four", "all"):
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    y = ctx.generate(synth_spec(RV_INT64, seed=46, length=n))
    z = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n))
if which in ("three", "all"):
    geoms = [("staged", {"direct": -1}), ("direct", {"direct": 1}), ("r4w8", {"direct": 1, "direct_r": 4, "direct_waves": 8}),
             ("r6w8", {"direct": 1, "direct_r": 6, "direct_waves": 8}), ("r4w16", {"direct": 1, "direct_r": 4, "direct_waves": 16}), ("r2w8", {"direct": 1, "direct_r": 2, "direct_waves": 8}),
             ("r8w4", {"direct": 1, "direct_r": 8, "direct_waves": 4})]
    sweep("x > t -> [x, y, z] 5e8", n, [x, y, z], lambda s: [Term(0, ">", 999 - 10 * s)], [0, 1, 2], geoms, sels)
if which in ("two", "all"):
    geoms = [("staged", {"direct": -1}), ("direct", {"direct": 1}), ("r12w8", {"direct": 1, "direct_r": 12, "direct_waves": 8})]
    sweep("x > t -> [x, y] 5e8", n, [x, y], lambda s: [Term(0, ">", 999 - 10 * s)], [0, 1], geoms, sels)
    sweep("x > t -> [y] 5e8", n, [x, y], lambda s: [Term(0, ">", 999 - 10 * s)], [1], geoms[:2], sels)
    sweep("x>t & y>=0 -> [x,y] 5e8", n, [x, y], lambda s: [Term(0, ">", 999 - 10 * s), Term(1, ">=", 0)], [0, 1], geoms[:2], sels)
if which in ("four", "all"):
    w = ctx.generate(synth_spec(RV_INT64, seed=49, length=n))
    geoms = [("staged", {"direct": -1}), ("direct", {"direct": 1}), ("r8w8", {"direct": 1, "direct_r": 8, "direct_waves": 8})]
    sweep("x > t -> [x,y,z,w] 5e8", n, [x, y, z, w], lambda s: [Term(0, ">", 999 - 10 * s)], [0, 1, 2, 3], geoms, sels)
