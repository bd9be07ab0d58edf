"""Where the direct kernel's cycles go: the FF_STAMP instantiations (one column R = 12, three columns R = 4; 8 waves) print,
per tile, the s_memtime sums of an ordinary wave's phases (stderr, "[stamp direct] ..."), next to the scanner / fallback counters."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

ctx = capi.Context(0)
n = 500_000_000
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
y = ctx.generate(synth_spec(RV_INT64, seed=46, length=n))
z = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n))
ctx.set_option("direct", 1)
for label, cols, proj, r in (("[x]", [x], [0], 12), ("[x, y, z]", [x, y, z], [0, 1, 2], 6)):
    for sel in (50, 90, 100):
        pred = Predicate([Term(0, ">", 999 - 10 * sel)])
        for wgs in (0, 1):
            ctx.set_option("direct_r", r)
            ctx.set_option("wgs_per_cu", wgs)
            ctx.set_option("stamp", 1)
            ctx.set_option("debug", 4)
            sys.stderr.write(f"x > t -> {label} sel {sel} % wgs/cu {wgs or 'auto'}: ")
            sys.stderr.flush()
            outs, rows, _ = ctx.filter_project(cols, pred, proj)
            sys.stderr.write(f"   {ctx.last_kernel()}\n")
            [o.free() for o in outs]
            ctx.set_option("stamp", 0)
            ctx.set_option("debug", 0)
