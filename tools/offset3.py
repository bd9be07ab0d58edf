"""Does the relative placement of the two input columns matter (HBM channel / bank conflicts between the streams)?
BASELINE config 3 with the x column starting `off` rows into its allocation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, RV_FLOAT64, Predicate, Term, synth_spec
n = 500_000_000
ctx = capi.Context(0)
f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
xa = ctx.generate(synth_spec(RV_INT64, seed=42, length=n + (1 << 21), validity_seed=45))
pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
ctx.set_option("profile_kernels", 1)
for off in [0, 64, 512, 2048, 8192, 1 << 15, 1 << 17, 1 << 19, (1 << 20) + 4096 + 64, 0]:
    x = xa.slice(off, n)
    for rep in range(2):
        outs, rows, _ = ctx.filter_project([f, x], pred, [0, 1]); [o.free() for o in outs]
    ctx.kernel_stats(reset=True)
    for rep in range(5):
        outs, rows, _ = ctx.filter_project([f, x], pred, [0, 1]); [o.free() for o in outs]
    ms, k = ctx.kernel_stats()
    print(f"x offset {off:8d} rows ({off * 8:9d} B): {ms / 5:7.3f} ms  {16.25 * n / (ms / 5) / 1e6 / 8000:.3f} of peak", flush=True)
