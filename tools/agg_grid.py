"""filter + SUM/COUNT (BASELINE configs[4] per GPU, 1e9 Int64 rows by default; `agg_grid.py 1e10` for the G = 1 table): one workgroup per tile against a grid-stride launch of
g workgroups per CU (option "agg_grid")."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000  # e.g. 1e10: BASELINE configs[4] at G = 1
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
pred = Predicate([Term(0, ">", 899)])
ctx.set_option("profile_kernels", 1)
ref = None
for g in [-1, 1, 2, 4, 8, 16, 32, 64, -1, 0]:  # -1: one workgroup per tile, 0: the default (32 per CU)
    ctx.set_option("agg_grid", g)
    for _ in range(2):
        r = ctx.filter_agg([x], pred, 0)
    ctx.kernel_stats(reset=True)
    for _ in range(10):
        r = ctx.filter_agg([x], pred, 0)
    ms, k = ctx.kernel_stats()
    ref = ref or (r[0], r[2])
    assert (r[0], r[2]) == ref
    t = ms / k
    print(f"agg_grid={g:2d}: {t:.3f} ms  {8 * n / t / 1e9:.2f} TB/s = {8 * n / t / 1e9 / 8 * 100:.1f} %", flush=True)
