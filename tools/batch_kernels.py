"""Rates of the S2 batch kernels next to the fused path: take (gather), concat, compare, Boolean ops."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_BOOLEAN, RV_INT64, synth_spec

ctx = capi.Context(0)
n = 500_000_000
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
b1 = ctx.generate(synth_spec(RV_BOOLEAN, seed=47, length=n, true_percent=30, validity_seed=48))
b2 = ctx.generate(synth_spec(RV_BOOLEAN, seed=49, length=n, true_percent=60, validity_seed=50))


def timed(label, fn, bytes_moved, reps=3):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:46s} {1e3*dt:8.3f} ms  {bytes_moved/dt/1e9:8.1f} GB/s", flush=True)


def free(cols):
    for c in cols if isinstance(cols, (list, tuple)) else [cols]:
        c.free()


m = 50_000_000
idx_sorted = np.sort(np.random.default_rng(1).integers(0, n, m)).astype(np.uint64)
idx_random = np.random.default_rng(2).integers(0, n, m).astype(np.uint64)
timed("take 5e7 ascending indices of 5e8 (incl. index upload)", lambda: free(ctx.take([x], idx_sorted)), m * (8 + 8 + 8))
timed("take 5e7 random indices of 5e8 (incl. index upload)", lambda: free(ctx.take([x], idx_random)), m * (8 + 8 + 8))
parts = [x.slice(i * (n // 16), n // 16) for i in range(16)]
timed("concat 16 x 3.1e7 nullable Int64", lambda: free(ctx.concat(parts)), 2 * n * 8.125)
timed("compare x > 899 -> nullable BooleanArray", lambda: free(ctx.compare(x, ">", 899)), n * 8.375)
timed("boolean and (values + validity)", lambda: free(ctx.boolean_and(b1, b2)), n * 0.75)
timed("boolean count", lambda: ctx.boolean_count(b1), n * 0.25)
timed("null_count of a slice", lambda: x.slice(7, n - 9).null_count(), n * 0.125)
