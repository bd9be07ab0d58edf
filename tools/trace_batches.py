"""Phase times of rv_filter_project_batches over handle-form batches (RV_TRACE_BATCHES=1 prints walk | pass | counts from
inside the library), next to rv_filter_project_chunked over the same rows.  Diagnostic; run on the GPU box.
usage: RV_TRACE_BATCHES=1 python tools/trace_batches.py [2|3] [rows_per_batch] [rows]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = capi.Context(0)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 28
if config == 2:
    cols = [ctx.generate(synth_spec(RV_INT64, seed=42, length=n))]
    pred, proj = Predicate([Term(0, ">", 899)]), [0]
else:
    cols = [ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44)), ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))]
    pred, proj = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)]), [0, 1]
bs = [[c.slice(i * b, min(b, n - i * b)) for c in cols] for i in range((n + b - 1) // b)]
h = ctx.batch_handles(bs)
kept = ctx.pinned_array(np.uint64, len(bs))
for rep in range(4):
    t0 = time.perf_counter()
    outs, rows, _, tot = ctx.filter_project_batches(None, pred, proj, want_nulls=False, handles=h, rows_buffer=kept)
    ctx.synchronize()
    print("batches call ms", (time.perf_counter() - t0) * 1e3, ctx.last_kernel(), flush=True)
    [o.free() for o in outs]
for rep in range(3):
    t0 = time.perf_counter()
    outs, rows, _, tot = ctx.filter_project_chunked(cols, b, pred, proj, want_nulls=False, rows_buffer=kept)
    ctx.synchronize()
    print("chunked call ms", (time.perf_counter() - t0) * 1e3, ctx.last_kernel(), flush=True)
    [o.free() for o in outs]
