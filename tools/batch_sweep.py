"""Throughput of collect_streaming()-style execution against the RecordBatch size R (config 2 and
config 3 fed through the stream seam S1 at R in {1024 (the reference's batch, streaming_planner.rs:32), 1 Mi, 64 Mi, ...}).
Device-resident batches = zero-copy slices of resident columns, as dataframe_to_batches hands them out.

Three ways to run the same batches:
  per batch      one rv_filter_project per batch (memset + launch + 512-byte read-back + sync: ~30 us fixed)
  two in flight  rv_filter_project_begin / _finish, one batch ahead
  batched        rv_filter_project_batches: a window of K batches per call (one pass, one read-back of K row counts);
                 the window is what a stream operator pulls ahead: at most WINDOW_ROWS rows / WINDOW_BATCHES batches
  chunked        rv_filter_project_chunked: the same windows handed over as one resident table + the batch size (the
                 reference's dataframe_to_batches source): no handles, no boundary table
Prints one JSON object per (workload, R) and a summary table; `--json path` also writes them to a file.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

WINDOW_ROWS = 1 << 28
WINDOW_BATCHES = 1 << 18
MAX_SINGLE_CALLS = 1500  # bound on the per-batch loops

out_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
ctx = capi.Context(0)
results = []


def sweep(name, cols, pred, proj, n, sizes):
    for b in sizes:
        nb_all = (n + b - 1) // b
        # ---- per batch / two in flight (bounded number of calls) ----
        nb = min(nb_all, MAX_SINGLE_CALLS)
        slices = [[c.slice(i * b, min(b, n - i * b)) for c in cols] for i in range(nb)]
        outs, _, _ = ctx.filter_project(slices[0], pred, proj)
        [o.free() for o in outs]
        ctx.synchronize()
        t0 = time.perf_counter()
        for s in slices:
            outs, rows, _ = ctx.filter_project(s, pred, proj)
            for o in outs:
                o.free()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        q = []
        for s in slices:
            q.append(ctx.filter_project_begin(s, pred, proj))
            if len(q) > 2:
                outs, rows = q.pop(0)()
                for o in outs:
                    o.free()
        for fin in q:
            outs, rows = fin()
            for o in outs:
                o.free()
        ctx.synchronize()
        dtp = time.perf_counter() - t0
        done = min(n, nb * b)
        # ---- batched: windows of K batches per call ----
        k = max(1, min(WINDOW_BATCHES, WINDOW_ROWS // b, nb_all))
        nwin = min((nb_all + k - 1) // k, 8)
        windows = []
        for w in range(nwin):
            bs = [[c.slice(i * b, min(b, n - i * b)) for c in cols] for i in range(w * k, min(nb_all, (w + 1) * k))]
            windows.append((bs, ctx.batch_handles(bs)))  # the handle array is assembled once, like a stream's batch list
        kept = ctx.pinned_array(np.uint64, k)  # the operator's own pinned array for the counts, as in the chunked form
        for _, h in windows:  # every window once untimed: a shorter last window has its own buffer sizes (first use = hipMalloc)
            outs, rows, _, _ = ctx.filter_project_batches(None, pred, proj, want_nulls=False, handles=h, rows_buffer=kept)
            [o.free() for o in outs]
        ctx.synchronize()
        t0 = time.perf_counter()
        total = 0
        for bs, h in windows:
            outs, rows, _, tot = ctx.filter_project_batches(None, pred, proj, want_nulls=False, handles=h, rows_buffer=kept)
            total += tot
            for o in outs:
                o.free()
        ctx.synchronize()
        dtb = time.perf_counter() - t0
        done_b = min(n, nwin * k * b)
        # ---- chunked: the same windows as (table, batch size) ----
        tables = [[c.slice(w * k * b, min(k * b, n - w * k * b)) for c in cols] for w in range(nwin)]
        # the per-batch counts land in a pinned array the stream operator keeps (rv_host_alloc): written by the device
        counts = ctx.pinned_array(np.uint64, k)
        outs, rows, _, _ = ctx.filter_project_chunked(tables[0], b, pred, proj, want_nulls=False, rows_buffer=counts)
        [o.free() for o in outs]
        ctx.synchronize()
        t0 = time.perf_counter()
        for tb in tables:
            outs, rows, _, tot = ctx.filter_project_chunked(tb, b, pred, proj, want_nulls=False, rows_buffer=counts)
            for o in outs:
                o.free()
        ctx.synchronize()
        dtc = time.perf_counter() - t0
        # ... and into an ordinary (pageable) array: staged + copied
        t0 = time.perf_counter()
        for tb in tables:
            outs, rows, _, tot = ctx.filter_project_chunked(tb, b, pred, proj, want_nulls=False)
            for o in outs:
                o.free()
        ctx.synchronize()
        dtcp = time.perf_counter() - t0
        r = {"workload": name, "rows_per_batch": b, "per_batch_rows_per_s": done / dt, "per_batch_us": dt / nb * 1e6, "chunked_rows_per_s": done_b / dtc, "chunked_pageable_counts_rows_per_s": done_b / dtcp,
             "two_in_flight_rows_per_s": done / dtp, "batched_rows_per_s": done_b / dtb, "batches_per_call": k,
             "batched_us_per_batch": dtb / (sum(len(bs) for bs, _ in windows)) * 1e6, "selectivity": total / max(1, done_b)}
        results.append(r)
        print(json.dumps(r), flush=True)
        del slices, windows


n = 1_000_000_000
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
sweep("config2: x > 899 -> [x]", [x], Predicate([Term(0, ">", 899)]), [0], n, [1024, 1 << 16, 1 << 20, 1 << 24, 1 << 26, 1 << 28])
x.free()
n3 = 500_000_000
f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n3, validity_seed=44))
xv = ctx.generate(synth_spec(RV_INT64, seed=42, length=n3, validity_seed=45))
sweep("config3: (f > 0.5) AND (x < 200) -> [f, x], nullable", [f, xv], Predicate([Term(0, ">", 0.5), Term(1, "<", 200)]), [0, 1], n3,
      [1024, 1 << 20, 1 << 26])

print(f"\n{'workload':58s} {'R':>10s} {'per batch':>10s} {'2 in flight':>11s} {'batched':>10s} {'chunked':>10s} {'K/call':>8s}")
for r in results:
    print(f"{r['workload']:58s} {r['rows_per_batch']:>10d} {r['per_batch_rows_per_s']:>10.2e} {r['two_in_flight_rows_per_s']:>11.2e} "
          f"{r['batched_rows_per_s']:>10.2e} {r['chunked_rows_per_s']:>10.2e} {r['batches_per_call']:>8d}")
if out_path:
    json.dump(results, open(out_path, "w"), indent=1)
