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
from rivulus_amd.capi import RV_BOOLEAN, RV_FLOAT64, RV_INT64, RV_STRING, Column, Predicate, Term, synth_spec  # noqa: E402

WINDOW_ROWS = 1 << 28
WINDOW_BATCHES = 1 << 18
MAX_SINGLE_CALLS = 1500  # bound on the per-batch loops

out_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
ctx = capi.Context(0)
results = []


def sweep(name, cols, pred, proj, n, sizes, bytes_per_row):
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
        # ---- two windows in flight (rv_filter_project_chunked_begin / _batches_begin + rv_filter_project_window_finish) ----
        cbufs = [counts, kept]
        def piped(begin):
            ctx.synchronize()
            t0 = time.perf_counter()
            q = [begin(0)]
            for w in range(nwin):
                if w + 1 < nwin:
                    q.append(begin(w + 1))
                outs, rows, _, tot = q.pop(0)(False)
                for o in outs:
                    o.free()
            ctx.synchronize()
            return time.perf_counter() - t0
        piped(lambda w: ctx.window_begin(pred, proj, cbufs[w % 2], cols=tables[w], chunk_rows=b))
        dtcq = piped(lambda w: ctx.window_begin(pred, proj, cbufs[w % 2], cols=tables[w], chunk_rows=b))
        piped(lambda w: ctx.window_begin(pred, proj, cbufs[w % 2], handles=windows[w][1]))
        dtbq = piped(lambda w: ctx.window_begin(pred, proj, cbufs[w % 2], handles=windows[w][1]))
        r = {"workload": name, "rows_per_batch": b, "chunked_two_in_flight_rows_per_s": done_b / dtcq, "batched_two_in_flight_rows_per_s": done_b / dtbq,
             "chunked_two_in_flight_frac_of_8TBps": done_b / dtcq * bytes_per_row / 8e12, "batched_two_in_flight_frac_of_8TBps": done_b / dtbq * bytes_per_row / 8e12, "per_batch_rows_per_s": done / dt, "per_batch_us": dt / nb * 1e6, "chunked_rows_per_s": done_b / dtc, "chunked_pageable_counts_rows_per_s": done_b / dtcp,
             "two_in_flight_rows_per_s": done / dtp, "batched_rows_per_s": done_b / dtb, "batches_per_call": k,
             "batched_us_per_batch": dtb / (sum(len(bs) for bs, _ in windows)) * 1e6, "selectivity": total / max(1, done_b),
             "algorithmic_read_bytes_per_row": bytes_per_row, "batched_frac_of_8TBps": done_b / dtb * bytes_per_row / 8e12, "chunked_frac_of_8TBps": done_b / dtc * bytes_per_row / 8e12,
             "kernel": ctx.last_kernel()}
        results.append(r)
        print(json.dumps(r), flush=True)
        del slices, windows


which = [a for a in sys.argv[1:] if not a.startswith("--") and a != out_path] or ["config2", "config3"]
sizes2 = [1024, 1 << 16] if "seam" in which else [1024, 1 << 16, 1 << 20, 1 << 24, 1 << 26, 1 << 28]
sizes3 = [1024] if "seam" in which else [1024, 1 << 20, 1 << 26]
if "config2" in which or "seam" in which:
    n = 1_000_000_000
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    sweep("config2: x > 899 -> [x]", [x], Predicate([Term(0, ">", 899)]), [0], n, sizes2, 8.0)
    x.free()
if "config3" in which or "seam" in which:
    n3 = 1_000_000_000 if "seam" in which else 500_000_000
    f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n3, validity_seed=44))
    xv = ctx.generate(synth_spec(RV_INT64, seed=42, length=n3, validity_seed=45))
    sweep("config3: (f > 0.5) AND (x < 200) -> [f, x], nullable", [f, xv], Predicate([Term(0, ">", 0.5), Term(1, "<", 200)]), [0, 1], n3,
          sizes3, 16.25)
    f.free(), xv.free()
if "bool" in which:
    # the reference's literal streaming query: the ONE predicate collect_streaming() accepts is a Boolean column
    # (streaming_planner.rs:137-168 -> FilterStream, stream.rs:136-158 -> RecordBatch::filter, record_batch.rs:221-243)
    nb_ = 1 << 30
    b = ctx.generate(synth_spec(RV_BOOLEAN, seed=47, length=nb_, true_percent=10, validity_seed=48))
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=nb_))
    sweep("bool_x: b is true -> [x]", [b, x], Predicate([Term(0, "is_true")]), [1], nb_, [1024, 1 << 16], 8.25)
    nb2 = 1 << 29
    fn = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=nb2, validity_seed=44))
    sweep("bool_xf: b is true -> [x, f_nullable]", [b.slice(0, nb2), x.slice(0, nb2), fn], Predicate([Term(0, "is_true")]), [1, 2], nb2, [1024], 16.375)
    fn.free()
    nb3 = 1 << 28
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 9, nb3).astype(np.int32)
    offs = np.zeros(nb3 + 1, dtype=np.int32)
    np.cumsum(lens, out=offs[1:])
    name = ctx.upload(Column(RV_STRING, rng.integers(97, 123, int(offs[-1])).astype(np.uint8), None, 0, nb3, offs))
    sweep("bool_xname: b is true -> [x, name]", [b.slice(0, nb3), x.slice(0, nb3), name], Predicate([Term(0, "is_true")]), [1, 2], nb3, [1024], 8.25 + 4.0)

print(f"\n{'workload':58s} {'R':>10s} {'per batch':>10s} {'2 in flight':>11s} {'batched':>10s} {'chunked':>10s} {'K/call':>8s}   batched / chunked share of 8 TB/s | two windows in flight: batched / chunked rows/s (share)")
for r in results:
    print(f"{r['workload']:58s} {r['rows_per_batch']:>10d} {r['per_batch_rows_per_s']:>10.2e} {r['two_in_flight_rows_per_s']:>11.2e} "
          f"{r['batched_rows_per_s']:>10.2e} {r['chunked_rows_per_s']:>10.2e} {r['batches_per_call']:>8d}   {r['batched_frac_of_8TBps']:.3f} / {r['chunked_frac_of_8TBps']:.3f} | {r['batched_two_in_flight_rows_per_s']:.2e} ({r['batched_two_in_flight_frac_of_8TBps']:.3f}) / {r['chunked_two_in_flight_rows_per_s']:.2e} ({r['chunked_two_in_flight_frac_of_8TBps']:.3f})")
if out_path:
    json.dump(results, open(out_path, "w"), indent=1)
