"""Seam S1 with two windows in flight (rv_filter_project_chunked_begin / _batches_begin + rv_filter_project_window_finish): 2^28-row
windows of 1024-row batches over a resident table, a rocprofv3 --kernel-trace target (profiles/r05_seam_*) and a timer.
    python3 tools/seam_pipeline.py config2|config3 chunked|handles [windows]
Prints one JSON line: rows/s and the share of 8 TB/s on the algorithmic bytes.
    python3 tools/seam_pipeline.py --timeline <dir with *kernel_trace.csv>     the last windows' kernels on one time axis"""
import csv
import glob
import json
import os
import sys
import time

if sys.argv[1] == "--timeline":
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    passes = [i for i, r in enumerate(rows) if "fused_" in r["Kernel_Name"] and "redo" not in r["Kernel_Name"]]
    first = passes[-5]
    t0 = int(rows[first]["Start_Timestamp"])
    busy_until = {}
    for r in rows[first:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get("Queue_Id", "?")
        idle = "" if q not in busy_until else f"  (queue idle {(s - busy_until[q]) / 1e3:6.1f})"
        print(f"q{q:>3s} {r['Kernel_Name'][:56]:56s} start {(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:8.1f}{idle}")
        busy_until[q] = e
    sys.exit(0)

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec  # noqa: E402

work, form = sys.argv[1], sys.argv[2]
nwin = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ctx = capi.Context(0)
W, R = 1 << 28, 1024
n = 4 * W  # the windows cycle over a 2^30-row table
if work == "config2":
    cols, pred, proj, bpr = [ctx.generate(synth_spec(RV_INT64, seed=42, length=n))], Predicate([Term(0, ">", 899)]), [0], 8.0
else:
    cols = [ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44)), ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))]
    pred, proj, bpr = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)]), [0, 1], 16.25
tables = [[c.slice(w * W, W) for c in cols] for w in range(4)]
batch_lists = [[[c.slice(o, R) for c in t] for o in range(0, W, R)] for t in tables] if form == "handles" else []  # (kept: the handles are theirs)
handles = [ctx.batch_handles(bl) for bl in batch_lists] if form == "handles" else None
bufs = [ctx.pinned_array(np.uint64, W // R) for _ in range(2)]


def begin(w):
    if form == "handles":
        return ctx.window_begin(pred, proj, bufs[w % 2], handles=handles[w % 4])
    return ctx.window_begin(pred, proj, bufs[w % 2], cols=tables[w % 4], chunk_rows=R)


stamps = []


def run(k):
    q = [begin(0)]
    total = 0
    for w in range(k):
        t0 = time.perf_counter()
        if w + 1 < k:
            q.append(begin(w + 1))
        t1 = time.perf_counter()
        outs, rows, _, tot = q.pop(0)(False)
        t2 = time.perf_counter()
        total += tot
        for o in outs:
            o.free()
        stamps.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
    return total


run(4)
ctx.synchronize()
t0 = time.perf_counter()
total = run(nwin)
ctx.synchronize()
dt = time.perf_counter() - t0
if os.environ.get("SEAM_STAMPS"):
    for b, f, r in stamps[-6:]:
        print(f"host: begin {b * 1e6:7.1f} us | finish {f * 1e6:7.1f} us | free {r * 1e6:6.1f} us", file=sys.stderr)
print(json.dumps({"workload": work, "form": form, "windows": nwin, "rows_per_window": W, "rows_per_batch": R, "rows_per_s": nwin * W / dt, "ms_per_window": dt / nwin * 1e3,
                  "frac_of_8TBps": nwin * W / dt * bpr / 8e12, "selectivity": total / (nwin * W), "kernel": ctx.last_kernel()}), flush=True)
