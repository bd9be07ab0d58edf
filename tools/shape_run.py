"""Run one query shape a few times (a rocprofv3 target for tools/profile_round.sh; prints one JSON line with HIP-event times).
    python3 tools/shape_run.py bool_xb      b is true -> [x, b], nullable Boolean column projected (5e8 rows)
    python3 tools/shape_run.py bool_x       b is true -> [x] (RecordBatch::filter by a BooleanArray, 5e8 rows)
    python3 tools/shape_run.py strings      x > 899 -> [x, name], 2e8 rows, strings of 0..16 bytes (strings_dense: x > 159, 84 % survive)
    python3 tools/shape_run.py or2          (f > 0.9 OR x < 50) AND y >= 100 -> [f, x], nullable columns (5e8 rows)
    python3 tools/shape_run.py bool_c       x > 899 -> [x, c], c a nullable Boolean column the predicate does not read (5e8 rows; bool_c_half / bool_c_dense: 50 % / 84 % survive)
    python3 tools/shape_run.py dense1       x > 99 -> [x], 90 % survive (5e8 rows)
    python3 tools/shape_run.py dense3       x > 99 -> [x, y, f], 90 % survive (5e8 rows)
    python3 tools/shape_run.py wide5|wide9[_dense]   x > t -> [x, c1 .. c4 | c8], 10 % (84 %) survive, 2e8 rows: the eager Filter's shape
    python3 tools/shape_run.py iid10|sorted10|sorteddesc10|clustered10|clustered10k|clustered10m   x > 899 -> [x] over 1e9 rows whose
                               survivors are spread evenly / the last tenth / the first tenth / runs of 1e5 / 1e3 / 1e7 rows (rv_synth_spec patterns);
                               iid50|sorted50|clustered50 (x > 499), iid84|sorted84|clustered84 (x > 159): the same at 50 % and 84 %
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_BOOLEAN, RV_FLOAT64, RV_INT64, RV_STRING, Column, Predicate, Term, synth_spec  # noqa: E402

shape = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = capi.Context(0)
n = 500_000_000
if shape in ("bool_xb", "bool_x"):
    b = ctx.generate(synth_spec(RV_BOOLEAN, seed=47, length=n, true_percent=10, validity_seed=48 if shape == "bool_xb" else None))
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    cols, pred, proj = [b, x], Predicate([Term(0, "is_true")]), ([1, 0] if shape == "bool_xb" else [1])
    bytes_per_row = 8.25 if shape == "bool_xb" else 8.125
elif shape in ("bool_c", "bool_c_half", "bool_c_dense"):
    c = ctx.generate(synth_spec(RV_BOOLEAN, seed=47, length=n, true_percent=30, validity_seed=48))
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    lit = {"bool_c": 899, "bool_c_half": 499, "bool_c_dense": 159}[shape]  # 10 % / 50 % / 84 % of the rows survive
    cols, pred, proj = [x, c], Predicate([Term(0, ">", lit)]), [0, 1]
    keep = (999 - lit) / 1000.0
    bytes_per_row = 8.0 + 0.25 + keep * (8.0 + 0.25) if keep > 0.3 else 8.0 + 0.1 * 0.25  # x once + the survivors' value and validity bits (dense: + x written, the bits read)
elif shape in ("strings", "strings_dense"):
    n = 200_000_000
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 17, n).astype(np.int32)
    offs = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=offs[1:])
    data = rng.integers(97, 123, int(offs[-1])).astype(np.uint8)
    cols = [ctx.upload(Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64))), ctx.upload(Column(RV_STRING, data, None, 0, n, offs))]
    # strings_dense: BASELINE configs[0]'s shape -- filter(age > 25) keeps 84 % of the rows, [name, age] projected
    keep = 0.1 if shape == "strings" else 0.84
    pred, proj = Predicate([Term(0, ">", 899 if shape == "strings" else 159)]), [0, 1]
    bytes_per_row = 8.0 + 4.0 * (keep > 0.5) + keep * (8 + 4 + 2 * 8.0)  # x once (+ every offset when most rows survive) + the survivors' x, offset and bytes in and out
elif shape == "or2":
    f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
    y = ctx.generate(synth_spec(RV_INT64, seed=46, length=n))
    cols, proj = [f, x, y], [0, 1]
    pred = Predicate([Term(0, ">", 0.9), Term(1, "<", 50), Term(2, ">=", 100)], "drops", ("and", ("or", 0, 1), 2))
    bytes_per_row = 24.25
elif shape in ("dense1", "dense3"):
    # dense selections (90 % survive): the launches after the first are sized from the selectivity the predicate had
    # (dense1: fused_direct_compact<1,16>; dense3: three projected columns, fused_direct_compact<3,4>)
    x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    cols, pred, proj = [x], Predicate([Term(0, ">", 99)]), [0]
    bytes_per_row = 8.0 + 0.9 * 8.0  # read once + the survivors written (the write side matters here)
    if shape == "dense3":
        cols += [ctx.generate(synth_spec(RV_INT64, seed=46, length=n)), ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n))]
        proj, bytes_per_row = [0, 1, 2], 24.0 + 0.9 * 24.0
elif shape in ("wide5", "wide9", "wide5_dense", "wide9_dense", "wide9n", "wide9n_half", "wide9n_dense"):
    # the eager Filter keeps EVERY column (plan.rs:132-147): x > t -> [x, c1 .. ck] over a wide frame; more than four 8-byte
    # columns are compacted in groups of four (query.hip, filter_by_groups), the later groups by the selection bitmap
    n = 200_000_000
    k = 5 if shape.startswith("wide5") else 9
    lit = 499 if shape.endswith("_half") else (899 if not shape.endswith("_dense") else 159)  # 10 % / 50 % / 84 % (BASELINE configs[0]'s age > 25 keeps 84 %)
    nullable = shape.startswith("wide9n")  # c1 .. c8 carry null bitmaps (as every column of a CSV scan does)
    cols = [ctx.generate(synth_spec(RV_INT64, seed=42, length=n))]
    cols += [ctx.generate(synth_spec(RV_INT64 if j % 2 else RV_FLOAT64, seed=50 + j, length=n, validity_seed=(70 + j) if nullable else None)) for j in range(1, k)]
    pred, proj = Predicate([Term(0, ">", lit)]), list(range(k))
    sel = (999 - lit) / 1000.0
    bytes_per_row = 8.0 * k * (1 + sel)
elif shape.rstrip("0123456789km") in ("iid", "sorted", "sorteddesc", "clustered"):
    # one Int64 column, BASELINE configs[1]'s query, over data that is NOT independent rows: the reference filters whatever order
    # its source has (plan.rs:112-147), and ids / timestamps come sorted or in runs
    n = 1_000_000_000
    kind = shape.rstrip("0123456789km")
    tail = shape[len(kind):]
    run = 1_000 if tail.endswith("k") else (10_000_000 if tail.endswith("m") else 100_000)
    pct = int(tail.rstrip("km"))
    lit = {10: 899, 50: 499, 84: 159}[pct]
    kw = {"iid": {}, "sorted": dict(pattern="sorted"), "sorteddesc": dict(pattern="sorted_desc"), "clustered": dict(pattern="clustered", run_rows=run)}[kind]
    cols, pred, proj = [ctx.generate(synth_spec(RV_INT64, seed=42, length=n, **kw))], Predicate([Term(0, ">", lit)]), [0]
    bytes_per_row = 8.0  # the HBM-read roofline, as BASELINE words it (the survivors' 8 * selectivity bytes are written on top)
else:
    raise SystemExit(f"unknown shape {shape}")

for _ in range(2):
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
print(json.dumps({"shape": shape, "kernel": ctx.last_kernel(), "rows": n, "survivors": rows, "call_ms": wall, "fused_kernel_ms": ms / max(1, reps), "rows_per_s": n / wall * 1e3,
                  "last_redo_ppm": ctx.get_option("last_redo_ppm"), "overflow_reruns": ctx.get_option("overflow_reruns"),
                  "algorithmic_read_bytes_per_row": bytes_per_row, "read_GBps_of_call": bytes_per_row * n / wall / 1e6,
                  "frac_of_8TBps_call": bytes_per_row * n / wall / 1e6 / 8000}), flush=True)
