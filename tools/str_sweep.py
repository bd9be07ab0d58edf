"""x > t -> [x, name] over 2e8 rows (strings of 0..16 bytes) at several selectivities: String side in block order (sel_str_lengths +
str_gather_copy: option str_tiles_from = -1) against source-tile order (sel_str_tile_sums + sel_str_tile_copy: 1), alternating on
one box.  Where the default (30 %) should sit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_STRING, Column, Predicate, Term  # noqa: E402

ctx = capi.Context(0)
n = 200_000_000
rng = np.random.default_rng(5)
lens = rng.integers(0, 17, n).astype(np.int32)
offs = np.zeros(n + 1, dtype=np.int32)
np.cumsum(lens, out=offs[1:])
data = rng.integers(97, 123, int(offs[-1])).astype(np.uint8)
cols = [ctx.upload(Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64))), ctx.upload(Column(RV_STRING, data, None, 0, n, offs))]


def timed(pred, reps=8):
    for _ in range(3):
        outs, rows, _ = ctx.filter_project(cols, pred, [0, 1])
        [o.free() for o in outs]
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        outs, rows, _ = ctx.filter_project(cols, pred, [0, 1])
        [o.free() for o in outs]
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, rows


for lit in (899, 799, 699, 599, 499, 299, 159, 49):
    pred = Predicate([Term(0, ">", lit)])
    out = {}
    for mode in (-1, 1, -1, 1):
        ctx.set_option("str_tiles_from", mode)
        ms, rows = timed(pred)
        out.setdefault(mode, []).append(ms)
    print(f"keep {rows / n:5.2f}  block order {min(out[-1]):6.3f} ms   tile order {min(out[1]):6.3f} ms   ({ctx.last_kernel()})", flush=True)
ctx.set_option("str_tiles_from", 0)
