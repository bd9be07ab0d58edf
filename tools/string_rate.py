"""Rate of filter+project carrying a String column: selection -> indices -> lengths ->
scan -> byte gather.  5e7 rows, strings of 0..16 bytes, 10 % selectivity, device-resident inputs."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_STRING, Column, Predicate, Term

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
rng = np.random.default_rng(5)
lens = rng.integers(0, 17, n).astype(np.int32)
offs = np.zeros(n + 1, dtype=np.int32)
np.cumsum(lens, out=offs[1:])
data = rng.integers(97, 123, int(offs[-1])).astype(np.uint8)
name = Column(RV_STRING, data, None, 0, n, offs)
x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64))
ctx = capi.Context(0)
d = [ctx.upload(x), ctx.upload(name)]
pred = Predicate([Term(0, ">", 899)])
for proj, label in (([0], "Int64 only"), ([1], "String only"), ([0, 1], "Int64 + String")):
    for rep in range(3):
        t0 = time.perf_counter()
        outs, rows, _ = ctx.filter_project(d, pred, proj)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        [o.free() for o in outs]
    in_bytes = 8 * n + (0 if proj == [0] else 4 * n + int(offs[-1]))
    print(f"{label:16s}: {1e3*dt:8.2f} ms  {n/dt:.3e} rows/s  ({rows} survivors, {in_bytes/1e9:.2f} GB of input columns)", flush=True)
