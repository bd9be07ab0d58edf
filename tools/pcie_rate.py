"""PCIe-inclusive rate of the hot path: host buffers in, host buffers out.
Never the bench `value` -- that is measured with the rows resident in HBM."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import Column, Predicate, Term

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rng = np.random.default_rng(1)
x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64))
ctx = capi.Context(0)
pred = Predicate([Term(0, ">", 899)])
for rep in range(4):
    t0 = time.perf_counter()
    d = ctx.upload(x)
    t1 = time.perf_counter()
    outs, rows, _ = ctx.filter_project([d], pred, [0])
    ctx.synchronize()
    t2 = time.perf_counter()
    got = outs[0].download()
    t3 = time.perf_counter()
    d.free(); [o.free() for o in outs]
    print(f"rep {rep}: n={n} rows={rows} upload {1e3*(t1-t0):.1f} ms ({8*n/(t1-t0)/1e9:.1f} GB/s)  filter+project {1e3*(t2-t1):.2f} ms  "
          f"download {1e3*(t3-t2):.1f} ms  end to end {n/(t3-t0):.3e} rows/s", flush=True)

# chunked pipeline (rv_filter_project_host): pageable and pinned source buffers
for label, arr in (("pageable", x), ("pinned", None)):
    if arr is None:
        px = ctx.pinned_array(np.int64, n)
        px[:] = x.values
        arr = Column.from_numpy(px)
    for chunk in (1 << 22, 1 << 24, 1 << 26):
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            outs, rows = ctx.filter_project_host([arr], pred, [0], chunk)
            ctx.synchronize()
            t1 = time.perf_counter()
            got = outs[0].download()
            t2 = time.perf_counter()
            [o.free() for o in outs]
            best = min(best, (t2 - t0, t1 - t0)) if best else (t2 - t0, t1 - t0)
        print(f"pipeline {label:8s} chunk {chunk:>9d}: in+filter {1e3*best[1]:.1f} ms ({8*n/best[1]/1e9:.1f} GB/s), "
              f"with download {1e3*best[0]:.1f} ms = {n/best[0]:.3e} rows/s", flush=True)
