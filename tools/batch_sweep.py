"""Throughput of collect_streaming()-style execution against the RecordBatch size (device-resident batches:
zero-copy slices of one 1e9-row column, one rv_filter_project per batch).  The reference's default batch is
1024 rows (memory_stream.rs); on the GPU the per-launch cost (descriptor memset + kernel launch + 384-byte
read-back, ~25 us) sets the floor, so the streaming layer wants batches of >= 1e7 rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, RV_FLOAT64, Predicate, Term, synth_spec

n = 1_000_000_000
ctx = capi.Context(0)
x = ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
pred = Predicate([Term(0, ">", 899)])
for b in [1024, 65536, 1 << 20, 1 << 24, 1 << 26, 1 << 28, n]:
    nb = min((n + b - 1) // b, 2000)  # bounded number of launches
    w = x.slice(0, min(b, n)); outs, _, _ = ctx.filter_project([w], pred, [0]); [o.free() for o in outs]; w.free()  # warm the buffer pool
    ctx.synchronize()
    t0 = time.perf_counter()
    total = 0
    for i in range(nb):
        s = x.slice(i * b, min(b, n - i * b))
        outs, rows, _ = ctx.filter_project([s], pred, [0])
        total += rows
        for o in outs:
            o.free()
        s.free()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    done = min(n, nb * b)
    # the same batches with two launches in flight (rv_filter_project_begin / _finish)
    t0 = time.perf_counter()
    q = []
    for i in range(nb):
        s = x.slice(i * b, min(b, n - i * b))
        q.append((s, ctx.filter_project_begin([s], pred, [0])))
        if len(q) > 2:
            s0, fin = q.pop(0)
            outs, rows = fin()
            for o in outs:
                o.free()
            s0.free()
    for s0, fin in q:
        outs, rows = fin()
        for o in outs:
            o.free()
        s0.free()
    ctx.synchronize()
    dtp = time.perf_counter() - t0
    print(f"batch {b:>10d} rows: {nb:5d} launches, {dt/nb*1e6:9.1f} us/batch, {done/dt:.3e} rows/s | two in flight: "
          f"{dtp/nb*1e6:9.1f} us/batch, {done/dtp:.3e} rows/s", flush=True)

# BASELINE config 3 through the same seam: (f > 0.5) AND (x < 200) over nullable Float64 + Int64, R rows per batch
del x
n3 = 500_000_000
f = ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n3, validity_seed=44))
xv = ctx.generate(synth_spec(RV_INT64, seed=42, length=n3, validity_seed=45))
pred3 = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
for b in [1024, 1 << 20, 1 << 26, n3]:
    nb = min((n3 + b - 1) // b, 1000)
    for warm in (True, False):
        t0 = time.perf_counter()
        for i in range(1 if warm else nb):
            sf, sx = f.slice(i * b, min(b, n3 - i * b)), xv.slice(i * b, min(b, n3 - i * b))
            outs, rows, _ = ctx.filter_project([sf, sx], pred3, [0, 1])
            for o in outs:
                o.free()
            sf.free(); sx.free()
        ctx.synchronize()
        dt = time.perf_counter() - t0
    done = min(n3, nb * b)
    print(f"config 3 batch {b:>10d} rows: {nb:5d} launches, {dt/nb*1e6:9.1f} us/batch, {done/dt:.3e} rows/s", flush=True)
