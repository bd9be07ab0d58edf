"""BASELINE configs[3] / configs[4] at G = 1 and their full size (-m gpu): ONE 1e10-row Int64 table -- more than 2^32 rows
and 80 GB on one MI355X -- through rv_filter_project, rv_filter_agg, rv_eval_predicate and, cut into two 5e9-row shards
on the same device, through rv_group_*.

No host can hold the table, so the checker streams it: oracle.synth_filter_checksums walks the generator on the host
threads and returns the exact COUNT, the wrapping SUM and an ORDER checksum sum(ordinal * value) mod 2^64 of the survivors
(the reference emits them in ascending row order, src/execution/record_batch.rs:235-240).  On top of that: oracle-exact
windows of the compacted output at the start, across the 2^32-row boundary of the input, at the shard boundary and at the
end, and the identity SUM/COUNT(filter(input)) == SUM/COUNT(output)."""
import os

import numpy as np
import pytest

from rivulus_amd import capi
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec

pytestmark = pytest.mark.gpu

N = int(float(os.environ.get("RV_HUGE_ROWS", "1e10")))
SEED, MOD, LIT = 42, 1000, 899
PRED = Predicate([Term(0, ">", LIT)])
WINDOW = 1 << 20


@pytest.fixture(scope="module")
def exact(oracle):
    """(sum, count, order checksum) of the survivors over all N rows, from the CPU."""
    return oracle.synth_filter_checksums(SEED, 0, N, MOD, LIT)


def _checksums(values: np.ndarray):
    """sum and sum(ordinal * value) mod 2^64 of an int64 array, in slabs (no second 8 GB temporary)."""
    s = w = 0
    slab = 1 << 26
    for at in range(0, len(values), slab):
        v = values[at:at + slab].astype(np.uint64)
        s = (s + int(v.sum(dtype=np.uint64))) & (2 ** 64 - 1)
        w = (w + int((np.arange(at, at + len(v), dtype=np.uint64) * v).sum(dtype=np.uint64))) & (2 ** 64 - 1)
    return s, w


def _window_starts():
    """Input-row windows worth an exact compare: start, just before / across / after row 2^32, the 2-rank shard boundary, end."""
    starts = {0, max(0, N - WINDOW)}
    for edge in (1 << 32, capi.shard_range(N, 2, 1)[0]):
        if WINDOW < edge < N - WINDOW:
            starts.update({edge - WINDOW // 2, edge - WINDOW, edge})
    return sorted(starts)


def _check_windows(oracle, out_values: np.ndarray, survivors_before):
    """out_values: the whole compacted output on the host; survivors_before(a): exact survivors among input rows [0, a)."""
    for a in _window_starts():
        w = min(WINDOW, N - a)
        hx = oracle.generate(synth_spec(RV_INT64, seed=SEED, length=w, first_row=a))
        want = oracle.filter_project([hx], PRED, [0])[0].values
        pos = survivors_before(a)
        got = out_values[pos:pos + len(want)]
        assert len(got) == len(want) and np.array_equal(got, want), f"window of input rows [{a}, {a + w}) differs at output row {pos}"


def test_one_context_filter_project_and_aggregate_over_the_whole_table(oracle, exact):
    want_sum, want_count, want_order = exact
    with capi.Context(0) as ctx:
        # (no "out_sizing": since round 5 a table this big gets outputs for what its predicate is known to keep x 1.2 + 2 % of the rows --
        # 80 GB in, ~11 GB out -- so BASELINE configs[3] at G = 1 fits one GPU as it is; rounds 2-4 needed out_sizing=120000 here)
        x = ctx.generate(synth_spec(RV_INT64, seed=SEED, length=N))
        # configs[4]: filter + SUM/COUNT, exact against the streamed CPU value
        si, _, cnt = ctx.filter_agg([x], PRED, 0)
        assert (si, cnt) == (want_sum, want_count)
        # the selection bitmap of N rows and its population count (more than 2^32 bits)
        sel, ecount = ctx.eval_predicate([x], PRED)
        assert ecount == want_count and ctx.boolean_count(sel) == (want_count, N - want_count)
        sel.free()
        # configs[3]: filter + project; count, sum, ORDER checksum and exact windows
        outs, rows, _ = ctx.filter_project([x], PRED, [0])
        assert rows == want_count and ctx.get_option("overflow_reruns") == 0
        # SUM/COUNT(filter(input)) == SUM/COUNT(output): the aggregate kernel over the compacted column, always-true predicate
        osi, _, ocnt = ctx.filter_agg([outs[0]], Predicate([Term(0, ">=", 0)]), 0)
        assert (osi, ocnt) == (want_sum, want_count)
        host = outs[0].download()
        assert host.validity is None and host.length == want_count
        s, w = _checksums(host.values)
        assert s == want_sum % (2 ** 64) and w == want_order, "sum / order checksum of the compacted output"

        def before(a):  # survivors among rows [0, a): the aggregate kernel on a zero-copy slice
            if a == 0:
                return 0
            v = x.slice(0, a)
            c = ctx.filter_agg([v], PRED, 0)[2]
            v.free()
            return c
        _check_windows(oracle, host.values, before)
        # a slice that STARTS past row 2^32 filters like the oracle's window (element offsets beyond 32 bits)
        a = min(N - WINDOW, (1 << 32) + 12_345) if N > (1 << 32) + WINDOW else N // 2
        v = x.slice(a, WINDOW)
        o2, r2, _ = ctx.filter_project([v], PRED, [0])
        hx = oracle.generate(synth_spec(RV_INT64, seed=SEED, length=WINDOW, first_row=a))
        assert o2[0].download().same_as(oracle.filter_project([hx], PRED, [0])[0]) is None and r2 == o2[0].length
        for d in (o2[0], v, outs[0], x):
            d.free()


def test_dense_selection_past_two_to_the_32_rows_through_the_direct_kernel(oracle):
    """90 % of a table of more than 2^32 rows survive: the direct (register-staged) kernel with 64-bit output offsets, a million and
    more tiles, the first call sized from the sample.  Exact COUNT and SUM against the streamed CPU values (the aggregate kernel over
    the compacted column), oracle-exact windows of the output at the start, across input row 2^32 and at the end."""
    n = min(N, 6_000_000_000)
    lit = 99
    pred = Predicate([Term(0, ">", lit)])
    want_sum, want_count, _ = oracle.synth_filter_checksums(SEED, 0, n, MOD, lit)
    with capi.Context(0) as ctx:
        x = ctx.generate(synth_spec(RV_INT64, seed=SEED, length=n))
        outs, rows, _ = ctx.filter_project([x], pred, [0])
        assert ctx.last_kernel().startswith("fused_direct_compact<1,0,"), ctx.last_kernel()  # an unseen predicate: sampled, then sized
        assert rows == want_count and ctx.get_option("last_redo_ppm") == 0
        osi, _, ocnt = ctx.filter_agg([outs[0]], Predicate([Term(0, ">=", 0)]), 0)
        assert (osi, ocnt) == (want_sum, want_count)
        starts = {0, max(0, n - WINDOW)}
        if n > (1 << 32) + WINDOW:
            starts.update({(1 << 32) - WINDOW // 2, 1 << 32})
        for a in sorted(starts):
            w = min(WINDOW, n - a)
            hx = oracle.generate(synth_spec(RV_INT64, seed=SEED, length=w, first_row=a))
            want = oracle.filter_project([hx], pred, [0])[0]
            pos = 0
            if a:
                v = x.slice(0, a)
                pos = ctx.filter_agg([v], pred, 0)[2]  # survivors among input rows [0, a)
                v.free()
            got = outs[0].slice(pos, want.length)
            assert got.download().same_as(want) is None, f"window of input rows [{a}, {a + w}) differs at output row {pos}"
            got.free()
        for d in (outs[0], x):
            d.free()


def test_two_shards_of_the_same_table_through_the_group(oracle, exact):
    want_sum, want_count, want_order = exact
    with capi.Group([0, 0]) as g:
        x = g.generate(synth_spec(RV_INT64, seed=SEED, length=N))
        b1 = capi.shard_range(N, 2, 1)[0]
        assert x.shard(0).length == b1 and x.shard(1).length == N - b1
        # configs[4] over two shards: partials + the reduction of 16 bytes
        si, _, cnt = g.filter_agg([x], PRED, 0)
        assert (si, cnt) == (want_sum, want_count)
        # configs[3]: per-shard pass, prefix sum, rank-order gather into one pinned host buffer
        res, rows = g.filter_project([x], PRED, [0])
        assert rows == want_count
        st = res.stats()
        first = oracle.synth_filter_checksums(SEED, 0, b1, MOD, LIT)[1]
        assert st["rank_rows"] == [first, want_count - first]
        values = res.values_view(0)
        s, w = _checksums(values)
        assert s == want_sum % (2 ** 64) and w == want_order, "sum / order checksum of the gathered output"

        def before(a):  # exact survivors in [0, a) from the per-rank counts + the aggregate kernel inside the rank's shard
            r = 0 if a < b1 else 1
            base, lo = (0, 0) if r == 0 else (first, b1)
            if a == lo:
                return base
            v = x.shard(r).slice(0, a - lo)
            c = g.context(r).filter_agg([v], PRED, 0)[2]
            v.free()
            return base + c
        _check_windows(oracle, values, before)
        res.free()
        x.free()
