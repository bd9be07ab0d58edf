"""BASELINE configs[2]'s predicate, an OR / NOT expression, RecordBatch::filter by a BooleanArray, a nine-column frame and the
filter + SUM/COUNT at 1e8 rows against numpy (-m gpu).  The oracle (a C++ restatement walking cells) checks these shapes up to a few
1e6 rows; here numpy restates each predicate in one line, is itself checked against the oracle on the first 300 000 rows, and then
stands in for it over ~50 000 tiles: every tile-offset, look-back and multi-pass path at a size where each has many rounds."""
import numpy as np
import pytest

from rivulus_amd import capi
from rivulus_amd.capi import Column, Predicate, Term

pytestmark = pytest.mark.gpu

N = 100_000_000
W = 300_000  # rows of the window on which numpy's restatement is pinned to the oracle


@pytest.fixture(scope="module")
def table(gpu_ctx):
    rng = np.random.default_rng(31337)
    f, fv = rng.random(N), rng.random(N) > 0.07
    x, xv = rng.integers(0, 1000, N).astype(np.int64), rng.random(N) > 0.05
    y = rng.integers(0, 1000, N).astype(np.int64)
    b, bv = rng.random(N) > 0.85, rng.random(N) > 0.1
    host = [Column.from_numpy(f, fv), Column.from_numpy(x, xv), Column.from_numpy(y), Column.from_numpy(b, bv)]
    dev = [gpu_ctx.upload(c) for c in host]
    yield (f, fv, x, xv, y, b, bv), host, dev
    for d in dev:
        d.free()


def _check(gpu_ctx, oracle, host, dev, pred, proj, keep, what):
    """keep: numpy's survivors.  Pinned to the oracle on the first W rows, then the GPU against numpy on all of them."""
    window = [c.slice(0, W) for c in host]
    assert oracle.eval_predicate(window, pred)[1] == int(keep[:W].sum()), f"{what}: numpy's restatement disagrees with the oracle"
    for call in range(2):
        outs, rows, _ = gpu_ctx.filter_project(dev, pred, proj)
        assert rows == int(keep.sum()), f"{what} call {call} ({gpu_ctx.last_kernel()})"
        for o, j in zip(outs, proj):
            col, src = o.download(), host[j]
            valid = src.logical_valid()
            if valid is not None and not valid[keep].all():
                assert np.array_equal(col.logical_valid(), valid[keep]), f"{what}: validity of column {j}, call {call}"
                want = np.where(valid[keep], src.logical_values()[keep], 0)
            else:
                assert col.validity is None, f"{what}: column {j} kept a bitmap without a null"
                want = src.logical_values()[keep]
            assert np.array_equal(col.logical_values() if col.dtype == 1 else col.values[:rows], want), f"{what}: column {j}, call {call} ({gpu_ctx.last_kernel()})"
            o.free()


def test_config2_predicate_both_null_policies(gpu_ctx, oracle, table):
    (f, fv, x, xv, y, b, bv), host, dev = table
    terms = [Term(0, ">", 0.5), Term(1, "<", 200)]
    _check(gpu_ctx, oracle, host, dev, Predicate(terms), [0, 1], fv & xv & (f > 0.5) & (x < 200), "drops")
    # eager ordering: Null is least (series.rs:105-107) -- a null f is not > 0.5, a null x is < 200
    _check(gpu_ctx, oracle, host, dev, Predicate(terms, "least"), [0, 1], (fv & (f > 0.5)) & (~xv | (x < 200)), "least")


def test_or_not_expression(gpu_ctx, oracle, table):
    (f, fv, x, xv, y, b, bv), host, dev = table
    terms = [Term(0, ">", 0.9), Term(1, "<", 50), Term(2, ">=", 100)]
    # strict null propagation (boolean.rs:120-165): a null on either side of OR is null, and null rows are dropped
    _check(gpu_ctx, oracle, host, dev, Predicate(terms, "drops", ("and", ("or", 0, 1), 2)), [0, 1], fv & xv & ((f > 0.9) | (x < 50)) & (y >= 100), "or")
    _check(gpu_ctx, oracle, host, dev, Predicate(terms, "drops", ("and", ("not", ("or", 0, 1)), 2)), [1, 2], fv & xv & ~((f > 0.9) | (x < 50)) & (y >= 100), "not or")


def test_filter_by_a_boolean_column(gpu_ctx, oracle, table):
    (f, fv, x, xv, y, b, bv), host, dev = table
    _check(gpu_ctx, oracle, host, dev, Predicate([Term(3, "is_true")]), [1, 3, 2], b & bv, "b is true")
    _check(gpu_ctx, oracle, host, dev, Predicate([Term(3, "is_true"), Term(2, "<", 700)]), [0, 2], b & bv & (y < 700), "b is true and y < 700")


def test_filter_by_a_boolean_column_without_a_pass(gpu_ctx, oracle, table):
    """`mask is true -> value columns` over a big table: no chained pass at all -- mask_select_kernel + a scan of its counts, then every
    column at those offsets (compact_ranges_kernel; validity bits by bits_compact_kernel)."""
    (f, fv, x, xv, y, b, bv), host, dev = table
    _check(gpu_ctx, oracle, host, dev, Predicate([Term(3, "is_true")]), [1, 2, 0], b & bv, "b is true, three columns")
    assert gpu_ctx.last_kernel().startswith("compact_ranges_kernel"), gpu_ctx.last_kernel()
    _check(gpu_ctx, oracle, host, dev, Predicate([Term(3, "is_true")]), [2, 0, 1, 2, 0, 1], b & bv, "b is true, six columns")


def test_nine_column_frame(gpu_ctx, oracle, table):
    (f, fv, x, xv, y, b, bv), host, dev = table
    _check(gpu_ctx, oracle, host, dev, Predicate([Term(2, ">", 899)]), [2, 0, 1, 2, 0, 1, 2, 0, 1], y > 899, "nine columns, 10 %")
    _check(gpu_ctx, oracle, host, dev, Predicate([Term(2, ">", 159)]), [2, 0, 1, 2, 0], y > 159, "five columns, 84 %")


def test_filter_sum_count(gpu_ctx, oracle, table):
    (f, fv, x, xv, y, b, bv), host, dev = table
    keep = xv & (x < 200) & (y >= 100)
    s, _, cnt = gpu_ctx.filter_agg(dev, Predicate([Term(1, "<", 200), Term(2, ">=", 100)]), 2)
    assert cnt == int(keep.sum()) and s == int(y[keep].sum())


def test_two_windows_in_flight_through_the_seam(gpu_ctx, oracle, table):
    """rv_filter_project_chunked_begin / _batches_begin + rv_filter_project_window_finish: window i + 1 is begun before window i is
    finished (what a stream operator that looks ahead does, stream.rs:25-28), its per-batch counts written by the device into the
    operator's pinned array on a side stream.  Config 3's shape at the reference's 1024-row batches, ten windows of 4 Mi rows in both
    forms, against numpy; then a window the walk does NOT confirm (the ordinary path at finish) and one with an error in batch 7."""
    (f, fv, x, xv, y, b, bv), host, dev = table
    rows_per, wrows, nwin = 1024, 4 * 1024 * 1024, 10
    keep = fv & xv & (f > 0.5) & (x < 200)
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
    bufs = [gpu_ctx.pinned_array(np.uint64, wrows // rows_per) for _ in range(2)]
    windows = [[dev[0].slice(w * wrows, wrows), dev[1].slice(w * wrows, wrows)] for w in range(nwin)]
    batch_lists = [[[c.slice(o, rows_per) for c in win] for o in range(0, wrows, rows_per)] for win in windows]  # (kept: the handles are theirs)
    handle_sets = [gpu_ctx.batch_handles(bl) for bl in batch_lists]
    for form in ("chunked", "handles"):
        spec = gpu_ctx.get_option("speculative_batch_passes")
        pending = []

        def begin(w):
            if form == "chunked":
                return gpu_ctx.window_begin(pred, [0, 1], bufs[w % 2], cols=windows[w], chunk_rows=rows_per)
            return gpu_ctx.window_begin(pred, [0, 1], bufs[w % 2], handles=handle_sets[w])
        pending.append(begin(0))
        for w in range(nwin):
            if w + 1 < nwin:
                pending.append(begin(w + 1))  # in flight before window w's counts are read
            outs, rows, nulls, total = pending.pop(0)()
            kw = keep[w * wrows:(w + 1) * wrows]
            assert total == int(kw.sum()) and np.array_equal(rows, kw.reshape(-1, rows_per).sum(axis=1).astype(np.uint64)), (form, w)
            assert np.array_equal(outs[0].download().values[:total], f[w * wrows:(w + 1) * wrows][kw]), (form, w)
            assert np.array_equal(outs[1].download().values[:total], x[w * wrows:(w + 1) * wrows][kw]), (form, w)
            assert not nulls.any(), (form, w)
            [o.free() for o in outs]
        if form == "handles":
            assert gpu_ctx.get_option("speculative_batch_passes") == spec + nwin
    # the reference's own streaming filter (a Boolean column) with two windows in flight: the first window of a predicate is sized by
    # waiting for its count (inside begin), the following ones are queued whole -- mask_select_kernel, the scan, the compaction into
    # outputs sized from what the window before kept -- and the count is read at finish
    keepb = b & bv
    wb = [[dev[3].slice(w * wrows, wrows), dev[2].slice(w * wrows, wrows), dev[1].slice(w * wrows, wrows)] for w in range(nwin)]
    predb = Predicate([Term(0, "is_true")])
    gpu_ctx.set_option("groups_by_ranges", 1)  # (4 Mi-row windows are below the size from which the mask path is taken by itself)
    try:
        reruns = gpu_ctx.get_option("overflow_reruns")
        wb_lists = [[[c.slice(o, rows_per) for c in win] for o in range(0, wrows, rows_per)] for win in wb[:4]]
        wb_handles = [gpu_ctx.batch_handles(bl) for bl in wb_lists]
        for form, count in (("chunked", nwin), ("handles", 4)):
            def beginb(w):
                if form == "chunked":
                    return gpu_ctx.window_begin(predb, [1], bufs[w % 2], cols=wb[w], chunk_rows=rows_per)
                return gpu_ctx.window_begin(predb, [1], bufs[w % 2], handles=wb_handles[w])
            pending = [beginb(0)]
            for w in range(count):
                if w + 1 < count:
                    pending.append(beginb(w + 1))
                outs, rows, nulls, total = pending.pop(0)()
                kw = keepb[w * wrows:(w + 1) * wrows]
                assert total == int(kw.sum()) and np.array_equal(rows, kw.reshape(-1, rows_per).sum(axis=1).astype(np.uint64)), ("bool", form, w)
                assert np.array_equal(outs[0].download().values[:total], y[w * wrows:(w + 1) * wrows][kw]) and not nulls.any(), ("bool", form, w)
                assert not gpu_ctx.last_kernel().startswith("fused_"), gpu_ctx.last_kernel()
                [o.free() for o in outs]
        assert gpu_ctx.get_option("overflow_reruns") == reruns
        # a window that keeps far more than the one before it (5 % -> 45 %): the outputs queued for 8 % of the rows do not hold it, the
        # count is exact, the compaction runs once more with outputs of that size
        rngb = np.random.default_rng(12)
        jump = np.concatenate([rngb.random(wrows) < 0.05, rngb.random(wrows) < 0.45])
        jd = gpu_ctx.upload(Column.from_numpy(jump))
        for w in range(2):
            fin = gpu_ctx.window_begin(predb, [1], bufs[w], cols=[jd.slice(w * wrows, wrows), dev[2].slice(w * wrows, wrows)], chunk_rows=rows_per)
            outs, rows, nulls, total = fin()
            kw = jump[w * wrows:(w + 1) * wrows]
            assert total == int(kw.sum()) and np.array_equal(rows, kw.reshape(-1, rows_per).sum(axis=1).astype(np.uint64)), ("jump", w)
            assert np.array_equal(outs[0].download().values[:total], y[w * wrows:(w + 1) * wrows][kw]), ("jump", w)
            [o.free() for o in outs]
        assert gpu_ctx.get_option("overflow_reruns") == reruns + 1
        jd.free()
    finally:
        gpu_ctx.set_option("groups_by_ranges", 0)
    # a nullable column that keeps its nulls: null counts per output batch at finish
    fin = gpu_ctx.window_begin(Predicate([Term(1, "<", 200)]), [0, 1], bufs[0], cols=windows[0], chunk_rows=rows_per)
    outs, rows, nulls, total = fin()
    kw = (xv & (x < 200))[:wrows]
    inv = (~fv[:wrows][kw]).astype(np.int64)
    bounds = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    assert total == int(kw.sum()) and np.array_equal(nulls[:, 0], np.add.reduceat(np.concatenate([inv, [0]]), bounds[:-1]) * (rows > 0)) and not nulls[:, 1].any()
    [o.free() for o in outs]
    # a window the walk does not confirm: batch 9 is shorter -- finish takes the ordinary path, same answer as the synchronous call
    small = [[c.slice(o, rows_per) for c in windows[0]] for o in range(0, 4000 * rows_per, rows_per)]
    small[9] = [c.slice(9 * rows_per, rows_per - 100) for c in windows[0]]
    buf = gpu_ctx.pinned_array(np.uint64, 4000)
    fin = gpu_ctx.window_begin(pred, [0, 1], buf, handles=gpu_ctx.batch_handles(small))
    outs, rows, nulls, total = fin()
    outs2, rows2, nulls2, total2 = gpu_ctx.filter_project_batches(small, pred, [0, 1])
    assert total == total2 and np.array_equal(rows, rows2) and np.array_equal(nulls, nulls2)
    assert outs[0].download().same_as(outs2[0].download()) is None and outs[1].download().same_as(outs2[1].download()) is None
    [o.free() for o in outs + outs2]
    # ... and one the walk rejects: the FIRST offending batch's error, at finish (record_batch.rs:31-40)
    small[7] = [windows[0][0].slice(7 * rows_per, rows_per), windows[0][1].slice(7 * rows_per, rows_per - 1)]
    fin = gpu_ctx.window_begin(pred, [0, 1], buf, handles=gpu_ctx.batch_handles(small))
    with pytest.raises(capi.RvError) as e:
        fin()
    assert e.value.message == "Column 1 has length 1023 but expected 1024"


def test_the_references_streaming_filter_through_the_seam_on_the_mask_path(gpu_ctx, oracle, table):
    """The ONE predicate collect_streaming() accepts is a Boolean column (streaming_planner.rs:137-168 -> FilterStream, stream.rs:136-158
    -> RecordBatch::filter, record_batch.rs:221-243), in 1024-row batches: 48 829 of them through rv_filter_project_batches (handles)
    and rv_filter_project_chunked.  No chained pass: mask_select_kernel's counts per 1024 rows ARE the per-batch survivor counts, every
    column follows at the scan's offsets -- plain, nullable (null counts per output batch), Boolean and String columns riding along.
    Then windows that are NOT regular: the ordinary path, same results."""
    (f, fv, x, xv, y, b, bv), host, dev = table
    n, rows_per = 50_000_000, 1024
    keep = (b & bv)[:n]
    want_rows = np.add.reduceat(keep.astype(np.uint64), np.arange(0, n, rows_per))
    pred = Predicate([Term(3, "is_true")])
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 9, n).astype(np.int32)
    offs = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=offs[1:])
    name_h = Column(4, rng.integers(97, 123, int(offs[-1])).astype(np.uint8), None, 0, n, offs)
    name_d = gpu_ctx.upload(name_h)
    cols_d = [d.slice(0, n) for d in dev] + [name_d]
    batches = [[c.slice(o, min(rows_per, n - o)) for c in cols_d] for o in range(0, n, rows_per)]
    handles = gpu_ctx.batch_handles(batches)

    def check(outs, rows, nulls, total, proj, what):
        assert total == int(keep.sum()) and np.array_equal(np.asarray(rows, dtype=np.uint64), want_rows), what
        for o, j in zip(outs, proj):
            col = o.download()
            if j == 4:
                idx = np.flatnonzero(keep)
                assert np.array_equal(col.offsets[1:total + 1] - col.offsets[:total], lens[idx]), what
                for k in range(0, total, 997 * 13):
                    i = idx[k]
                    assert bytes(col.values[col.offsets[k]:col.offsets[k + 1]]) == bytes(name_h.values[offs[i]:offs[i + 1]]), (what, k)
            elif j == 3:  # b itself: all true, never null among the survivors (record_batch.rs:237)
                assert col.validity is None and col.logical_values().all(), what
            else:
                src = host[j].slice(0, n)
                valid = src.logical_valid()
                if valid is not None:
                    assert np.array_equal(col.logical_valid(), valid[keep]), (what, j)
                    assert np.array_equal(col.values[:total], np.where(valid[keep], src.logical_values()[keep], 0)), (what, j)
                    got = np.asarray([row[proj.index(j)] for row in nulls[:3000]])
                    bounds = np.concatenate([[0], np.cumsum(want_rows[:3000])]).astype(np.int64)
                    inv = (~valid[keep]).astype(np.int64)
                    want_nulls = np.add.reduceat(np.concatenate([inv[:bounds[-1]], [0]]), bounds[:-1]) * (want_rows[:3000] > 0)
                    assert np.array_equal(got, want_nulls), (what, j)
                else:
                    assert col.validity is None and np.array_equal(col.values[:total], src.logical_values()[keep]), (what, j)
            o.free()

    try:
        for proj in ([2], [2, 0], [1, 2, 4], [2, 3]):  # -> [y] | [y, fn] | [xn, y, name] | [y, b]
            before, spec = gpu_ctx.get_option("batch_counts_in_pass"), gpu_ctx.get_option("speculative_batch_passes")
            outs, rows, nulls, total = gpu_ctx.filter_project_batches(None, pred, proj, handles=handles)
            kern = gpu_ctx.last_kernel()
            check(outs, rows, nulls, total, proj, f"handles {proj}")
            assert gpu_ctx.get_option("speculative_batch_passes") == spec + 1
            outs, rows, nulls, total = gpu_ctx.filter_project_chunked(cols_d, rows_per, pred, proj)
            check(outs, rows, nulls, total, proj, f"chunked {proj}")
            assert gpu_ctx.get_option("batch_counts_in_pass") == before + 2, proj  # both forms took their counts from mask_select_kernel
            assert not kern.startswith("fused_"), (proj, kern)  # no chained pass ran
        # an irregular window: one batch is a handle of its own buffers, one is shorter than the rest -- the walk says so, the
        # speculative result is dropped and the ordinary path concatenates (record_batch.rs:277-342)
        m = 6000
        small = [[c.slice(o, rows_per) for c in cols_d[:4]] for o in range(0, m * rows_per, rows_per)]
        odd = [gpu_ctx.upload(host[j].slice(5 * rows_per, rows_per - 24)) for j in range(4)]
        small[5] = odd
        k2 = np.concatenate([keep[:5 * rows_per], keep[5 * rows_per:6 * rows_per - 24], keep[6 * rows_per:m * rows_per]])
        yv = np.concatenate([y[:5 * rows_per], y[5 * rows_per:6 * rows_per - 24], y[6 * rows_per:m * rows_per]])
        spec = gpu_ctx.get_option("speculative_batch_passes")
        outs, rows, nulls, total = gpu_ctx.filter_project_batches(small, pred, [2])
        assert gpu_ctx.get_option("speculative_batch_passes") == spec
        assert total == int(k2.sum()) and int(rows[5]) == int(keep[5 * rows_per:6 * rows_per - 24].sum()) and int(rows[6]) == int(want_rows[6])
        assert np.array_equal(outs[0].download().values[:total], yv[k2])
        outs[0].free()
        [o.free() for o in odd]
    finally:
        name_d.free()


def test_a_wide_frame_through_the_seam(gpu_ctx, oracle, table):
    """The eager Filter's shape (every column kept, plan.rs:132-147) as a stream's window: x > t -> nine columns, five of them nullable, in
    1024-row batches through rv_filter_project_chunked.  The pass reads what the predicate reads and counts the survivors of every
    batch; every other column follows at its wave offsets (round 5: windows too); the null counts per output batch come from the
    compacted outputs."""
    (f, fv, x, xv, y, b, bv), host, dev = table
    n, rows_per = 40_000_000, 1024
    proj = [2, 0, 1, 2, 0, 1, 2, 0, 1]
    for lit, what in ((899, "10 %"), (399, "60 %")):
        keep = (y > lit)[:n]
        want_rows = np.add.reduceat(keep.astype(np.uint64), np.arange(0, n, rows_per))
        outs, rows, nulls, total = gpu_ctx.filter_project_chunked([d.slice(0, n) for d in dev[:3]], rows_per, Predicate([Term(2, ">", lit)]), proj)
        assert total == int(keep.sum()) and np.array_equal(np.asarray(rows, dtype=np.uint64), want_rows), what
        bounds = np.concatenate([[0], np.cumsum(want_rows[:2000])]).astype(np.int64)
        for o, j in zip(outs, proj):
            col, src = o.download(), host[j].slice(0, n)
            valid = src.logical_valid()
            if valid is not None:
                assert np.array_equal(col.logical_valid(), valid[keep]) and np.array_equal(col.values[:total], np.where(valid[keep], src.logical_values()[keep], 0)), (what, j)
            else:
                assert col.validity is None and np.array_equal(col.values[:total], src.logical_values()[keep]), (what, j)
            o.free()
        for k, j in enumerate(proj):
            valid = host[j].slice(0, n).logical_valid()
            inv = np.zeros(bounds[-1] + 1, dtype=np.int64) if valid is None else np.concatenate([(~valid[keep][:bounds[-1]]).astype(np.int64), [0]])
            assert np.array_equal(np.asarray([row[k] for row in nulls[:2000]]), np.add.reduceat(inv, bounds[:-1]) * (want_rows[:2000] > 0)), (what, j)


def test_config3_through_the_stream_seam_at_the_references_batch_size(gpu_ctx, oracle, table):
    """Seam S1 at 1024-row batches (FilterStream over MemoryStream, stream.rs:58-163): 48 829 handles per column in ONE window of
    rv_filter_project_batches -- launched speculatively, validated by the handle walk meanwhile -- and the chunked form over the same
    rows; per-batch survivor counts and the concatenated outputs against numpy."""
    (f, fv, x, xv, y, b, bv), host, dev = table
    n, rows_per = 50_000_000, 1024
    keep = (fv & xv & (f > 0.5) & (x < 200))[:n]
    want_rows = np.add.reduceat(keep.astype(np.uint64), np.arange(0, n, rows_per))
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
    batches = [[dev[0].slice(o, min(rows_per, n - o)), dev[1].slice(o, min(rows_per, n - o))] for o in range(0, n, rows_per)]
    before = gpu_ctx.get_option("speculative_batch_passes")
    for form in ("handles", "chunked"):
        if form == "handles":
            outs, rows, nulls, total = gpu_ctx.filter_project_batches(batches, pred, [0, 1])
            assert gpu_ctx.get_option("speculative_batch_passes") == before + 1
        else:
            outs, rows, nulls, total = gpu_ctx.filter_project_chunked([dev[0].slice(0, n), dev[1].slice(0, n)], rows_per, pred, [0, 1])
        assert total == int(keep.sum()) and np.array_equal(np.asarray(rows, dtype=np.uint64), want_rows), form
        assert np.array_equal(outs[0].download().values[:total], f[:n][keep]) and np.array_equal(outs[1].download().values[:total], x[:n][keep]), form
        assert all(int(v) == 0 for row in nulls[:2000] for v in row), form  # both columns are tested by terms that drop their nulls
        for o in outs:
            o.free()
