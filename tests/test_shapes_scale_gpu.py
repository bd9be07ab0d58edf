"""BASELINE configs[2]'s predicate, an OR / NOT expression, RecordBatch::filter by a BooleanArray, a nine-column frame and the
filter + SUM/COUNT at 1e8 rows against numpy (-m gpu).  The oracle (a C++ restatement walking cells) checks these shapes up to a few
1e6 rows; here numpy restates each predicate in one line, is itself checked against the oracle on the first 300 000 rows, and then
stands in for it over ~50 000 tiles: every tile-offset, look-back and multi-pass path at a size where each has many rounds."""
import numpy as np
import pytest

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
