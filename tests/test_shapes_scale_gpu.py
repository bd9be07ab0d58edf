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


def test_nine_column_frame(gpu_ctx, oracle, table):
    (f, fv, x, xv, y, b, bv), host, dev = table
    _check(gpu_ctx, oracle, host, dev, Predicate([Term(2, ">", 899)]), [2, 0, 1, 2, 0, 1, 2, 0, 1], y > 899, "nine columns, 10 %")
    _check(gpu_ctx, oracle, host, dev, Predicate([Term(2, ">", 159)]), [2, 0, 1, 2, 0], y > 159, "five columns, 84 %")


def test_filter_sum_count(gpu_ctx, oracle, table):
    (f, fv, x, xv, y, b, bv), host, dev = table
    keep = xv & (x < 200) & (y >= 100)
    s, _, cnt = gpu_ctx.filter_agg(dev, Predicate([Term(1, "<", 200), Term(2, ">=", 100)]), 2)
    assert cnt == int(keep.sum()) and s == int(y[keep].sum())
