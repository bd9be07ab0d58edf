"""Boolean columns riding along a filter at a size where bits_compact_kernel's workgroups stride over many chunks (-m gpu): 2e8
rows (the oracle-checked tests stop at 1e6, one chunk per workgroup), nullable and plain columns, off the word grid by a slice,
at a sparse and a dense selectivity -- the loop over set bits and the one that deletes clear bits.  Expectation by numpy:
BooleanArray take keeps Some(v) / None in row order, a null's value bit is false (record_batch.rs:171-175, boolean.rs:29-32)."""
import numpy as np
import pytest

from rivulus_amd.capi import Column, Predicate, Term

pytestmark = pytest.mark.gpu

N = 200_000_000
CUT = 37  # rows sliced off the front: bit offsets off the word grid


@pytest.fixture(scope="module")
def table(gpu_ctx):
    rng = np.random.default_rng(404)
    x = rng.integers(0, 1000, N).astype(np.int64)
    bv, bn, cv = rng.random(N) > 0.5, rng.random(N) > 0.2, rng.random(N) > 0.3
    host = [Column.from_numpy(x), Column.from_numpy(bv, bn), Column.from_numpy(cv)]
    dev = [gpu_ctx.upload(c.slice(CUT, N - CUT)) for c in host]
    yield x[CUT:], bv[CUT:], bn[CUT:], cv[CUT:], dev
    for d in dev:
        d.free()


@pytest.mark.parametrize("lit", [899, 159])
def test_boolean_ride_along_at_scale_matches_numpy(gpu_ctx, table, lit):
    x, bv, bn, cv, dev = table
    keep = x > lit
    for call in range(2):  # the second call sizes the output bitmaps by the selectivity of the first
        outs, rows, _ = gpu_ctx.filter_project(dev, Predicate([Term(0, ">", lit)]), [1, 2, 0])
        assert rows == int(keep.sum())
        b, c = outs[0].download(), outs[1].download()
        assert b.length == rows and c.length == rows
        assert np.array_equal(b.logical_valid(), bn[keep]), f"b validity, call {call}"
        assert np.array_equal(b.logical_values(), (bv & bn)[keep]), f"b values, call {call}"
        assert c.validity is None and np.array_equal(c.logical_values(), cv[keep]), f"c, call {call}"
        assert np.array_equal(outs[2].download().values[:rows], x[keep]), f"x, call {call}"
        for o in outs:
            o.free()
