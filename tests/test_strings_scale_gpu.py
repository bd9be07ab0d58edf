"""filter() of a StringArray at a size where the byte prefixes take their multi-workgroup paths (-m gpu): 6e7 rows, 0.48 GB of
strings.  The tests against the oracle stop at a few 1e5 rows (Python strings), where every prefix is one workgroup's scan:
here the block sums / tile sums go through str_group_sums (more than 64 groups of 256) and the outputs run to hundreds of
megabytes, in BOTH orders of the String side (block order: sel_str_lengths + str_gather_copy; source-tile order:
sel_str_tile_sums + sel_str_tile_copy) at a sparse and a dense selectivity.  The expectation is numpy's: the reference's
filter keeps the survivors in row order, a null survivor spans no bytes (record_batch.rs:163-170, string.rs:19-57)."""
import numpy as np
import pytest

from rivulus_amd.capi import RV_STRING, Column, Predicate, Term, pack_bits

pytestmark = pytest.mark.gpu

N = 60_000_000


@pytest.fixture(scope="module")
def table(gpu_ctx):
    rng = np.random.default_rng(2026)
    lens = rng.integers(0, 17, N).astype(np.int32)
    offs = np.zeros(N + 1, dtype=np.int32)
    np.cumsum(lens, out=offs[1:])
    data = rng.integers(97, 123, int(offs[-1])).astype(np.uint8)
    valid = rng.random(N) > 0.05  # (a null row keeps its bytes in the source: the output must not)
    x = rng.integers(0, 1000, N).astype(np.int64)
    dev = [gpu_ctx.upload(Column.from_numpy(x)), gpu_ctx.upload(Column(RV_STRING, data, pack_bits(valid), 0, N, offs))]
    yield x, lens, data, valid, dev
    for d in dev:
        d.free()


@pytest.mark.parametrize("lit", [899, 159])
@pytest.mark.parametrize("order", ["block", "tile"])
def test_string_filter_at_scale_matches_numpy(gpu_ctx, table, lit, order):
    x, lens, data, valid, dev = table
    keep = x > lit
    want_valid = valid[keep]
    want_lens = np.where(want_valid, lens[keep], 0).astype(np.int64)
    want_offsets = np.zeros(len(want_lens) + 1, dtype=np.int64)
    np.cumsum(want_lens, out=want_offsets[1:])
    want_data = data[np.repeat(keep & valid, lens)]
    gpu_ctx.set_option("str_tiles_from", 1 if order == "tile" else -1)
    try:
        for call in range(2):  # the second call is sized from the selectivity of the first
            outs, rows, _ = gpu_ctx.filter_project(dev, Predicate([Term(0, ">", lit)]), [1, 0])
            assert rows == int(keep.sum())
            got = outs[0].download()
            assert got.dtype == RV_STRING and got.length == rows and got.offset == 0
            assert np.array_equal(got.offsets[: rows + 1].astype(np.int64), want_offsets), f"offsets, call {call}"
            assert np.array_equal(got.values[: int(want_offsets[-1])], want_data), f"bytes, call {call}"
            assert np.array_equal(got.logical_valid(), want_valid), f"validity, call {call}"
            assert np.array_equal(outs[1].download().values[:rows], x[keep]), f"x, call {call}"
            for o in outs:
                o.free()
    finally:
        gpu_ctx.set_option("str_tiles_from", 0)


def test_string_take_at_scale_matches_numpy(gpu_ctx, table):
    """RecordBatch::take (record_batch.rs:131-178) of 8e6 rows in random order, repeats included: the copy's block sums pass their
    64 groups too."""
    x, lens, data, valid, dev = table
    rng = np.random.default_rng(7)
    idx = rng.integers(0, N, 8_000_000).astype(np.uint64)
    want_valid = valid[idx]
    want_lens = np.where(want_valid, lens[idx], 0).astype(np.int64)
    want_offsets = np.zeros(len(idx) + 1, dtype=np.int64)
    np.cumsum(want_lens, out=want_offsets[1:])
    starts = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    src = np.repeat(starts[idx], want_lens) + (np.arange(int(want_offsets[-1]), dtype=np.int64) - np.repeat(want_offsets[:-1], want_lens))
    got = gpu_ctx.take([dev[1]], idx)[0]
    col = got.download()
    assert col.length == len(idx)
    assert np.array_equal(col.offsets[: len(idx) + 1].astype(np.int64), want_offsets)
    assert np.array_equal(col.values[: int(want_offsets[-1])], data[src])
    assert np.array_equal(col.logical_valid(), want_valid)
    got.free()
