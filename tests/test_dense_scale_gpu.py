"""Dense selections with columns that keep nulls, at 1e8 rows against numpy (-m gpu): the direct kernel's FF_OUTVALID path (a validity
byte per survivor next to its value in the LDS slot, packed to words on the way out, edge words OR-ed) and the staged geometries it
replaces below its thresholds, over ~50 000 tiles -- the oracle-checked cases stop at a few 1e6 rows.  Reference behaviour:
survivors in row order, a null slot holds 0, the bitmap is dropped when no null survives (record_batch.rs:131-178, primitive.rs:155-197)."""
import numpy as np
import pytest

from rivulus_amd.capi import Column, Predicate, Term

pytestmark = pytest.mark.gpu

N = 100_000_000


@pytest.fixture(scope="module")
def table(gpu_ctx):
    rng = np.random.default_rng(77)
    x = rng.integers(0, 1000, N).astype(np.int64)
    y = rng.integers(-5, 5, N).astype(np.int64)
    f = rng.random(N)
    fn = rng.random(N) > 0.08
    dev = [gpu_ctx.upload(Column.from_numpy(x)), gpu_ctx.upload(Column.from_numpy(y)), gpu_ctx.upload(Column.from_numpy(f, fn))]
    yield x, y, f, fn, dev
    for d in dev:
        d.free()


@pytest.mark.parametrize("mode", [-1, 0])
@pytest.mark.parametrize("lit,proj", [(99, [0, 2]), (499, [0, 1, 2]), (99, [2]), (899, [0, 2])])
def test_dense_selection_with_nulls_at_scale_matches_numpy(gpu_ctx, table, lit, proj, mode):
    """mode -1: every column through the pass (the direct kernel's output bitmaps at scale); 0, the default: the nullable column the
    predicate does not read is compacted after the pass at its wave offsets, its validity bits by bits_compact_kernel."""
    x, y, f, fn, dev = table
    host = [x, y, f]
    keep = x > lit
    seen = set()
    gpu_ctx.set_option("groups_by_ranges", mode)
    for call in range(3):  # unprepared, then sized from the selectivity seen (direct kernel from its thresholds on)
        outs, rows, _ = gpu_ctx.filter_project(dev, Predicate([Term(0, ">", lit)]), proj)
        seen.add(gpu_ctx.last_kernel())
        assert rows == int(keep.sum())
        for o, j in zip(outs, proj):
            col = o.download()
            want = host[j][keep]
            if j == 2:
                assert np.array_equal(col.logical_valid(), fn[keep]), f"validity of f, call {call}, {gpu_ctx.last_kernel()}"
                want = np.where(fn[keep], want, 0.0)  # a null slot holds the placeholder 0
            assert np.array_equal(col.values[:rows], want), f"column {j}, call {call}, {gpu_ctx.last_kernel()}"
            o.free()
    gpu_ctx.set_option("groups_by_ranges", 0)
    if lit == 99 and mode == -1:
        assert any(k.startswith("fused_direct_compact") for k in seen), seen
    if mode == 0 and len(proj) > 1:
        assert any(k.startswith("compact_ranges_kernel") for k in seen), seen
