"""Which kernel / path a query takes on either side of every switch of rivulus_amd/csrc/thresholds.hpp (-m gpu).  The values there are
measured crossovers (the files each row names); this test pins the MECHANISM -- the selectivity a predicate is known to have over these
buffers (its last pass, or the first call's strided sample) picks the path, the same way on the first call of a big table as on the
second of any -- so that a change of a constant, or of the code that reads it, shows up as a path that moved."""
import re

import numpy as np
import pytest

from rivulus_amd.capi import RV_BOOLEAN, RV_FLOAT64, RV_INT64, Predicate, Term, synth_spec

pytestmark = pytest.mark.gpu
HDR = open(__file__.replace("tests/test_paths_gpu.py", "rivulus_amd/csrc/thresholds.hpp")).read()


def const(name):
    m = re.search(rf"\b{name} = ([^;,]+)[;,]", HDR)
    v = m.group(1).strip()
    if "<<" in v:
        a, b = re.findall(r"\d+", v)[-2:]
        return int(a) << int(b)
    return float(v) if "." in v else int(v)


def kernel_after(ctx, cols, pred, proj, calls=2):
    names = []
    for _ in range(calls):
        outs, rows, _ = ctx.filter_project(cols, pred, proj)
        names.append(ctx.last_kernel())
        [o.free() for o in outs]
    return names


def lit_for(sel):  # x = splitmix % 1000: x > lit keeps (999 - lit) / 1000
    return 999 - int(round(sel * 1000))


@pytest.fixture(scope="module")
def big(gpu_ctx):
    n = 40_000_000  # past kSampleFromRows: first calls are sized by the sample
    cols = [gpu_ctx.generate(synth_spec(RV_INT64, seed=42, length=n)), gpu_ctx.generate(synth_spec(RV_INT64, seed=46, length=n)),
            gpu_ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n)), gpu_ctx.generate(synth_spec(RV_BOOLEAN, seed=47, length=n, true_percent=10, validity_seed=48)),
            gpu_ctx.generate(synth_spec(RV_BOOLEAN, seed=49, length=n, true_percent=70))]
    yield cols
    [c.free() for c in cols]


@pytest.mark.parametrize("row,proj,name", [("kDirectFromOneColumn", [0], "one column"), ("kDirectFromTwoProjected", [0, 1], "two projected"),
                                           ("kDirectFromThreeProjected", [0, 1, 2], "three projected")])
def test_direct_kernel_from_the_tables_selectivity(gpu_ctx, big, row, proj, name):
    at = const(row)
    gpu_ctx.set_option("groups_by_ranges", -1)  # (the columns stay in the pass: the rule under test is the pass's kernel)
    try:
        for sel, want in ((at - 0.06, "fused_filter_compact"), (at + 0.06, "fused_direct_compact")):
            first, second = kernel_after(gpu_ctx, big[:len(proj)] if len(proj) > 1 else big[:1], Predicate([Term(0, ">", lit_for(sel))]), proj)
            assert first.startswith(want) and second.startswith(want), (name, sel, first, second)
    finally:
        gpu_ctx.set_option("groups_by_ranges", 0)


def test_one_projected_of_several_loaded(gpu_ctx, big):
    at = const("kDirectFromOneProjectedOfSeveral")
    gpu_ctx.set_option("groups_by_ranges", -1)
    try:
        for sel, want in ((at - 0.04, "fused_filter_compact"), (at + 0.05, "fused_direct_compact")):
            pred = Predicate([Term(0, ">", lit_for(sel)), Term(1, ">=", 0)])  # the pass loads x and y, projects y
            first, second = kernel_after(gpu_ctx, big[:2], pred, [1])
            assert first.startswith(want) and second.startswith(want), (sel, first, second)
    finally:
        gpu_ctx.set_option("groups_by_ranges", 0)


def test_plain_columns_the_predicate_does_not_read_follow_at_the_wave_offsets(gpu_ctx, big):
    at = const("kDeferPlainUpTo")
    assert big[0].length >= const("kRangesFromRows")
    first, second = kernel_after(gpu_ctx, big[:2], Predicate([Term(0, ">", lit_for(at - 0.05))]), [0, 1])
    assert first.startswith("compact_ranges_kernel<1>") and second.startswith("compact_ranges_kernel<1>"), (first, second)
    first, second = kernel_after(gpu_ctx, big[:2], Predicate([Term(0, ">", lit_for(at + 0.05))]), [0, 1])
    assert second.startswith("fused_"), (first, second)


def test_filter_by_a_boolean_column_takes_the_mask_path_while_sparse(gpu_ctx, big):
    sparse, dense = big[3], big[4]  # 9.5 % / 70 % of the rows survive
    first, second = kernel_after(gpu_ctx, [sparse, big[0]], Predicate([Term(0, "is_true")]), [1])
    assert first.startswith("compact_ranges_kernel") and second.startswith("compact_ranges_kernel")
    # a plain column of a selection denser than kMaskPathPlainUpTo: the direct kernel's pass -- from the first call on (the sample taken
    # to decide which columns the pass carries has told how dense the selection is)
    assert const("kMaskPathPlainUpTo") < 0.7
    first, second = kernel_after(gpu_ctx, [dense, big[0]], Predicate([Term(0, "is_true")]), [1])
    assert first.startswith("fused_direct_compact") and second.startswith("fused_direct_compact"), (first, second)


def test_the_first_call_is_sampled_from_kSampleFromRows_on(gpu_ctx):
    edge = const("kSampleFromRows")
    for n, sampled in ((edge, 1), (edge - 64, 0)):
        x = gpu_ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
        before = gpu_ctx.get_option("samples_taken")
        kernel_after(gpu_ctx, [x], Predicate([Term(0, ">", 899)]), [0], calls=1)
        assert gpu_ctx.get_option("samples_taken") == before + sampled, n
        x.free()


def test_columns_that_keep_nulls_switch_later(gpu_ctx):
    """[x, fn] with fn nullable and NOT tested by the predicate (its nulls survive): on a table too small for the ranges path the pass
    carries the output bitmap -- the direct kernel from kDirectFromTwoProjectedNullable on (second call: a small table is not sampled)."""
    n = 8_000_000
    assert n < const("kRangesFromRows")
    x, fn = gpu_ctx.generate(synth_spec(RV_INT64, seed=42, length=n)), gpu_ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
    at = const("kDirectFromTwoProjectedNullable")
    for sel, want in ((at - 0.06, "fused_filter_compact"), (at + 0.06, "fused_direct_compact")):
        _, second = kernel_after(gpu_ctx, [x, fn], Predicate([Term(0, ">", lit_for(sel))]), [0, 1])
        assert second.startswith(want), (sel, second)
    x.free(), fn.free()


def test_a_table_is_cut_into_stretches_by_the_samples_profile(gpu_ctx):
    """kStretchSparseUpTo / kStretchDenseFrom / kStretchLeastBlocks / kStretchesMost: a sorted table (two or three long stretches, each all
    sparse or all dense) is filtered stretch by stretch; a dense stretch shorter than kStretchLeastBlocks sample blocks, a table of many
    runs, or a "dense" stretch in which an independent second term leaves less than kStretchDenseFrom of the rows is not."""
    n = 40_000_000
    least, most = const("kStretchLeastBlocks"), const("kStretchesMost")
    dense_from, sparse_to = const("kStretchDenseFrom"), const("kStretchSparseUpTo")
    x = gpu_ctx.generate(synth_spec(RV_INT64, seed=42, length=n, pattern="sorted"))
    y = gpu_ctx.generate(synth_spec(RV_INT64, seed=46, length=n))
    runs = gpu_ctx.generate(synth_spec(RV_INT64, seed=42, length=n, pattern="clustered", run_rows=n // 64))  # 64 runs of 16 blocks
    try:
        cut = lambda cols, pred, proj: kernel_after(gpu_ctx, cols, pred, proj)[1].startswith("stretches: ")
        assert const("kStretchFromRows") > n and not cut([x], Predicate([Term(0, ">", 499)]), [0])  # (a stretch more costs ~55 us: big tables only)
        gpu_ctx.set_option("segments", n)
        assert cut([x], Predicate([Term(0, ">", 499)]), [0])
        assert cut([x], Predicate([Term(0, ">", 299), Term(0, "<", 700)]), [0]) and most >= 3
        # the dense stretch is (999 - lit) / 1000 of the table = that many of the 1024 blocks
        assert cut([x], Predicate([Term(0, ">", 999 - (least + 8))]), [0]) and not cut([x], Predicate([Term(0, ">", 999 - (least - 8))]), [0])
        # x > 499 and y < t: the dense stretch keeps t / 1000 of its rows
        assert cut([x, y], Predicate([Term(0, ">", 499), Term(1, "<", int(dense_from * 1000) + 40)]), [0])
        assert not cut([x, y], Predicate([Term(0, ">", 499), Term(1, "<", int(dense_from * 1000) - 40)]), [0])
        assert sparse_to < dense_from
        assert not cut([runs], Predicate([Term(0, ">", 499)]), [0])
    finally:
        gpu_ctx.set_option("segments", 0)
        [c.free() for c in (x, y, runs)]


def test_one_column_takes_the_taller_tiles_of_the_direct_kernel_while_fewer_survive(gpu_ctx, big):
    """kDirectTallBelow: the direct kernel over ONE loaded column keeps 16 rows per lane (8192-row tiles) while fewer than that share
    survives -- a table of runs at 30-50 %, a selection just past kDirectFromOneColumn -- and 12 (6144) above it."""
    at, first = const("kDirectTallBelow"), const("kDirectFromOneColumn")
    assert first < at - 0.04
    geometry = lambda name: tuple(int(v) for v in name[name.index("<") + 1:name.index(">")].split(",")[2:4])
    for sel, rows_per_lane in ((at - 0.04, 16), (at + 0.04, 12)):
        _, second = kernel_after(gpu_ctx, big[:1], Predicate([Term(0, ">", lit_for(sel))]), [0])
        assert second.startswith("fused_direct_compact<1,0,") and geometry(second) == (rows_per_lane, 8), (sel, second)
    # two loaded columns: the one geometry there is
    gpu_ctx.set_option("groups_by_ranges", -1)
    try:
        _, second = kernel_after(gpu_ctx, big[:2], Predicate([Term(0, ">", lit_for(0.5))]), [0, 1])
        assert second.startswith("fused_direct_compact<1,1,8,8"), second
    finally:
        gpu_ctx.set_option("groups_by_ranges", 0)
