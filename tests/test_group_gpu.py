"""rv_group_* (-m gpu): the single-process multi-GPU driver through the real kernels.

A one-GPU box can list device 0 several times (contexts are independent), so the whole protocol of BASELINE
configs[3] / configs[4] -- row-range shards, per-rank generator with the GLOBAL row index, per-rank fused pass,
prefix sum of the survivor counts, rank-order gather into pinned host memory, {SUM, COUNT} reduction -- runs
through librivulus_gpu.so here and is compared with (a) the oracle on the unsharded table and (b) the
unsharded single-context GPU result.  With one rank the aggregate goes through RCCL (ncclCommInitAll)."""
import threading

import numpy as np
import pytest

from helpers import assert_columns_equal
from rivulus_amd import capi
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Column, Predicate, Term, synth_spec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[1, 2, 3, 4], ids=lambda n: f"ranks{n}")
def group(request):
    g = capi.Group([0] * request.param)
    yield g
    g.close()


@pytest.mark.parametrize("n_rows", [0, 1, 100, 4096, 1_000_037])
def test_config3_sharded_filter_project_equals_unsharded(group, gpu_ctx, oracle, n_rows):
    """configs[3] shape: filter(x > 899).select([x]) over row-range shards generated in place."""
    spec = synth_spec(RV_INT64, seed=42, length=n_rows, first_row=7_000_000_000)
    x = group.generate(spec)
    pred = Predicate([Term(0, ">", 899)])
    res, rows = group.filter_project([x], pred, [0])
    got = res.column(0)
    hx = oracle.generate(spec)
    want = oracle.filter_project([hx], pred, [0])
    assert rows == want[0].length
    assert_columns_equal([got], want, f"n={n_rows} ranks={group.n}")
    # the unsharded single-context GPU result, byte for byte
    single, srows, _ = gpu_ctx.filter_project([gpu_ctx.generate(spec)], pred, [0])
    assert srows == rows and single[0].download().same_as(got) is None
    # the per-rank counts are the counts of the row ranges rv_shard_range hands out
    st = res.stats()
    assert sum(st["rank_rows"]) == rows and st["filter_ms"] >= 0 and st["gather_ms"] >= 0
    for r in range(group.n):
        b, e = capi.shard_range(n_rows, group.n, r)
        assert b % 64 == 0 or b == n_rows
        assert st["rank_rows"][r] == oracle.eval_predicate([hx.slice(b, e - b)], pred)[1]
        assert x.shard(r).length == e - b
    res.free()
    x.free()


@pytest.mark.parametrize("nulls", ["drops", "least"])
def test_config3_shape_with_null_bitmaps_sharded(group, oracle, nulls):
    """(f > 0.5) AND (x < 200) over nullable Float64 + Int64 shards; validity gathered at arbitrary bit offsets."""
    n = 300_011
    fs = synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44)
    xs = synth_spec(RV_INT64, seed=42, length=n, validity_seed=45)
    f, x = group.generate(fs), group.generate(xs)
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)], nulls)
    res, rows = group.filter_project([f, x], pred, [1, 0])
    want = oracle.filter_project([oracle.generate(fs), oracle.generate(xs)], pred, [1, 0])
    assert rows == want[0].length
    assert_columns_equal([res.column(0), res.column(1)], want, f"{nulls} ranks={group.n}")
    for j in range(2):
        assert res.null_count(j) == (0 if want[j].validity is None else int((~want[j].logical_valid()).sum()))


def test_uploaded_table_every_array_type(group, oracle):
    """rv_group_upload cuts a host table (sliced arrays, every array type) into row-range shards; the gather
    re-joins Int64 / Float64 / Boolean / String / Null columns like RecordBatch::concat (record_batch.rs:245-342)."""
    n = 50_021
    rng = np.random.default_rng(group.n)
    words = ["", "a", "Bob", "Ünï", "名前", "0123456789abcdef"]
    pad = 13
    x = Column.from_numpy(rng.integers(0, 1000, n + pad).astype(np.int64), rng.random(n + pad) > 0.1).slice(pad, n)
    f = Column.from_numpy(rng.random(n + pad)).slice(pad, n)
    b = Column.from_numpy(rng.random(n + pad) > 0.5, rng.random(n + pad) > 0.2).slice(pad, n)
    s = Column.from_strings([None if rng.random() < 0.1 else words[k] for k in rng.integers(0, len(words), n + pad)]).slice(pad, n)
    cols = [x, f, b, s]
    shards = [group.upload(c) for c in cols]
    for r in range(group.n):  # every shard is the row range of the host array
        lo, hi = capi.shard_range(n, group.n, r)
        for c, sh in zip(cols, shards):
            assert sh.shard(r).download().same_as(c.slice(lo, hi - lo)) is None
    for pred in (Predicate([Term(0, "<", 300), Term(1, ">", 0.25)]), Predicate([Term(0, "<", 300)], "least"),
                 Predicate([Term(2, "is_true")]), Predicate([Term(3, ">=", "Bob")]), Predicate([Term(0, ">", 5000)])):
        proj = [3, 0, 2, 1, 0]
        res, rows = group.filter_project(shards, pred, proj)
        want = oracle.filter_project(cols, pred, proj)
        assert rows == want[0].length
        assert_columns_equal([res.column(j) for j in range(len(proj))], want, f"{pred.terms} ranks={group.n}")
    # a NullArray column rides along
    res, rows = group.filter_project(shards + [group.upload(Column.nulls(n))], Predicate([Term(0, "<", 300)]), [4, 0])
    want = oracle.filter_project(cols + [Column.nulls(n)], Predicate([Term(0, "<", 300)]), [4, 0])
    assert_columns_equal([res.column(0), res.column(1)], want, "null column")


@pytest.mark.parametrize("n_rows", [0, 77, 2_000_003])
def test_config4_sharded_sum_count_equals_unsharded(group, gpu_ctx, oracle, n_rows):
    """configs[4] shape: filter + global SUM/COUNT; per-rank partials + one reduction of 16 bytes.  One rank:
    RCCL (ncclCommInitAll + ncclAllReduce); device 0 listed several times: host sum (RCCL refuses that group)."""
    spec = synth_spec(RV_INT64, seed=42, length=n_rows, first_row=123)
    x = group.generate(spec)
    pred = Predicate([Term(0, ">", 899)])
    si, _, cnt = group.filter_agg([x], pred, 0)
    hx = oracle.generate(spec)
    assert (si, cnt) == oracle.filter_agg([hx], pred, 0)[::2]
    ssi, _, scnt = gpu_ctx.filter_agg([gpu_ctx.generate(spec)], pred, 0)
    assert (si, cnt) == (ssi, scnt)
    # wrapping Int64 sums and a Float64 sum (tolerance 1e-12 relative: the order of additions differs)
    rng = np.random.default_rng(n_rows)
    big = Column.from_numpy(rng.integers(-(2 ** 63), 2 ** 63 - 1, n_rows, dtype=np.int64), rng.random(n_rows) > 0.1)
    fl = Column.from_numpy(rng.random(n_rows), rng.random(n_rows) > 0.1)
    sb, sf = group.upload(big), group.upload(fl)
    p2 = Predicate([Term(1, "<", 0.5)])
    si, _, cnt = group.filter_agg([sb, sf], p2, 0)
    assert (si, cnt) == oracle.filter_agg([big, fl], p2, 0)[::2]
    _, sfl, cnt = group.filter_agg([sb, sf], p2, 1)
    _, wfl, wcnt = oracle.filter_agg([big, fl], p2, 1)
    assert cnt == wcnt and abs(sfl - wfl) <= 1e-12 * max(1.0, abs(wfl))


def test_group_errors_carry_the_ranks_message(group):
    x = group.generate(synth_spec(RV_INT64, seed=42, length=1000))
    with pytest.raises(capi.RvError) as e:
        group.filter_project([x], Predicate([Term(3, ">", 1)]), [0])
    assert "references column 3" in e.value.message
    with pytest.raises(capi.RvError):
        capi.Group([])
    with pytest.raises(capi.RvError):
        capi.Group([4096])


def test_two_contexts_on_two_host_threads(oracle):
    """include/rivulus_gpu.h: 'different contexts are independent' -- two host threads, each with its own context,
    run queries at the same time and both get the reference result."""
    n = 700_003
    specs = [synth_spec(RV_INT64, seed=42 + t, length=n) for t in range(2)]
    wants = [oracle.filter_project([oracle.generate(s)], Predicate([Term(0, ">", 499 + 100 * t)]), [0])[0] for t, s in enumerate(specs)]
    errors = []

    def work(t):
        try:
            with capi.Context(0) as ctx:
                x = ctx.generate(specs[t])
                for _ in range(20):
                    outs, rows, _ = ctx.filter_project([x], Predicate([Term(0, ">", 499 + 100 * t)]), [0])
                    diff = outs[0].download().same_as(wants[t])
                    if diff is not None or rows != wants[t].length:
                        errors.append(f"thread {t}: {diff}")
                    outs[0].free()
                x.free()
        except Exception as ex:  # noqa: BLE001
            errors.append(f"thread {t}: {ex!r}")
    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_config3_and_config4_at_1e9_rows_two_ranks(gpu_ctx):
    """configs[3] / configs[4] at a BASELINE-sized shard count per rank (5e8 rows each, two ranks on this device):
    size-independent properties -- the gathered rows are exactly the survivors, in row order, and the group's
    {SUM, COUNT} equals the single-context aggregate and the sum over the gathered host column."""
    n = 1_000_000_000
    g = capi.Group([0, 0])
    try:
        spec = synth_spec(RV_INT64, seed=42, length=n)
        x = g.generate(spec)
        pred = Predicate([Term(0, ">", 899)])
        res, rows = g.filter_project([x], pred, [0])
        si, _, cnt = g.filter_agg([x], pred, 0)
        assert rows == cnt
        got = res.column(0)
        vals = np.asarray(got.values)
        assert got.length == rows and got.validity is None and vals.min() > 899
        assert int(vals.sum(dtype=np.int64)) == si
        # the unsharded single-context run: same count, same sum, same first / last window of the output
        one = gpu_ctx.generate(spec)
        ssi, _, scnt = gpu_ctx.filter_agg([one], pred, 0)
        assert (ssi, scnt) == (si, cnt)
        outs, srows, _ = gpu_ctx.filter_project([one], pred, [0])
        w = 1_000_000
        assert np.array_equal(np.asarray(outs[0].slice(0, w).download().values), vals[:w])
        assert np.array_equal(np.asarray(outs[0].slice(srows - w, w).download().values), vals[-w:])
        # the seam between the two ranks: the output rows around rank 0's last survivor
        r0 = res.stats()["rank_rows"][0]
        assert np.array_equal(np.asarray(outs[0].slice(r0 - 1000, 2000).download().values), vals[r0 - 1000:r0 + 1000])
        res.free()
        x.free()
    finally:
        g.close()


def test_resident_outputs_then_gather_equal_the_one_call_form(group, oracle):
    """rv_group_filter_project_resident + rv_group_gather == rv_group_filter_project; the resident outputs are ordinary
    device columns of their rank's context."""
    n = 700_001
    spec = synth_spec(RV_INT64, seed=42, length=n, validity_seed=45)
    x = group.generate(spec)
    pred = Predicate([Term(0, "<", 300)], "least")
    resident, rows = group.filter_project_resident([x], pred, [0])
    hx = oracle.generate(spec)
    want = oracle.filter_project([hx], pred, [0])
    assert rows == want[0].length and sum(resident.rank_rows) == rows
    at = 0
    for r in range(group.n):
        part = resident.column(r, 0).download()
        assert part.length == resident.rank_rows[r]
        b, e = capi.shard_range(n, group.n, r)
        assert part.same_as(oracle.filter_project([hx.slice(b, e - b)], pred, [0])[0]) is None
        at += part.length
    gathered = group.gather(resident)
    assert_columns_equal([gathered.column(0)], want, f"resident + gather ranks={group.n}")
    gathered.free()
    resident.free()
    x.free()


@pytest.mark.parametrize("chunk_rows", [0, 10_000, 4096])
def test_host_table_through_every_ranks_own_chunk_pipeline(group, gpu_ctx, oracle, chunk_rows):
    """rv_group_filter_project_host: a host-resident table (sliced arrays, every array type) cut into row ranges, every range streamed
    through its device's own double-buffered chunk pipeline, survivors gathered in rank order -- StreamingPhysicalPlan::collect()
    over N devices (streaming.rs:71-133, :343-352).  == rv_filter_project_host on one device == the oracle on the whole table."""
    n = 70_003
    rng = np.random.default_rng(7 + group.n)
    words = ["", "a", "Bob", "Ünï", "名前", "0123456789abcdef"]
    pad = 11
    x = Column.from_numpy(rng.integers(0, 1000, n + pad).astype(np.int64), rng.random(n + pad) > 0.1).slice(pad, n)
    f = Column.from_numpy(rng.random(n + pad)).slice(pad, n)
    b = Column.from_numpy(rng.random(n + pad) > 0.5, rng.random(n + pad) > 0.2).slice(pad, n)
    s = Column.from_strings([None if rng.random() < 0.1 else words[k] for k in rng.integers(0, len(words), n + pad)]).slice(pad, n)
    cols = [x, f, b, s]
    for pred, proj in ((Predicate([Term(0, "<", 300), Term(1, ">", 0.25)]), [3, 0, 2, 1]), (Predicate([Term(0, "<", 300)], "least"), [0, 1]),
                       (Predicate([Term(2, "is_true")]), [1, 2, 0]), (Predicate([Term(0, ">", 5000)]), [0, 3])):
        res, rows, gbs = group.filter_project_host(cols, pred, proj, chunk_rows)
        want = oracle.filter_project(cols, pred, proj)
        assert rows == want[0].length and len(gbs) == group.n and all(g >= 0.0 for g in gbs)
        assert_columns_equal([res.column(j) for j in range(len(proj))], want, f"{pred.terms} ranks={group.n} chunk={chunk_rows}")
        one, rows1 = gpu_ctx.filter_project_host(cols, pred, proj, chunk_rows)
        assert rows1 == rows
        assert_columns_equal([res.column(j) for j in range(len(proj))], [o.download() for o in one], "one device")
        res.free()
    # an empty table, and fewer rows than ranks x 64
    for m in (0, 100):
        res, rows, _ = group.filter_project_host([c.slice(0, m) for c in cols], Predicate([Term(1, ">=", 0.0)]), [0, 3])
        want = oracle.filter_project([c.slice(0, m) for c in cols], Predicate([Term(1, ">=", 0.0)]), [0, 3])
        assert rows == m and res.column(0).same_as(want[0]) is None and res.column(1).same_as(want[1]) is None


@pytest.mark.parametrize("bad_rank", ["first", "last"])
def test_a_failing_rank_fails_the_host_table_call(group, bad_rank):
    n = 300_000
    x = Column.from_numpy((np.arange(n) % 1000).astype(np.int64))
    pred = Predicate([Term(0, ">", 899)])
    bad = 0 if bad_rank == "first" else group.n - 1
    group.context(bad).set_option("inject_failure", 1)
    with pytest.raises(capi.RvError) as err:
        group.filter_project_host([x], pred, [0], 50_000)
    assert "injected failure" in str(err.value) and group.context(bad).get_option("inject_failure") == 0
    res, rows, _ = group.filter_project_host([x], pred, [0], 50_000)  # the group is usable again
    assert rows == n // 10 and np.array_equal(res.column(0).values[:rows], x.values[x.values > 899])


@pytest.mark.parametrize("bad_rank", ["first", "last"])
@pytest.mark.parametrize("call", ["filter_agg", "filter_project", "resident"])
def test_a_failing_rank_fails_the_call_and_keeps_every_rank_out_of_the_collective(group, call, bad_rank):
    """One rank's query fails (fault injection: option inject_failure): the group call returns that error after ALL ranks
    have finished, nothing hangs, no rank has entered the all-reduce, and the group works again afterwards."""
    n = 500_003
    x = group.generate(synth_spec(RV_INT64, seed=42, length=n))
    pred = Predicate([Term(0, ">", 899)])
    good = group.filter_agg([x], pred, 0)
    before = group.stat("allreduce_calls")
    bad = 0 if bad_rank == "first" else group.n - 1
    group.context(bad).set_option("inject_failure", 1)
    with pytest.raises(capi.RvError) as err:
        if call == "filter_agg":
            group.filter_agg([x], pred, 0)
        elif call == "filter_project":
            group.filter_project([x], pred, [0])
        else:
            group.filter_project_resident([x], pred, [0])
    assert "injected failure" in str(err.value)
    assert group.stat("allreduce_calls") == before and group.stat("comm_aborts") == 0
    assert group.context(bad).get_option("inject_failure") == 0
    assert group.filter_agg([x], pred, 0) == good  # the group is usable again
    res, rows = group.filter_project([x], pred, [0])
    assert rows == good[2]
    res.free()
    x.free()


def test_group_counters_tell_which_reduction_ran(group):
    x = group.generate(synth_spec(RV_INT64, seed=42, length=100_000))
    group.filter_agg([x], Predicate([Term(0, ">", 899)]), 0)
    if group.n == 1:  # distinct devices: RCCL formed the communicator (ncclCommInitAll) and ran the all-reduce
        assert group.stat("distinct_devices") == 1 and group.stat("rccl_ranks") == 1 and group.stat("allreduce_calls") >= 1
    else:             # device 0 listed several times: host sum
        assert group.stat("distinct_devices") == 0 and group.stat("rccl_ranks") == 0 and group.stat("allreduce_calls") == 0
    assert group.stat("last_agg_filter_us") > 0
    with pytest.raises(capi.RvError):
        group.stat("no_such_counter")
    x.free()


def test_distinct_devices_all_reduce_over_rccl(oracle):
    """The path a one-GPU box cannot take: N distinct devices, ncclCommInitAll + the grouped all-reduce.  Runs wherever the
    box has >= 2 GPUs (the driver's multi-GPU node); skipped otherwise."""
    ndev = capi.device_count()  # through the library: importing torch here would load a second HIP runtime into the process
    if ndev < 2:
        pytest.skip("needs >= 2 GPUs")
    ndev = min(ndev, 8)
    with capi.Group(list(range(ndev))) as g:
        n = 4_000_037
        spec = synth_spec(RV_INT64, seed=42, length=n)
        x = g.generate(spec)
        pred = Predicate([Term(0, ">", 899)])
        si, _, cnt = g.filter_agg([x], pred, 0)
        assert (si, cnt) == oracle.filter_agg([oracle.generate(spec)], pred, 0)[::2]
        assert g.stat("rccl_ranks") == ndev and g.stat("allreduce_calls") == 1
        g.context(ndev - 1).set_option("inject_failure", 1)
        with pytest.raises(capi.RvError):
            g.filter_agg([x], pred, 0)
        assert g.stat("allreduce_calls") == 1 and g.stat("comm_aborts") == 0
        assert g.filter_agg([x], pred, 0)[2] == cnt and g.stat("allreduce_calls") == 2
        res, rows = g.filter_project([x], pred, [0])
        assert rows == cnt
        assert_columns_equal([res.column(0)], oracle.filter_project([oracle.generate(spec)], pred, [0]), "distinct devices")
        x.free()


def test_a_result_may_outlive_its_group(oracle):
    """rv_gather_free after rv_group_destroy (what a garbage-collected binding does): the result's pinned blocks are shared
    with the group's pool, not borrowed from it."""
    g = capi.Group([0, 0])
    spec = synth_spec(RV_INT64, seed=42, length=300_000)
    x = g.generate(spec)
    pred = Predicate([Term(0, ">", 899)])
    res, rows = g.filter_project([x], pred, [0])
    res2, _ = g.filter_project([x], pred, [0])
    res2.free()  # one block back in the pool, one still out
    x.free()
    g.close()
    got = res.column(0)  # still readable: the memory belongs to the result
    assert_columns_equal([got], oracle.filter_project([oracle.generate(spec)], pred, [0]), "after the group is gone")
    assert rows == got.length
    res.free()


def test_a_sorted_table_sharded_over_two_ranks(gpu_ctx):
    """configs[3] over a table SORTED on the predicate's column, two ranks of 5e7 rows: the rank whose row range holds the edge cuts its
    shard into stretches (fused_launch.hip, run_segmented_pass), the other one keeps everything or nothing; the gathered rows are the
    unsharded single-context result, byte for byte, and the group's {SUM, COUNT} agrees."""
    n = 100_000_000
    g = capi.Group([0, 0])
    try:
        for r in range(2):
            g.context(r).set_option("segments", 1 << 25)  # (by default from 2^28 rows a shard)
        gpu_ctx.set_option("segments", 1 << 25)
        for pattern in ("sorted", "sorted_desc"):
            spec = synth_spec(RV_INT64, seed=42, length=n, pattern=pattern)
            x = g.generate(spec)
            one = gpu_ctx.generate(spec)
            for lit in (249, 749):  # the edge lies in rank 0's / rank 1's rows
                pred = Predicate([Term(0, ">", lit)])
                cut = [g.context(r).get_option("segmented_passes") for r in range(2)]
                res, rows = g.filter_project([x], pred, [0])
                si, _, cnt = g.filter_agg([x], pred, 0)
                got = res.column(0)
                outs, srows, _ = gpu_ctx.filter_project([one], pred, [0])
                assert rows == cnt == srows and abs(rows - (999 - lit) * (n // 1000)) <= 1, (pattern, lit, rows, cnt, srows)
                assert outs[0].download().same_as(got) is None, (pattern, lit)
                assert int(np.asarray(got.values).sum(dtype=np.int64)) == si
                now = [g.context(r).get_option("segmented_passes") for r in range(2)]
                edge_rank = (0 if lit == 249 else 1) if pattern == "sorted" else (1 if lit == 249 else 0)
                assert now[edge_rank] == cut[edge_rank] + 1 and now[1 - edge_rank] == cut[1 - edge_rank], (pattern, lit, cut, now)
                [o.free() for o in outs]
                res.free()
            x.free()
            one.free()
    finally:
        gpu_ctx.set_option("segments", 0)
        g.close()
