"""Sorted, clustered, periodic and drifting tables at 1e8 rows (-m gpu).

The reference filters whatever order its source has (plan.rs:112-147; file_stream.rs:122-198 reads a CSV in file order), and
the only contract on the output is ascending row order (record_batch.rs:235-240) -- `col > lit` columns are ids and timestamps:
sorted or in runs.  Every other large test of this repo draws independent rows, so a tile's selectivity is the table's.  Here
it is not: whole tiles survive next to tiles where nothing does, which is what the launch geometry (LDS slots sized from ONE
global selectivity), the strided first-call sample, the redo kernel for wave ranges that outgrow their slot, the speculative
output sizing and the direct / ranges / mask switches all have to live with.

numpy restates each predicate in one line; it is pinned to the oracle on windows that straddle the edges of the survivor
runs, then stands in for it over the whole table."""
import numpy as np
import pytest

from rivulus_amd.capi import RV_BOOLEAN, RV_FLOAT64, RV_INT64, Column, Predicate, Term, synth_spec

pytestmark = pytest.mark.gpu

N = 100_000_000
W = 200_000  # rows of an oracle window


def _gen(gpu_ctx, spec):
    d = gpu_ctx.generate(spec)
    return d, d.download()


def _edges(keep, limit=3):
    """Row indices where the survive mask flips (the edges of the survivor runs): the first few, the last, one in the middle."""
    flips = np.flatnonzero(keep[1:] != keep[:-1]) + 1
    if len(flips) == 0:
        return [len(keep) // 2]
    pick = list(flips[:limit]) + [int(flips[len(flips) // 2]), int(flips[-1])]
    return sorted(set(int(q) for q in pick))


def _pin_to_oracle(oracle, host, pred, keep, what):
    """numpy's survivors == the oracle's on windows across the run edges (and at the table's ends)."""
    n = len(keep)
    for at in [W // 2, n - W // 2] + _edges(keep):
        lo = max(0, min(n - W, at - W // 2))
        window = [c.slice(lo, W) for c in host]
        assert oracle.eval_predicate(window, pred)[1] == int(keep[lo:lo + W].sum()), f"{what}: numpy disagrees with the oracle on rows [{lo}, {lo + W})"


def _compare(gpu_ctx, outs, rows, host, proj, keep, what):
    assert rows == int(keep.sum()), f"{what}: {rows} rows, numpy keeps {int(keep.sum())} ({gpu_ctx.last_kernel()})"
    for o, j in zip(outs, proj):
        col, src = o.download(), host[j]
        if src.dtype == 4:  # String: offsets + bytes of the survivors
            idx = np.flatnonzero(keep)
            lens = (src.offsets[1:] - src.offsets[:-1])[idx]
            assert np.array_equal(col.offsets[1:rows + 1] - col.offsets[:rows], lens), f"{what}: string lengths of column {j}"
            sample = idx[:: max(1, len(idx) // 5000)]
            pos = np.flatnonzero(keep)[:: max(1, len(idx) // 5000)]
            for k, i in zip(range(0, rows, max(1, len(idx) // 5000)), sample):
                assert bytes(col.values[col.offsets[k]:col.offsets[k + 1]]) == bytes(src.values[src.offsets[i]:src.offsets[i + 1]]), f"{what}: string {k}"
            del pos
            o.free()
            continue
        valid = src.logical_valid()
        if valid is not None and not valid[keep].all():
            assert np.array_equal(col.logical_valid(), valid[keep]), f"{what}: validity of column {j} ({gpu_ctx.last_kernel()})"
            want = np.where(valid[keep], src.logical_values()[keep], 0)
        else:
            assert col.validity is None, f"{what}: column {j} kept a bitmap without a null"
            want = src.logical_values()[keep]
        got = col.logical_values() if col.dtype == 1 else col.values[:rows]
        assert np.array_equal(got, want), f"{what}: column {j} ({gpu_ctx.last_kernel()})"
        o.free()


def _run(gpu_ctx, oracle, host, dev, pred, proj, keep, what, calls=2, log=None, reruns_allowed=0):
    _pin_to_oracle(oracle, host, pred, keep, what)
    reruns = gpu_ctx.get_option("overflow_reruns")
    for call in range(calls):
        outs, rows, _ = gpu_ctx.filter_project(dev, pred, proj)
        redo = gpu_ctx.get_option("last_redo_ppm")
        _compare(gpu_ctx, outs, rows, host, proj, keep, f"{what} call {call}")
        if log is not None:
            log.append((what, call, gpu_ctx.last_kernel(), redo))
        print(f"[skew] {what} call {call}: {gpu_ctx.last_kernel()} last_redo_ppm {redo}")
    # outputs sized from the sample / the predicate's last pass (+ 20 % + 2 % of the rows) hold what a sorted or clustered table keeps
    assert gpu_ctx.get_option("overflow_reruns") - reruns <= reruns_allowed, f"{what}: the default output sizing needed a re-run"
    return gpu_ctx.get_option("overflow_reruns") - reruns


PATTERNS = {
    "sorted": dict(pattern="sorted"),
    "sorted_desc": dict(pattern="sorted_desc"),
    "runs_1e3": dict(pattern="clustered", run_rows=1_000),
    "runs_1e5": dict(pattern="clustered", run_rows=100_000),
    "runs_1e7": dict(pattern="clustered", run_rows=10_000_000),
}


def test_generator_patterns_match_the_oracle(gpu_ctx, oracle):
    """rv_generate == orc_generate bit for bit for every pattern, dtype and a shard that starts inside the table."""
    for name, kw in PATTERNS.items():
        for dtype in (RV_INT64, RV_FLOAT64, RV_BOOLEAN):
            for first, length in ((0, 70_001), (123_457, 65_537)):
                kw2 = dict(kw)
                if "sorted" in name:
                    kw2["table_rows"] = 200_000
                if "run_rows" in kw2:
                    kw2["run_rows"] = 777
                spec = synth_spec(dtype, seed=42, length=length, first_row=first, true_percent=30, validity_seed=44, **kw2)
                d = gpu_ctx.generate(spec)
                diff = d.download().same_as(oracle.generate(spec))
                assert diff is None, f"{name} dtype {dtype} first {first}: {diff}"
                d.free()
    # what the patterns promise: order, runs
    asc = oracle.generate(synth_spec(RV_INT64, seed=1, length=100_000, pattern="sorted")).values
    assert (np.diff(asc) >= 0).all() and asc[0] == 0 and asc[-1] == 999 and abs(int((asc > 899).sum()) - 10_000) <= 1
    desc = oracle.generate(synth_spec(RV_FLOAT64, seed=1, length=100_000, pattern="sorted_desc")).values
    assert (np.diff(desc) <= 0).all() and 0.0 <= desc[-1] < desc[0] < 1.0
    runs = oracle.generate(synth_spec(RV_INT64, seed=1, length=100_000, pattern="clustered", run_rows=1000)).values
    assert all(len(set(runs[i:i + 1000])) == 1 for i in range(0, 100_000, 1000)) and len(set(runs[::1000])) > 50


@pytest.mark.parametrize("name", list(PATTERNS))
def test_one_column_over_sorted_and_clustered_tables(gpu_ctx, oracle, name):
    """x > t -> [x] at 10 / 50 / 84 % global selectivity: whole tiles survive next to empty ones.  First call (sized by the strided
    sample) and second (sized by what the first one saw)."""
    d, h = _gen(gpu_ctx, synth_spec(RV_INT64, seed=42, length=N, **PATTERNS[name]))
    x = h.values
    try:
        for lit in (899, 499, 159):
            pred = Predicate([Term(0, ">", lit)])
            _run(gpu_ctx, oracle, [h], [d], pred, [0], x > lit, f"{name} x > {lit} -> [x]")
        # the other side of the order: survivors first (ascending) / last (descending)
        pred = Predicate([Term(0, "<", 100)])
        _run(gpu_ctx, oracle, [h], [d], pred, [0], x < 100, f"{name} x < 100 -> [x]", calls=1)
    finally:
        d.free()


def test_runs_of_survivors_pick_the_kernel_by_what_the_redo_would_cost(gpu_ctx):
    """Independent rows at 50 %: the staged pass.  The same selectivity in runs: half of its wave ranges would outgrow their slots
    and be re-read, so the direct kernel (rows wait in registers) runs instead -- from the FIRST call on a big table (the sample's
    histogram of blocks), from the third on a small one (the first pass, sized blind, and the second, with the roomier slots of a
    50 % selection, both left half of their ranges to the redo kernel: independent rows would have stopped at the second).  At 10 % the redo kernel is
    the cheaper way and follows the pass on the stream."""
    pred = Predicate([Term(0, ">", 499)])
    for n, first_is_direct in ((N, True), (20_000_000, False)):
        iid = gpu_ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
        runs = gpu_ctx.generate(synth_spec(RV_INT64, seed=42, length=n, pattern="clustered", run_rows=50_000))
        for call in range(3):
            outs, _, _ = gpu_ctx.filter_project([iid], pred, [0])
            # (a small table's first call is sized blind: the default geometry's slots hold 37.5 % of a wave's rows, half of them survive)
            assert gpu_ctx.last_kernel().startswith("fused_filter_compact") and (gpu_ctx.get_option("last_redo_ppm") == 0 or (call == 0 and not first_is_direct)), (n, call, gpu_ctx.last_kernel())
            [o.free() for o in outs]
            outs, _, _ = gpu_ctx.filter_project([runs], pred, [0])
            direct = gpu_ctx.last_kernel().startswith("fused_direct_compact")
            assert direct == (first_is_direct or call > 1), (n, call, gpu_ctx.last_kernel(), gpu_ctx.get_option("last_redo_ppm"))
            [o.free() for o in outs]
        outs, _, _ = gpu_ctx.filter_project([runs], Predicate([Term(0, ">", 899)]), [0])
        assert gpu_ctx.last_kernel().startswith("fused_filter_compact") and gpu_ctx.get_option("last_redo_ppm") > 50_000, gpu_ctx.last_kernel()
        [o.free() for o in outs]
        gpu_ctx.set_option("skew", -1)  # the rule switched off: the staged pass + the redo kernel, whatever it costs
        outs, _, _ = gpu_ctx.filter_project([runs], pred, [0])
        assert gpu_ctx.last_kernel().startswith("fused_filter_compact") and gpu_ctx.get_option("last_redo_ppm") > 300_000
        [o.free() for o in outs]
        gpu_ctx.set_option("skew", 0)
        iid.free()
        runs.free()


@pytest.fixture(scope="module")
def riders(gpu_ctx):
    """Columns that ride along: a nullable Float64, a plain Int64, a nullable Boolean (independent rows)."""
    specs = [synth_spec(RV_FLOAT64, seed=43, length=N, validity_seed=44), synth_spec(RV_INT64, seed=46, length=N),
             synth_spec(RV_BOOLEAN, seed=47, length=N, true_percent=30, validity_seed=48)]
    pairs = [_gen(gpu_ctx, s) for s in specs]
    yield [p[1] for p in pairs], [p[0] for p in pairs]
    for p in pairs:
        p[0].free()


@pytest.mark.parametrize("name", ["sorted", "runs_1e3", "runs_1e5"])
def test_riding_columns_and_null_policies(gpu_ctx, oracle, riders, name):
    """A nullable predicate column under both null policies, nullable / Boolean columns riding along, the nine-column frame of the
    eager Filter (columns compacted after the pass at its wave offsets), the filter + SUM / COUNT."""
    rh, rd = riders
    d, h = _gen(gpu_ctx, synth_spec(RV_INT64, seed=42, length=N, validity_seed=45, **PATTERNS[name]))
    x, xv = h.values, h.logical_valid()
    host, dev = [h] + rh, [d] + rd
    try:
        for lit in (899, 159):
            terms = [Term(0, ">", lit)]
            _run(gpu_ctx, oracle, host, dev, Predicate(terms), [0, 1], xv & (x > lit), f"{name} nullable x > {lit} -> [x, fn] drops", calls=1)
            _run(gpu_ctx, oracle, host, dev, Predicate(terms), [0, 1, 2, 3], xv & (x > lit), f"{name} nullable x > {lit} -> [x, fn, y, c] drops", calls=2)
        # eager ordering: Null is least (series.rs:105-107): a null x is < 100, and its rows keep their null
        _run(gpu_ctx, oracle, host, dev, Predicate([Term(0, "<", 100)], "least"), [0, 1], ~xv | (x < 100), f"{name} nullable x < 100 least -> [x, fn]", calls=2)
        _run(gpu_ctx, oracle, host, dev, Predicate([Term(0, ">", 899), Term(2, ">=", 100)]), [2, 0], xv & (x > 899) & (rh[1].values >= 100), f"{name} x > 899 and y >= 100 -> [y, x]", calls=1)
        # the eager Filter keeps every column (plan.rs:132-147)
        _run(gpu_ctx, oracle, host, dev, Predicate([Term(0, ">", 899)]), [0, 1, 2, 1, 2, 0, 1, 2, 1], xv & (x > 899), f"{name} nine columns 10 %", calls=2)
        _run(gpu_ctx, oracle, host, dev, Predicate([Term(0, ">", 159)]), [0, 1, 2, 1, 2], xv & (x > 159), f"{name} five columns 84 %", calls=1)
        keep = xv & (x > 899)
        s, _, cnt = gpu_ctx.filter_agg(dev, Predicate([Term(0, ">", 899)]), 2)
        assert cnt == int(keep.sum()) and s == int(rh[1].values[keep].sum(dtype=np.int64)), f"{name}: filter + SUM / COUNT"
    finally:
        d.free()


@pytest.mark.parametrize("name", ["sorted", "runs_1e3", "runs_1e5"])
def test_filter_by_a_clustered_boolean_column(gpu_ctx, oracle, riders, name):
    """RecordBatch::filter by a BooleanArray (record_batch.rs:221-243) whose true cells come in runs: the mask path
    (mask_select_kernel + compact_ranges_kernel) and the chained pass by a Boolean term."""
    rh, rd = riders
    d, h = _gen(gpu_ctx, synth_spec(RV_BOOLEAN, seed=49, length=N, true_percent=10, validity_seed=50, **PATTERNS[name]))
    keep = h.logical_values() & h.logical_valid()
    host, dev = [h] + rh, [d] + rd
    try:
        _run(gpu_ctx, oracle, host, dev, Predicate([Term(0, "is_true")]), [2, 1], keep, f"{name} b is true -> [y, fn]", calls=2)
        _run(gpu_ctx, oracle, host, dev, Predicate([Term(0, "is_true")]), [2, 3], keep, f"{name} b is true -> [y, c]", calls=1)
        _run(gpu_ctx, oracle, host, dev, Predicate([Term(0, "is_true"), Term(2, "<", 700)]), [1, 2], keep & (rh[1].values < 700), f"{name} b is true and y < 700 -> [fn, y]", calls=1)
    finally:
        d.free()


def test_strings_ride_along_a_clustered_selection(gpu_ctx, oracle):
    """x > t -> [x, name] with x in runs / sorted: the String gather's block order and source-tile order both see tiles where every
    string survives next to tiles where none does."""
    n = 30_000_000
    rng = np.random.default_rng(11)
    lens = rng.integers(0, 13, n).astype(np.int32)
    offs = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=offs[1:])
    data = rng.integers(97, 123, int(offs[-1])).astype(np.uint8)
    name_h = Column(4, data, None, 0, n, offs)
    name_d = gpu_ctx.upload(name_h)
    try:
        for pat in ("runs_1e3", "sorted"):
            d, h = _gen(gpu_ctx, synth_spec(RV_INT64, seed=42, length=n, **PATTERNS[pat]))
            for lit in (899, 159):
                keep = h.values > lit
                _pin_to_oracle(oracle, [h], Predicate([Term(0, ">", lit)]), keep, f"strings {pat} {lit}")
                for call in range(2):
                    outs, rows, _ = gpu_ctx.filter_project([d, name_d], Predicate([Term(0, ">", lit)]), [0, 1])
                    _compare(gpu_ctx, outs, rows, [h, name_h], [0, 1], keep, f"strings {pat} x > {lit} call {call}")
            d.free()
    finally:
        name_d.free()


def test_a_column_periodic_in_the_samples_stride(gpu_ctx, oracle):
    """The first call of a predicate over a big table is sized by 1024 blocks of 1024 rows strided over it (fused_launch.hip,
    sample_selectivity).  A column that repeats with exactly that stride shows the sample the same rows every time: it sees 100 %
    where 0.1 % survive, or 0 % where 99.9 % do.  Results never depend on the estimate."""
    n = N
    stride = (n // 1024) & ~63
    phase = np.arange(n, dtype=np.int64) % stride
    for burst_is_high in (True, False):
        x = np.where(phase < 1024, 999 if burst_is_high else 0, 0 if burst_is_high else 999).astype(np.int64)
        h = Column.from_numpy(x)
        d = gpu_ctx.upload(h)
        try:
            taken = gpu_ctx.get_option("samples_taken")
            reran = _run(gpu_ctx, oracle, [h], [d], Predicate([Term(0, ">", 899)]), [0], x > 899, f"periodic burst_is_high={burst_is_high}", calls=2, reruns_allowed=1)
            assert gpu_ctx.get_option("samples_taken") == taken + 1  # the first call sampled (and was fooled); the second remembered the truth
            # fooled into outputs for 3 % of the rows where 99.9 % survive: the pass counted exactly and ran once more with outputs of that size
            assert reran == (0 if burst_is_high else 1)
        finally:
            d.free()


def test_a_stream_whose_selectivity_drifts(gpu_ctx, oracle, riders):
    """Windows of 1024-row batches (stream.rs:136-158) over a table whose selectivity drifts from 5 % to 95 %: every window is sized by
    what the predicate did on the window before it.  x = splitmix % 1000 + a ramp, so `x > 999` keeps more and more rows."""
    rh, rd = riders
    n = 64 * 1024 * 1024
    base = rh[1].values[:n]
    ramp = (np.arange(n, dtype=np.int64) * 900) // n + 50  # 50 .. 949
    x = base + ramp  # x > 999  <=>  base > 999 - ramp: 5 % of the rows at the start, 95 % at the end
    h = Column.from_numpy(x)
    fn_h, fn_d = rh[0].slice(0, n), rd[0].slice(0, n)
    d = gpu_ctx.upload(h)
    pred = Predicate([Term(0, ">", 999)])
    keep = x > 999
    window = 8 * 1024 * 1024
    try:
        _pin_to_oracle(oracle, [h, fn_h], pred, keep, "drifting stream")
        reruns = gpu_ctx.get_option("overflow_reruns")
        for w0 in range(0, n, window):
            cols = [d.slice(w0, window), fn_d.slice(w0, window)]
            outs, rows, nulls, total = gpu_ctx.filter_project_chunked(cols, 1024, pred, [0, 1])
            kw = keep[w0:w0 + window]
            assert total == int(kw.sum()) and np.array_equal(rows, kw.reshape(-1, 1024).sum(axis=1).astype(np.uint64)), f"window at {w0}"
            _compare(gpu_ctx, outs, total, [h.slice(w0, window), fn_h.slice(w0, window)], [0, 1], kw, f"drifting window at {w0}")
            print(f"[skew] drifting window at {w0}: {kw.mean():.3f} kept, {gpu_ctx.last_kernel()} last_redo_ppm {gpu_ctx.get_option('last_redo_ppm')}")
            for c in cols:
                c.free()
        assert gpu_ctx.get_option("overflow_reruns") == reruns
    finally:
        d.free()
        fn_d.free()


def test_a_sorted_table_is_filtered_stretch_by_stretch(gpu_ctx, oracle, riders):
    """A table sorted on the predicate's column keeps its survivors in one long stretch: the sample's profile (its 1024 blocks in table
    order) shows it, and the table is cut there -- the staged pass over the stretch where nothing survives, the direct kernel over the
    one where everything does, both into ONE set of outputs (fused_launch.hip, run_segmented_pass).  From the first call; the same
    rows in the same order as one pass; independent rows, many runs, a nullable projected column or option "segments" = -1: one pass."""
    rh, rd = riders
    took = lambda: gpu_ctx.get_option("segmented_passes")
    gpu_ctx.set_option("segments", FROM)  # (by default from 2^28 rows on: a stretch more costs ~55 us per query)
    try:
        _stretches(gpu_ctx, oracle, rh, rd, took)
    finally:
        gpu_ctx.set_option("segments", 0)


FROM = 1 << 25


def _stretches(gpu_ctx, oracle, rh, rd, took):
    for name in ("sorted", "sorted_desc"):
        d, h = _gen(gpu_ctx, synth_spec(RV_INT64, seed=42, length=N, **PATTERNS[name]))
        x = h.values
        try:
            for lit in (899, 499, 159):
                before = took()
                _run(gpu_ctx, oracle, [h, rh[1]], [d, rd[1]], Predicate([Term(0, ">", lit)]), [0], x > lit, f"stretches {name} x > {lit} -> [x]", calls=2)
                assert took() == before + 2 and gpu_ctx.last_kernel().startswith("stretches: "), (name, lit, gpu_ctx.last_kernel())
                assert ("fused_direct_compact" in gpu_ctx.last_kernel()) and ("fused_filter_compact" in gpu_ctx.last_kernel()), gpu_ctx.last_kernel()
            # a plain column riding along, two terms (the second one independent of the order: the dense stretch keeps 70 % of its rows)
            before = took()
            y = rh[1].values
            _run(gpu_ctx, oracle, [h, rh[1]], [d, rd[1]], Predicate([Term(0, ">", 499)]), [1, 0], x > 499, f"stretches {name} x > 499 -> [y, x]", calls=2)
            _run(gpu_ctx, oracle, [h, rh[1]], [d, rd[1]], Predicate([Term(0, ">", 499), Term(1, "<", 700)]), [0, 1], (x > 499) & (y < 700), f"stretches {name} x > 499 and y < 700 -> [x, y]", calls=2)
            assert took() == before + 4, gpu_ctx.last_kernel()
            # a between: sparse, dense, sparse
            before = took()
            _run(gpu_ctx, oracle, [h], [d], Predicate([Term(0, ">", 299), Term(0, "<", 600)]), [0], (x > 299) & (x < 600), f"stretches {name} 299 < x < 600 -> [x]", calls=2)
            assert took() == before + 2 and gpu_ctx.last_kernel().count("+") == 2, gpu_ctx.last_kernel()
            # not this way: a column that follows at the pass's wave offsets (query.hip: nullable and not read by the predicate), the option
            before = took()
            _run(gpu_ctx, oracle, [h, rh[0]], [d, rd[0]], Predicate([Term(0, ">", 899)]), [0, 1], x > 899, f"stretches {name} x > 899 -> [x, fn]", calls=1)
            gpu_ctx.set_option("segments", -1)
            _run(gpu_ctx, oracle, [h], [d], Predicate([Term(0, ">", 899)]), [0], x > 899, f"stretches {name} switched off", calls=1)
            gpu_ctx.set_option("segments", 0)  # the default: not a table of 1e8 rows
            _run(gpu_ctx, oracle, [h], [d], Predicate([Term(0, ">", 899)]), [0], x > 899, f"stretches {name} by default", calls=1)
            gpu_ctx.set_option("segments", FROM)
            assert took() == before and not gpu_ctx.last_kernel().startswith("stretches"), gpu_ctx.last_kernel()
        finally:
            d.free()
    # a nullable predicate column, both null policies (the nulls are independent rows: under "least" they survive inside the sparse stretch)
    d, h = _gen(gpu_ctx, synth_spec(RV_INT64, seed=42, length=N, validity_seed=45, pattern="sorted"))
    x, xv = h.values, h.logical_valid()
    try:
        before = took()
        _run(gpu_ctx, oracle, [h, rh[1]], [d, rd[1]], Predicate([Term(0, ">", 499)]), [1], xv & (x > 499), "stretches nullable x > 499 -> [y] drops", calls=2)
        _run(gpu_ctx, oracle, [h, rh[1]], [d, rd[1]], Predicate([Term(0, "<", 300)], "least"), [1], ~xv | (x < 300), "stretches nullable x < 300 least -> [y]", calls=2)
        assert took() == before + 4, gpu_ctx.last_kernel()
        # ... projected itself: under "least" its nulls survive, every stretch writes a bitmap of its own and the table's is put together from them
        before = took()
        _run(gpu_ctx, oracle, [h, rh[1]], [d, rd[1]], Predicate([Term(0, "<", 300)], "least"), [0], ~xv | (x < 300), "stretches nullable x < 300 least -> [x]", calls=2)
        _run(gpu_ctx, oracle, [h, rh[1]], [d, rd[1]], Predicate([Term(0, "<", 600)], "least"), [1, 0], ~xv | (x < 600), "stretches nullable x < 600 least -> [y, x]", calls=2)
        _run(gpu_ctx, oracle, [h, rh[1]], [d, rd[1]], Predicate([Term(0, ">", 499)]), [0, 1], xv & (x > 499), "stretches nullable x > 499 -> [x, y] drops", calls=1)
        assert took() == before + 5, gpu_ctx.last_kernel()
    finally:
        d.free()
    # independent rows, short runs: nothing to cut
    before = took()
    for kw in ({}, dict(pattern="clustered", run_rows=1000)):
        d, h = _gen(gpu_ctx, synth_spec(RV_INT64, seed=42, length=N, **kw))
        _run(gpu_ctx, oracle, [h], [d], Predicate([Term(0, ">", 499)]), [0], h.values > 499, f"stretches: none in {kw}", calls=2)
        d.free()
    assert took() == before


def test_a_profile_that_lies_about_the_stretches(gpu_ctx, oracle):
    """The sample sees 1024 rows out of every ~97 000.  A table whose SAMPLED rows say "everything survives in the first half, nothing
    in the second" while every other row survives: the stretches are planned for half of the rows, the shared outputs overflow in the
    second one, and the query falls back to one pass -- which counts exactly and sizes its re-run by the count."""
    n = N
    stride = (n // 1024) & ~63
    rows = np.arange(n, dtype=np.int64)
    sampled = (rows % stride) < 1024
    x = np.where(sampled & (rows >= n // 2), 0, 999).astype(np.int64)
    del rows, sampled
    h = Column.from_numpy(x)
    d = gpu_ctx.upload(h)
    gpu_ctx.set_option("segments", FROM)
    try:
        before, reruns, fell = gpu_ctx.get_option("segmented_passes"), gpu_ctx.get_option("overflow_reruns"), gpu_ctx.get_option("segment_fallbacks")
        _run(gpu_ctx, oracle, [h], [d], Predicate([Term(0, ">", 899)]), [0], x > 899, "lying profile", calls=2, reruns_allowed=1)
        assert gpu_ctx.get_option("segmented_passes") == before, gpu_ctx.last_kernel()  # planned once, abandoned; the second call knows the table's real selectivity
        assert gpu_ctx.get_option("overflow_reruns") == reruns + 1 and gpu_ctx.get_option("segment_fallbacks") == fell + 1
    finally:
        gpu_ctx.set_option("segments", 0)
        d.free()
