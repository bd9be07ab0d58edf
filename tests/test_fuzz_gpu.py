"""Randomised differential test (-m gpu): random schemas (Int64 / Float64 / Boolean / String columns, null
bitmaps, sliced inputs), random AND-of-compare predicates (every operator, Int64 / Float64 / Boolean / String /
Null / cross-type literals, both null policies) and random projections, GPU against the oracle.  Fixed seeds:
the cases are the same on every run.  Exercises every kernel-selection path (single-term fast path, FF_PROJALL /
FF_NONULL, Boolean predicate columns, bit streams, selection bitmap, column groups, String gather/compare)."""
import numpy as np
import pytest

from helpers import assert_columns_equal
from rivulus_amd import capi
from rivulus_amd.capi import Column, Predicate, Term

import os

pytestmark = pytest.mark.gpu
N_QUERY_CASES = int(os.environ.get("RV_FUZZ_CASES", 400))  # more seeds for a soak run (round 3, final code: 5000 + 400 + 400 + 1500 over the four tests and 180 large cases, green; mid-round 12 000 + 480; round 5, final code: RV_FUZZ_CASES=4000 RV_FUZZ_BATCH_CASES=300 RV_FUZZ_STREAM_CASES=300 -> 4680 green, and RV_FUZZ_BIG=1 with 240 / 20 / 20 -> 360 green)
N_BATCH_CASES = int(os.environ.get("RV_FUZZ_BATCH_CASES", 60))
OPS = ["==", "!=", "<", ">", "<=", ">="]
WORDS = ["", "a", "ab", "b", "Bob", "Ünï", "zz", "名前"]


def _column(rng, kind, n, pad):
    total = n + pad + int(rng.integers(0, 9))
    nulls = rng.random() < 0.6
    valid = (rng.random(total) > rng.choice([0.02, 0.3, 0.9])) if nulls else None
    if kind == "i":
        c = Column.from_numpy(rng.integers(-3, 12, total).astype(np.int64), valid)
    elif kind == "f":
        vals = rng.choice([0.0, -0.0, 0.5, 1.5, -2.25, np.nan, np.inf, -np.inf, 7.0], total)
        c = Column.from_numpy(vals.astype(np.float64), valid)
    elif kind == "b":
        c = Column.from_numpy(rng.random(total) > 0.4, valid)
    else:
        vals = [None if (valid is not None and not valid[i]) else WORDS[k] for i, k in enumerate(rng.integers(0, len(WORDS), total))]
        c = Column.from_strings(vals)
    return c.slice(pad, n)


def _literal(rng, kind):
    r = rng.random()
    if r < 0.08:
        return None                                     # Literal(AnyValue::Null)
    if r < 0.16:                                        # cross-type literal
        return {"i": 1.5, "f": 3, "b": 1, "s": 4}[kind]
    if kind == "i":
        return int(rng.integers(-3, 12))
    if kind == "f":
        return float(rng.choice([0.0, 0.5, 1.5, -2.25, np.nan, np.inf, 7.0]))
    if kind == "b":
        return bool(rng.random() > 0.5)
    return WORDS[int(rng.integers(0, len(WORDS)))]


def _random_tree(rng, nterms, depth=0):
    """Random AND / OR / NOT tree over term indices; every term may appear several times or not at all."""
    r = rng.random()
    if depth >= 3 or r < 0.3:
        leaf = int(rng.integers(0, nterms))
        return ("not", leaf) if rng.random() < 0.25 else leaf
    if r < 0.4:
        return ("not", _random_tree(rng, nterms, depth + 1))
    op = "and" if rng.random() < 0.45 else "or"
    return (op, *[_random_tree(rng, nterms, depth + 1) for _ in range(int(rng.integers(2, 4)))])


@pytest.mark.parametrize("seed", range(N_QUERY_CASES))
def test_random_query_matches_oracle(gpu_ctx, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    sizes = [300_007, 1_000_003, 3_000_017] if os.environ.get("RV_FUZZ_BIG") else [0, 1, 63, 64, 65, 777, 4096, 20_011, 70_003]
    n = int(rng.choice(sizes))
    ncols = int(rng.integers(1, 8))
    kinds = [str(rng.choice(list("iifbs"))) for _ in range(ncols)]
    pad = int(rng.choice([0, 0, 2, 8, 64, 67]))
    cols = [_column(rng, k, n, pad) for k in kinds]
    terms = []
    for _ in range(int(rng.integers(1, 5))):
        c = int(rng.integers(0, ncols))
        if kinds[c] == "b" and rng.random() < 0.4:
            terms.append(Term(c, "is_true"))
        else:
            terms.append(Term(c, str(rng.choice(OPS)), _literal(rng, kinds[c])))
    tree = _random_tree(rng, len(terms)) if rng.random() < 0.5 else None  # OR / NOT over the terms (rv_predicate::expr)
    pred = Predicate(terms, str(rng.choice(["drops", "least"])), tree)
    proj = [int(c) for c in rng.integers(0, ncols, int(rng.integers(0, ncols + 2)))]
    want_sel = bool(rng.random() < 0.4)
    vec = int(rng.choice([0, 0, 1, 2]))
    cap_rows = int(rng.choice([0, 0, 0, 32]))   # 32: most waves outgrow their LDS slot -> the redo kernel
    depth = int(rng.choice([0, 0, 1, 2]))
    # outputs sized by a bound (rows per million): small bounds overflow -> exact counts, one re-run; String / Boolean
    # columns queued behind the pass with the same bound take the scan path then
    sizing = int(rng.choice([0, 0, 0, 1, 2_000, 300_000]))
    d = [gpu_ctx.upload(c) for c in cols]
    gpu_ctx.set_option("vec", vec)
    gpu_ctx.set_option("cap_rows", cap_rows)
    gpu_ctx.set_option("depth", depth)
    gpu_ctx.set_option("out_sizing", sizing)
    try:
        outs, rows, sel = gpu_ctx.filter_project(d, pred, proj, want_sel)
    finally:
        gpu_ctx.set_option("vec", 0)
        gpu_ctx.set_option("cap_rows", 0)
        gpu_ctx.set_option("depth", 0)
        gpu_ctx.set_option("out_sizing", 0)
    what = f"seed={seed} n={n} vec={vec} cap_rows={cap_rows} depth={depth} out_sizing={sizing} kinds={kinds} pad={pad} terms={[(t.column, t.op, t.literal) for t in terms]} nulls={pred.nulls} expr={tree} proj={proj}"
    osel, ocnt = oracle.eval_predicate(cols, pred)
    assert rows == ocnt, what
    want = oracle.filter_project(cols, pred, proj) if proj else []
    if proj:
        assert_columns_equal([o.download() for o in outs], want, what)
    if want_sel:
        assert sel.download().same_as(osel) is None, what
    # the same query again, default options: the launch is now sized from the selectivity this predicate just had (a dense
    # one walks down to geometries with fewer rows per lane / roomier LDS slots)
    # (every second seed, and with option "direct" = 1 on every fourth: the unstaged kernel wherever the launch is eligible)
    if seed % 2 == 0:
        gpu_ctx.set_option("direct", 1 if seed % 4 == 0 else 0)
        gpu_ctx.set_option("str_tiles_from", 1 if seed % 8 != 4 else 0)  # String columns in source-tile order whatever the selectivity
        gpu_ctx.set_option("groups_by_ranges", 1 if seed % 8 in (2, 6) else 0)  # 8-byte columns the predicate does not read: after the pass, at its wave offsets
        try:
            outs2, rows2, sel2 = gpu_ctx.filter_project(d, pred, proj, want_sel)
        finally:
            gpu_ctx.set_option("direct", 0)
            gpu_ctx.set_option("str_tiles_from", 0)
            gpu_ctx.set_option("groups_by_ranges", 0)
        assert rows2 == ocnt, "second call " + what
        if proj:
            assert_columns_equal([o.download() for o in outs2], want, "second call " + what)
        if want_sel:
            assert sel2.download().same_as(osel) is None, "second call " + what
        for o in outs2:
            o.free()
    # the same predicate through filter + SUM/COUNT over the first Int64 column, if there is one
    if "i" in kinds:
        a = kinds.index("i")
        si, _, cnt = gpu_ctx.filter_agg(d, pred, a)
        wi, _, wcnt = oracle.filter_agg(cols, pred, a)
        assert (si, cnt) == (wi, wcnt), "filter_agg " + what
    for o in outs:
        o.free()
    for c in d:
        c.free()


@pytest.mark.parametrize("seed", range(N_BATCH_CASES))
def test_random_batch_kernels_match_oracle(gpu_ctx, oracle, seed):
    """rv_filter (BooleanArray predicate), rv_take, rv_concat and the host chunk pipeline on random batches."""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([1, 64, 65, 1000, 30_001]))
    ncols = int(rng.integers(1, 7))
    kinds = [str(rng.choice(list("ifbs"))) for _ in range(ncols)]
    pad = int(rng.choice([0, 3, 64]))
    cols = [_column(rng, k, n, pad) for k in kinds]
    d = [gpu_ctx.upload(c) for c in cols]
    what = f"seed={seed} n={n} kinds={kinds} pad={pad}"
    # filter by a nullable BooleanArray (record_batch.rs:221-243)
    p = _column(rng, "b", n, int(rng.choice([0, 5])))
    outs, rows = gpu_ctx.filter(d, gpu_ctx.upload(p))
    want = oracle.filter(cols, p)
    assert rows == want[0].length, what
    assert_columns_equal([o.download() for o in outs], want, "filter " + what)
    # take
    idx = rng.integers(0, n, int(rng.integers(0, 2 * n + 1))).astype(np.uint64)
    assert_columns_equal([c.download() for c in gpu_ctx.take(d, idx)], oracle.take(cols, idx), "take " + what)
    # concat of random slices of column 0
    parts = []
    for _ in range(int(rng.integers(1, 6))):
        a = int(rng.integers(0, n))
        parts.append(cols[0].slice(a, int(rng.integers(0, n - a + 1))))
    assert gpu_ctx.concat([gpu_ctx.upload(q) for q in parts]).download().same_as(oracle.concat(parts)) is None, "concat " + what
    # host chunk pipeline with a compare on the first fixed-width / String column
    tcol = int(rng.integers(0, ncols))
    lit = _literal(rng, kinds[tcol])
    pred = Predicate([Term(tcol, "is_true") if kinds[tcol] == "b" and lit is None else Term(tcol, str(rng.choice(OPS)), lit)],
                     str(rng.choice(["drops", "least"])))
    proj = list(range(ncols))
    outs, rows = gpu_ctx.filter_project_host(cols, pred, proj, int(rng.choice([0, 64, 640, 4096])))
    assert_columns_equal([o.download() for o in outs], oracle.filter_project(cols, pred, proj), "host pipeline " + what)


N_STREAM_CASES = int(os.environ.get("RV_FUZZ_STREAM_CASES", 60))


def _random_string_column(rng, n, pad):
    """Strings of very different lengths (the gather's per-wave LDS window holds 4 KiB: long elements take the fallback),
    many empty ones, multi-byte UTF-8, nulls."""
    total = n + pad
    mode = str(rng.choice(["short", "mixed", "long", "empty", "same"]))
    hi = {"short": 17, "mixed": 120, "long": 400, "empty": 3, "same": 9}[mode]
    alphabet = ["a", "b", "z", "Q", "é", "名", " ", "0"]
    null_p = float(rng.choice([0.0, 0.05, 0.6]))
    out = []
    for _ in range(total):
        if rng.random() < null_p:
            out.append(None)
            continue
        ln = 8 if mode == "same" else int(rng.integers(0, hi))
        if mode == "mixed" and rng.random() < 0.02:
            ln = int(rng.integers(300, 900))
        out.append("".join(alphabet[k] for k in rng.integers(0, len(alphabet), ln)))
    return Column.from_strings(out).slice(pad, n)


@pytest.mark.parametrize("seed", range(N_STREAM_CASES))
def test_random_strings_batches_and_shards_match_oracle(gpu_ctx, oracle, seed):
    """String columns of any length profile through the filter path (selection -> (start, length) -> copy) at every
    selectivity, the same query cut into random RecordBatches through rv_filter_project_batches, the device-resident
    selection -> indices -> take chain, and the table cut into row-range shards (rv_group_* on this device)."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([0, 1, 63, 64, 65, 1000, 4097, 20_011]))
    pad = int(rng.choice([0, 3, 64, 67]))
    x = Column.from_numpy(rng.integers(0, 1000, n + pad).astype(np.int64), (rng.random(n + pad) > 0.1) if rng.random() < 0.5 else None).slice(pad, n)
    s = _random_string_column(rng, n, int(rng.choice([0, 5])))
    b = Column.from_numpy(rng.random(n) > 0.5, (rng.random(n) > 0.2) if rng.random() < 0.5 else None)
    cols = [x, s, b]
    thr = int(rng.choice([-1, 1000, 899, 990, 500, 998]))  # everything / nothing / 10 % / 1 % / half / ~0.1 %
    pred = Predicate([Term(0, ">", thr)], str(rng.choice(["drops", "least"])))
    proj = [[1], [0, 1], [1, 1, 2], [2, 1, 0]][int(rng.integers(0, 4))]
    what = f"seed={seed} n={n} pad={pad} thr={thr} nulls={pred.nulls} proj={proj}"
    d = [gpu_ctx.upload(c) for c in cols]
    outs, rows, sel = gpu_ctx.filter_project(d, pred, proj, True)
    want = oracle.filter_project(cols, pred, proj)
    assert rows == want[0].length, what
    assert_columns_equal([o.download() for o in outs], want, what)
    # selection -> indices -> take on the device gives the same table
    idx = gpu_ctx.selection_indices(sel)
    assert_columns_equal([c.download() for c in gpu_ctx.take_device([d[j] for j in proj], idx)], want, "take_device " + what)
    # random RecordBatches (zero-copy slices and separately uploaded ones), one launch
    if n:
        cuts = sorted(set([0, n] + [int(v) for v in rng.integers(0, n + 1, int(rng.integers(0, 9)))]))
        batches, host_batches = [], []
        for a, e in zip(cuts[:-1], cuts[1:]):
            hb = [c.slice(a, e - a) for c in cols]
            host_batches.append(hb)
            batches.append([gpu_ctx.upload(c) for c in hb] if rng.random() < 0.4 else [w.slice(a, e - a) for w in d])
        bouts, brows, bnulls, total = gpu_ctx.filter_project_batches(batches, pred, proj)
        assert total == rows, what
        at = 0
        for k, hb in enumerate(host_batches):
            wb = oracle.filter_project(hb, pred, proj)
            assert int(brows[k]) == wb[0].length, f"batch {k} " + what
            got = [gpu_ctx.slice_known(o, at, int(brows[k]), int(bnulls[k][j])).download() for j, o in enumerate(bouts)]
            assert_columns_equal(got, wb, f"batch {k} " + what)
            at += int(brows[k])
        # the chunker form: equal batches of a random size over the resident table
        chunk = int(rng.choice([1, 7, 64, 100, 1024, 5000])) if n <= 5000 else int(rng.choice([64, 100, 1024, 5000, 70_001]))
        couts, crows, cnulls, ctotal = gpu_ctx.filter_project_chunked(d, chunk, pred, proj)
        assert ctotal == rows and len(crows) == (n + chunk - 1) // chunk, what
        assert_columns_equal([o.download() for o in couts], want, f"chunked {chunk} " + what)
        k = int(rng.integers(0, len(crows)))
        wk = oracle.filter_project([c.slice(k * chunk, min(chunk, n - k * chunk)) for c in cols], pred, proj)
        at = int(crows[:k].sum())
        assert int(crows[k]) == wk[0].length, f"chunk {k} of size {chunk} " + what
        assert_columns_equal([gpu_ctx.slice_known(o, at, int(crows[k]), int(cnulls[k][j])).download() for j, o in enumerate(couts)], wk,
                             f"chunk {k} of size {chunk} " + what)
    # row-range shards on this device, gathered in rank order
    g = capi.Group([0] * int(rng.integers(1, 4)))
    try:
        sh = [g.upload(c) for c in cols]
        res, grows = g.filter_project(sh, pred, proj)
        assert grows == rows, what
        assert_columns_equal([res.column(j) for j in range(len(proj))], want, "group " + what)
        res.free()
        si, _, cnt = g.filter_agg(sh, pred, 0)
        wi, _, wcnt = oracle.filter_agg(cols, pred, 0)
        assert (si, cnt) == (wi, wcnt), "group filter_agg " + what
    finally:
        g.close()


N_EXPR_CASES = int(os.environ.get("RV_FUZZ_EXPR_CASES", 80))


@pytest.mark.parametrize("seed", range(N_EXPR_CASES))
def test_random_large_expressions_match_oracle(gpu_ctx, oracle, seed):
    """Expressions over up to twelve terms and seven columns: conjunctive normal forms at and beyond the 16-literal
    limit (either polarity), more Boolean / String predicate columns than one pass reads, value columns past the
    four-slot budget -- every lowering of normalize_predicate including the composed BooleanArray fallback."""
    rng = np.random.default_rng(13000 + seed)
    n = int(rng.choice([1, 65, 4096, 20_011]))
    ncols = int(rng.integers(1, 8))
    kinds = [str(rng.choice(list("iifbbs"))) for _ in range(ncols)]
    pad = int(rng.choice([0, 2, 64, 67]))
    cols = [_column(rng, k, n, pad) for k in kinds]
    terms = []
    for _ in range(int(rng.integers(2, 13))):
        c = int(rng.integers(0, ncols))
        if kinds[c] == "b" and rng.random() < 0.4:
            terms.append(Term(c, "is_true"))
        else:
            terms.append(Term(c, str(rng.choice(OPS)), _literal(rng, kinds[c])))
    tree = _random_tree(rng, len(terms), depth=-1 if rng.random() < 0.5 else 0)  # one level deeper half of the time
    pred = Predicate(terms, str(rng.choice(["drops", "least"])), tree)
    proj = [int(c) for c in rng.integers(0, ncols, int(rng.integers(1, 4)))]
    what = f"seed={seed} n={n} kinds={kinds} pad={pad} terms={[(t.column, t.op, t.literal) for t in terms]} nulls={pred.nulls} expr={tree} proj={proj}"
    d = [gpu_ctx.upload(c) for c in cols]
    outs, rows, sel = gpu_ctx.filter_project(d, pred, proj, True)
    osel, ocnt = oracle.eval_predicate(cols, pred)
    assert rows == ocnt, what
    assert_columns_equal([o.download() for o in outs], oracle.filter_project(cols, pred, proj), what)
    assert sel.download().same_as(osel) is None, what
    assert gpu_ctx.eval_predicate(d, pred)[1] == ocnt, "eval_predicate " + what
    if "i" in kinds:
        a = kinds.index("i")
        assert gpu_ctx.filter_agg(d, pred, a)[::2] == oracle.filter_agg(cols, pred, a)[::2], "filter_agg " + what
