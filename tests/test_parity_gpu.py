"""GPU parity suite (-m gpu): every call goes through the C ABI (librivulus_gpu.so) and is
compared bit for bit with the CPU oracle on the same inputs, and with the committed golden
vectors.  Float64 SUM is the only tolerance-based comparison (documented at the test)."""
import math

import numpy as np
import pytest

from helpers import assert_columns_equal, load_golden
from rivulus_amd import capi
from rivulus_amd.capi import (RV_BOOLEAN, RV_FLOAT64, RV_INT64, Column, Predicate, Term, synth_spec)

pytestmark = pytest.mark.gpu
CASES = load_golden()
I64_MIN, I64_MAX = -(2 ** 63), 2 ** 63 - 1


def gpu_filter_project(ctx, cols, pred, proj, want_selection=False):
    d = [ctx.upload(c) for c in cols]
    outs, rows, sel = ctx.filter_project(d, pred, proj, want_selection)
    got = [o.download() for o in outs]
    assert all(g.length == rows for g in got)
    return got, rows, (sel.download() if sel is not None else None)


# ---- golden vectors ------------------------------------------------------------------------
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_matches_golden(gpu_ctx, case):
    got, rows, _ = gpu_filter_project(gpu_ctx, case["columns"], case["predicate"], case["projection"])
    assert rows == case["rows"]
    assert_columns_equal(got, case["expected"], case["name"])


# ---- generator --------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 100_003])
@pytest.mark.parametrize("dtype", [RV_INT64, RV_FLOAT64, RV_BOOLEAN])
def test_generate_bit_exact(gpu_ctx, oracle, n, dtype):
    spec = synth_spec(dtype, seed=42, length=n, first_row=12345678901, validity_seed=44, null_percent=5,
                      true_percent=30)
    got = gpu_ctx.generate(spec).download()
    assert got.same_as(oracle.generate(spec)) is None


# ---- BASELINE config 2 shape: filter(x > lit).select([x]) ------------------------------------------
@pytest.mark.parametrize("n", [1, 64, 1023, 1024, 1025, 4097, 16383, 16384, 16385, 100_000, 1_000_003])
@pytest.mark.parametrize("lit", [-1, 899, 499, 999])  # 100 %, 10 %, 50 %, 0 %
def test_config2_sizes_and_selectivities(gpu_ctx, oracle, n, lit):
    x = oracle.generate(synth_spec(RV_INT64, seed=42, length=n))
    pred = Predicate([Term(0, ">", lit)])
    got, rows, sel = gpu_filter_project(gpu_ctx, [x], pred, [0], want_selection=True)
    assert_columns_equal(got, oracle.filter_project([x], pred, [0]), f"n={n} lit={lit}")
    osel, ocnt = oracle.eval_predicate([x], pred)
    assert rows == ocnt and sel.same_as(osel) is None


@pytest.mark.parametrize("geometry", [16 | (16 << 8), 8 | (16 << 8), 16 | (8 << 8), 32 | (8 << 8)])
@pytest.mark.parametrize("vec", [1, 2])
def test_config2_every_kernel_geometry(gpu_ctx, oracle, geometry, vec):
    n = 300_007
    x = oracle.generate(synth_spec(RV_INT64, seed=42, length=n))
    pred = Predicate([Term(0, ">", 899)])
    gpu_ctx.set_option("rows_per_lane", geometry)
    gpu_ctx.set_option("vec", vec)
    try:
        got, _, _ = gpu_filter_project(gpu_ctx, [x], pred, [0])
    finally:
        gpu_ctx.set_option("rows_per_lane", 0)
        gpu_ctx.set_option("vec", 0)
    assert_columns_equal(got, oracle.filter_project([x], pred, [0]), f"geometry={geometry:#x} vec={vec}")


@pytest.mark.parametrize("cap_rows", [64, 128, 1024])
def test_staging_rounds(gpu_ctx, oracle, cap_rows):
    """LDS staging smaller than a tile's survivors: several scatter/flush rounds per tile."""
    n = 70_001
    x = oracle.generate(synth_spec(RV_INT64, seed=42, length=n))
    v = oracle.generate(synth_spec(RV_INT64, seed=9, length=n, validity_seed=10, null_percent=20))
    pred = Predicate([Term(0, ">", 99)], "least")
    gpu_ctx.set_option("cap_rows", cap_rows)
    try:
        got, _, _ = gpu_filter_project(gpu_ctx, [x, v], pred, [0, 1])
    finally:
        gpu_ctx.set_option("cap_rows", 0)
    assert_columns_equal(got, oracle.filter_project([x, v], pred, [0, 1]), f"cap_rows={cap_rows}")


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("shape", ["config2", "config3", "f64_one_term"])
def test_write_out_depth_and_scanner(gpu_ctx, oracle, depth, shape):
    """Two / three LDS stages (option "depth"), many tiles per workgroup so that the scanner wave,
    the single-descriptor lookup and the fallback look-back all run (5e6 rows = 305+ tiles)."""
    n = 5_000_011
    if shape == "config2":
        cols = [oracle.generate(synth_spec(RV_INT64, seed=42, length=n))]
        pred, proj = Predicate([Term(0, ">", 899)]), [0]
    elif shape == "f64_one_term":
        cols = [oracle.generate(synth_spec(RV_FLOAT64, seed=43, length=n))]
        pred, proj = Predicate([Term(0, "<=", 0.125)]), [0]
    else:
        cols = [oracle.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44)),
                oracle.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))]
        pred, proj = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)]), [0, 1]
    gpu_ctx.set_option("depth", depth)
    try:
        got, rows, sel = gpu_filter_project(gpu_ctx, cols, pred, proj, want_selection=True)
    finally:
        gpu_ctx.set_option("depth", 0)
    assert_columns_equal(got, oracle.filter_project(cols, pred, proj), f"{shape} depth={depth}")
    osel, ocnt = oracle.eval_predicate(cols, pred)
    assert rows == ocnt and sel.same_as(osel) is None


@pytest.mark.parametrize("depth", [1, 2])
def test_results_do_not_depend_on_the_scanner_wave(gpu_ctx, oracle, depth):
    """Diagnostic instantiation with the scanner wave switched off (option "debug" bit 3): every tile takes the
    decoupled look-back fallback and the result is still the reference's."""
    n = 6_000_013
    x = oracle.generate(synth_spec(RV_INT64, seed=42, length=n))
    pred = Predicate([Term(0, ">", 899)])
    gpu_ctx.set_option("debug", 8)
    gpu_ctx.set_option("depth", depth)
    try:
        got, rows, sel = gpu_filter_project(gpu_ctx, [x], pred, [0], want_selection=False)
    finally:
        gpu_ctx.set_option("debug", 0)
        gpu_ctx.set_option("depth", 0)
    assert_columns_equal(got, oracle.filter_project([x], pred, [0]), f"no scanner, depth={depth}")


def test_bounded_spin_gives_up_with_a_device_error(gpu_ctx, oracle):
    """Fault injection in the diagnostic instantiation (option "debug" bit 4): tile 1 never publishes its count.  The
    scanner wave and the look-back of the tile behind it must give up after "spin_limit" polls instead of waiting forever,
    the call returns RV_ERR_DEVICE, and the context keeps working afterwards."""
    n = 1_000_003  # 62 tiles of 16 384 rows
    x = oracle.generate(synth_spec(RV_INT64, seed=42, length=n))
    pred = Predicate([Term(0, ">", 899)])
    gpu_ctx.set_option("debug", 16)
    gpu_ctx.set_option("spin_limit", 2000)
    try:
        with pytest.raises(capi.RvError) as e:
            gpu_filter_project(gpu_ctx, [x], pred, [0], want_selection=False)
        assert e.value.status == capi.STATUS_NAMES.index("RV_ERR_DEVICE") and "spin limit" in str(e.value)
    finally:
        gpu_ctx.set_option("debug", 0)
        gpu_ctx.set_option("spin_limit", 0)
    got, rows, _ = gpu_filter_project(gpu_ctx, [x], pred, [0], want_selection=False)
    assert_columns_equal(got, oracle.filter_project([x], pred, [0]), "after the injected fault")


@pytest.mark.parametrize("vec", [1, 2])
@pytest.mark.parametrize("nulls", ["drops", "least"])
def test_two_column_kernels_both_load_widths(gpu_ctx, oracle, vec, nulls):
    """FF_PROJALL and plain two-column instantiations, 8- and 16-byte loads, unaligned bitmap offsets."""
    n = 400_003
    rng = np.random.default_rng(17)
    f = Column.from_numpy(rng.random(n + 70), rng.random(n + 70) > 0.07).slice(66, n)  # 16-byte aligned, bit offset 2
    x = Column.from_numpy(rng.integers(0, 1000, n + 70).astype(np.int64), rng.random(n + 70) > 0.05).slice(4, n)
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)], nulls)
    gpu_ctx.set_option("vec", vec)
    try:
        both, _, _ = gpu_filter_project(gpu_ctx, [f, x], pred, [0, 1])   # every loaded column projected
        one, _, _ = gpu_filter_project(gpu_ctx, [f, x], pred, [1])       # predicate column f not projected
    finally:
        gpu_ctx.set_option("vec", 0)
    assert_columns_equal(both, oracle.filter_project([f, x], pred, [0, 1]), f"vec={vec} {nulls} both")
    assert_columns_equal(one, oracle.filter_project([f, x], pred, [1]), f"vec={vec} {nulls} one")


@pytest.mark.parametrize("lit", [949, 499, 49, -1])
@pytest.mark.parametrize("shape", ["one", "one_f64", "two_tested_nullable", "three_unprojected", "four", "bool_predicate", "or_tree", "bounded",
                                   "nullable_out", "nullable_three", "nulls_are_least", "nullable_pair_sliced"])
def test_direct_kernel_matches_oracle(gpu_ctx, oracle, shape, lit):
    """Option "direct" = 1: every launch whose outputs are plain value columns goes through the unstaged kernel for dense
    selections (direct_kernel.hpp) -- at any selectivity, ragged last tile, > 1 tile, nulls dropped by the predicate, a
    Boolean predicate column, an OR tree, and outputs sized too small (exact count, one re-run)."""
    n = 1_300_021
    xs, ys = synth_spec(RV_INT64, seed=42, length=n), synth_spec(RV_INT64, seed=46, length=n)
    fs, xns = synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44), synth_spec(RV_INT64, seed=42, length=n, validity_seed=45)
    bs = synth_spec(RV_BOOLEAN, seed=47, length=n, true_percent=70, validity_seed=48)
    hx, hy, hf, hxn, hb = (oracle.generate(q) for q in (xs, ys, fs, xns, bs))
    dx, dy, df, dxn, db = (gpu_ctx.generate(q) for q in (xs, ys, fs, xns, bs))
    flit = (lit + 1) / 1000.0
    host, dev, proj, pred = {
        "one": ([hx], [dx], [0], Predicate([Term(0, ">", lit)])),
        "one_f64": ([hf], [df], [0], Predicate([Term(0, ">", flit)])),  # nullable, tested: no null survives
        "two_tested_nullable": ([hf, hxn], [df, dxn], [1, 0], Predicate([Term(1, ">", lit), Term(0, ">=", 0.0)])),
        "three_unprojected": ([hx, hy, hf], [dx, dy, df], [1], Predicate([Term(0, ">", lit), Term(2, ">=", 0.0)])),
        "four": ([hx, hy, hf, hxn], [dx, dy, df, dxn], [3, 2, 1, 0], Predicate([Term(0, ">", lit), Term(2, ">=", 0.0), Term(3, ">=", 0)])),
        "bool_predicate": ([hb, hx, hy], [db, dx, dy], [2, 1], Predicate([Term(0, "is_true"), Term(1, ">", lit)])),
        "or_tree": ([hx, hy], [dx, dy], [0, 1], Predicate([Term(0, ">", lit), Term(1, "<", 100)], "drops", ("or", 0, 1))),
        "bounded": ([hx, hy], [dx, dy], [0, 1], Predicate([Term(0, ">", lit)])),
        # projected columns that KEEP nulls among the survivors: validity bits compacted with the rows, placeholder 0 under a null
        # (record_batch.rs:142-146), the bitmap dropped when no null survived (primitive.rs:179-185)
        "nullable_out": ([hx, hf], [dx, df], [0, 1], Predicate([Term(0, ">", lit)])),
        "nullable_three": ([hx, hxn, hf], [dx, dxn, df], [2, 0, 1], Predicate([Term(0, ">", lit)])),
        "nulls_are_least": ([hxn, hy], [dxn, dy], [0, 1], Predicate([Term(0, "<", lit + 1)], "least")),   # null < any value: null rows survive
        "nullable_pair_sliced": ([hxn.slice(37, n - 100), hf.slice(37, n - 100)], None, [1, 0], Predicate([Term(0, "!=", lit)], "least")),
    }[shape]
    if dev is None:
        dev = [gpu_ctx.upload(c) for c in host]
    want = oracle.filter_project(host, pred, proj)
    gpu_ctx.set_option("direct", 1)
    if shape == "bounded":
        gpu_ctx.set_option("out_sizing", 20_000)  # room for 2 % of the rows
    try:
        reruns = gpu_ctx.get_option("overflow_reruns")
        for call in range(2):
            outs, rows, sel = gpu_ctx.filter_project(dev, pred, proj, want_selection=call == 1)
            assert gpu_ctx.last_kernel().startswith("fused_direct_compact<"), gpu_ctx.last_kernel()
            assert rows == want[0].length
            assert_columns_equal([o.download() for o in outs], want, f"{shape} lit {lit} call {call}")
            if sel is not None:  # the selection bitmap written on the way: RecordBatch::filter's mask (record_batch.rs:235-240)
                assert_columns_equal([sel.download()], [oracle.eval_predicate(host, pred)[0]], f"{shape} lit {lit} selection")
                sel.free()
            [o.free() for o in outs]
        if shape == "bounded" and want[0].length > 0.03 * n:
            assert gpu_ctx.get_option("overflow_reruns") == reruns + 2
    finally:
        gpu_ctx.set_option("direct", 0)
        gpu_ctx.set_option("out_sizing", 0)


def test_direct_kernel_is_taken_for_dense_plain_selections_only(gpu_ctx, oracle):
    """Automatic choice: from kDirectFromOneColumn (thresholds.hpp: one column; 22 % / 15 % with two / more projected) known for the predicate, for value columns -- also ones that
    keep nulls among the survivors; the selection bitmap is written on the way when asked for."""
    n = 900_001
    xs, fs = synth_spec(RV_INT64, seed=42, length=n), synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44)
    hx, hf = oracle.generate(xs), oracle.generate(fs)
    dx, df = gpu_ctx.generate(xs), gpu_ctx.generate(fs)
    dense, sparse = Predicate([Term(0, ">", 199)]), Predicate([Term(0, ">", 899)])  # 80 % / 10 %

    def kernels(cols, pred, proj, sel=False):
        names = []
        for call in range(2):
            outs, rows, s = gpu_ctx.filter_project(cols, pred, proj, sel)
            names.append(gpu_ctx.last_kernel())
            [o.free() for o in outs]
            if s is not None:
                s.free()
        return names

    first, second = kernels([dx], dense, [0])
    assert first.startswith("fused_filter_compact<") and second.startswith("fused_direct_compact<1,"), (first, second)
    assert all(k.startswith("fused_filter_compact<") for k in kernels([dx], sparse, [0]))
    ks = kernels([dx, df], dense, [0, 1])  # f keeps its nulls: the direct kernel compacts the validity bits with the rows (FF_OUTVALID = 2048)
    assert ks[1].startswith("fused_direct_compact<1,1,") and int(ks[1][ks[1].index("<") + 1:-1].split(",")[4]) & 2048, ks
    got, rows, _ = gpu_ctx.filter_project([dx, df], dense, [0, 1])
    assert_columns_equal([o.download() for o in got], oracle.filter_project([hx, hf], dense, [0, 1]), "dense, nullable output")
    assert all(k.startswith("fused_direct_compact<") for k in kernels([dx], dense, [0], sel=True)[1:])  # it writes the selection bitmap too
    got, rows, _ = gpu_ctx.filter_project([dx], dense, [0])
    assert_columns_equal([o.download() for o in got], oracle.filter_project([hx], dense, [0]), "dense, direct kernel")


@pytest.mark.parametrize("shape", ["one", "three", "nullable_and", "bool_term", "string_term"])
def test_first_call_of_an_unseen_predicate_is_sized_from_a_sample(gpu_ctx, oracle, shape):
    """The reference's operators have no warm-up call (stream.rs:136-158): a predicate the context has not run over this data
    gets its selectivity from a strided sample before the launch is sized, so a dense selection's FIRST call leaves no tile to the
    redo kernel and lands on the geometry the following calls use.  The same predicate text over another table, or another
    String literal over the same table, is a predicate of its own."""
    n = 2_400_011
    xs, ys = synth_spec(RV_INT64, seed=42, length=n), synth_spec(RV_INT64, seed=46, length=n)
    fs, xns = synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44), synth_spec(RV_INT64, seed=42, length=n, validity_seed=45)
    bs = synth_spec(RV_BOOLEAN, seed=47, length=n, true_percent=85)
    hx, hy, hf, hxn, hb = (oracle.generate(q) for q in (xs, ys, fs, xns, bs))
    dx, dy, df, dxn, db = (gpu_ctx.generate(q) for q in (xs, ys, fs, xns, bs))
    rng = np.random.default_rng(3)
    hs = Column.from_strings([("keep" if r < 0.8 else "drop") + str(k) for r, k in zip(rng.random(200_000), rng.integers(0, 3, 200_000))] * 12 + ["keep0"] * 11)
    host, dev, proj, preds = {
        "one": ([hx], [dx], [0], [Predicate([Term(0, ">", 149)]), Predicate([Term(0, ">", 899)])]),
        "three": ([hx, hy, hf], [dx, dy, df], [0, 1, 2], [Predicate([Term(0, ">", 249)])]),
        "nullable_and": ([hf, hxn], [df, dxn], [0, 1], [Predicate([Term(0, ">", 0.1), Term(1, "<", 900)])]),
        "bool_term": ([hb, hx], [db, dx], [1], [Predicate([Term(0, "is_true")])]),
        "string_term": ([hs, hx], None, [1], [Predicate([Term(0, "<", "keep1")]), Predicate([Term(0, ">=", "keep2")])]),
    }[shape]
    if dev is None:
        host = [host[0], Column(hx.dtype, hx.values[:host[0].length], None, 0, host[0].length)]
        dev = [gpu_ctx.upload(c) for c in host]
    gpu_ctx.set_option("sample", 1_000_000)  # the default threshold is 2^25 rows: a pass over less costs no more than the sample
    try:
        for pred in preds:
            want = oracle.filter_project(host, pred, proj)
            taken = gpu_ctx.get_option("samples_taken")
            kernels = []
            for call in range(3):
                outs, rows, _ = gpu_ctx.filter_project(dev, pred, proj)
                kernels.append(gpu_ctx.last_kernel())
                assert gpu_ctx.get_option("last_redo_ppm") == 0, f"{shape} call {call}: tiles left to the redo kernel ({kernels})"
                assert rows == want[0].length
                assert_columns_equal([o.download() for o in outs], want, f"{shape} call {call}")
                [o.free() for o in outs]
            assert gpu_ctx.get_option("samples_taken") == taken + 1, "one sample, on the first call only"
            assert kernels[0] == kernels[1] == kernels[2], kernels
        # the same predicate over another table of the same shape is sampled again, not assumed
        other = [gpu_ctx.upload(c) for c in host]
        taken = gpu_ctx.get_option("samples_taken")
        outs, rows, _ = gpu_ctx.filter_project(other, preds[0], proj)
        assert gpu_ctx.get_option("samples_taken") == taken + 1
        assert_columns_equal([o.download() for o in outs], oracle.filter_project(host, preds[0], proj), f"{shape} other table")
    finally:
        gpu_ctx.set_option("sample", 0)


def test_dense_selection_with_string_and_boolean_columns_riding_along(gpu_ctx, oracle):
    """The columns compacted after the pass (String, Boolean) find their output rows through the pass's wave offsets: with
    the small wave ranges of the dense geometries (512 / 256 rows) as with the default's 1024."""
    n = 700_003
    rng = np.random.default_rng(31)
    words = ["", "a", "Bob", "Ünï", "zz", "a longer string value"]
    x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64))
    s = Column.from_strings([None if r < 0.1 else words[k] for r, k in zip(rng.random(n), rng.integers(0, len(words), n))])
    b = Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.2)
    host = [x, s, b]
    dev = [gpu_ctx.upload(c) for c in host]
    seen = set()
    for lit in (899, 599, 299, 49):
        pred = Predicate([Term(0, ">", lit)])
        want = oracle.filter_project(host, pred, [0, 1, 2])
        for call in range(3):
            outs, rows, _ = gpu_ctx.filter_project(dev, pred, [0, 1, 2])
            assert rows == want[0].length
            assert_columns_equal([o.download() for o in outs], want, f"x > {lit} call {call}")
            [o.free() for o in outs]
        seen.add(gpu_ctx.last_kernel())
        if lit == 49:  # 95 % survive: BASELINE configs[0]'s [name, age] shape runs on the direct kernel, which writes the selection
            assert gpu_ctx.last_kernel().startswith("fused_direct_compact<1,0,16,"), gpu_ctx.last_kernel()  # bitmap and wave offsets on the way
    assert len(seen) >= 3, seen  # the staged default, a staged geometry with fewer rows per lane, the direct kernel


@pytest.mark.parametrize("shape", ["two_nonull", "two_nullable_out", "three_nullable_out", "three_unprojected", "four"])
def test_dense_selections_on_several_columns_take_roomier_geometries(gpu_ctx, oracle, shape):
    """Several 8-byte columns: a selectivity the default geometry's LDS slots cannot hold.  The first call meets it unprepared
    (its dense tiles go to the redo kernel), the next ones size the launch from the context's last selectivity: the same or
    a smaller 16-wave geometry with the dense sizing, then the 8-wave one whose slots hold every row -- no tile redone.
    Every call gives the reference result; back at 10 % the default geometry returns."""
    n = 1_500_007
    xs, ys = synth_spec(RV_INT64, seed=42, length=n), synth_spec(RV_INT64, seed=46, length=n)
    fs, xns = synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44), synth_spec(RV_INT64, seed=42, length=n, validity_seed=45)
    hx, hy, hf, hxn = (oracle.generate(q) for q in (xs, ys, fs, xns))
    dx, dy, df, dxn = (gpu_ctx.generate(q) for q in (xs, ys, fs, xns))
    host, dev, proj, mk = {
        "two_nonull": ([hf, hxn], [df, dxn], [0, 1], lambda lit: [Term(1, ">", lit), Term(0, ">=", 0.0)]),
        "two_nullable_out": ([hx, hf], [dx, df], [0, 1], lambda lit: [Term(0, ">", lit)]),
        "three_nullable_out": ([hx, hy, hf], [dx, dy, df], [0, 1, 2], lambda lit: [Term(0, ">", lit)]),
        "three_unprojected": ([hx, hxn, hf], [dx, dxn, df], [1, 2], lambda lit: [Term(0, ">", lit)]),
        "four": ([hx, hy, hf, hxn], [dx, dy, df, dxn], [0, 1, 2, 3], lambda lit: [Term(0, ">", lit)]),
    }[shape]
    kernels = {}
    for lit in (899, 699, 399, 49, 899):
        pred = Predicate(mk(lit))
        want = oracle.filter_project(host, pred, proj)
        for call in range(3):
            outs, rows, _ = gpu_ctx.filter_project(dev, pred, proj)
            assert rows == want[0].length
            assert_columns_equal([o.download() for o in outs], want, f"{shape} x > {lit} call {call}")
            [o.free() for o in outs]
        assert gpu_ctx.get_option("last_redo_ppm") == 0, f"{shape} x > {lit}: tiles still go to the redo kernel on the third call"
        kernels.setdefault(lit, []).append(gpu_ctx.last_kernel())
    def geometry(name):  # (rows per lane, waves); the direct kernel is <np, nq, rows per lane, waves, flags>
        q = [int(t) for t in name[name.index("<") + 1:name.index(">")].split(",")]
        return (q[2], q[3]) if name.startswith("fused_direct_compact<") else (q[1], q[3])
    sparse, dense = geometry(kernels[899][0]), geometry(kernels[49][0])
    assert kernels[49][0].startswith("fused_direct_compact<") or dense[0] * dense[1] < sparse[0] * sparse[1], (
        kernels, "95 % selectivity should run on the direct kernel or on smaller tiles than 10 %")
    assert kernels[899][0] == kernels[899][1], "back at 10 % the default geometry returns"


def test_dense_tiles_take_the_redo_kernel_then_the_dense_geometry(gpu_ctx, oracle):
    """90 % selectivity: the first launch overflows the LDS slots (redo kernel rewrites those tiles), the
    context then switches to the dense geometry; both launches must give the reference result."""
    n = 2_000_003
    x = oracle.generate(synth_spec(RV_INT64, seed=42, length=n))
    pred = Predicate([Term(0, ">", 99)])
    want = oracle.filter_project([x], pred, [0])
    sparse = Predicate([Term(0, ">", 989)])
    gpu_filter_project(gpu_ctx, [x], sparse, [0])  # resets the dense-mode memory of the context
    for launch in range(3):
        got, _, _ = gpu_filter_project(gpu_ctx, [x], pred, [0])
        assert_columns_equal(got, want, f"dense launch {launch}")
    got, _, _ = gpu_filter_project(gpu_ctx, [x], sparse, [0])
    assert_columns_equal(got, oracle.filter_project([x], sparse, [0]), "sparse after dense")


# ---- host-resident table through the chunk pipeline (streaming.rs:71-133, :135-233, :343-352) ---------
@pytest.mark.parametrize("chunk_rows", [0, 64, 4096, 100_000, 1 << 20])
def test_host_chunk_pipeline_equals_whole_batch(gpu_ctx, oracle, chunk_rows):
    """rv_filter_project_host: chunked upload on a second stream + per-chunk fused pass + device concat ==
    the reference result on the whole table, for any chunk size and with sliced (offset != 0) host arrays."""
    n = 300_003
    rng = np.random.default_rng(chunk_rows + 5)
    f = Column.from_numpy(rng.random(n + 100), rng.random(n + 100) > 0.06).slice(37, n)
    x = Column.from_numpy(rng.integers(0, 1000, n + 100).astype(np.int64), rng.random(n + 100) > 0.05).slice(3, n)
    y = Column.from_numpy(rng.integers(-5, 5, n + 100).astype(np.int64)).slice(11, n)        # no bitmap
    b = Column.from_numpy(rng.random(n + 100) > 0.5, rng.random(n + 100) > 0.1).slice(21, n)  # nullable Boolean
    cols = [f, x, y, b]
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 400)])
    proj = [1, 0, 2, 3]
    outs, rows = gpu_ctx.filter_project_host(cols, pred, proj, chunk_rows)
    got = [o.download() for o in outs]
    want = oracle.filter_project(cols, pred, proj)
    assert rows == want[0].length
    assert_columns_equal(got, want, f"chunk_rows={chunk_rows}")
    if chunk_rows in (64, 4096):  # the reference-shaped pull loop over the same batches gives the same table
        assert_columns_equal(got, oracle.stream_filter_project(cols, max(chunk_rows, 64), pred, proj), "streamed")


def test_host_chunk_pipeline_edges(gpu_ctx, oracle):
    empty = Column.from_numpy(np.zeros(0, dtype=np.int64))
    outs, rows = gpu_ctx.filter_project_host([empty], Predicate([Term(0, ">", 1)]), [0], 128)
    assert rows == 0 and outs[0].download().length == 0
    # no survivor in any chunk; a null-free result drops the bitmap (primitive.rs:179-185)
    x = Column.from_numpy(np.arange(1000, dtype=np.int64), np.arange(1000) % 7 != 0)
    outs, rows = gpu_ctx.filter_project_host([x], Predicate([Term(0, ">", 5000)]), [0], 128)
    assert rows == 0
    outs, rows = gpu_ctx.filter_project_host([x], Predicate([Term(0, ">=", 0)]), [0], 192)
    assert_columns_equal([outs[0].download()], oracle.filter_project([x], Predicate([Term(0, ">=", 0)]), [0]), "all rows")
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.filter_project_host([x, empty], Predicate([Term(0, ">", 1)]), [0], 128)
    assert "same length" in e.value.message
    # pinned source buffers (rv_host_alloc): same result
    px = gpu_ctx.pinned_array(np.int64, 50_000)
    px[:] = np.arange(50_000) % 1000
    pc = Column.from_numpy(px)
    pred = Predicate([Term(0, ">", 899)])
    outs, rows = gpu_ctx.filter_project_host([pc], pred, [0], 8192)
    assert_columns_equal([outs[0].download()], oracle.filter_project([pc], pred, [0]), "pinned")


# ---- StringArray on the device (string.rs:8-147; take -> record_batch.rs:163-170) -----------------------
def _random_strings(rng, n, null_share=0.1):
    alphabet = ["a", "b", "c", "xyz", "", "Ünï", "名前", "0123456789abcdef", " "]
    out = []
    for _ in range(n):
        if rng.random() < null_share:
            out.append(None)
        else:
            out.append("".join(alphabet[k] for k in rng.integers(0, len(alphabet), rng.integers(0, 5))))
    return out


def test_string_upload_download_and_slices(gpu_ctx, oracle):
    rng = np.random.default_rng(11)
    host = Column.from_strings(_random_strings(rng, 5000))
    d = gpu_ctx.upload(host)
    assert d.download().same_as(host) is None
    s = d.slice(37, 1000).slice(5, 900)
    assert s.download().same_as(host.slice(42, 900)) is None
    assert s.null_count() == oracle.null_count(host.slice(42, 900))
    empty = gpu_ctx.upload(Column.from_strings([]))
    assert empty.download().length == 0


@pytest.mark.parametrize("n", [1, 63, 64, 65, 5000, 200_003])
def test_string_take_matches_reference(gpu_ctx, oracle, n):
    rng = np.random.default_rng(n)
    names = Column.from_strings(_random_strings(rng, n)).slice(n // 7, n - n // 7)
    ids = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64), rng.random(n) > 0.1).slice(n // 7, n - n // 7)
    idx = rng.integers(0, names.length, min(3 * n, 100_000)).astype(np.uint64)
    got = [c.download() for c in gpu_ctx.take([gpu_ctx.upload(ids), gpu_ctx.upload(names)], idx)]
    assert_columns_equal(got, oracle.take([ids, names], idx), f"take n={n}")
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.take([gpu_ctx.upload(names)], np.array([names.length], dtype=np.uint64))
    assert "out of bounds" in e.value.message


@pytest.mark.parametrize("n", [0, 5, 64, 1000, 300_001])
def test_filter_project_carries_string_columns(gpu_ctx, oracle, n):
    """Config-1 shape on the device: the predicate tests a numeric column, String columns ride along
    (selection bitmap -> row indices -> gather of offsets + bytes)."""
    rng = np.random.default_rng(n + 3)
    name = Column.from_strings(_random_strings(rng, n))
    city = Column.from_strings(_random_strings(rng, n, null_share=0.0))
    age = Column.from_numpy((18 + rng.integers(0, 50, n)).astype(np.int64))
    score = Column.from_numpy(rng.random(n), rng.random(n) > 0.08)
    cols = [name, age, score, city]
    d = [gpu_ctx.upload(c) for c in cols]
    pred = Predicate([Term(1, ">", 25), Term(2, "<", 0.75)])
    for proj in ([0], [0, 1], [3, 2, 0, 1], [1]):
        outs, rows, _ = gpu_ctx.filter_project(d, pred, proj)
        got = [o.download() for o in outs]
        want = oracle.filter_project(cols, pred, proj)
        assert rows == want[0].length
        assert_columns_equal(got, want, f"n={n} proj={proj}")


@pytest.mark.parametrize("op", ["==", "!=", "<", ">", "<=", ">="])
def test_string_compare_gives_nullable_boolean_array(gpu_ctx, oracle, op):
    rng = np.random.default_rng(3)
    col = Column.from_strings(_random_strings(rng, 10_007)).slice(3, 10_000)
    nonull = Column.from_strings(_random_strings(rng, 500, null_share=0.0))
    for c in (col, nonull):
        d = gpu_ctx.upload(c)
        for lit in ("b", "", None, 5):
            assert gpu_ctx.compare(d, op, lit).download().same_as(oracle.compare(c, op, lit)) is None, (op, lit)


@pytest.mark.parametrize("nulls", ["drops", "least"])
@pytest.mark.parametrize("op", ["==", "!=", "<", ">", "<=", ">="])
def test_string_compare_terms(gpu_ctx, oracle, op, nulls):
    """`name <op> "literal"` (plan.rs:527-547): byte-wise str ordering, AnyValue null / cross-type rules."""
    n = 50_003
    rng = np.random.default_rng(len(op) * 7 + len(nulls))
    pool = ["Bob", "Bo", "Bobby", "", "Alice", "bob", "Ünï", "名前", "B", "Bob "]
    vals = [None if rng.random() < 0.1 else pool[k] for k in rng.integers(0, len(pool), n)]
    name = Column.from_strings(vals).slice(5, n - 9)
    x = Column.from_numpy(rng.integers(0, 100, n).astype(np.int64), rng.random(n) > 0.1).slice(5, n - 9)
    cols = [name, x]
    d = [gpu_ctx.upload(c) for c in cols]
    for lit in ("Bob", "", None, 7):  # String, empty String, Null and cross-type literals
        pred = Predicate([Term(0, op, lit), Term(1, "<", 60)], nulls)
        outs, rows, sel = gpu_ctx.filter_project(d, pred, [1, 0], want_selection=True)
        want = oracle.filter_project(cols, pred, [1, 0])
        assert rows == want[0].length
        assert_columns_equal([o.download() for o in outs], want, f"{op} {lit!r} {nulls}")
        osel, ocnt = oracle.eval_predicate(cols, pred)
        assert ocnt == rows and sel.download().same_as(osel) is None


def test_filter_by_boolean_array_keeps_strings(gpu_ctx, oracle):
    n = 10_000
    rng = np.random.default_rng(8)
    name = Column.from_strings(_random_strings(rng, n))
    x = Column.from_numpy(rng.integers(-5, 5, n).astype(np.int64))
    p = Column.from_numpy(rng.random(n) > 0.6, rng.random(n) > 0.1)
    outs, rows = gpu_ctx.filter([gpu_ctx.upload(x), gpu_ctx.upload(name)], gpu_ctx.upload(p))
    want = oracle.filter([x, name], p)
    assert rows == want[0].length
    assert_columns_equal([o.download() for o in outs], want, "filter keeps strings")


@pytest.mark.parametrize("nparts", [1, 2, 17])
def test_string_concat(gpu_ctx, oracle, nparts):
    rng = np.random.default_rng(nparts)
    parts = []
    for k in range(nparts):
        m = int(rng.integers(0, 400))
        c = Column.from_strings(_random_strings(rng, m + 10, null_share=0.0 if k % 3 == 0 else 0.2))
        parts.append(c.slice(int(rng.integers(0, 10)), m))
    got = gpu_ctx.concat([gpu_ctx.upload(p) for p in parts]).download()
    assert got.same_as(oracle.concat(parts)) is None


@pytest.mark.parametrize("n", [0, 1, 64, 65, 10_007])
def test_fill_nulls_is_the_chunkers_rule(gpu_ctx, n):
    """dataframe_to_batches (streaming.rs:135-233): null -> 0 / 0.0 / false, no bitmap; String keeps its nulls."""
    rng = np.random.default_rng(n)
    valid = rng.random(n + 9) > 0.3
    xi = rng.integers(-5, 5, n + 9).astype(np.int64)
    xf = rng.random(n + 9)
    xb = rng.random(n + 9) > 0.5
    for vals in (xi, xf, xb):
        col = Column.from_numpy(vals, valid).slice(5, n)
        got = gpu_ctx.upload(col).fill_nulls().download()
        want = Column.from_numpy(np.where(valid, vals, np.zeros(1, dtype=vals.dtype))[5:5 + n])
        assert got.same_as(want) is None
        plain = Column.from_numpy(vals).slice(5, n)
        assert gpu_ctx.upload(plain).fill_nulls().download().same_as(Column.from_numpy(vals[5:5 + n])) is None
    s = Column.from_strings([None if not v else "s" for v in valid]).slice(5, n)
    assert gpu_ctx.upload(s).fill_nulls().download().same_as(s) is None


def test_long_strings_and_mixed_lengths(gpu_ctx, oracle):
    """Elements of 0..300 bytes: multi-word copies, waves whose run outgrows the LDS window (byte-wise fallback),
    odd alignments of source and destination."""
    rng = np.random.default_rng(77)
    n = 20_011
    vals = []
    for i in range(n):
        r = rng.random()
        if r < 0.1:
            vals.append(None)
        else:
            ln = int(rng.integers(0, 300)) if r < 0.5 else int(rng.integers(0, 9))
            vals.append("".join(chr(97 + (i + k) % 26) for k in range(ln)))
    name = Column.from_strings(vals).slice(7, n - 11)
    x = Column.from_numpy(rng.integers(0, 100, n).astype(np.int64)).slice(7, n - 11)
    d = [gpu_ctx.upload(x), gpu_ctx.upload(name)]
    for lit in (49, 89, 5):
        pred = Predicate([Term(0, ">", lit)])
        outs, rows, _ = gpu_ctx.filter_project(d, pred, [1, 0])
        assert_columns_equal([o.download() for o in outs], oracle.filter_project([x, name], pred, [1, 0]), f"lit={lit}")
    idx = rng.integers(0, n - 11, 5000).astype(np.uint64)
    got = gpu_ctx.take([d[1]], idx)
    assert_columns_equal([got[0].download()], oracle.take([name], idx), "take")
    again = gpu_ctx.take([got[0]], np.arange(0, 5000, 3, dtype=np.uint64))  # gather of a gathered column
    assert_columns_equal([again[0].download()], oracle.take(oracle.take([name], idx), np.arange(0, 5000, 3, dtype=np.uint64)), "take of take")


def test_strings_that_grow_along_the_rows(gpu_ctx, oracle):
    """The copy pass sizes its LDS window from the column's AVERAGE element (strings.hip, launch_str_copy): a column whose first
    rows hold short strings and whose last rows hold long ones has blocks far past that window -- they take the kernel's direct
    (not LDS-staged) path, at every alignment of source and destination."""
    rng = np.random.default_rng(78)
    n = 60_000
    vals = ["".join(chr(97 + (i + k) % 26) for k in range(int(rng.integers(0, 4)))) for i in range(n - 3000)]
    vals += ["".join(chr(65 + (i + k) % 26) for k in range(int(rng.integers(0, 90)))) for i in range(3000)]  # average stays < 5 bytes
    name = Column.from_strings(vals)
    x = Column.from_numpy(rng.integers(0, 100, n).astype(np.int64))
    d = [gpu_ctx.upload(x), gpu_ctx.upload(name)]
    for lit in (9, 49, 94):
        pred = Predicate([Term(0, ">", lit)])
        outs, rows, _ = gpu_ctx.filter_project(d, pred, [1, 0])
        assert_columns_equal([o.download() for o in outs], oracle.filter_project([x, name], pred, [1, 0]), f"lit={lit}")
    idx = rng.integers(n - 6000, n, 7000).astype(np.uint64)
    assert_columns_equal([gpu_ctx.take([d[1]], idx)[0].download()], oracle.take([name], idx), "take")


@pytest.mark.parametrize("kind", ["short", "nullable", "long", "sliced", "growing"])
def test_string_columns_in_source_tile_order(gpu_ctx, oracle, kind):
    """Dense selections copy String columns tile by tile of 512 SOURCE rows (sel_str_tile_sums / sel_str_tile_copy, queued behind the
    pass with no survivor count in between) instead of block by block of survivors: forced here at every selectivity, from nothing
    to everything, over the ranges of every geometry the pass picks on the way (1024 / 512 / 256 rows per wave), with nulls, with
    strings past the LDS window, with sliced columns, and with a second String column launched after the pass has finished."""
    rng = np.random.default_rng({"short": 1, "nullable": 2, "long": 3, "sliced": 4, "growing": 5}[kind])
    n = 300_007
    if kind == "long":
        vals = ["".join(chr(97 + (i + k) % 26) for k in range(int(rng.integers(0, 120)) if rng.random() < 0.3 else int(rng.integers(0, 9)))) for i in range(n)]
    elif kind == "growing":
        vals = ["ab"[: int(rng.integers(0, 3))] for _ in range(n - 4000)] + ["".join(chr(65 + (i + k) % 26) for k in range(int(rng.integers(0, 200)))) for i in range(4000)]
    else:
        vals = _random_strings(rng, n, null_share=0.15 if kind in ("nullable", "sliced") else 0.0)
    name = Column.from_strings(vals)
    other = Column.from_strings(_random_strings(rng, n, null_share=0.05))
    x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64))
    cols = [x, name, other]
    if kind == "sliced":
        cols = [c.slice(13, n - 40) for c in cols]
    d = [gpu_ctx.upload(c) for c in cols]
    gpu_ctx.set_option("str_tiles_from", 1)
    try:
        for lit in (-1, 9, 159, 499, 899, 997, 1000):
            pred = Predicate([Term(0, ">", lit)])
            want = oracle.filter_project(cols, pred, [1, 0, 2])
            for call in range(2):  # the second call is sized from the selectivity of the first: another geometry, other ranges
                outs, rows, _ = gpu_ctx.filter_project(d, pred, [1, 0, 2])
                assert rows == want[0].length
                assert_columns_equal([o.download() for o in outs], want, f"{kind} x > {lit} call {call}")
                [o.free() for o in outs]
    finally:
        gpu_ctx.set_option("str_tiles_from", 0)


def test_boolean_columns_ride_along_at_every_selectivity(gpu_ctx, oracle):
    """Projected Boolean columns (values + validity) are compacted by the selection bitmap after the pass (bits_compact_kernel, at the
    pass's wave offsets): nullable and plain columns, a sliced frame (bit offsets off the word grid), from no survivor to all of
    them, and RecordBatch::filter by a nullable BooleanArray over the same frame."""
    rng = np.random.default_rng(91)
    n = 1_000_003
    x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64))
    b = Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.2)
    c = Column.from_numpy(rng.random(n) > 0.3)
    host = [col.slice(37, n - 100) for col in (x, b, c)]
    dev = [gpu_ctx.upload(col) for col in host]
    for lit in (-1, 99, 499, 899, 998, 1000):
        pred = Predicate([Term(0, ">", lit)])
        want = oracle.filter_project(host, pred, [1, 0, 2])
        outs, rows, _ = gpu_ctx.filter_project(dev, pred, [1, 0, 2])
        assert rows == want[0].length
        assert_columns_equal([o.download() for o in outs], want, f"x > {lit}")
        [o.free() for o in outs]
    # the bitmaps are sized for the survivors the pass expects + 25 %; more than that (forced: 1000 rows) and the column takes the
    # scan path after the pass instead
    gpu_ctx.set_option("bool_cap", 1000)
    try:
        for lit in (499, 998):
            pred = Predicate([Term(0, ">", lit)])
            outs, rows, _ = gpu_ctx.filter_project(dev, pred, [1, 0, 2])
            assert_columns_equal([o.download() for o in outs], oracle.filter_project(host, pred, [1, 0, 2]), f"capped, x > {lit}")
            [o.free() for o in outs]
    finally:
        gpu_ctx.set_option("bool_cap", 0)
    mask = Column.from_numpy(rng.random(n - 100) > 0.4, rng.random(n - 100) > 0.1)
    got, _ = gpu_ctx.filter(dev, gpu_ctx.upload(mask))
    assert_columns_equal([o.download() for o in got], oracle.filter(host, mask), "filter by mask")


def test_null_array_columns(gpu_ctx, oracle):
    """NullArray (null.rs:5-66) columns ride through filter_project / filter / take / concat."""
    n = 5000
    rng = np.random.default_rng(2)
    x = Column.from_numpy(rng.integers(0, 100, n).astype(np.int64), rng.random(n) > 0.1)
    nothing = Column.nulls(n)
    cols = [x, nothing]
    d = [gpu_ctx.upload(c) for c in cols]
    assert d[1].null_count() == n and d[1].slice(10, 100).null_count() == 100
    pred = Predicate([Term(0, "<", 30)])
    outs, rows, _ = gpu_ctx.filter_project(d, pred, [1, 0, 1])
    assert_columns_equal([o.download() for o in outs], oracle.filter_project(cols, pred, [1, 0, 1]), "filter_project")
    p = Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.1)
    outs, rows = gpu_ctx.filter(d, gpu_ctx.upload(p))
    assert_columns_equal([o.download() for o in outs], oracle.filter(cols, p), "filter")
    idx = rng.integers(0, n, 77).astype(np.uint64)
    assert_columns_equal([c.download() for c in gpu_ctx.take(d, idx)], oracle.take(cols, idx), "take")
    parts = [nothing.slice(0, 10), nothing.slice(5, 0), nothing]
    assert gpu_ctx.concat([gpu_ctx.upload(q) for q in parts]).download().same_as(oracle.concat(parts)) is None


def test_string_edge_cases(gpu_ctx, oracle):
    rng = np.random.default_rng(21)
    n = 4097
    all_null = Column.from_strings([None] * n)
    empties = Column.from_strings([""] * n)
    wide = [Column.from_numpy(rng.integers(0, 10, n).astype(np.int64)) for _ in range(6)]  # > 4 columns: several groups
    flag = Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.2)                      # nullable Boolean rides along
    name = Column.from_strings(_random_strings(rng, n))
    cols = wide + [all_null, empties, flag, name]
    d = [gpu_ctx.upload(c) for c in cols]
    pred = Predicate([Term(0, ">", 4), Term(9, "!=", "a")], "least")
    proj = [9, 6, 0, 1, 2, 3, 4, 5, 8, 7]
    outs, rows, _ = gpu_ctx.filter_project(d, pred, proj)
    want = oracle.filter_project(cols, pred, proj)
    assert rows == want[0].length
    assert_columns_equal([o.download() for o in outs], want, "wide batch with strings")
    # take with repeated indices, zero indices
    idx = np.array([5, 5, 5, 0, n - 1, 5], dtype=np.uint64)
    got = [c.download() for c in gpu_ctx.take([d[9], d[6], d[7]], idx)]
    assert_columns_equal(got, oracle.take([name, all_null, empties], idx), "repeated indices")
    got = [c.download() for c in gpu_ctx.take([d[9]], np.zeros(0, dtype=np.uint64))]
    assert got[0].length == 0 and got[0].validity is None
    # host chunk pipeline with String columns (cut at element boundaries, offsets rebased per chunk), sliced inputs
    hcols = [c.slice(3, n - 5) for c in cols]
    for chunk in (64, 1000, 0):
        outs, rows = gpu_ctx.filter_project_host(hcols, pred, proj, chunk)
        assert_columns_equal([o.download() for o in outs], oracle.filter_project(hcols, pred, proj), f"host pipeline chunk={chunk}")


# ---- the fused pass in two halves: several launches in flight ------------------------------------------
@pytest.mark.parametrize("depth_in_flight", [1, 2, 5])
def test_begin_finish_pipelined_batches(gpu_ctx, oracle, depth_in_flight):
    """rv_filter_project_begin / _finish: batch k+1 is queued before batch k is finished; every batch gives the
    reference result, whatever the number of launches in flight (each has its own control block)."""
    n, b = 1_000_003, 70_001
    rng = np.random.default_rng(depth_in_flight)
    f = Column.from_numpy(rng.random(n), rng.random(n) > 0.05)
    x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64), rng.random(n) > 0.05)
    name = Column.from_strings(_random_strings(rng, n))
    df, dx, dn = gpu_ctx.upload(f), gpu_ctx.upload(x), gpu_ctx.upload(name)
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 300)])
    starts = list(range(0, n, b))
    queue, results = [], []
    for k, s0 in enumerate(starts):
        ln = min(b, n - s0)
        cols = [df.slice(s0, ln), dx.slice(s0, ln)]
        proj = [1, 0]
        if k % 4 == 3:  # a String column rides along: completed inside begin (several passes)
            cols.append(dn.slice(s0, ln))
            proj = [2, 1, 0]
        queue.append((k, s0, ln, proj, gpu_ctx.filter_project_begin(cols, pred, proj)))
        if len(queue) > depth_in_flight:
            results.append(queue.pop(0))
    results += queue
    total = 0
    for k, s0, ln, proj, finish in results:
        outs, rows = finish()
        host = [f.slice(s0, ln), x.slice(s0, ln), name.slice(s0, ln)]
        want = oracle.filter_project(host, pred, proj)
        assert rows == want[0].length
        assert_columns_equal([o.download() for o in outs], want, f"batch {k} in flight {depth_in_flight}")
        total += rows
    assert total == oracle.eval_predicate([f, x], pred)[1]


# ---- many RecordBatches, one launch (seam S1 at the reference's 1024-row batch size) -----------------------------------
@pytest.mark.parametrize("layout", ["slices", "separate", "mixed"])
@pytest.mark.parametrize("nulls", ["drops", "least"])
def test_filter_project_batches_equals_per_batch_calls(gpu_ctx, oracle, layout, nulls):
    """rv_filter_project_batches: K batches in, one pass, K output batches back to back; every output batch (cut out
    with rv_slice_known) equals RecordBatch::filter + select on that input batch, bitmap dropped where no null survived,
    and the concatenation equals the reference-shaped pull loop (streaming.rs:343-352)."""
    rng = np.random.default_rng(len(layout))
    words = ["", "a", "Bob", "Ünï", "zz"]
    lengths = [1024] * 9 + [0, 1, 63, 1024, 777, 0, 1024, 5000, 64, 1024]
    n = sum(lengths)
    valid_x = rng.random(n) > 0.1
    valid_x[1024 * 3:1024 * 5] = True   # two batches without any null: their output bitmaps are dropped
    f = Column.from_numpy(rng.random(n), rng.random(n) > 0.05)
    x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64), valid_x)
    s = Column.from_strings([None if rng.random() < 0.1 else words[k] for k in rng.integers(0, len(words), n)])
    b = Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.2)
    host = [f, x, s, b]
    whole = [gpu_ctx.upload(c) for c in host]
    starts = np.concatenate([[0], np.cumsum(lengths)])
    batches, host_batches = [], []
    for k, ln in enumerate(lengths):
        hb = [c.slice(int(starts[k]), ln) for c in host]
        host_batches.append(hb)
        separate = layout == "separate" or (layout == "mixed" and k % 5 in (2, 3))
        batches.append([gpu_ctx.upload(c) for c in hb] if separate else [w.slice(int(starts[k]), ln) for w in whole])
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 600)], nulls)
    proj = [1, 0, 2, 3]
    outs, rows, nulls_out, total = gpu_ctx.filter_project_batches(batches, pred, proj)
    assert len(rows) == len(lengths) and int(rows.sum()) == total
    at = 0
    for k, hb in enumerate(host_batches):
        want = oracle.filter_project(hb, pred, proj)
        assert int(rows[k]) == want[0].length, f"batch {k}"
        got = [gpu_ctx.slice_known(o, at, int(rows[k]), int(nulls_out[k][j])).download() for j, o in enumerate(outs)]
        assert_columns_equal(got, want, f"{layout} {nulls} batch {k}")
        at += int(rows[k])
    assert_columns_equal([o.download() for o in outs], oracle.stream_filter_project(host, 1024, pred, proj)
                         if lengths == [1024] * len(lengths) else oracle.filter_project(host, pred, proj), "all batches")
    # one batch: the same as rv_filter_project
    outs1, rows1, n1, t1 = gpu_ctx.filter_project_batches([batches[0]], pred, proj)
    assert_columns_equal([o.download() for o in outs1], oracle.filter_project(host_batches[0], pred, proj), "one batch")
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.filter_project_batches([batches[0], [batches[1][1], batches[1][0], batches[1][2], batches[1][3]]], pred, proj)
    assert "same schema" in e.value.message


def test_filter_project_batches_many_small_batches(gpu_ctx, oracle):
    """More than 16 384 batches: the handle walk runs on several host threads; same result, and the error reported is the
    FIRST offending batch's, as from the sequential walk."""
    n = 20_000
    rng = np.random.default_rng(77)
    x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64), rng.random(n) > 0.1)
    y = Column.from_numpy(rng.random(n))
    dx, dy = gpu_ctx.upload(x), gpu_ctx.upload(y)
    pred = Predicate([Term(0, ">", 500), Term(1, "<", 0.8)])
    batches = [[dx.slice(i, 1), dy.slice(i, 1)] for i in range(n)]
    outs, rows, nulls_out, total = gpu_ctx.filter_project_batches(batches, pred, [1, 0])
    want = oracle.filter_project([x, y], pred, [1, 0])
    assert total == want[0].length and int(rows.sum()) == total and rows.max() <= 1
    assert_columns_equal([o.download() for o in outs], want, "20 000 one-row batches")
    sel, _ = oracle.eval_predicate([x, y], pred)
    assert np.array_equal(rows.astype(bool), sel.logical_values())
    # two batches out of place: the runs split, the output follows the batch order given
    order = list(range(n))
    order[100], order[15_000] = order[15_000], order[100]
    outs2, rows2, _, total2 = gpu_ctx.filter_project_batches([batches[i] for i in order], pred, [1, 0])
    perm = np.array(order)
    assert total2 == total and np.array_equal(rows2, rows[perm])
    want2 = oracle.filter_project([oracle.take([x], perm.astype(np.uint64))[0], oracle.take([y], perm.astype(np.uint64))[0]], pred, [1, 0])
    assert_columns_equal([o.download() for o in outs2], want2, "permuted batches")
    # errors: batch 12 345 (length 3) comes before batch 17 000 (length 2)
    bad = list(batches)
    bad[17_000] = [dx.slice(17_000, 1), dy.slice(0, 2)]
    bad[12_345] = [dx.slice(12_345, 1), dy.slice(0, 3)]
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.filter_project_batches(bad, pred, [1, 0])
    assert e.value.message == "Column 1 has length 3 but expected 1"


def test_filter_project_batches_reference_batch_size_config3(gpu_ctx, oracle):
    """BASELINE configs[2] through the batched seam: 1024-row batches (streaming_planner.rs:32) cut from resident columns."""
    n = 1_000_003
    fs, xs = synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44), synth_spec(RV_INT64, seed=42, length=n, validity_seed=45)
    df, dx = gpu_ctx.generate(fs), gpu_ctx.generate(xs)
    batches = [[df.slice(o, min(1024, n - o)), dx.slice(o, min(1024, n - o))] for o in range(0, n, 1024)]
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
    outs, rows, nulls_out, total = gpu_ctx.filter_project_batches(batches, pred, [0, 1])
    hf, hx = oracle.generate(fs), oracle.generate(xs)
    want = oracle.stream_filter_project([hf, hx], 1024, pred, [0, 1])
    assert total == want[0].length and not nulls_out.any()
    assert_columns_equal([o.download() for o in outs], want, "config 3, 1024-row batches")
    sel, _ = oracle.eval_predicate([hf, hx], pred)
    bits = sel.logical_values()
    assert [int(r) for r in rows] == [int(bits[o:o + 1024].sum()) for o in range(0, n, 1024)]


def test_filter_project_batches_regular_stream_on_walk_threads(gpu_ctx, oracle):
    """17 000 batches of 1024 rows, the last one short: a REGULAR stream (adjacent slices of one length) -- validated on
    several host threads without per-batch tables, counts out of the pass (counter batch_counts_in_pass), written into a
    pinned array.  Then the same stream with its last two batches merged into one of 1524 rows (irregular lengths: the
    recorded walk + the selection-bitmap count) and with one batch taken from a copy of the column (two runs + concat)."""
    b, nb = 1024, 17_000
    n = nb * b - 500
    spec = synth_spec(RV_INT64, seed=42, length=n)
    dx = gpu_ctx.generate(spec)
    hx = oracle.generate(spec)
    vals = hx.logical_values()
    pred = Predicate([Term(0, ">", 899)])
    keep = vals > 899
    want_rows = np.add.reduceat(keep.astype(np.uint64), np.arange(0, n, b))
    batches = [[dx.slice(o, min(b, n - o))] for o in range(0, n, b)]
    kept = gpu_ctx.pinned_array(np.uint64, nb)
    before = gpu_ctx.get_option("batch_counts_in_pass")
    spec_passes = gpu_ctx.get_option("speculative_batch_passes")
    outs, rows, _, total = gpu_ctx.filter_project_batches(batches, pred, [0], want_nulls=False, rows_buffer=kept)
    assert gpu_ctx.get_option("batch_counts_in_pass") == before + 1
    assert gpu_ctx.get_option("speculative_batch_passes") == spec_passes + 1  # 4096 batches and more: the pass ran while the walk validated
    assert total == int(keep.sum()) and np.array_equal(rows, want_rows)
    assert np.array_equal(outs[0].download().logical_values(), vals[keep])
    # irregular lengths: ... | 1024 | 1524
    merged = batches[:-2] + [[dx.slice((nb - 2) * b, n - (nb - 2) * b)]]
    outs2, rows2, _, total2 = gpu_ctx.filter_project_batches(merged, pred, [0], want_nulls=False)
    assert total2 == total and np.array_equal(rows2[:-1], want_rows[:-2]) and int(rows2[-1]) == int(want_rows[-2] + want_rows[-1])
    assert np.array_equal(outs2[0].download().logical_values(), vals[keep])
    # one batch from another buffer: three runs, joined by a device concat in front of the pass
    other = gpu_ctx.generate(spec)
    mixed = list(batches)
    mixed[9_000] = [other.slice(9_000 * b, b)]
    outs3, rows3, _, total3 = gpu_ctx.filter_project_batches(mixed, pred, [0], want_nulls=False)
    assert gpu_ctx.get_option("speculative_batch_passes") == spec_passes + 1  # (launched on its regular looks, dropped by the walk's verdict)
    assert total3 == total and np.array_equal(rows3, want_rows)
    assert np.array_equal(outs3[0].download().logical_values(), vals[keep])


@pytest.mark.parametrize("chunk", [1, 63, 64, 1000, 1024, 4096, 262_144 + 5, 10_000_000])
@pytest.mark.parametrize("nulls", ["drops", "least"])
def test_filter_project_chunked_equals_the_references_chunker(gpu_ctx, oracle, chunk, nulls):
    """rv_filter_project_chunked: DataFrameSource -> Filter -> Select at batch size `chunk` over one resident table
    (dataframe_to_batches, streaming.rs:135-233), one pass; every output batch equals filter + select on the input batch,
    bitmap dropped where no null survived; the whole equals the reference-shaped pull loop."""
    n = 300_007 if chunk > 1 else 3000
    rng = np.random.default_rng(chunk)
    words = ["", "a", "Bob", "Ünï", "zz"]
    valid_x = rng.random(n + 70) > 0.1
    valid_x[70 + 2048:70 + 4096] = True  # batches without a null: their output bitmaps are dropped
    f = Column.from_numpy(rng.random(n), rng.random(n) > 0.05)
    x = Column.from_numpy(rng.integers(0, 1000, n + 70).astype(np.int64), valid_x).slice(70, n)
    s = Column.from_strings([None if rng.random() < 0.1 else words[k] for k in rng.integers(0, len(words), n)])
    b = Column.from_numpy(rng.random(n + 3) > 0.5, rng.random(n + 3) > 0.2).slice(3, n)
    host = [f, x, s, b]
    d = [gpu_ctx.upload(c) for c in host]
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 600)], nulls)
    proj = [1, 0, 2, 3]
    outs, rows, nulls_out, total = gpu_ctx.filter_project_chunked(d, chunk, pred, proj)
    nb = (n + chunk - 1) // chunk
    assert len(rows) == nb and int(rows.sum()) == total
    want_all = oracle.filter_project(host, pred, proj)
    assert total == want_all[0].length
    assert_columns_equal([o.download() for o in outs], want_all, f"chunk {chunk} {nulls}: all batches")
    # the same numbers as the handle-array form over rv_slice views
    step = max(1, nb // 40)   # every batch when there are few, a sample otherwise
    starts = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    for k in list(range(0, nb, step)) + [nb - 1]:
        hb = [c.slice(k * chunk, min(chunk, n - k * chunk)) for c in host]
        want = oracle.filter_project(hb, pred, proj)
        assert int(rows[k]) == want[0].length, f"batch {k}"
        got = [gpu_ctx.slice_known(o, int(starts[k]), int(rows[k]), int(nulls_out[k][j])).download() for j, o in enumerate(outs)]
        assert_columns_equal(got, want, f"chunk {chunk} {nulls} batch {k}")
    if chunk == 1024:
        assert_columns_equal([o.download() for o in outs], oracle.stream_filter_project(host, 1024, pred, proj), "pull loop")
        batches = [[w.slice(k * chunk, min(chunk, n - k * chunk)) for w in d] for k in range(nb)]
        outs2, rows2, nulls2, total2 = gpu_ctx.filter_project_batches(batches, pred, proj)
        assert total2 == total and np.array_equal(rows2, rows) and np.array_equal(nulls2, nulls_out)


@pytest.mark.parametrize("chunk", [512, 1024, 2048, 3072, 16_384, 65_536, 1 << 20, 1000])
@pytest.mark.parametrize("shape", ["config2", "config3", "nullable_out", "dense", "bounded_outputs"])
def test_per_batch_counts_out_of_the_pass(gpu_ctx, oracle, chunk, shape):
    """Batches that are a whole number of the pass's 1024-row wave ranges get their survivor counts from the fused pass
    itself (no selection bitmap); the caller's array may be pinned (the device writes it) or pageable.  Same numbers as
    counting the oracle's selection per batch; other batch sizes (1000, 512, 3072 with 1024-row ranges) take the bitmap."""
    n = 5_000_123
    if shape == "config3":
        specs = [synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44), synth_spec(RV_INT64, seed=42, length=n, validity_seed=45)]
        pred, proj = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)]), [0, 1]
    elif shape == "nullable_out":  # the projected column keeps nulls: per-batch null counts are taken from the compacted bitmap
        specs = [synth_spec(RV_INT64, seed=42, length=n), synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44)]
        pred, proj = Predicate([Term(0, ">", 799)]), [1, 0]
    else:
        specs = [synth_spec(RV_INT64, seed=42, length=n)]
        pred, proj = Predicate([Term(0, ">", -1 if shape == "dense" else 899)]), [0]
    d = [gpu_ctx.generate(s) for s in specs]
    host = [oracle.generate(s) for s in specs]
    nb = (n + chunk - 1) // chunk
    # the launch geometry follows the context's last selectivity: make it this query's, so both calls below take the same one
    primed, _, _ = gpu_ctx.filter_project(d, pred, proj)
    [o.free() for o in primed]
    if shape == "bounded_outputs":
        gpu_ctx.set_option("out_sizing", 20_000)  # 2 % bound, 10 % survive: the pass overflows, counts, and is re-run
    try:
        before = gpu_ctx.get_option("batch_counts_in_pass")
        pinned = gpu_ctx.pinned_array(np.uint64, nb)
        pinned[:] = 0xDEAD
        outs, rows, nulls_out, total = gpu_ctx.filter_project_chunked(d, chunk, pred, proj, rows_buffer=pinned)
        outs2, rows2, nulls2, total2 = gpu_ctx.filter_project_chunked(d, chunk, pred, proj)  # pageable array
    finally:
        gpu_ctx.set_option("out_sizing", 0)
    in_pass = gpu_ctx.get_option("batch_counts_in_pass") - before
    # the geometries of these shapes stage 1024-, 512- or 256-row wave ranges (16 / 8 / 4 rows per lane; dense data takes the
    # small ones)
    assert in_pass in (0, 2)
    if chunk % 1024 == 0:
        assert in_pass == 2
    if chunk % 512:
        assert in_pass == 0
    bits = oracle.eval_predicate(host, pred)[0].logical_values()
    want_rows = np.add.reduceat(bits.astype(np.uint64), np.arange(0, n, chunk))
    assert np.array_equal(rows[:nb], want_rows) and np.array_equal(rows2, want_rows) and total == total2 == int(want_rows.sum())
    want = oracle.filter_project(host, pred, proj)
    assert_columns_equal([o.download() for o in outs], want, f"{shape} chunk {chunk}")
    assert np.array_equal(nulls_out, nulls2)
    starts = np.concatenate([[0], np.cumsum(want_rows)]).astype(np.int64)
    for j, w in enumerate(want):
        if w.validity is None:
            assert not nulls_out[:, j].any()
        else:
            inv = (~w.logical_valid()).astype(np.int64)
            want_nulls = np.array([int(inv[starts[k]:starts[k + 1]].sum()) for k in range(nb)])
            assert np.array_equal(nulls_out[:, j], want_nulls)
    for o in outs + outs2 + d:
        o.free()


def test_filter_project_chunked_edge_cases(gpu_ctx, oracle):
    x = gpu_ctx.upload(Column.from_numpy(np.arange(10, dtype=np.int64)))
    pred = Predicate([Term(0, ">=", 4)])
    outs, rows, nulls_out, total = gpu_ctx.filter_project_chunked([x], 4, pred, [0])
    assert [int(r) for r in rows] == [0, 4, 2] and total == 6 and not nulls_out.any()
    empty = gpu_ctx.upload(Column.from_numpy(np.zeros(0, dtype=np.int64)))
    outs, rows, _, total = gpu_ctx.filter_project_chunked([empty], 1024, pred, [0])   # an empty frame yields no batch
    assert len(rows) == 0 and total == 0 and outs[0].length == 0
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.filter_project_chunked([x], 0, pred, [0])
    assert "chunk_rows" in e.value.message


@pytest.mark.parametrize("n", [1, 64, 65, 1000, 16_385, 300_007])
def test_boolean_columns_compacted_inside_the_pass(gpu_ctx, oracle, n):
    """Option "bools_in_pass": projected Boolean columns ride through the fused pass as bit streams (lane-form PEXT per
    64-row word) instead of the bit-compaction kernel after it; same arrays either way (boolean.rs:29-32, :255-297)."""
    rng = np.random.default_rng(n)
    pad = 67
    x = Column.from_numpy(rng.integers(0, 100, n + pad).astype(np.int64), rng.random(n + pad) > 0.1).slice(pad, n)
    b1 = Column.from_numpy(rng.random(n + pad) > 0.5, rng.random(n + pad) > 0.2).slice(pad, n)
    b2 = Column.from_numpy(rng.random(n + 3) > 0.3).slice(3, n)
    f = Column.from_numpy(rng.random(n))
    cols = [x, b1, b2, f]
    d = [gpu_ctx.upload(c) for c in cols]
    gpu_ctx.set_option("bools_in_pass", 1)
    try:
        for pred, proj in [(Predicate([Term(1, "is_true")]), [0, 1]), (Predicate([Term(0, "<", 30)], "least"), [1, 0, 2]),
                           (Predicate([Term(3, ">", 0.5), Term(2, "==", False)]), [2, 1, 3, 0]), (Predicate([Term(0, ">=", 0)]), [1, 2]),
                           (Predicate([Term(0, "<", 40)]), [1])]:
            outs, rows, sel = gpu_ctx.filter_project(d, pred, proj, want_selection=True)
            want = oracle.filter_project(cols, pred, proj)
            assert rows == want[0].length
            assert_columns_equal([o.download() for o in outs], want, f"n={n} {pred.terms} proj={proj}")
            assert sel.download().same_as(oracle.eval_predicate(cols, pred)[0]) is None
    finally:
        gpu_ctx.set_option("bools_in_pass", 0)


def test_speculative_output_sizing_reruns_on_overflow(gpu_ctx, oracle):
    """Option "out_sizing": outputs sized from a bound (or the last selectivity) instead of for every row.  A launch
    that overflows still counts exactly and is re-run with outputs of that size: the result never depends on the bound."""
    n = 2_000_003
    x = oracle.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
    f = oracle.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
    b = Column.from_numpy(np.arange(n) % 3 == 0, np.arange(n) % 7 != 0)
    cols = [x, f, b]
    d = [gpu_ctx.upload(c) for c in cols]
    queries = [(Predicate([Term(0, ">", 899)]), [0]), (Predicate([Term(0, "<", 500)], "least"), [0, 1, 2]),
               (Predicate([Term(1, ">", 0.5), Term(0, "<", 200)]), [1, 0]), (Predicate([Term(0, ">=", 0)]), [0, 2])]
    try:
        for sizing, expect_rerun in [(20_000, True), (990_000, False), (2, True), (1, None)]:   # 2 %, 99 %, 2 ppm, adaptive
            gpu_ctx.set_option("out_sizing", sizing)
            for pred, proj in queries:
                before = gpu_ctx.get_option("overflow_reruns")
                outs, rows, sel = gpu_ctx.filter_project(d, pred, proj, want_selection=True)
                want = oracle.filter_project(cols, pred, proj)
                assert rows == want[0].length
                assert_columns_equal([o.download() for o in outs], want, f"out_sizing={sizing} {pred.terms}")
                assert sel.download().same_as(oracle.eval_predicate(cols, pred)[0]) is None
                reran = gpu_ctx.get_option("overflow_reruns") - before
                if expect_rerun is not None:
                    assert (reran > 0) == expect_rerun, (sizing, pred.terms, reran)
                assert abs(gpu_ctx.get_option("last_selectivity_ppm") - rows / n * 1e6) < 2
        # adaptive: the second identical query fits without a re-run
        gpu_ctx.set_option("out_sizing", 1)
        gpu_ctx.filter_project(d, queries[0][0], [0])
        before = gpu_ctx.get_option("overflow_reruns")
        gpu_ctx.filter_project(d, queries[0][0], [0])
        assert gpu_ctx.get_option("overflow_reruns") == before
    finally:
        gpu_ctx.set_option("out_sizing", 0)
    with pytest.raises(capi.RvError):
        gpu_ctx.set_option("out_sizing", -5)


# ---- offsets / slices (primitive.rs:107-117, bitmap.rs:104-112) --------------------------------------
@pytest.mark.parametrize("offset", [1, 7, 9, 63, 64, 65, 130])
def test_sliced_inputs(gpu_ctx, oracle, offset):
    n = 20_000
    rng = np.random.default_rng(offset)
    x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64), rng.random(n) > 0.1).slice(offset, n - offset - 3)
    f = Column.from_numpy(rng.random(n), rng.random(n) > 0.1).slice(offset, n - offset - 3)
    pred = Predicate([Term(1, ">", 0.5), Term(0, "<", 800)])
    got, _, sel = gpu_filter_project(gpu_ctx, [x, f], pred, [1, 0], want_selection=True)
    assert_columns_equal(got, oracle.filter_project([x, f], pred, [1, 0]), f"offset={offset}")
    assert sel.same_as(oracle.eval_predicate([x, f], pred)[0]) is None


def test_device_slice_is_zero_copy_view(gpu_ctx, oracle):
    n = 5000
    rng = np.random.default_rng(3)
    host = Column.from_numpy(rng.integers(-50, 50, n).astype(np.int64), rng.random(n) > 0.2)
    d = gpu_ctx.upload(host)
    s = d.slice(37, 1000).slice(5, 900)  # chained slice == single slice (bitmap.rs:283-309)
    assert s.download().same_as(oracle.slice_(host, 42, 900)) is None
    assert s.null_count() == oracle.null_count(host.slice(42, 900))
    assert s.device_ptrs().values == d.device_ptrs().values
    with pytest.raises(capi.RvError) as e:
        d.slice(4000, 1001)
    assert e.value.status == 4 and "Slice out of bounds" in e.value.message


# ---- BASELINE config 3 shape: AND of two compares + null bitmaps over Float64 + Int64 ------------------
@pytest.mark.parametrize("n", [1000, 65_537, 1_000_000])
@pytest.mark.parametrize("nulls", ["drops", "least"])
def test_config3_and_of_compares_with_nulls(gpu_ctx, oracle, n, nulls):
    f = oracle.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
    x = oracle.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)], nulls)
    got, rows, _ = gpu_filter_project(gpu_ctx, [f, x], pred, [0, 1])
    assert_columns_equal(got, oracle.filter_project([f, x], pred, [0, 1]), f"n={n} {nulls}")
    if n <= 65_537:  # the reference-shaped 1024-row pull loop gives the same batch
        assert_columns_equal(got, oracle.stream_filter_project([f, x], 1024, pred, [0, 1]), "streamed")


def test_float_specials_and_int_extremes(gpu_ctx, oracle):
    fs = np.tile(np.array([0.0, -0.0, 1.5, -1.5, math.inf, -math.inf, math.nan, 5e-324, 1e308]), 300)
    xs = np.tile(np.array([I64_MIN, -1, 0, 1, I64_MAX, I64_MIN + 1, I64_MAX - 1, 7, 9], np.int64), 300)
    f, x = Column.from_numpy(fs), Column.from_numpy(xs)
    for op in ["==", "!=", "<", ">", "<=", ">="]:
        for lit in [0.0, -0.0, math.nan, math.inf, -math.inf, 1.5]:
            pred = Predicate([Term(0, op, lit)])
            assert_columns_equal(gpu_filter_project(gpu_ctx, [f, x], pred, [0, 1])[0],
                                 oracle.filter_project([f, x], pred, [0, 1]), f"f64 {op} {lit}")
        for lit in [I64_MIN, I64_MAX, 0, -1]:
            pred = Predicate([Term(1, op, lit)])
            assert_columns_equal(gpu_filter_project(gpu_ctx, [f, x], pred, [1])[0],
                                 oracle.filter_project([f, x], pred, [1]), f"i64 {op} {lit}")


# ---- OR / NOT in the predicate (rv_predicate::expr; expr.rs:27-28, boolean.rs:120-165) ------------------------------
def _xor(a, b):
    return ("and", ("or", a, b), ("not", ("and", a, b)))


EXPR_TREES = {
    "or2": ("or", 0, 1),
    "or_and": ("or", ("and", 0, 1), 3),
    "and_or_not": ("and", ("or", 0, 1), ("not", 3)),
    "dnf3": ("or", ("and", 0, 1), ("and", 3, 4), ("and", 1, 3)),       # small as the negation of a CNF
    "not_top": ("not", ("or", 0, ("and", 1, 3))),
    "bool_in_or": ("or", 2, ("and", 0, ("not", 2))),
    "three_bool_columns": ("or", ("and", 2, 5), ("not", 6), 0),          # more Boolean columns than one pass reads: composed
    "string_in_or": ("or", 7, ("and", 0, 1)),
    "not_string": ("and", ("not", 7), 3),                                 # a null name drops the row even under NOT (strict)
    "two_strings": ("or", 7, 8, 4),
    "xor4": _xor(_xor(0, 1), _xor(3, 4)),                                 # 32 literals either way: composed from BooleanArrays
    "tautology": ("or", 0, ("not", 0)),                                   # true where x is valid (drops) / everywhere (least)
    "contradiction": ("and", 1, ("not", 1)),
    "pure_and": ("and", 0, ("and", 1, 3)),                                # falls back to the plain term list
    "and_after_tautology": ("and", 3, ("or", 2, ("not", 2), 1)),          # simplifies to term 3, but b's and f's nulls still drop rows
    "and_after_tautology_string": ("and", 0, ("or", 7, ("not", 7))),      # ... and the name's
    "single": 4,
}


@pytest.mark.parametrize("nulls", ["drops", "least"])
@pytest.mark.parametrize("tree", list(EXPR_TREES), ids=list(EXPR_TREES))
def test_predicate_expressions(gpu_ctx, oracle, tree, nulls):
    n = 70_003
    rng = np.random.default_rng(len(tree))
    words = ["Bob", "Bo", "", "Alice", "bob", "Ünï"]
    cols = [Column.from_numpy(rng.integers(0, 100, n + 5).astype(np.int64), rng.random(n + 5) > 0.1).slice(5, n),   # 0 x
            Column.from_numpy(rng.random(n), rng.random(n) > 0.1),                                                    # 1 f
            Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.1),                                              # 2 b
            Column.from_numpy(rng.integers(0, 100, n).astype(np.int64)),                                              # 3 y (no nulls)
            Column.from_numpy(rng.random(n) > 0.3, rng.random(n) > 0.2),                                              # 4 b2
            Column.from_numpy(rng.random(n) > 0.7),                                                                   # 5 b3
            Column.from_strings([None if rng.random() < 0.1 else words[k] for k in rng.integers(0, len(words), n)])]  # 6 name
    terms = [Term(0, "<", 30), Term(1, ">", 0.6), Term(2, "is_true"), Term(3, ">=", 80), Term(3, "<", 10),
             Term(4, "is_true"), Term(5, "==", False), Term(6, "==", "Bob"), Term(6, ">", "a")]
    pred = Predicate(terms, nulls, EXPR_TREES[tree])
    d = [gpu_ctx.upload(c) for c in cols]
    proj = [0, 6, 2, 3]
    outs, rows, sel = gpu_ctx.filter_project(d, pred, proj, want_selection=True)
    want = oracle.filter_project(cols, pred, proj)
    osel, ocnt = oracle.eval_predicate(cols, pred)
    assert rows == ocnt == want[0].length
    assert_columns_equal([o.download() for o in outs], want, f"{tree} {nulls}")
    assert sel.download().same_as(osel) is None
    assert gpu_ctx.eval_predicate(d, pred)[1] == ocnt
    assert gpu_ctx.filter_agg(d, pred, 3)[::2] == oracle.filter_agg(cols, pred, 3)[::2]
    # the reference-shaped 1024-row pull loop over the same batches gives the same table
    if tree in ("or_and", "not_string"):
        assert_columns_equal([o.download() for o in outs], oracle.stream_filter_project(cols, 1024, pred, proj), "streamed")
        outs2, rows2 = gpu_ctx.filter_project_host(cols, pred, proj, 4096)
        assert rows2 == rows
        assert_columns_equal([o.download() for o in outs2], want, "host chunks")
    if tree == "pure_and":  # the same rows as the plain AND of those terms
        plain, prow, _ = gpu_ctx.filter_project(d, Predicate([terms[0], terms[1], terms[3]], nulls), proj)
        assert prow == rows
        assert_columns_equal([o.download() for o in plain], want, "plain AND")


def test_predicate_expression_sizes_and_errors(gpu_ctx, oracle):
    """Small / ragged batches through the expression kernels (every NCOLS instantiation), malformed programs."""
    for n in [0, 1, 63, 64, 65, 4097]:
        rng = np.random.default_rng(n)
        cols = [Column.from_numpy(rng.integers(0, 10, n).astype(np.int64), rng.random(n) > 0.2) for _ in range(5)]
        d = [gpu_ctx.upload(c) for c in cols]
        for k in range(1, 6):  # k value columns in the predicate (5: composed), all projected
            terms = [Term(c, "<", 5) for c in range(k)]
            tree = ("or", *range(k)) if k > 1 else ("not", 0)
            for nulls in ("drops", "least"):
                pred = Predicate(terms, nulls, tree)
                outs, rows, _ = gpu_ctx.filter_project(d, pred, list(range(5)))
                assert_columns_equal([o.download() for o in outs], oracle.filter_project(cols, pred, list(range(5))), f"n={n} k={k} {nulls}")
    x = gpu_ctx.upload(Column.from_numpy(np.arange(10, dtype=np.int64)))
    for tree, text in [(("and", 0, 7), "pushes term 7"), (("not", ("not", 0)), None)]:
        pred = Predicate([Term(0, ">", 3)], "drops", tree)
        if text is None:
            assert gpu_ctx.filter_project([x], pred, [0])[1] == 6
        else:
            with pytest.raises(capi.RvError) as e:
                gpu_ctx.filter_project([x], pred, [0])
            assert e.value.status == 1 and text in e.value.message

    class Raw(Predicate):  # hand-written postfix programs
        def __init__(self, code):
            super().__init__([Term(0, ">", 3)])
            self.code = code

        def as_struct(self):
            p, keep = super().as_struct()
            import ctypes
            prog = (ctypes.c_uint8 * len(self.code))(*self.code)
            keep.append(prog)
            p.expr, p.n_expr = prog, len(self.code)
            return p, keep
    for code, text in [([capi.RV_EXPR_AND], "fewer than two operands"), ([0, 0], "exactly one value"), ([capi.RV_EXPR_NOT], "no operand"), ([0, 0x90], "unknown entry")]:
        with pytest.raises(capi.RvError) as e:
            gpu_ctx.filter_project([x], Raw(code), [0])
        assert text in e.value.message


# ---- boundary hardening: malformed host arrays, rv_wrap ---------------------------------------------------------------
def test_string_offsets_are_validated_before_upload(gpu_ctx):
    """string.rs:126-147: negative, decreasing or out-of-range offsets never reach the device."""
    good = Column.from_strings(["ab", "c", "", "def"])
    for bad_offsets in ([0, 2, 1, 3, 6], [-1, 2, 3, 3, 6], [0, 2, 3, 3, 7], [0, 2, 9, 3, 6]):
        bad = Column(good.dtype, good.values, None, 0, 4, np.asarray(bad_offsets, dtype=np.int32))
        with pytest.raises(capi.RvError) as e:
            gpu_ctx.upload(bad)
        assert e.value.status == 1 and e.value.message == "Offset out of bounds"
        with pytest.raises(capi.RvError) as e:
            gpu_ctx.filter_project_host([bad], Predicate([Term(0, "==", "c")]), [0], 64)
        assert e.value.message == "Offset out of bounds"
    assert gpu_ctx.upload(good).download().same_as(good) is None


def test_wrap_adopts_caller_owned_device_memory(gpu_ctx, oracle):
    """rv_wrap: a column over device memory the library does not own (here: buffers of another column, at an
    element offset) behaves like any other column and is not freed with the handle."""
    n = 10_000
    rng = np.random.default_rng(4)
    host = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64), rng.random(n) > 0.1)
    owner = gpu_ctx.upload(host)
    desc = owner.device_ptrs()
    desc.offset, desc.length = 64, n - 100
    view = gpu_ctx.wrap(desc)
    pred = Predicate([Term(0, ">", 500)], "least")
    outs, rows, _ = gpu_ctx.filter_project([view], pred, [0])
    assert_columns_equal([outs[0].download()], oracle.filter_project([host.slice(64, n - 100)], pred, [0]), "wrapped")
    view.free()
    assert owner.download().same_as(host) is None  # the buffers are still the owner's
    bad = owner.device_ptrs()
    bad.values += 4
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.wrap(bad)
    assert "8-byte aligned" in e.value.message


# ---- 3, 4 and more columns; duplicates; predicate-only columns -------------------------------------------
@pytest.mark.parametrize("ncols", [3, 4, 5, 9])
def test_many_columns(gpu_ctx, oracle, ncols):
    n = 33_333
    rng = np.random.default_rng(ncols)
    cols = []
    for c in range(ncols):
        if c % 3 == 0:
            cols.append(Column.from_numpy(rng.integers(0, 100, n).astype(np.int64), rng.random(n) > 0.1))
        elif c % 3 == 1:
            cols.append(Column.from_numpy(rng.random(n)))
        else:
            cols.append(Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.1))
    pred = Predicate([Term(0, ">=", 40), Term(1, "<", 0.7)])
    proj = list(range(ncols))[::-1] + [0]
    got, rows, _ = gpu_filter_project(gpu_ctx, cols, pred, proj)
    assert_columns_equal(got, oracle.filter_project(cols, pred, proj), f"ncols={ncols}")


@pytest.mark.parametrize("mode", [1, -1])
def test_wide_frames_later_groups_at_the_first_passs_wave_offsets(gpu_ctx, oracle, mode):
    """More than four 8-byte columns are compacted in groups of four: the groups after the first at the FIRST pass's wave offsets by
    compact_ranges_kernel (option groups_by_ranges: 1 = always, forced here; the default stops at two rows in three surviving), or as
    passes of their own (-1).  Plain and nullable columns mixed (a nullable column's validity bits are compacted by bits_compact_kernel
    at the same offsets, its null slots zeroed), sliced frames, from no survivor to all, each query twice (the second call's first pass is sized from the first: other wave
    ranges -- 1024, 512, 256 rows -- and the direct kernel)."""
    rng = np.random.default_rng(5)
    n = 400_009
    cols = [Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64))]
    for c in range(1, 11):
        vals = rng.integers(-50, 50, n).astype(np.int64) if c % 2 else rng.random(n)
        cols.append(Column.from_numpy(vals, rng.random(n) > 0.1) if c in (2, 6, 7) else Column.from_numpy(vals))
    cols = [c.slice(21, n - 60) for c in cols]
    d = [gpu_ctx.upload(c) for c in cols]
    gpu_ctx.set_option("groups_by_ranges", mode)
    try:
        seen = set()
        for lit in (-1, 99, 499, 899, 998, 1000):
            pred = Predicate([Term(0, ">", lit)])
            for proj in (list(range(9)), [3, 0, 10, 9, 8, 7, 6, 5, 4, 2, 1], [1, 2, 3, 4, 5]):
                want = oracle.filter_project(cols, pred, proj)
                for call in range(2):
                    outs, rows, _ = gpu_ctx.filter_project(d, pred, proj)
                    seen.add(gpu_ctx.last_kernel().split("<")[0])
                    assert rows == want[0].length
                    assert_columns_equal([o.download() for o in outs], want, f"mode={mode} x > {lit} proj={proj} call {call}")
                    [o.free() for o in outs]
        assert ("compact_ranges_kernel" in seen) == (mode == 1), seen
    finally:
        gpu_ctx.set_option("groups_by_ranges", 0)


@pytest.mark.parametrize("n", [0, 1, 63, 1024, 1025, 70_001, 300_000])
def test_filter_by_boolean_array_without_a_pass(gpu_ctx, oracle, n):
    """RecordBatch::filter by a BooleanArray through mask_select_kernel + scan + compact_ranges_kernel (forced: the default takes it from
    2^24 rows): nullable mask, sliced frame, plain and nullable columns, masks from all-false to all-true."""
    rng = np.random.default_rng(n)
    cut = max(0, min(5, n - 1))
    cols = [Column.from_numpy(rng.integers(-9, 9, n).astype(np.int64), rng.random(n) > 0.2), Column.from_numpy(rng.random(n)),
            Column.from_numpy(rng.integers(0, 5, n).astype(np.int64))]
    gpu_ctx.set_option("groups_by_ranges", 1)
    try:
        for share in (0.0, 0.1, 0.6, 1.0):
            mask = Column.from_numpy(rng.random(n) < share, rng.random(n) > 0.1 if share not in (0.0, 1.0) else None)
            host = [c.slice(cut, n - cut) for c in cols]
            m = mask.slice(cut, n - cut)
            got, rows = gpu_ctx.filter([gpu_ctx.upload(c) for c in host], gpu_ctx.upload(m))
            assert_columns_equal([o.download() for o in got], oracle.filter(host, m), f"n={n} share={share}")
            d = [gpu_ctx.upload(c) for c in host] + [gpu_ctx.upload(m)]
            outs, rows2, _ = gpu_ctx.filter_project(d, Predicate([Term(3, "is_true")]), [2, 0, 1, 0])
            assert rows2 == rows
            assert_columns_equal([o.download() for o in outs], [oracle.filter(host, m)[j] for j in (2, 0, 1, 0)], f"n={n} share={share} projected")
    finally:
        gpu_ctx.set_option("groups_by_ranges", 0)


def test_predicate_column_not_projected(gpu_ctx, oracle):
    n = 10_000
    rng = np.random.default_rng(1)
    a = Column.from_numpy(rng.integers(0, 10, n).astype(np.int64))
    b = Column.from_numpy(rng.random(n), rng.random(n) > 0.5)
    pred = Predicate([Term(0, "==", 3)])
    assert_columns_equal(gpu_filter_project(gpu_ctx, [a, b], pred, [1])[0], oracle.filter_project([a, b], pred, [1]))


# ---- RecordBatch::filter with a BooleanArray predicate (the reference streaming form) ------------------------
@pytest.mark.parametrize("n", [0, 2, 3, 64, 1000, 70_000])
def test_filter_by_boolean_array(gpu_ctx, oracle, n):
    rng = np.random.default_rng(n + 1)
    ids = Column.from_numpy(np.arange(n, dtype=np.int64))
    score = Column.from_numpy(rng.random(n), rng.random(n) > 0.2)
    active = Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.1)
    predicate = Column.from_numpy(rng.random(n) > 0.7, rng.random(n) > 0.1)  # nulls are dropped (:237)
    d = [gpu_ctx.upload(c) for c in (ids, score, active)]
    outs, rows = gpu_ctx.filter(d, gpu_ctx.upload(predicate))
    got = [o.download() for o in outs]
    exp = oracle.filter([ids, score, active], predicate)
    assert rows == exp[0].length
    assert_columns_equal(got, exp, f"n={n}")


def test_filter_error_texts_are_the_references(gpu_ctx):
    ids = gpu_ctx.upload(Column.from_numpy(np.arange(3, dtype=np.int64)))
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.filter([ids], gpu_ctx.upload(Column.from_numpy(np.array([True, False, True, True]))))
    assert e.value.status == 2 and e.value.message == "Predicate length 4 doesn't match batch length 3"
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.filter([ids], gpu_ctx.upload(Column.from_numpy(np.array([1, 0, 1], np.int64))))
    assert e.value.status == 3 and e.value.message == "Predicate must be a BooleanArray"


# ---- K1 / compare / BooleanArray logic -----------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 63, 64, 65, 10_000])
@pytest.mark.parametrize("op", ["==", "!=", "<", ">", "<=", ">="])
def test_compare_gives_nullable_boolean_array(gpu_ctx, oracle, n, op):
    rng = np.random.default_rng(n)
    for col, lit in [(Column.from_numpy(rng.integers(0, 10, n).astype(np.int64), rng.random(n) > 0.3), 5),
                     (Column.from_numpy(rng.random(n)), 0.5),
                     (Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.3), True),
                     (Column.from_numpy(rng.integers(0, 10, n).astype(np.int64), rng.random(n) > 0.3), None),
                     (Column.from_numpy(rng.integers(0, 10, n).astype(np.int64)), 5.0)]:
        got = gpu_ctx.compare(gpu_ctx.upload(col), op, lit).download()
        assert got.same_as(oracle.compare(col, op, lit)) is None, f"{op} {lit}"


@pytest.mark.parametrize("n", [5, 64, 65, 1000, 100_001])
def test_boolean_and_or_not_count(gpu_ctx, oracle, n):
    rng = np.random.default_rng(n)
    a = Column.from_numpy(rng.random(n + 9) > 0.5, rng.random(n + 9) > 0.2).slice(9, n)
    b = Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.2)
    c = Column.from_numpy(rng.random(n) > 0.5)
    da, db, dc = (gpu_ctx.upload(x) for x in (a, b, c))
    assert gpu_ctx.boolean_and(da, db).download().same_as(oracle.boolean_op("and", a, b)) is None
    assert gpu_ctx.boolean_or(da, db).download().same_as(oracle.boolean_op("or", a, b)) is None
    assert gpu_ctx.boolean_not(da).download().same_as(oracle.boolean_op("not", a)) is None
    assert gpu_ctx.boolean_and(dc, dc).download().same_as(oracle.boolean_op("and", c, c)) is None  # no validity out
    assert gpu_ctx.boolean_count(da) == oracle.boolean_count(a)
    assert da.null_count() == oracle.null_count(a)


def test_boolean_reference_vectors(gpu_ctx):
    """boolean.rs:625-690 on the device."""
    def arr(vals):
        return gpu_ctx.upload(Column.from_numpy(np.array([bool(v) for v in vals]), np.array([v is not None for v in vals])))

    def opt(col):
        c = col.download()
        v, m = c.logical_values(), c.logical_valid()
        return [None if (m is not None and not m[i]) else bool(v[i]) for i in range(c.length)]
    a, b = arr([True, False, True, None, False]), arr([True, True, False, True, None])
    assert opt(gpu_ctx.boolean_and(a, b)) == [True, False, False, None, None]
    assert opt(gpu_ctx.boolean_or(a, arr([False, True, False, True, None]))) == [True, True, True, None, None]
    assert opt(gpu_ctx.boolean_not(arr([True, False, None, True]))) == [False, True, None, False]
    assert gpu_ctx.boolean_count(arr([True, False, True, None, False, True])) == (3, 2)
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.boolean_and(arr([True, False]), arr([True]))
    assert e.value.message == "Array lengths must match for logical operations"


# ---- take / concat --------------------------------------------------------------------------------------------------------
def test_take_arbitrary_indices(gpu_ctx, oracle):
    n = 5000
    rng = np.random.default_rng(5)
    cols = [Column.from_numpy(rng.integers(0, 100, n).astype(np.int64), rng.random(n) > 0.3),
            Column.from_numpy(rng.random(n)),
            Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.3)]
    d = [gpu_ctx.upload(c) for c in cols]
    for idx in [[2, 0, 1], [], list(rng.integers(0, n, 777)), [n - 1] * 70]:
        got = [o.download() for o in gpu_ctx.take(d, idx)]
        assert_columns_equal(got, oracle.take(cols, idx), f"{len(idx)} indices")
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.take(d, [0, n + 5, 1])
    assert e.value.status == 4 and e.value.message == f"Index {n + 5} out of bounds for {n} rows"


def test_selection_to_indices_to_take_stays_on_the_device(gpu_ctx, oracle):
    """RecordBatch::filter is `index list of Some(true) rows` + take (record_batch.rs:221-243); with rv_selection_indices and
    rv_take_device the chain runs without the indices ever visiting the host."""
    n = 100_003
    rng = np.random.default_rng(12)
    cols = [Column.from_numpy(rng.integers(0, 100, n + 3).astype(np.int64), rng.random(n + 3) > 0.2).slice(3, n),
            Column.from_numpy(rng.random(n) > 0.5, rng.random(n) > 0.2),
            Column.from_strings([None if rng.random() < 0.1 else "s%d" % (k % 7) for k in range(n)])]
    d = [gpu_ctx.upload(c) for c in cols]
    for pcol in (Column.from_numpy(rng.random(n + 70) > 0.7, rng.random(n + 70) > 0.1).slice(67, n),   # nullable, bit offset 3
                 Column.from_numpy(rng.random(n) > 0.9),                                                   # no bitmap
                 Column.from_numpy(np.zeros(n, bool))):                                                    # nothing selected
        idx = gpu_ctx.selection_indices(gpu_ctx.upload(pcol))
        sel = pcol.logical_values() & (pcol.logical_valid() if pcol.validity is not None else True)
        want_idx = np.nonzero(sel)[0].astype(np.int64)
        assert np.array_equal(idx.download().logical_values(), want_idx)
        got = [c.download() for c in gpu_ctx.take_device(d, idx)]
        assert_columns_equal(got, oracle.filter(cols, pcol), "selection -> indices -> take == filter")
        assert_columns_equal(got, oracle.take(cols, want_idx.astype(np.uint64)), "== take")
    # arbitrary device-resident indices, repeated and unordered; the first out-of-range index is the one reported
    idx = Column.from_numpy(rng.integers(0, n, 5000).astype(np.int64))
    got = [c.download() for c in gpu_ctx.take_device(d, gpu_ctx.upload(idx))]
    assert_columns_equal(got, oracle.take(cols, idx.logical_values().astype(np.uint64)), "device indices")
    bad = idx.logical_values().copy()
    bad[4000], bad[123] = n + 9, n + 5
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.take_device(d, gpu_ctx.upload(Column.from_numpy(bad)))
    assert e.value.status == 4 and e.value.message == f"Index {n + 5} out of bounds for {n} rows"
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.take_device(d, gpu_ctx.upload(Column.from_numpy(bad, np.ones(5000, bool) & (np.arange(5000) != 7))))
    assert "nulls" in e.value.message
    with pytest.raises(capi.RvError) as e:
        gpu_ctx.take(d, [0, n + 5, n + 9])  # the host-list form reports the same way
    assert e.value.message == f"Index {n + 5} out of bounds for {n} rows"


@pytest.mark.parametrize("nparts", [1, 2, 17, 300])
def test_concat(gpu_ctx, oracle, nparts):
    rng = np.random.default_rng(nparts)
    for kind in ["i", "f", "b"]:
        parts = []
        for p in range(nparts):
            n = int(rng.integers(0, 200))
            vals = {"i": rng.integers(0, 9, n).astype(np.int64), "f": rng.random(n), "b": rng.random(n) > 0.5}[kind]
            valid = (rng.random(n) > 0.3) if p % 3 == 0 else None
            parts.append(Column.from_numpy(vals, valid))
        got = gpu_ctx.concat([gpu_ctx.upload(p) for p in parts]).download()
        assert got.same_as(oracle.concat(parts)) is None, kind


# ---- K4: filter + SUM / COUNT (no reference operator: parity unpinned) --------------------------------
@pytest.mark.parametrize("n", [1, 4096, 4097, 1_000_003])
def test_filter_agg_int64_bit_exact(gpu_ctx, oracle, n):
    rng = np.random.default_rng(n)
    x = Column.from_numpy(rng.integers(I64_MIN, I64_MAX, n, dtype=np.int64), rng.random(n) > 0.1)  # wraps
    k = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64))
    pred = Predicate([Term(1, ">", 899)])
    si, _, cnt = gpu_ctx.filter_agg([gpu_ctx.upload(x), gpu_ctx.upload(k)], pred, 0)
    osi, _, ocnt = oracle.filter_agg([x, k], pred, 0)
    assert (si, cnt) == (osi, ocnt)


def test_filter_agg_float64_within_tolerance(gpu_ctx, oracle):
    """Float64 SUM depends on the reduction order; tolerance 1e-12 relative (documented in DESIGN.md)."""
    n = 500_000
    f = oracle.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
    pred = Predicate([Term(0, ">", 0.5)])
    _, sf, cnt = gpu_ctx.filter_agg([gpu_ctx.upload(f)], pred, 0)
    _, osf, ocnt = oracle.filter_agg([f], pred, 0)
    assert cnt == ocnt and abs(sf - osf) <= 1e-12 * abs(osf)
    assert gpu_ctx.filter_agg([gpu_ctx.upload(f)], pred, 0)[1] == sf  # reproducible run to run


def test_float64_sum_tree_depends_on_the_row_count_only(gpu_ctx, oracle):
    """The default launch of the aggregate is 8192 workgroups whatever the device's CU count, so a Float64 SUM's order of
    additions -- hence its bits -- follows from the row count alone.  On an MI355X (256 CUs) that is the geometry
    `agg_grid` = 32 per CU spells out; a different grid gives a differently rounded (still 1e-12-close) sum."""
    n = 60_000_037  # more tiles than workgroups: the stride matters
    spec = synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44)
    f = gpu_ctx.generate(spec)
    pred = Predicate([Term(0, ">", 0.25)])
    _, base, cnt = gpu_ctx.filter_agg([f], pred, 0)
    try:
        gpu_ctx.set_option("agg_grid", 32)
        cus = gpu_ctx.device_info()["compute_units"]
        _, same, _ = gpu_ctx.filter_agg([f], pred, 0)
        gpu_ctx.set_option("agg_grid", 7)
        _, other, cnt7 = gpu_ctx.filter_agg([f], pred, 0)
    finally:
        gpu_ctx.set_option("agg_grid", 0)
    if cus == 256:
        assert same == base
    assert cnt7 == cnt and abs(other - base) <= 1e-12 * abs(base)
    hf = oracle.generate(spec)
    want = float(hf.values[hf.logical_valid() & (hf.values > 0.25)].sum())
    assert abs(base - want) <= 1e-11 * abs(want)
    f.free()


def test_rccl_allreduce_single_rank(gpu_ctx):
    """rv_comm_*: ncclAllReduce(count=2, ncclInt64, ncclSum) through RCCL; with one rank the payload comes back
    unchanged (the 8-GPU run is the driver's; the protocol is covered at world_size 2 over gloo on CPU)."""
    comm = capi.Comm(gpu_ctx, capi.comm_unique_id(), 1, 0)
    try:
        assert comm.allreduce_sum_count(-(2 ** 62) - 12345, 2 ** 40 + 7) == (-(2 ** 62) - 12345, 2 ** 40 + 7)
    finally:
        comm.close()


# ---- BASELINE.json full size: size-independent properties at 1e9 rows --------------------------------------------------
def test_full_size_properties_1e9(gpu_ctx, oracle):
    n = 1_000_000_000
    x = gpu_ctx.generate(synth_spec(RV_INT64, seed=42, length=n))
    pred = Predicate([Term(0, ">", 899)])
    outs, rows, _ = gpu_ctx.filter_project([x], pred, [0])
    out = outs[0]
    si, _, cnt = gpu_ctx.filter_agg([x], pred, 0)
    assert rows == cnt and abs(rows / n - 0.1) < 1e-3
    # every survivor satisfies the predicate, none is lost: SUM/COUNT of the output == masked SUM/COUNT of the input
    so, _, co = gpu_ctx.filter_agg([out], Predicate([Term(0, ">=", I64_MIN)]), 0)
    assert (so, co) == (si, cnt)
    assert gpu_ctx.filter_agg([out], Predicate([Term(0, "<=", 899)]), 0)[2] == 0
    # idempotence: filtering the output again keeps everything
    again, rows2, _ = gpu_ctx.filter_project([out], pred, [0])
    assert rows2 == rows
    # order: windows of the output equal the oracle on the matching input windows
    w = 2_000_000
    for start in [0, 123_456_789 // 64 * 64, n - w]:
        part = x.slice(start, w)
        before = gpu_ctx.filter_agg([x.slice(0, start)], pred, 0)[2] if start else 0
        host = part.download()
        exp = oracle.filter_project([host], pred, [0])[0]
        got = out.slice(before, exp.length).download()
        assert got.same_as(exp) is None, f"window at {start}"
        assert again[0].slice(before, exp.length).download().same_as(exp) is None


def test_full_size_properties_config3_1e9(gpu_ctx, oracle):
    """BASELINE configs[2] at its full size: (f > 0.5) AND (x < 200) -> [f, x] over two nullable columns, 1e9 rows."""
    n = 1_000_000_000
    f = gpu_ctx.generate(synth_spec(RV_FLOAT64, seed=43, length=n, validity_seed=44))
    x = gpu_ctx.generate(synth_spec(RV_INT64, seed=42, length=n, validity_seed=45))
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
    outs, rows, _ = gpu_ctx.filter_project([f, x], pred, [0, 1])
    si, _, cnt = gpu_ctx.filter_agg([f, x], pred, 1)
    assert rows == cnt and outs[0].length == rows and outs[1].length == rows
    # survivors of a null-dropping conjunction over both columns carry no nulls: the builder drops both bitmaps
    assert outs[0].null_count() == 0 and outs[1].null_count() == 0
    # none lost, none invented: SUM / COUNT of the x output == masked SUM / COUNT of the input; every survivor passes again
    so, _, co = gpu_ctx.filter_agg([outs[1]], Predicate([Term(0, ">=", I64_MIN)]), 0)
    assert (so, co) == (si, cnt)
    again, rows2, _ = gpu_ctx.filter_project(outs, pred, [0, 1])
    assert rows2 == rows
    # order: windows of the output equal the oracle on the matching input windows
    w = 2_000_000
    for start in [0, 387_654_321 // 64 * 64, n - w]:
        before = gpu_ctx.filter_agg([f.slice(0, start), x.slice(0, start)], pred, 1)[2] if start else 0
        host = [f.slice(start, w).download(), x.slice(start, w).download()]
        exp = oracle.filter_project(host, pred, [0, 1])
        for j in range(2):
            got = outs[j].slice(before, exp[j].length).download()
            assert got.same_as(exp[j]) is None, f"column {j}, window at {start}"
            assert again[j].slice(before, exp[j].length).download().same_as(exp[j]) is None
