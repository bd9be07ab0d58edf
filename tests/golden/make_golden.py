"""Generates tests/golden/cases.json -- small input/expected-output vectors for the
filter/project path.

The reference is Rust and cannot run in this pipeline, so the expected outputs are
produced by the pinned CPU oracle (oracle/, checked against the reference's own inline
test vectors by oracle/kat_tests) and CROSS-CHECKED here by an independent numpy
restatement written from the reference text (series.rs:87-117, record_batch.rs:131-243):
a case is only written when both agree.  The file holds data only.

Run:  python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from rivulus_amd.capi import RV_BOOLEAN, RV_FLOAT64, RV_INT64, Column, Predicate, Term  # noqa: E402

I64_MIN, I64_MAX = -(2 ** 63), 2 ** 63 - 1


def np_cell_compare(op, vals, lit, kind):
    """valid-cell compare, AnyValue semantics; kind in {'i','f','b'}; lit python value or None."""
    n = len(vals)
    if lit is None:  # any value > Null
        return np.full(n, op in (">", ">=", "!="))
    lit_kind = "b" if isinstance(lit, bool) else ("i" if isinstance(lit, int) else "f")
    if lit_kind != kind:  # cross-type partial_cmp == None
        return np.full(n, op == "!=")
    with np.errstate(invalid="ignore"):
        return {"==": vals == lit, "!=": vals != lit, "<": vals < lit, ">": vals > lit, "<=": vals <= lit,
                ">=": vals >= lit}[op]


def np_filter_project(cols, pred, proj):
    """cols: list of (kind, values ndarray, valid ndarray|None).  Returns rows + per projected column
    (values, valid|None) following take_array: null slot -> 0, validity dropped when no null survives."""
    n = len(cols[0][1])

    def term_truth(t):
        """(truth, valid): valid is all True under "least" (the eager mask is a definite bool per row)."""
        kind, vals, valid = cols[t.column]
        v = np.ones(n, bool) if valid is None else valid
        if t.op == "is_true":
            r = np.where(v, vals.astype(bool), False)
        else:
            cell = np_cell_compare(t.op, vals, t.literal, kind)
            if pred.nulls == "drops":
                null_res = False
            elif t.literal is None:
                null_res = t.op in ("==", "<=", ">=")
            else:
                null_res = t.op in ("<", "<=", "!=")
            r = np.where(v, cell, null_res)
        return r, (v if pred.nulls == "drops" else np.ones(n, bool))

    def evaluate(tree):
        """Three-valued, strict: the result is null wherever an operand is null (boolean.rs:120-165)."""
        if isinstance(tree, int):
            return term_truth(pred.terms[tree])
        op, *args = tree
        if op == "not":
            r, v = evaluate(args[0])
            return ~r, v
        r, v = evaluate(args[0])
        for a in args[1:]:
            r2, v2 = evaluate(a)
            r = (r & r2) if op == "and" else (r | r2)
            v = v & v2
        return r, v

    if getattr(pred, "expr", None) is not None:
        r, v = evaluate(pred.expr)
        keep = r & v  # RecordBatch::filter keeps Some(true)
    else:
        keep = np.ones(n, bool)
        for t in pred.terms:
            keep &= term_truth(t)[0]
    out = []
    for c in proj:
        kind, vals, valid = cols[c]
        v = np.ones(n, bool) if valid is None else valid
        kv = v[keep]
        vals_k = np.where(kv, vals[keep], np.zeros(1, vals.dtype)[0])
        out.append((kind, vals_k, None if kv.all() else kv))
    return int(keep.sum()), out


def to_column(kind, vals, valid):
    return Column.from_numpy(vals, valid)


def encode_vals(kind, vals):
    if kind == "f":
        return [float(x).hex() for x in vals]
    if kind == "b":
        return [bool(x) for x in vals]
    return [int(x) for x in vals]


def main():
    rng = np.random.default_rng(20251003)
    cases = []

    def add(name, cols, pred, proj):
        o_cols = [to_column(*c) for c in cols]
        got = pyoracle.filter_project(o_cols, pred, proj)
        rows, exp = np_filter_project(cols, pred, proj)
        assert len(got) == len(exp)
        for g, (kind, ev, evalid) in zip(got, exp):
            assert g.length == rows, (name, g.length, rows)
            gv = g.logical_values()
            if kind == "f":
                assert np.array_equal(gv.view(np.uint64), ev.view(np.uint64)), name
            else:
                assert np.array_equal(gv, ev), name
            assert (g.validity is None) == (evalid is None), name
            if evalid is not None:
                assert np.array_equal(g.logical_valid(), evalid), name
        cases.append({
            "name": name,
            "columns": [{"kind": k, "values": encode_vals(k, v), "valid": None if m is None else [bool(x) for x in m]}
                        for (k, v, m) in cols],
            "predicate": {"nulls": pred.nulls, "expr": pred.expr,
                          "terms": [{"column": t.column, "op": t.op,
                                     "literal": (t.literal.hex() if isinstance(t.literal, float) else t.literal),
                                     "literal_is_float": isinstance(t.literal, float)} for t in pred.terms]},
            "projection": list(proj),
            "rows": rows,
            "expected": [{"kind": k, "values": encode_vals(k, v), "valid": None if m is None else [bool(x) for x in m]}
                         for (k, v, m) in exp],
        })

    # reference fixture: ages [25,30,35] > 25 (plan.rs:504-525)
    add("ref_age_gt_25", [("i", np.array([25, 30, 35], np.int64), None)], Predicate([Term(0, ">", 25)]), [0])
    # record_batch.rs:821-840: ids [1,2,3], predicate [T,F,T]
    add("ref_filter_bool_TFT", [("i", np.array([1, 2, 3], np.int64), None), ("b", np.array([True, False, True]), None)],
        Predicate([Term(1, "is_true")]), [0, 1])
    # record_batch.rs:868-879: predicate [T, null, F]
    add("ref_filter_bool_with_null",
        [("i", np.array([1, 2, 3], np.int64), None), ("b", np.array([True, False, False]), np.array([True, False, True]))],
        Predicate([Term(1, "is_true")]), [0])

    sizes = [0, 1, 7, 8, 9, 63, 64, 65, 127, 129, 1023, 1025]
    for n in sizes:
        x = rng.integers(0, 1000, n).astype(np.int64)
        for op in ["==", "!=", "<", ">", "<=", ">="]:
            add(f"i64_n{n}_{op}", [("i", x, None)], Predicate([Term(0, op, 500)]), [0])
    # int64 extremes
    ext = np.array([I64_MIN, -1, 0, 1, I64_MAX, I64_MIN + 1, I64_MAX - 1], np.int64)
    for op in ["<", ">", "==", "<=", ">=", "!="]:
        for lit in [I64_MIN, 0, I64_MAX]:
            add(f"i64_extreme_{op}_{lit}", [("i", ext, None)], Predicate([Term(0, op, lit)]), [0])
    # float specials
    fs = np.array([0.0, -0.0, 1.5, -1.5, math.inf, -math.inf, math.nan, 5e-324, 1e308], np.float64)
    for op in ["==", "!=", "<", ">", "<=", ">="]:
        for lit in [0.0, -0.0, math.nan, math.inf, 1.5]:
            add(f"f64_special_{op}_{lit!r}", [("f", fs, None)], Predicate([Term(0, op, lit)]), [0])
    # nulls, both policies, null literal, cross-type literal
    for n in [1, 9, 64, 65, 200]:
        x = rng.integers(0, 100, n).astype(np.int64)
        f = rng.random(n)
        vx = rng.random(n) > 0.3
        vf = rng.random(n) > 0.3
        for nulls in ["drops", "least"]:
            for op in ["==", "!=", "<", ">", "<=", ">="]:
                add(f"nulls_{nulls}_n{n}_{op}", [("i", x, vx), ("f", f, vf)], Predicate([Term(0, op, 50)], nulls), [0, 1])
                add(f"nulllit_{nulls}_n{n}_{op}", [("i", x, vx)], Predicate([Term(0, op, None)], nulls), [0])
                add(f"crosstype_{nulls}_n{n}_{op}", [("i", x, vx)], Predicate([Term(0, op, 50.0)], nulls), [0])
            add(f"and2_{nulls}_n{n}", [("f", f, vf), ("i", x, vx)],
                Predicate([Term(0, ">", 0.5), Term(1, "<", 60)], nulls), [0, 1])
    # all-null survivors, no-null survivors (validity dropped)
    x = np.arange(70, dtype=np.int64)
    v = np.ones(70, bool)
    v[10:20] = False
    add("validity_dropped_when_no_null_survives", [("i", x, v)], Predicate([Term(0, ">=", 20)]), [0])
    add("only_nulls_survive_least", [("i", x, v)], Predicate([Term(0, "<", -5)], "least"), [0])
    # boolean columns travelling through a filter
    b = rng.random(130) > 0.5
    vb = rng.random(130) > 0.2
    sel = rng.random(130) > 0.6
    add("bool_column_compacted", [("b", b, vb), ("b", sel, None), ("i", np.arange(130, dtype=np.int64), None)],
        Predicate([Term(1, "is_true")]), [0, 2, 1])
    for op in ["==", "!=", "<", ">", "<=", ">="]:
        for lit in [True, False]:
            for nulls in ["drops", "least"]:
                add(f"bool_cmp_{op}_{lit}_{nulls}", [("b", b, vb)], Predicate([Term(0, op, lit)], nulls), [0])

    # OR / NOT expressions (rv_predicate::expr): BinaryOperator::Or (expr.rs:28), BooleanArray::{and,or,not}
    # (boolean.rs:120-165) -- reference vectors first: [T,F,T,null,F] AND [T,T,F,T,null] etc. (boolean.rs:625-666)
    ba = (np.array([True, False, True, False, False]), np.array([True, True, True, False, True]))
    bb = (np.array([True, True, False, True, False]), np.array([True, True, True, True, False]))
    ids = np.arange(5, dtype=np.int64)
    for nulls in ["drops", "least"]:
        for name, tree in [("and", ("and", 0, 1)), ("or", ("or", 0, 1)), ("not", ("not", 0)), ("nor", ("not", ("or", 0, 1)))]:
            add(f"expr_ref_boolean_{name}_{nulls}", [("b",) + ba, ("b",) + bb, ("i", ids, None)],
                Predicate([Term(0, "is_true"), Term(1, "is_true")], nulls, tree), [2, 0, 1])
    trees = [("or", 0, 1), ("or", ("and", 0, 1), 2), ("and", ("or", 0, 1), ("not", 2)), ("not", ("and", 0, ("or", 1, 2))),
             ("or", ("and", 0, 1), ("and", 2, 3)), ("or", ("not", 0), ("and", 1, ("not", 3))), ("or", 0, ("not", 0)),
             ("and", 0, ("not", 0)), ("or", ("and", 0, 1), ("and", 2, 3), ("and", 1, 2))]
    for n in [1, 9, 64, 65, 300]:
        x = rng.integers(0, 100, n).astype(np.int64)
        f = rng.random(n)
        bcol = rng.random(n) > 0.5
        vx, vf, vbm = rng.random(n) > 0.3, rng.random(n) > 0.3, rng.random(n) > 0.3
        for nulls in ["drops", "least"]:
            for k, tree in enumerate(trees):
                add(f"expr_{nulls}_n{n}_{k}", [("i", x, vx), ("f", f, vf), ("b", bcol, vbm), ("i", x[::-1].copy(), None)],
                    Predicate([Term(0, "<", 50), Term(1, ">", 0.5), Term(2, "is_true"), Term(3, ">=", 70)], nulls, tree), [0, 1, 3])
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "cases": cases}, f, separators=(",", ":"))
    print(f"wrote {len(cases)} cases")


if __name__ == "__main__":
    main()
