"""CPU: the oracle's StringArray surface (C API export / import) against plain Python list semantics of the
reference: take -> StringArray::new (record_batch.rs:163-170, string.rs:19-57), filter keeps Some(true) rows
(record_batch.rs:235-240), concat re-appends every element (:277-342), compare terms follow AnyValue ordering
(series.rs:87-117)."""
import numpy as np
import pytest

from rivulus_amd.capi import Column, Predicate, Term


def _strings(rng, n, null_share=0.15):
    pool = ["Alice", "Bob", "Charlie", "", "Ünï", "名前", "Bo", "Bobby", "bob"]
    return [None if rng.random() < null_share else pool[k] for k in rng.integers(0, len(pool), n)]


def test_reference_fixture_take_filter(oracle):
    # record_batch.rs:594-604 fixture: id [1,2,3], name [Alice, null, Charlie], active [T,F,T]
    ids = Column.from_numpy(np.array([1, 2, 3], dtype=np.int64))
    name = Column.from_strings(["Alice", None, "Charlie"])
    t = oracle.take([ids, name], [2, 0, 1])  # :751-770
    assert list(t[0].logical_values()) == [3, 1, 2] and t[1].to_strings() == ["Charlie", "Alice", None]
    f = oracle.filter([ids, name], Column.from_numpy(np.array([True, False, True])))  # :821-840
    assert list(f[0].logical_values()) == [1, 3] and f[1].to_strings() == ["Alice", "Charlie"]
    assert f[1].validity is None  # no null survived: bitmap dropped (string.rs:41-45)
    assert list(f[1].offsets) == [0, 5, 12] and bytes(f[1].values) == b"AliceCharlie"


@pytest.mark.parametrize("n", [0, 1, 9, 1000])
def test_take_filter_concat_against_python_lists(oracle, n):
    rng = np.random.default_rng(n)
    vals = _strings(rng, n + 6)
    col = Column.from_strings(vals).slice(3, n)  # offset != 0
    logical = vals[3:3 + n]
    assert col.to_strings() == logical
    idx = rng.integers(0, max(n, 1), 2 * n) if n else np.zeros(0, dtype=np.int64)
    got = oracle.take([col], idx)[0]
    assert got.to_strings() == [logical[i] for i in idx]
    assert (got.validity is not None) == any(logical[i] is None for i in idx)
    mask = rng.random(n) > 0.5
    mvalid = rng.random(n) > 0.2
    kept = oracle.filter([col], Column.from_numpy(mask, mvalid))[0]
    assert kept.to_strings() == [s for s, m, v in zip(logical, mask, mvalid) if m and v]
    parts = [col.slice(0, n // 2), col.slice(n // 2, n - n // 2), col]
    assert oracle.concat(parts).to_strings() == logical[:n // 2] + logical[n // 2:] + logical
    # a null spans no bytes in anything the oracle builds
    out = oracle.concat(parts)
    assert int(out.offsets[-1]) == sum(len(s.encode()) for s in out.to_strings() if s is not None)


@pytest.mark.parametrize("nulls", ["drops", "least"])
@pytest.mark.parametrize("op", ["==", "!=", "<", ">", "<=", ">="])
def test_string_compare_terms_against_python(oracle, op, nulls):
    rng = np.random.default_rng(5)
    vals = _strings(rng, 400)
    col, ids = Column.from_strings(vals), Column.from_numpy(np.arange(400, dtype=np.int64))
    lit = "Bob"
    py = {"==": lambda a: a == lit, "!=": lambda a: a != lit, "<": lambda a: a < lit, ">": lambda a: a > lit,
          "<=": lambda a: a <= lit, ">=": lambda a: a >= lit}[op]
    want = []
    for i, s in enumerate(vals):
        if s is None:
            keep = nulls == "least" and op in ("<", "<=", "!=")  # Null < any value (series.rs:105-107)
        else:
            keep = py(s.encode()) if False else {"==": s.encode() == b"Bob", "!=": s.encode() != b"Bob", "<": s.encode() < b"Bob",
                                                  ">": s.encode() > b"Bob", "<=": s.encode() <= b"Bob", ">=": s.encode() >= b"Bob"}[op]
        if keep:
            want.append(i)
    got = oracle.filter_project([col, ids], Predicate([Term(0, op, lit)], nulls), [1])[0]
    assert list(got.logical_values()) == want
