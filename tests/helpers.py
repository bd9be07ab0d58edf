"""Shared helpers: golden-case loading and column comparison."""
import json
import os

import numpy as np

from rivulus_amd.capi import Column, Predicate, Term

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases.json")
_NP = {"i": np.int64, "f": np.float64, "b": np.bool_}


def _decode_col(c):
    if c["kind"] == "f":
        vals = np.array([float.fromhex(x) for x in c["values"]], np.float64)
    else:
        vals = np.array(c["values"], _NP[c["kind"]])
    valid = None if c["valid"] is None else np.array(c["valid"], bool)
    return Column.from_numpy(vals, valid)


def load_golden():
    with open(GOLDEN) as f:
        doc = json.load(f)
    out = []
    for case in doc["cases"]:
        terms = []
        for t in case["predicate"]["terms"]:
            lit = t["literal"]
            if t["literal_is_float"]:
                lit = float.fromhex(lit)
            terms.append(Term(t["column"], t["op"], lit))
        out.append({
            "name": case["name"],
            "columns": [_decode_col(c) for c in case["columns"]],
            "predicate": Predicate(terms, case["predicate"]["nulls"], case["predicate"].get("expr")),
            "projection": case["projection"],
            "rows": case["rows"],
            "expected": [_decode_col(c) for c in case["expected"]],
        })
    return out


def assert_columns_equal(got, expected, what=""):
    assert len(got) == len(expected), f"{what}: {len(got)} columns != {len(expected)}"
    for j, (g, e) in enumerate(zip(got, expected)):
        diff = g.same_as(e)
        assert diff is None, f"{what} column {j}: {diff}"
