"""CPU suite: the oracle against the committed golden vectors, its streaming restatement
against its one-shot form, and the synthetic generator's published constants."""
import numpy as np
import pytest

from helpers import assert_columns_equal, load_golden
from rivulus_amd.capi import RV_FLOAT64, RV_INT64, Column, Predicate, Term, synth_spec

CASES = load_golden()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_golden(oracle, case):
    got = oracle.filter_project(case["columns"], case["predicate"], case["projection"])
    assert all(g.length == case["rows"] for g in got)
    assert_columns_equal(got, case["expected"], case["name"])


@pytest.mark.parametrize("batch_rows", [1, 3, 64, 1024])
def test_streaming_restatement_equals_one_shot(oracle, batch_rows):
    rng = np.random.default_rng(7)
    n = 5000
    f = Column.from_numpy(rng.random(n), rng.random(n) > 0.05)
    x = Column.from_numpy(rng.integers(0, 1000, n).astype(np.int64), rng.random(n) > 0.05)
    pred = Predicate([Term(0, ">", 0.5), Term(1, "<", 200)])
    one = oracle.filter_project([f, x], pred, [0, 1])
    streamed = oracle.stream_filter_project([f, x], batch_rows, pred, [0, 1])
    assert_columns_equal(streamed, one, f"batch_rows={batch_rows}")


def test_generator_known_values(oracle):
    # splitmix64 reference values (public test vector of the algorithm, seed 0 stream)
    def sm64(z):
        z = (z + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)
    assert sm64(0) == 0xE220A8397B1DCDAF
    col = oracle.generate(synth_spec(RV_INT64, seed=42, length=100, first_row=5))
    assert [int(v) for v in col.values] == [sm64(42 + 5 + i) % 1000 for i in range(100)]
    f = oracle.generate(synth_spec(RV_FLOAT64, seed=43, length=10, validity_seed=44))
    assert [float(v) for v in f.values] == [(sm64(43 + i) >> 11) * 2.0**-53 for i in range(10)]
    assert list(f.logical_valid()) == [sm64(44 + i) % 100 >= 5 for i in range(10)]


def test_selectivity_of_config2_predicate(oracle):
    x = oracle.generate(synth_spec(RV_INT64, seed=42, length=200_000))
    sel, cnt = oracle.eval_predicate([x], Predicate([Term(0, ">", 899)]))
    assert cnt == int((x.values > 899).sum())
    assert abs(cnt / 200_000 - 0.10) < 0.005
