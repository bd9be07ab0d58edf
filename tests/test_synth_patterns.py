"""The synthetic generator's patterns (include/rivulus_gpu.h, rv_synth_spec: independent rows, runs, ascending / descending) in the
CPU oracle against an independent numpy restatement of the header's formulas, and what each pattern promises.  No GPU."""
import numpy as np
import pytest

from rivulus_amd.capi import RV_BOOLEAN, RV_FLOAT64, RV_INT64, synth_spec, unpack_bits

M64 = (1 << 64) - 1


def splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def numpy_generate(dtype, seed, first, length, modulus, true_percent, pattern, run_rows, table_rows):
    g = np.arange(first, first + length, dtype=np.uint64)
    table = table_rows or first + length
    with np.errstate(over="ignore"):
        if pattern == "iid":
            h = splitmix64(g + np.uint64(seed))
        elif pattern == "clustered":
            h = splitmix64(g // np.uint64(run_rows) + np.uint64(seed))
        else:
            step = np.uint64(M64 // table)
            h = (g if pattern == "sorted" else np.uint64(table - 1) - g) * step
    sorted_ = pattern.startswith("sorted")
    mulhi = lambda a, b: np.array([(int(x) * b) >> 64 for x in a], dtype=np.uint64)  # noqa: E731
    if dtype == RV_INT64:
        return (mulhi(h, modulus) if sorted_ else h % np.uint64(modulus)).astype(np.int64)
    if dtype == RV_FLOAT64:
        return (h >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    return (mulhi(h, 100) if sorted_ else h % np.uint64(100)) < np.uint64(true_percent)


@pytest.mark.parametrize("pattern,kw", [("iid", {}), ("clustered", dict(run_rows=37)), ("clustered", dict(run_rows=4096)),
                                        ("sorted", {}), ("sorted", dict(table_rows=50_000)), ("sorted_desc", dict(table_rows=50_000))])
@pytest.mark.parametrize("dtype", [RV_INT64, RV_FLOAT64, RV_BOOLEAN])
def test_oracle_generator_is_the_headers_formula(oracle, pattern, kw, dtype):
    for first, length in ((0, 20_011), (12_345, 9_999)):
        spec = synth_spec(dtype, seed=42, length=length, first_row=first, true_percent=30, validity_seed=44, pattern=pattern, **kw)
        got = oracle.generate(spec)
        want = numpy_generate(dtype, 42, first, length, 1000, 30, pattern, kw.get("run_rows", 0), kw.get("table_rows", 0))
        have = got.logical_values()
        assert np.array_equal(have.view(np.uint64) if dtype == RV_FLOAT64 else have, want.view(np.uint64) if dtype == RV_FLOAT64 else want), (pattern, kw, dtype, first)
        with np.errstate(over="ignore"):
            valid = splitmix64(np.arange(first, first + length, dtype=np.uint64) + np.uint64(44)) % np.uint64(100) >= np.uint64(5)
        assert np.array_equal(unpack_bits(got.validity, length), valid)  # independent rows under every pattern


def test_what_the_patterns_promise(oracle):
    asc = oracle.generate(synth_spec(RV_INT64, seed=1, length=100_000, pattern="sorted")).values
    assert (np.diff(asc) >= 0).all() and asc[0] == 0 and asc[-1] == 999 and abs(int((asc > 899).sum()) - 10_000) <= 1
    desc = oracle.generate(synth_spec(RV_FLOAT64, seed=1, length=100_000, pattern="sorted_desc")).values
    assert (np.diff(desc) <= 0).all() and 0.0 <= desc[-1] < desc[0] < 1.0
    runs = oracle.generate(synth_spec(RV_INT64, seed=1, length=100_000, pattern="clustered", run_rows=1000)).values
    assert all(len(set(runs[i:i + 1000])) == 1 for i in range(0, 100_000, 1000)) and len(set(runs[::1000])) > 50
    b = oracle.generate(synth_spec(RV_BOOLEAN, seed=1, length=100_000, true_percent=10, pattern="sorted")).logical_values()
    assert b[:9_999].all() and not b[10_001:].any()  # the first tenth is true
    # shards of a sorted table agree with the unsharded column
    whole = oracle.generate(synth_spec(RV_INT64, seed=1, length=100_000, pattern="sorted")).values
    part = oracle.generate(synth_spec(RV_INT64, seed=1, length=30_000, first_row=50_000, pattern="sorted", table_rows=100_000)).values
    assert np.array_equal(whole[50_000:80_000], part)


def test_generator_rejects_what_it_cannot_make(oracle):
    with pytest.raises(oracle.OracleError):
        oracle.generate(synth_spec(RV_INT64, seed=1, length=10, pattern="clustered", run_rows=0))
    with pytest.raises(oracle.OracleError):
        oracle.generate(synth_spec(RV_INT64, seed=1, length=10, first_row=95, pattern="sorted", table_rows=100))
    with pytest.raises(oracle.OracleError):
        oracle.generate(synth_spec(RV_INT64, seed=1, length=10, pattern=9))
