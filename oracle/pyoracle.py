"""ORACLE -- TEST INFRASTRUCTURE ONLY (ctypes binding of oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
It takes and returns the same `Column` / `Predicate` objects as rivulus_amd.capi so a
parity test is: same inputs -> oracle result == GPU result, byte for byte.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from typing import List, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from rivulus_amd.capi import (RV_BOOLEAN, RV_FLOAT64, RV_INT64, RV_STRING, Column, Predicate, RvColumn, RvPredicate,  # noqa: E402
                              RvSynthSpec, Term)

LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-j4"], check=True, stdout=subprocess.DEVNULL)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.orc_last_error.restype = C.c_char_p
        lib.orc_result_rows.restype = C.c_uint64
        lib.orc_result_rows.argtypes = [C.c_void_p]
        lib.orc_result_ncols.restype = C.c_uint32
        lib.orc_result_ncols.argtypes = [C.c_void_p]
        lib.orc_result_free.argtypes = [C.c_void_p]
        lib.orc_result_column.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(RvColumn), C.POINTER(C.c_int),
                                          C.POINTER(C.c_uint64)]
        _lib = lib
    return _lib


class OracleError(RuntimeError):
    pass


def _check(rc: int):
    if rc != 0:
        raise OracleError(load().orc_last_error().decode())


def _structs(cols: Sequence[Column]):
    arr = (RvColumn * max(1, len(cols)))()
    for i, c in enumerate(cols):
        arr[i] = c.as_struct()
    return arr


def _collect(res) -> List[Column]:
    lib = load()
    out = []
    try:
        for j in range(lib.orc_result_ncols(res)):
            s, has, nulls = RvColumn(), C.c_int(), C.c_uint64()
            _check(lib.orc_result_column(res, j, C.byref(s), C.byref(has), C.byref(nulls)))
            n = int(s.length)
            if s.dtype == RV_BOOLEAN:
                vals = np.ctypeslib.as_array(C.cast(s.values, C.POINTER(C.c_uint8)), ((n + 7) // 8,)).copy() if n else np.zeros(0, np.uint8)
            elif s.dtype in (RV_INT64, RV_FLOAT64):
                ct, nt = (C.c_int64, np.int64) if s.dtype == RV_INT64 else (C.c_double, np.float64)
                vals = np.ctypeslib.as_array(C.cast(s.values, C.POINTER(ct)), (n,)).copy() if n else np.zeros(0, nt)
            else:
                vals = np.zeros(0, np.uint8)
            offs = None
            if s.dtype == RV_STRING:
                nb = int(s.data_bytes)
                vals = np.ctypeslib.as_array(C.cast(s.values, C.POINTER(C.c_uint8)), (nb,)).copy() if nb else np.zeros(0, np.uint8)
                offs = np.ctypeslib.as_array(C.cast(s.offsets, C.POINTER(C.c_int32)), (n + 1,)).copy()
            valid = None
            if has.value:
                valid = np.ctypeslib.as_array(C.cast(s.validity, C.POINTER(C.c_uint8)), ((n + 7) // 8,)).copy() if n else np.zeros(0, np.uint8)
            out.append(Column(s.dtype, vals, valid, 0, n, offs))
    finally:
        lib.orc_result_free(res)
    return out


def generate(spec: RvSynthSpec) -> Column:
    n = int(spec.length)
    if spec.dtype == RV_BOOLEAN:
        vals = np.zeros((n + 7) // 8, np.uint8)
    else:
        vals = np.zeros(n, np.int64 if spec.dtype == RV_INT64 else np.float64)
    valid = np.zeros((n + 7) // 8, np.uint8) if spec.with_validity else None
    _check(load().orc_generate(C.byref(spec), C.c_void_p(vals.ctypes.data if vals.size else None),
                               C.c_void_p(valid.ctypes.data if valid is not None and valid.size else None)))
    return Column(spec.dtype, vals, valid, 0, n)


def eval_predicate(cols: Sequence[Column], pred: Predicate):
    p, _keep = pred.as_struct()
    n = cols[0].length
    bits = np.zeros((n + 7) // 8, np.uint8)
    cnt = C.c_uint64()
    _check(load().orc_eval_predicate(_structs(cols), len(cols), C.byref(p), C.c_void_p(bits.ctypes.data if bits.size else None),
                                     C.byref(cnt)))
    return Column(RV_BOOLEAN, bits, None, 0, n), cnt.value


def compare(col: Column, op: str, literal) -> Column:
    _p, keep = Predicate([Term(0, op, literal)]).as_struct()
    t = keep[0][0]
    if isinstance(literal, str) or col.dtype == RV_STRING:
        res = C.c_void_p()
        s = col.as_struct()
        _check(load().orc_compare_term(C.byref(s), C.byref(t), C.byref(res)))
        return _collect(res)[0]
    res = C.c_void_p()
    s = col.as_struct()
    _check(load().orc_compare(C.byref(s), t.op, t.lit_type, C.c_int64(t.lit.i if t.lit_type != RV_FLOAT64 else 0),
                              C.c_double(t.lit.f if t.lit_type == RV_FLOAT64 else 0.0), C.byref(res)))
    return _collect(res)[0]


def boolean_op(kind: str, a: Column, b: Column = None) -> Column:
    res = C.c_void_p()
    sa = a.as_struct()
    sb = (b or a).as_struct()
    _check(load().orc_boolean_op({"and": 0, "or": 1, "not": 2}[kind], C.byref(sa), C.byref(sb), C.byref(res)))
    return _collect(res)[0]


def boolean_count(a: Column):
    t, f = C.c_uint64(), C.c_uint64()
    s = a.as_struct()
    _check(load().orc_boolean_count(C.byref(s), C.byref(t), C.byref(f)))
    return t.value, f.value


def null_count(a: Column) -> int:
    out = C.c_uint64()
    s = a.as_struct()
    _check(load().orc_null_count(C.byref(s), C.byref(out)))
    return out.value


def filter(cols: Sequence[Column], predicate: Column) -> List[Column]:
    res = C.c_void_p()
    s = predicate.as_struct()
    _check(load().orc_filter(_structs(cols), len(cols), C.byref(s), C.byref(res)))
    return _collect(res)


def take(cols: Sequence[Column], indices) -> List[Column]:
    idx = np.asarray(indices, dtype=np.uint64)
    res = C.c_void_p()
    _check(load().orc_take(_structs(cols), len(cols), idx.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(len(idx)),
                           C.byref(res)))
    return _collect(res)


def slice_(col: Column, offset: int, length: int) -> Column:
    res = C.c_void_p()
    s = col.as_struct()
    _check(load().orc_slice(C.byref(s), C.c_uint64(offset), C.c_uint64(length), C.byref(res)))
    return _collect(res)[0]


def concat(parts: Sequence[Column]) -> Column:
    res = C.c_void_p()
    _check(load().orc_concat(_structs(parts), len(parts), C.byref(res)))
    return _collect(res)[0]


def filter_project(cols: Sequence[Column], pred: Predicate, proj: Sequence[int]) -> List[Column]:
    p, _keep = pred.as_struct()
    pj = (C.c_uint32 * max(1, len(proj)))(*proj)
    res = C.c_void_p()
    _check(load().orc_filter_project(_structs(cols), len(cols), C.byref(p), pj, len(proj), C.byref(res)))
    return _collect(res)


def stream_filter_project(cols: Sequence[Column], batch_rows: int, pred: Predicate, proj: Sequence[int]) -> List[Column]:
    p, _keep = pred.as_struct()
    pj = (C.c_uint32 * max(1, len(proj)))(*proj)
    res = C.c_void_p()
    _check(load().orc_stream_filter_project(_structs(cols), len(cols), C.c_uint64(batch_rows), C.byref(p), pj, len(proj),
                                            C.byref(res)))
    return _collect(res)


def filter_agg(cols: Sequence[Column], pred: Predicate, agg_col: int):
    p, _keep = pred.as_struct()
    si, sf, cnt = C.c_int64(), C.c_double(), C.c_uint64()
    _check(load().orc_filter_agg(_structs(cols), len(cols), C.byref(p), agg_col, C.byref(si), C.byref(sf), C.byref(cnt)))
    return si.value, sf.value, cnt.value


def synth_filter_checksums(seed: int, first_row: int, n_rows: int, modulus: int, literal: int, threads: int = 0):
    """Exact COUNT, wrapping SUM and order checksum sum(ordinal * value) mod 2^64 of the survivors of `x > literal` over the
    synthetic Int64 column, streamed (nothing is materialised): the full-size checker of the 1e10-row tests."""
    import os
    threads = threads or min(os.cpu_count() or 1, 32)
    s, c, w = C.c_int64(), C.c_uint64(), C.c_uint64()
    _check(load().orc_synth_filter_checksums(C.c_uint64(seed), C.c_uint64(first_row), C.c_uint64(n_rows), C.c_uint64(modulus),
                                             C.c_int64(literal), C.c_uint32(threads), C.byref(s), C.byref(c), C.byref(w)))
    return s.value, c.value, w.value


def bench_eager_collect(n_rows: int, seed: int, modulus: int, literal: int):
    sec, rows, cs = C.c_double(), C.c_uint64(), C.c_int64()
    _check(load().orc_bench_eager_collect(C.c_uint64(n_rows), C.c_uint64(seed), C.c_uint64(modulus), C.c_int64(literal),
                                          C.byref(sec), C.byref(rows), C.byref(cs)))
    return sec.value, rows.value, cs.value


def bench_stream(n_rows: int, seed: int, modulus: int, literal: int, batch_rows: int = 1024):
    sec, rows, cs = C.c_double(), C.c_uint64(), C.c_int64()
    _check(load().orc_bench_stream(C.c_uint64(n_rows), C.c_uint64(seed), C.c_uint64(modulus), C.c_int64(literal),
                                   C.c_uint64(batch_rows), C.byref(sec), C.byref(rows), C.byref(cs)))
    return sec.value, rows.value, cs.value


def bench_threads(n_rows: int, seed: int, modulus: int, literal: int, threads: int):
    """Courtesy figure: typed compress loop on `threads` host threads (same result, not the reference's algorithm)."""
    sec, rows, cs = C.c_double(), C.c_uint64(), C.c_int64()
    _check(load().orc_bench_threads(C.c_uint64(n_rows), C.c_uint64(seed), C.c_uint64(modulus), C.c_int64(literal),
                                    C.c_uint32(threads), C.byref(sec), C.byref(rows), C.byref(cs)))
    return sec.value, rows.value, cs.value
