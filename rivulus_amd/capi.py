"""ctypes binding of include/rivulus_gpu.h -- the driver used by tests and bench.py.

The product is the C-ABI shared library (rivulus_amd/csrc/librivulus_gpu.so) plus the
C++ host layer; this module only marshals numpy buffers into `rv_column` structs.  There
is NO fallback: if the library is missing, or no gfx950 device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RIVULUS_GPU_LIB: another build of the library (same-box A/B runs of two builds; development aid)
LIB_PATH = os.environ.get("RIVULUS_GPU_LIB") or os.path.join(_HERE, "csrc", "librivulus_gpu.so")

# ---- enums (include/rivulus_gpu.h) -----------------------------------------------------
RV_OK = 0
RV_NULL, RV_BOOLEAN, RV_INT64, RV_FLOAT64, RV_STRING = 0, 1, 2, 3, 4
RV_EQ, RV_NE, RV_LT, RV_GT, RV_LE, RV_GE, RV_IS_TRUE = range(7)
RV_NULL_DROPS, RV_NULL_IS_LEAST = 0, 1
RV_COMM_ID_BYTES = 128
OPS = {"==": RV_EQ, "!=": RV_NE, "<": RV_LT, ">": RV_GT, "<=": RV_LE, ">=": RV_GE, "is_true": RV_IS_TRUE}
STATUS_NAMES = ["RV_OK", "RV_ERR_INVALID_ARG", "RV_ERR_LENGTH_MISMATCH", "RV_ERR_TYPE_MISMATCH",
                "RV_ERR_OUT_OF_BOUNDS", "RV_ERR_UNSUPPORTED", "RV_ERR_DEVICE", "RV_ERR_OOM", "RV_ERR_INTERNAL"]


class RvColumn(C.Structure):
    _fields_ = [("dtype", C.c_int), ("values", C.c_void_p), ("validity", C.c_void_p),
                ("offset", C.c_uint64), ("length", C.c_uint64),
                ("offsets", C.c_void_p), ("data_bytes", C.c_uint64)]  # RV_STRING: int32 offsets, bytes in values


class _LitStr(C.Structure):
    _fields_ = [("ptr", C.c_char_p), ("len", C.c_uint64)]


class _Lit(C.Union):
    _fields_ = [("i", C.c_int64), ("f", C.c_double), ("s", _LitStr)]


class RvTerm(C.Structure):
    _fields_ = [("column", C.c_uint32), ("op", C.c_int), ("lit_type", C.c_int), ("lit", _Lit)]


class RvPredicate(C.Structure):
    _fields_ = [("terms", C.POINTER(RvTerm)), ("n_terms", C.c_uint32), ("nulls", C.c_int),
                ("expr", C.POINTER(C.c_uint8)), ("n_expr", C.c_uint32)]  # postfix AND / OR / NOT over the terms, or NULL


RV_EXPR_AND, RV_EXPR_OR, RV_EXPR_NOT = 0x80, 0x81, 0x82


def postfix(tree) -> List[int]:
    """Nested expression over term indices -> rv_predicate::expr bytes.
    tree: int (term index) | ("and" | "or", a, b, ...) | ("not", a)."""
    if isinstance(tree, (int, np.integer)):
        return [int(tree)]
    op, *args = tree
    if op == "not":
        (a,) = args
        return postfix(a) + [RV_EXPR_NOT]
    code = {"and": RV_EXPR_AND, "or": RV_EXPR_OR}[op]
    out = postfix(args[0])
    for a in args[1:]:
        out += postfix(a) + [code]
    return out


class RvSynthSpec(C.Structure):
    _fields_ = [("dtype", C.c_int), ("seed", C.c_uint64), ("first_row", C.c_uint64), ("length", C.c_uint64),
                ("modulus", C.c_uint64), ("true_percent", C.c_uint32), ("with_validity", C.c_int32),
                ("validity_seed", C.c_uint64), ("null_percent", C.c_uint32), ("pattern", C.c_uint32),
                ("run_rows", C.c_uint64), ("table_rows", C.c_uint64)]


RV_SYNTH_IID, RV_SYNTH_CLUSTERED, RV_SYNTH_SORTED_ASC, RV_SYNTH_SORTED_DESC = 0, 1, 2, 3
SYNTH_PATTERNS = {"iid": RV_SYNTH_IID, "clustered": RV_SYNTH_CLUSTERED, "sorted": RV_SYNTH_SORTED_ASC, "sorted_desc": RV_SYNTH_SORTED_DESC}


class RvColumnInfo(C.Structure):
    _fields_ = [("dtype", C.c_int), ("length", C.c_uint64), ("offset", C.c_uint64),
                ("has_validity", C.c_int32), ("null_count", C.c_int64), ("data_bytes", C.c_uint64)]


# every symbol include/rivulus_gpu.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_U64P = C.POINTER(C.c_uint64)
PROTOTYPES = {
    "rv_abi_version": (C.c_uint32, []),
    "rv_last_error": (C.c_char_p, []),
    "rv_status_name": (C.c_char_p, [C.c_int]),
    "rv_device_count": (C.c_int, []),
    "rv_ctx_create": (C.c_int, [C.c_int, _PP]),
    "rv_ctx_destroy": (C.c_int, [_P]),
    "rv_ctx_synchronize": (C.c_int, [_P]),
    "rv_ctx_stream": (C.c_void_p, [_P]),
    "rv_ctx_device_info": (C.c_int, [_P, C.POINTER(C.c_int), _U64P, C.c_char_p, C.c_size_t]),
    "rv_ctx_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "rv_ctx_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "rv_ctx_kernel_stats": (C.c_int, [_P, C.POINTER(C.c_double), _U64P, C.c_int]),
    "rv_ctx_last_kernel": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "rv_timer_start": (C.c_int, [_P]),
    "rv_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "rv_upload": (C.c_int, [_P, C.POINTER(RvColumn), _PP]),
    "rv_wrap": (C.c_int, [_P, C.POINTER(RvColumn), _PP]),
    "rv_generate": (C.c_int, [_P, C.POINTER(RvSynthSpec), _PP]),
    "rv_free": (C.c_int, [_P, _P]),
    "rv_slice": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _PP]),
    "rv_column_info_get": (C.c_int, [_P, _P, C.POINTER(RvColumnInfo)]),
    "rv_null_count": (C.c_int, [_P, _P, _U64P]),
    "rv_download": (C.c_int, [_P, _P, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "rv_device_ptrs": (C.c_int, [_P, _P, C.POINTER(RvColumn)]),
    "rv_eval_predicate": (C.c_int, [_P, _PP, C.c_uint32, C.POINTER(RvPredicate), _PP, _U64P]),
    "rv_compare": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, C.c_double, _PP]),
    "rv_boolean_and": (C.c_int, [_P, _P, _P, _PP]),
    "rv_boolean_or": (C.c_int, [_P, _P, _P, _PP]),
    "rv_boolean_not": (C.c_int, [_P, _P, _PP]),
    "rv_boolean_count": (C.c_int, [_P, _P, _U64P, _U64P]),
    "rv_filter": (C.c_int, [_P, _PP, C.c_uint32, _P, _PP, _U64P]),
    "rv_take": (C.c_int, [_P, _PP, C.c_uint32, _U64P, C.c_uint64, _PP]),
    "rv_take_device": (C.c_int, [_P, _PP, C.c_uint32, _P, _PP]),
    "rv_selection_indices": (C.c_int, [_P, _P, _PP]),
    "rv_concat": (C.c_int, [_P, _PP, C.c_uint32, _PP]),
    "rv_filter_project": (C.c_int, [_P, _PP, C.c_uint32, C.POINTER(RvPredicate), C.POINTER(C.c_uint32), C.c_uint32,
                                    _PP, _U64P, _PP]),
    "rv_download_string": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(C.c_int)]),
    "rv_compare_term": (C.c_int, [_P, _P, C.POINTER(RvTerm), _PP]),
    "rv_fill_nulls": (C.c_int, [_P, _P, _PP]),
    "rv_filter_project_begin": (C.c_int, [_P, _PP, C.c_uint32, C.POINTER(RvPredicate), C.POINTER(C.c_uint32), C.c_uint32, _PP]),
    "rv_filter_project_finish": (C.c_int, [_P, _P, _PP, _U64P]),
    "rv_filter_project_batches": (C.c_int, [_P, _PP, C.c_uint32, C.c_uint32, C.POINTER(RvPredicate), C.POINTER(C.c_uint32), C.c_uint32,
                                            _PP, _U64P, C.POINTER(C.c_int64), _U64P]),
    "rv_filter_project_chunked": (C.c_int, [_P, _PP, C.c_uint32, C.c_uint64, C.POINTER(RvPredicate), C.POINTER(C.c_uint32), C.c_uint32,
                                            _PP, _U64P, C.c_uint64, C.POINTER(C.c_int64), _U64P]),
    "rv_filter_project_chunked_begin": (C.c_int, [_P, _PP, C.c_uint32, C.c_uint64, C.POINTER(RvPredicate), C.POINTER(C.c_uint32), C.c_uint32, _U64P, C.c_uint64, _PP]),
    "rv_filter_project_batches_begin": (C.c_int, [_P, _PP, C.c_uint32, C.c_uint32, C.POINTER(RvPredicate), C.POINTER(C.c_uint32), C.c_uint32, _U64P, _PP]),
    "rv_filter_project_window_finish": (C.c_int, [_P, _P, _PP, C.POINTER(C.c_int64), _U64P]),
    "rv_slice_known": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_int64, _PP]),
    "rv_host_alloc": (C.c_int, [_P, C.c_size_t, _PP]),
    "rv_host_free": (C.c_int, [_P, _P]),
    "rv_filter_project_host": (C.c_int, [_P, C.POINTER(RvColumn), C.c_uint32, C.POINTER(RvPredicate), C.POINTER(C.c_uint32),
                                         C.c_uint32, C.c_uint64, _PP, _U64P]),
    "rv_filter_agg": (C.c_int, [_P, _PP, C.c_uint32, C.POINTER(RvPredicate), C.c_uint32, C.POINTER(C.c_int64),
                                C.POINTER(C.c_double), _U64P]),
    "rv_shard_range": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, _U64P, _U64P]),
    "rv_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rv_comm_create": (C.c_int, [_P, C.c_char_p, C.c_uint32, C.c_uint32, _PP]),
    "rv_comm_allreduce_sum_count": (C.c_int, [_P, C.POINTER(C.c_int64), _U64P]),
    "rv_comm_destroy": (C.c_int, [_P]),
    "rv_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_uint32, _PP]),
    "rv_group_destroy": (C.c_int, [_P]),
    "rv_group_size": (C.c_uint32, [_P]),
    "rv_group_ctx": (C.c_void_p, [_P, C.c_uint32]),
    "rv_group_generate": (C.c_int, [_P, C.POINTER(RvSynthSpec), _PP]),
    "rv_group_upload": (C.c_int, [_P, C.POINTER(RvColumn), _PP]),
    "rv_group_free": (C.c_int, [_P, _PP]),
    "rv_group_filter_project_host": (C.c_int, [_P, C.POINTER(RvColumn), C.c_uint32, C.POINTER(RvPredicate), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint64, _PP, _U64P, C.POINTER(C.c_double)]),
    "rv_group_filter_project": (C.c_int, [_P, _PP, C.c_uint32, C.POINTER(RvPredicate), C.POINTER(C.c_uint32), C.c_uint32, _PP, _U64P]),
    "rv_group_filter_project_resident": (C.c_int, [_P, _PP, C.c_uint32, C.POINTER(RvPredicate), C.POINTER(C.c_uint32), C.c_uint32, _PP,
                                                   _U64P, _U64P]),
    "rv_group_gather": (C.c_int, [_P, _PP, C.c_uint32, _PP]),
    "rv_gather_column": (C.c_int, [_P, C.c_uint32, C.POINTER(RvColumn), C.POINTER(C.c_int64)]),
    "rv_gather_stats": (C.c_int, [_P, _U64P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rv_gather_free": (C.c_int, [_P]),
    "rv_group_filter_agg": (C.c_int, [_P, _PP, C.c_uint32, C.POINTER(RvPredicate), C.c_uint32, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_double), _U64P]),
    "rv_group_stat": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "rv_host_register": (C.c_int, [_P, _P, C.c_size_t]),
    "rv_host_unregister": (C.c_int, [_P, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load librivulus_gpu.so (built by __graft_entry__.build()).  Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the MI355X backend has no Python/CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class RvError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES[status] if status < len(STATUS_NAMES) else status}: {message}")
        self.status = status
        self.message = message


def _check(status: int):
    if status != RV_OK:
        raise RvError(status, load().rv_last_error().decode())


# ---- host-side column description ----------------------------------------------------------
def pack_bits(mask: np.ndarray) -> np.ndarray:
    """bool array -> LSB-first packed bytes (reference BitMap::from_bool_slice, bitmap.rs:44-59)."""
    return np.packbits(np.asarray(mask, dtype=bool), bitorder="little")


def unpack_bits(buf: np.ndarray, n: int, offset: int = 0) -> np.ndarray:
    return np.unpackbits(np.asarray(buf, dtype=np.uint8), bitorder="little")[offset:offset + n].astype(bool)


_NP_DTYPE = {RV_INT64: np.int64, RV_FLOAT64: np.float64}


@dataclass
class Column:
    """Host view of PrimitiveArray<i64|f64> / BooleanArray: buffers + offset + length."""
    dtype: int
    values: np.ndarray            # int64 / float64 elements, or packed uint8 bits for RV_BOOLEAN
    validity: Optional[np.ndarray]  # packed uint8 bits or None
    offset: int
    length: int
    offsets: Optional[np.ndarray] = None  # RV_STRING: int32[>= offset + length + 1]; values = uint8 UTF-8 bytes

    @staticmethod
    def nulls(n: int) -> "Column":
        """NullArray::new(n) (null.rs:5-66): a length, no buffers."""
        return Column(RV_NULL, np.zeros(0, dtype=np.uint8), None, 0, int(n))

    @staticmethod
    def from_strings(strings: Sequence[Optional[str]]) -> "Column":
        """StringArray::new (string.rs:19-57): a null spans no bytes; bitmap only if there is a null."""
        data = bytearray()
        offs = [0]
        for x in strings:
            if x is not None:
                data += x.encode("utf-8")
            offs.append(len(data))
        valid = None
        if any(x is None for x in strings):
            valid = pack_bits(np.array([x is not None for x in strings], dtype=bool))
        return Column(RV_STRING, np.frombuffer(bytes(data), dtype=np.uint8).copy(), valid, 0, len(strings),
                      np.asarray(offs, dtype=np.int32))

    def to_strings(self) -> List[Optional[str]]:
        assert self.dtype == RV_STRING
        valid = self.logical_valid()
        out = []
        for i in range(self.length):
            if valid is not None and not valid[i]:
                out.append(None)
            else:
                a, b = int(self.offsets[self.offset + i]), int(self.offsets[self.offset + i + 1])
                out.append(bytes(self.values[a:b]).decode("utf-8"))
        return out

    @staticmethod
    def from_numpy(values: np.ndarray, valid: Optional[np.ndarray] = None) -> "Column":
        values = np.asarray(values)
        n = len(values)
        v = None if valid is None else pack_bits(valid)
        if values.dtype == np.bool_:
            return Column(RV_BOOLEAN, pack_bits(values), v, 0, n)
        if values.dtype == np.int64:
            return Column(RV_INT64, np.ascontiguousarray(values), v, 0, n)
        if values.dtype == np.float64:
            return Column(RV_FLOAT64, np.ascontiguousarray(values), v, 0, n)
        raise TypeError(f"unsupported numpy dtype {values.dtype}")

    def slice(self, offset: int, length: int) -> "Column":
        assert offset + length <= self.length
        return Column(self.dtype, self.values, self.validity, self.offset + offset, length, self.offsets)

    # logical content
    def logical_values(self) -> np.ndarray:
        if self.dtype == RV_BOOLEAN:
            return unpack_bits(self.values, self.length, self.offset)
        return self.values[self.offset:self.offset + self.length]

    def logical_valid(self) -> Optional[np.ndarray]:
        return None if self.validity is None else unpack_bits(self.validity, self.length, self.offset)

    def as_struct(self) -> RvColumn:
        s = RvColumn()
        s.dtype = self.dtype
        s.values = self.values.ctypes.data if self.values is not None and self.values.size else None
        s.validity = self.validity.ctypes.data if self.validity is not None else None
        s.offset = self.offset
        s.length = self.length
        s.offsets = self.offsets.ctypes.data if self.offsets is not None else None
        s.data_bytes = int(self.values.size) if self.dtype == RV_STRING else 0
        return s

    def same_as(self, other: "Column") -> Optional[str]:
        """None when bit-identical in the sense of the parity bar, else a description."""
        if self.dtype != other.dtype:
            return f"dtype {self.dtype} != {other.dtype}"
        if self.length != other.length:
            return f"length {self.length} != {other.length}"
        if (self.validity is None) != (other.validity is None):
            return f"has_validity {self.validity is not None} != {other.validity is not None}"
        if self.dtype == RV_NULL:
            return None
        if self.dtype == RV_STRING:  # logical elements (None under a null); nulls span no bytes on both sides
            a, b = self.to_strings(), other.to_strings()
            for i, (x, y) in enumerate(zip(a, b)):
                if x != y:
                    return f"string element {i}: {x!r} != {y!r}"
            la = int(self.offsets[self.offset + self.length]) - int(self.offsets[self.offset])
            lb = int(other.offsets[other.offset + other.length]) - int(other.offsets[other.offset])
            return None if la == lb else f"logical bytes {la} != {lb}"
        a, b = self.logical_values(), other.logical_values()
        if self.dtype == RV_FLOAT64:
            a, b = a.view(np.uint64), b.view(np.uint64)
        if not np.array_equal(a, b):
            idx = int(np.nonzero(a != b)[0][0])
            return f"values differ first at row {idx}: {a[idx]} != {b[idx]}"
        if self.validity is not None and not np.array_equal(self.logical_valid(), other.logical_valid()):
            return "validity bits differ"
        return None


@dataclass
class Term:
    column: int
    op: str
    literal: object = None  # None -> Literal(AnyValue::Null); int / float / bool / str


@dataclass
class Predicate:
    terms: List[Term]
    nulls: str = "drops"  # "drops" (streaming composition) | "least" (eager AnyValue ordering)
    expr: object = None   # None: AND of the terms; else a nested ("and" | "or" | "not", ...) tree over term indices

    def as_struct(self):
        arr = (RvTerm * len(self.terms))()
        keep = [arr]
        for i, t in enumerate(self.terms):
            arr[i].column = t.column
            arr[i].op = OPS[t.op]
            lit = t.literal
            if t.op == "is_true" or lit is None:
                arr[i].lit_type = RV_NULL
            elif isinstance(lit, (bool, np.bool_)):
                arr[i].lit_type = RV_BOOLEAN
                arr[i].lit.i = int(lit)
            elif isinstance(lit, (int, np.integer)):
                arr[i].lit_type = RV_INT64
                arr[i].lit.i = int(lit)
            elif isinstance(lit, (float, np.floating)):
                arr[i].lit_type = RV_FLOAT64
                arr[i].lit.f = float(lit)
            elif isinstance(lit, str):
                raw = lit.encode("utf-8")
                keep.append(raw)
                arr[i].lit_type = RV_STRING
                arr[i].lit.s.ptr = raw
                arr[i].lit.s.len = len(raw)
            else:
                raise TypeError(f"unsupported literal {lit!r}")
        p = RvPredicate()
        p.terms = arr
        p.n_terms = len(self.terms)
        p.nulls = RV_NULL_IS_LEAST if self.nulls == "least" else RV_NULL_DROPS
        if self.expr is not None:
            code = postfix(self.expr)
            prog = (C.c_uint8 * len(code))(*code)
            keep.append(prog)
            p.expr = prog
            p.n_expr = len(code)
        return p, keep  # keeps the term array, the literal bytes and the expression alive


def synth_spec(dtype: int, seed: int, length: int, first_row: int = 0, modulus: int = 1000, true_percent: int = 50,
               validity_seed: Optional[int] = None, null_percent: int = 5, pattern="iid", run_rows: int = 0,
               table_rows: int = 0) -> RvSynthSpec:
    """pattern: "iid" (independent rows), "clustered" (runs of run_rows equal cells), "sorted" / "sorted_desc" (values grow /
    fall with the row index over a table of table_rows rows; 0 = first_row + length): include/rivulus_gpu.h, rv_synth_spec."""
    s = RvSynthSpec()
    s.dtype, s.seed, s.first_row, s.length, s.modulus = dtype, seed, first_row, length, modulus
    s.true_percent = true_percent
    s.with_validity = 0 if validity_seed is None else 1
    s.validity_seed = validity_seed or 0
    s.null_percent = null_percent
    s.pattern = SYNTH_PATTERNS[pattern] if isinstance(pattern, str) else int(pattern)
    s.run_rows, s.table_rows = run_rows, table_rows
    return s


# ---- device side ------------------------------------------------------------------------------
class DeviceColumn:
    def __init__(self, ctx: "Context", handle):
        self.ctx, self.handle = ctx, handle

    def info(self) -> RvColumnInfo:
        i = RvColumnInfo()
        _check(load().rv_column_info_get(self.ctx.handle, self.handle, C.byref(i)))
        return i

    @property
    def length(self) -> int:
        return int(self.info().length)

    def null_count(self) -> int:
        out = C.c_uint64()
        _check(load().rv_null_count(self.ctx.handle, self.handle, C.byref(out)))
        return out.value

    def slice(self, offset: int, length: int) -> "DeviceColumn":
        out = C.c_void_p()
        _check(load().rv_slice(self.ctx.handle, self.handle, offset, length, C.byref(out)))
        return DeviceColumn(self.ctx, out)

    def download(self) -> Column:
        i = self.info()
        n = int(i.length)
        if i.dtype == RV_STRING:
            offs = np.zeros(n + 1, dtype=np.int32)
            data = np.zeros(int(i.data_bytes), dtype=np.uint8)
            valid = np.zeros((n + 7) // 8, dtype=np.uint8) if i.has_validity else None
            has = C.c_int()
            _check(load().rv_download_string(self.ctx.handle, self.handle, offs.ctypes.data, data.ctypes.data if data.size else None,
                                             valid.ctypes.data if valid is not None and valid.size else None, C.byref(has)))
            return Column(RV_STRING, data, valid, 0, n, offs)
        if i.dtype == RV_NULL:
            return Column(RV_NULL, np.zeros(0, dtype=np.uint8), None, 0, n)
        if i.dtype == RV_BOOLEAN:
            vals = np.zeros((n + 7) // 8, dtype=np.uint8)
        else:
            vals = np.zeros(n, dtype=_NP_DTYPE[i.dtype])
        valid = np.zeros((n + 7) // 8, dtype=np.uint8) if i.has_validity else None
        has = C.c_int()
        _check(load().rv_download(self.ctx.handle, self.handle, vals.ctypes.data if vals.size else None,
                                  valid.ctypes.data if valid is not None and valid.size else None, C.byref(has)))
        return Column(i.dtype, vals, valid, 0, n)

    def fill_nulls(self) -> "DeviceColumn":
        out = C.c_void_p()
        _check(load().rv_fill_nulls(self.ctx.handle, self.handle, C.byref(out)))
        return DeviceColumn(self.ctx, out)

    def device_ptrs(self) -> RvColumn:
        s = RvColumn()
        _check(load().rv_device_ptrs(self.ctx.handle, self.handle, C.byref(s)))
        return s

    def free(self):
        if self.handle is not None:
            load().rv_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            if self.ctx.handle is not None:
                self.free()
        except Exception:
            pass


def _handles(cols: Sequence[DeviceColumn]):
    arr = (C.c_void_p * max(1, len(cols)))()
    for i, c in enumerate(cols):
        arr[i] = c.handle
    return arr


class Context:
    def __init__(self, device: int = 0):
        self.handle = None
        h = C.c_void_p()
        _check(load().rv_ctx_create(device, C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle is not None:
            for ptr in getattr(self, "_pinned", []):  # arrays from pinned_array() dangle after this
                load().rv_host_free(self.handle, ptr)
            self._pinned = []
            load().rv_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_info(self):
        cu, mem, name = C.c_int(), C.c_uint64(), C.create_string_buffer(256)
        _check(load().rv_ctx_device_info(self.handle, C.byref(cu), C.byref(mem), name, 256))
        return {"compute_units": cu.value, "hbm_bytes": mem.value, "name": name.value.decode()}

    def set_option(self, key: str, value: int):
        _check(load().rv_ctx_set_option(self.handle, key.encode(), value))

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        _check(load().rv_ctx_get_option(self.handle, key.encode(), C.byref(v)))
        return v.value

    def kernel_stats(self, reset: bool = False):
        ms, n = C.c_double(), C.c_uint64()
        _check(load().rv_ctx_kernel_stats(self.handle, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def last_kernel(self) -> str:
        """Instantiation of the last hot-path launch, e.g. 'fused_filter_compact<1,16,2,16,32>'."""
        buf = C.create_string_buffer(256)
        _check(load().rv_ctx_last_kernel(self.handle, buf, len(buf)))
        return buf.value.decode()

    def synchronize(self):
        _check(load().rv_ctx_synchronize(self.handle))

    def timer_start(self):
        _check(load().rv_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        _check(load().rv_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def upload(self, col: Column) -> DeviceColumn:
        s = col.as_struct()
        out = C.c_void_p()
        _check(load().rv_upload(self.handle, C.byref(s), C.byref(out)))
        return DeviceColumn(self, out)

    def wrap(self, desc: RvColumn) -> DeviceColumn:
        out = C.c_void_p()
        _check(load().rv_wrap(self.handle, C.byref(desc), C.byref(out)))
        return DeviceColumn(self, out)

    def generate(self, spec: RvSynthSpec) -> DeviceColumn:
        out = C.c_void_p()
        _check(load().rv_generate(self.handle, C.byref(spec), C.byref(out)))
        return DeviceColumn(self, out)

    def eval_predicate(self, cols: Sequence[DeviceColumn], pred: Predicate):
        p, _keep = pred.as_struct()
        sel, cnt = C.c_void_p(), C.c_uint64()
        _check(load().rv_eval_predicate(self.handle, _handles(cols), len(cols), C.byref(p), C.byref(sel), C.byref(cnt)))
        return DeviceColumn(self, sel), cnt.value

    def compare(self, col: DeviceColumn, op: str, literal) -> DeviceColumn:
        _p, keep = Predicate([Term(0, op, literal)]).as_struct()
        t = keep[0][0]
        out = C.c_void_p()
        if isinstance(literal, str) or col.info().dtype == RV_STRING:
            _check(load().rv_compare_term(self.handle, col.handle, C.byref(t), C.byref(out)))
            return DeviceColumn(self, out)
        _check(load().rv_compare(self.handle, col.handle, t.op, t.lit_type, t.lit.i if t.lit_type != RV_FLOAT64 else 0,
                                 t.lit.f if t.lit_type == RV_FLOAT64 else 0.0, C.byref(out)))
        return DeviceColumn(self, out)

    def boolean_and(self, a, b):
        out = C.c_void_p()
        _check(load().rv_boolean_and(self.handle, a.handle, b.handle, C.byref(out)))
        return DeviceColumn(self, out)

    def boolean_or(self, a, b):
        out = C.c_void_p()
        _check(load().rv_boolean_or(self.handle, a.handle, b.handle, C.byref(out)))
        return DeviceColumn(self, out)

    def boolean_not(self, a):
        out = C.c_void_p()
        _check(load().rv_boolean_not(self.handle, a.handle, C.byref(out)))
        return DeviceColumn(self, out)

    def boolean_count(self, a):
        t, f = C.c_uint64(), C.c_uint64()
        _check(load().rv_boolean_count(self.handle, a.handle, C.byref(t), C.byref(f)))
        return t.value, f.value

    def filter(self, cols: Sequence[DeviceColumn], predicate: DeviceColumn):
        out = (C.c_void_p * max(1, len(cols)))()
        rows = C.c_uint64()
        _check(load().rv_filter(self.handle, _handles(cols), len(cols), predicate.handle, out, C.byref(rows)))
        return [DeviceColumn(self, C.c_void_p(out[i])) for i in range(len(cols))], rows.value

    def take(self, cols: Sequence[DeviceColumn], indices: Sequence[int]):
        idx = np.asarray(indices, dtype=np.uint64)
        out = (C.c_void_p * max(1, len(cols)))()
        _check(load().rv_take(self.handle, _handles(cols), len(cols),
                              idx.ctypes.data_as(C.POINTER(C.c_uint64)), len(idx), out))
        return [DeviceColumn(self, C.c_void_p(out[i])) for i in range(len(cols))]

    def take_device(self, cols: Sequence[DeviceColumn], indices: DeviceColumn):
        out = (C.c_void_p * max(1, len(cols)))()
        _check(load().rv_take_device(self.handle, _handles(cols), len(cols), indices.handle, out))
        return [DeviceColumn(self, C.c_void_p(out[i])) for i in range(len(cols))]

    def selection_indices(self, selection: DeviceColumn) -> DeviceColumn:
        out = C.c_void_p()
        _check(load().rv_selection_indices(self.handle, selection.handle, C.byref(out)))
        return DeviceColumn(self, out)

    def concat(self, parts: Sequence[DeviceColumn]) -> DeviceColumn:
        out = C.c_void_p()
        _check(load().rv_concat(self.handle, _handles(parts), len(parts), C.byref(out)))
        return DeviceColumn(self, out)

    def filter_project(self, cols: Sequence[DeviceColumn], pred: Predicate, proj: Sequence[int],
                       want_selection: bool = False):
        p, _keep = pred.as_struct()
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        out = (C.c_void_p * max(1, len(proj)))()
        rows, sel = C.c_uint64(), C.c_void_p()
        _check(load().rv_filter_project(self.handle, _handles(cols), len(cols), C.byref(p), pj, len(proj), out,
                                        C.byref(rows), C.byref(sel) if want_selection else None))
        outs = [DeviceColumn(self, C.c_void_p(out[i])) for i in range(len(proj))]
        return outs, rows.value, (DeviceColumn(self, sel) if want_selection else None)

    def prepared_filter_project(self, cols: Sequence[DeviceColumn], pred: Predicate, proj: Sequence[int]):
        """The same call with its argument structs built ONCE (what a caller that repeats a query does): returns
        run(keep: bool = False) -> (outs | None, rows).  keep=False releases the outputs at once (rv_free) and returns None
        for them -- the form a throughput loop wants; the Python side of a step is then two C calls and nothing else."""
        lib = load()
        p, _keep = pred.as_struct()
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        out = (C.c_void_p * max(1, len(proj)))()
        handles, ncols, nproj = _handles(cols), len(cols), len(proj)
        rows = C.c_uint64()
        p_ref, rows_ref, ctx_h = C.byref(p), C.byref(rows), self.handle
        fn, free = lib.rv_filter_project, lib.rv_free

        def run(keep: bool = False):
            _check(fn(ctx_h, handles, ncols, p_ref, pj, nproj, out, rows_ref, None))
            if keep:
                return [DeviceColumn(self, C.c_void_p(out[i])) for i in range(nproj)], rows.value
            for i in range(nproj):
                free(ctx_h, out[i])
            return None, rows.value
        run._keep = (_keep, cols, p, pj, out, handles, rows)  # the structs live as long as the closure
        return run

    def filter_project_begin(self, cols: Sequence[DeviceColumn], pred: Predicate, proj: Sequence[int]):
        """Queue the launch and return a callable that finishes it: finish() -> (outs, rows).  The input columns
        are kept alive by the returned closure."""
        p, _keep = pred.as_struct()
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        pend = C.c_void_p()
        _check(load().rv_filter_project_begin(self.handle, _handles(cols), len(cols), C.byref(p), pj, len(proj), C.byref(pend)))
        inputs = list(cols)

        def finish():
            out = (C.c_void_p * max(1, len(proj)))()
            rows = C.c_uint64()
            _check(load().rv_filter_project_finish(self.handle, pend, out, C.byref(rows)))
            inputs.clear()
            return [DeviceColumn(self, C.c_void_p(out[i])) for i in range(len(proj))], rows.value

        return finish

    def batch_handles(self, batches: Sequence[Sequence[DeviceColumn]]):
        """The K x ncols handle array rv_filter_project_batches takes (build once, reuse: a stream's batches)."""
        k, ncols = len(batches), len(batches[0])
        arr = (C.c_void_p * (k * ncols))()
        for b, cols in enumerate(batches):
            for c, col in enumerate(cols):
                arr[b * ncols + c] = col.handle.value if isinstance(col.handle, C.c_void_p) else col.handle
        return arr, k, ncols

    def filter_project_batches(self, batches, pred: Predicate, proj: Sequence[int], want_nulls: bool = True, handles=None,
                               rows_buffer: Optional[np.ndarray] = None):
        """rv_filter_project_batches: K input batches, ONE launch.  Returns (outs, rows_per_batch, nulls, total): outs are
        the nproj back-to-back outputs; nulls[b][j] the null count of output batch b, column j (None if not asked)."""
        arr, k, ncols = handles if handles is not None else self.batch_handles(batches)
        p, _keep = pred.as_struct()
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        out = (C.c_void_p * max(1, len(proj)))()
        # rows_buffer: a caller-kept uint64 array for the per-batch counts (Context.pinned_array: written by the device)
        rows = rows_buffer[:k] if rows_buffer is not None else np.zeros(k, dtype=np.uint64)
        assert rows.dtype == np.uint64 and len(rows) == k
        nulls = np.zeros(k * max(1, len(proj)), dtype=np.int64) if want_nulls else None
        total = C.c_uint64()
        _check(load().rv_filter_project_batches(self.handle, arr, k, ncols, C.byref(p), pj, len(proj), out,
                                                rows.ctypes.data_as(_U64P),
                                                nulls.ctypes.data_as(C.POINTER(C.c_int64)) if want_nulls else None, C.byref(total)))
        outs = [DeviceColumn(self, C.c_void_p(out[i])) for i in range(len(proj))]
        return outs, rows, (nulls.reshape(k, max(1, len(proj))) if want_nulls else None), total.value

    def filter_project_chunked(self, cols: Sequence[DeviceColumn], chunk_rows: int, pred: Predicate, proj: Sequence[int],
                               want_nulls: bool = True, rows_buffer: Optional[np.ndarray] = None):
        """rv_filter_project_chunked: one resident table cut into chunk_rows-row RecordBatches the way dataframe_to_batches
        does, ONE launch.  Returns (outs, rows_per_batch, nulls, total) like filter_project_batches."""
        n = cols[0].length
        k = (n + chunk_rows - 1) // chunk_rows if chunk_rows > 0 else 0  # 0: the library reports the argument
        p, _keep = pred.as_struct()
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        out = (C.c_void_p * max(1, len(proj)))()
        # rows_buffer: a caller-kept uint64 array for the per-batch counts (e.g. Context.pinned_array: written by the device)
        rows = rows_buffer if rows_buffer is not None else np.zeros(max(1, k), dtype=np.uint64)
        assert rows.dtype == np.uint64 and len(rows) >= k
        nulls = np.zeros(max(1, k) * max(1, len(proj)), dtype=np.int64) if want_nulls else None
        total = C.c_uint64()
        _check(load().rv_filter_project_chunked(self.handle, _handles(cols), len(cols), chunk_rows, C.byref(p), pj, len(proj), out,
                                                rows.ctypes.data_as(_U64P), k,
                                                nulls.ctypes.data_as(C.POINTER(C.c_int64)) if want_nulls else None, C.byref(total)))
        outs = [DeviceColumn(self, C.c_void_p(out[i])) for i in range(len(proj))]
        return outs, rows[:k], (nulls.reshape(max(1, k), max(1, len(proj)))[:k] if want_nulls else None), total.value

    def window_begin(self, pred: Predicate, proj: Sequence[int], rows_buffer: np.ndarray, cols: Optional[Sequence[DeviceColumn]] = None,
                     chunk_rows: int = 0, handles=None):
        """rv_filter_project_chunked_begin (cols + chunk_rows) / rv_filter_project_batches_begin (handles = batch_handles(...)):
        queues a window's pass and returns finish(want_nulls=True) -> (outs, rows_per_batch, nulls, total).  rows_buffer: the
        caller's uint64 array for the per-batch counts (Context.pinned_array: written by the device beside the next window's
        pass); it must not be reused before finish."""
        p, _keep = pred.as_struct()
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        pend = C.c_void_p()
        assert rows_buffer.dtype == np.uint64
        if handles is None:
            n = cols[0].length
            k = (n + chunk_rows - 1) // chunk_rows if chunk_rows > 0 else 0
            assert len(rows_buffer) >= k
            hs = _handles(cols)
            _check(load().rv_filter_project_chunked_begin(self.handle, hs, len(cols), chunk_rows, C.byref(p), pj, len(proj),
                                                          rows_buffer.ctypes.data_as(_U64P), k, C.byref(pend)))
            keep = (hs, list(cols))
        else:
            arr, k, ncols = handles
            assert len(rows_buffer) >= k
            _check(load().rv_filter_project_batches_begin(self.handle, arr, k, ncols, C.byref(p), pj, len(proj),
                                                          rows_buffer.ctypes.data_as(_U64P), C.byref(pend)))
            keep = (arr,)
        nproj = len(proj)

        def finish(want_nulls: bool = True):
            out = (C.c_void_p * max(1, nproj))()
            nulls = np.zeros(max(1, k) * max(1, nproj), dtype=np.int64) if want_nulls else None
            total = C.c_uint64()
            _check(load().rv_filter_project_window_finish(self.handle, pend, out, nulls.ctypes.data_as(C.POINTER(C.c_int64)) if want_nulls else None,
                                                          C.byref(total)))
            outs = [DeviceColumn(self, C.c_void_p(out[i])) for i in range(nproj)]
            return outs, rows_buffer[:k], (nulls.reshape(max(1, k), max(1, nproj))[:k] if want_nulls else None), total.value
        finish._keep = (p, _keep, pj, keep)  # the argument structs live until finish
        return finish

    def slice_known(self, col: DeviceColumn, offset: int, length: int, null_count: int) -> DeviceColumn:
        out = C.c_void_p()
        _check(load().rv_slice_known(self.handle, col.handle, offset, length, null_count, C.byref(out)))
        return DeviceColumn(self, out)

    def filter_project_host(self, cols: Sequence[Column], pred: Predicate, proj: Sequence[int], chunk_rows: int = 0):
        """rv_filter_project_host: host columns in, device columns out (chunked, overlapped upload)."""
        p, _keep = pred.as_struct()
        hc = (RvColumn * len(cols))(*[c.as_struct() for c in cols])
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        out = (C.c_void_p * max(1, len(proj)))()
        rows = C.c_uint64()
        _check(load().rv_filter_project_host(self.handle, hc, len(cols), C.byref(p), pj, len(proj), chunk_rows, out,
                                             C.byref(rows)))
        return [DeviceColumn(self, C.c_void_p(out[i])) for i in range(len(proj))], rows.value

    def pinned_array(self, dtype, n: int) -> np.ndarray:
        """numpy array over rv_host_alloc (pinned) memory; freed when the context is closed."""
        nbytes = max(8, int(n) * np.dtype(dtype).itemsize)
        ptr = C.c_void_p()
        _check(load().rv_host_alloc(self.handle, nbytes, C.byref(ptr)))
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(ptr)
        buf = (C.c_uint8 * nbytes).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype, count=int(n))

    def filter_agg(self, cols: Sequence[DeviceColumn], pred: Predicate, agg_col: int):
        p, _keep = pred.as_struct()
        si, sf, cnt = C.c_int64(), C.c_double(), C.c_uint64()
        _check(load().rv_filter_agg(self.handle, _handles(cols), len(cols), C.byref(p), agg_col, C.byref(si),
                                    C.byref(sf), C.byref(cnt)))
        return si.value, sf.value, cnt.value


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(RV_COMM_ID_BYTES)
    _check(load().rv_comm_unique_id(buf))
    return buf.raw


class Comm:
    """RCCL communicator of one rank (rv_comm): the 16-byte {SUM, COUNT} all-reduce of config 5."""

    def __init__(self, ctx: Context, unique_id: bytes, world: int, rank: int):
        self.handle = None
        h = C.c_void_p()
        _check(load().rv_comm_create(ctx.handle, unique_id, world, rank, C.byref(h)))
        self.handle = h

    def allreduce_sum_count(self, total: int, count: int):
        s, c = C.c_int64(total), C.c_uint64(count)
        _check(load().rv_comm_allreduce_sum_count(self.handle, C.byref(s), C.byref(c)))
        return s.value, c.value

    def close(self):
        if self.handle is not None:
            load().rv_comm_destroy(self.handle)
            self.handle = None


class ShardedColumn:
    """One column of a row-range sharded table: shard r lives on rank r's device of a Group."""

    def __init__(self, group: "Group", handles):
        self.group, self.handles = group, handles  # ctypes array of n rv_dcolumn*

    def shard(self, rank: int) -> "DeviceColumn":
        """Borrowed view of rank r's shard (not freed by the returned object)."""
        d = DeviceColumn(self.group.context(rank), C.c_void_p(self.handles[rank]))
        d.free = lambda: None
        return d

    def free(self):
        if self.handles is not None and self.group.handle is not None:
            load().rv_group_free(self.group.handle, self.handles)
        self.handles = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GatherResult:
    """rv_gather: a query result gathered on the host in rank order (pinned memory, copied out by column())."""

    def __init__(self, handle, ncols: int, nranks: int):
        self.handle, self.ncols, self.nranks = handle, ncols, nranks

    def column(self, j: int) -> Column:
        v, nulls = RvColumn(), C.c_int64()
        _check(load().rv_gather_column(self.handle, j, C.byref(v), C.byref(nulls)))
        n = int(v.length)

        def copy(ptr, ctype, count, npdtype):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (count,)).copy() if count else np.zeros(0, npdtype)
        valid = copy(v.validity, C.c_uint8, (n + 7) // 8, np.uint8) if v.validity else None
        if v.dtype == RV_BOOLEAN:
            return Column(RV_BOOLEAN, copy(v.values, C.c_uint8, (n + 7) // 8, np.uint8), valid, 0, n)
        if v.dtype == RV_INT64:
            return Column(RV_INT64, copy(v.values, C.c_int64, n, np.int64), valid, 0, n)
        if v.dtype == RV_FLOAT64:
            return Column(RV_FLOAT64, copy(v.values, C.c_double, n, np.float64), valid, 0, n)
        if v.dtype == RV_STRING:
            return Column(RV_STRING, copy(v.values, C.c_uint8, int(v.data_bytes), np.uint8), valid, 0, n,
                          copy(v.offsets, C.c_int32, n + 1, np.int32))
        return Column(RV_NULL, np.zeros(0, dtype=np.uint8), None, 0, n)

    def null_count(self, j: int) -> int:
        v, nulls = RvColumn(), C.c_int64()
        _check(load().rv_gather_column(self.handle, j, C.byref(v), C.byref(nulls)))
        return nulls.value

    def values_view(self, j: int) -> np.ndarray:
        """The 8-byte values of column j WITHOUT a copy: a view of the result's pinned memory, valid until free()."""
        v = RvColumn()
        _check(load().rv_gather_column(self.handle, j, C.byref(v), None))
        assert v.dtype in (RV_INT64, RV_FLOAT64)
        ctype, n = (C.c_int64, int(v.length)) if v.dtype == RV_INT64 else (C.c_double, int(v.length))
        return np.ctypeslib.as_array(C.cast(v.values, C.POINTER(ctype)), (max(1, n),))[:n]

    def stats(self):
        rows = (C.c_uint64 * self.nranks)()
        f, g = C.c_double(), C.c_double()
        _check(load().rv_gather_stats(self.handle, rows, C.byref(f), C.byref(g)))
        return {"rank_rows": [int(x) for x in rows], "filter_ms": f.value, "gather_ms": g.value}

    def free(self):
        if self.handle is not None:
            load().rv_gather_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ResidentResult:
    """Outputs of rv_group_filter_project_resident: handles[r * nproj + j] lives on rank r's device."""

    def __init__(self, group: "Group", handles, nproj: int, rank_rows):
        self.group, self.handles, self.nproj, self.rank_rows = group, handles, nproj, rank_rows

    def column(self, rank: int, j: int) -> "DeviceColumn":
        d = DeviceColumn(self.group.context(rank), C.c_void_p(self.handles[rank * self.nproj + j]))
        d.free = lambda: None
        return d

    def free(self):
        if self.handles is not None and self.group.handle is not None:
            for r in range(self.group.n):
                for j in range(self.nproj):
                    h = self.handles[r * self.nproj + j]
                    if h:
                        load().rv_free(self.group.context(r).handle, h)
        self.handles = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Group:
    """rv_group: N contexts + N worker threads in ONE process; row-range shards, host gather in rank order."""

    def __init__(self, devices: Sequence[int]):
        self.handle = None
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(load().rv_group_create(arr, len(devices), C.byref(h)))
        self.handle = h
        self.n = len(devices)
        self._ctx = {}

    def context(self, rank: int) -> Context:
        """Borrowed Context of rank r (owned by the group)."""
        if rank not in self._ctx:
            c = Context.__new__(Context)
            c.handle = C.c_void_p(load().rv_group_ctx(self.handle, rank))
            c.close = lambda: None
            self._ctx[rank] = c
        return self._ctx[rank]

    def generate(self, spec: RvSynthSpec) -> ShardedColumn:
        out = (C.c_void_p * self.n)()
        _check(load().rv_group_generate(self.handle, C.byref(spec), out))
        return ShardedColumn(self, out)

    def upload(self, col: Column) -> ShardedColumn:
        s = col.as_struct()
        out = (C.c_void_p * self.n)()
        _check(load().rv_group_upload(self.handle, C.byref(s), out))
        return ShardedColumn(self, out)

    def _shards(self, cols: Sequence[ShardedColumn]):
        arr = (C.c_void_p * max(1, self.n * len(cols)))()
        for r in range(self.n):
            for c, col in enumerate(cols):
                arr[r * len(cols) + c] = col.handles[r]
        return arr

    def filter_project(self, cols: Sequence[ShardedColumn], pred: Predicate, proj: Sequence[int]):
        p, _keep = pred.as_struct()
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        res, rows = C.c_void_p(), C.c_uint64()
        _check(load().rv_group_filter_project(self.handle, self._shards(cols), len(cols), C.byref(p), pj, len(proj),
                                              C.byref(res), C.byref(rows)))
        return GatherResult(res, len(proj), self.n), rows.value

    def filter_project_host(self, cols: Sequence[Column], pred: Predicate, proj: Sequence[int], chunk_rows: int = 0):
        """rv_group_filter_project_host: a host table cut into row ranges, every range streamed through its device's own chunk
        pipeline at once, survivors gathered in rank order.  Returns (GatherResult, rows, upload GB/s per rank)."""
        p, _keep = pred.as_struct()
        hc = (RvColumn * len(cols))(*[c.as_struct() for c in cols])
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        out, rows = C.c_void_p(), C.c_uint64()
        gbs = (C.c_double * self.n)()
        _check(load().rv_group_filter_project_host(self.handle, hc, len(cols), C.byref(p), pj, len(proj), chunk_rows, C.byref(out), C.byref(rows), gbs))
        return GatherResult(out, len(proj), self.n), rows.value, list(gbs)

    def filter_project_resident(self, cols: Sequence[ShardedColumn], pred: Predicate, proj: Sequence[int]):
        """Phase 1 only: the outputs stay in HBM.  Returns (ResidentResult, total rows)."""
        p, _keep = pred.as_struct()
        pj = (C.c_uint32 * max(1, len(proj)))(*proj)
        outs = (C.c_void_p * max(1, self.n * len(proj)))()
        rank_rows, rows = (C.c_uint64 * self.n)(), C.c_uint64()
        _check(load().rv_group_filter_project_resident(self.handle, self._shards(cols), len(cols), C.byref(p), pj, len(proj),
                                                       outs, rank_rows, C.byref(rows)))
        return ResidentResult(self, outs, len(proj), [int(v) for v in rank_rows]), rows.value

    def gather(self, resident: "ResidentResult") -> GatherResult:
        res = C.c_void_p()
        _check(load().rv_group_gather(self.handle, resident.handles, resident.nproj, C.byref(res)))
        return GatherResult(res, resident.nproj, self.n)

    def filter_agg(self, cols: Sequence[ShardedColumn], pred: Predicate, agg_col: int):
        p, _keep = pred.as_struct()
        si, sf, cnt = C.c_int64(), C.c_double(), C.c_uint64()
        _check(load().rv_group_filter_agg(self.handle, self._shards(cols), len(cols), C.byref(p), agg_col,
                                          C.byref(si), C.byref(sf), C.byref(cnt)))
        return si.value, sf.value, cnt.value

    def stat(self, key: str) -> int:
        v = C.c_int64()
        _check(load().rv_group_stat(self.handle, key.encode(), C.byref(v)))
        return v.value

    def close(self):
        if self.handle is not None:
            for c in self._ctx.values():
                c.handle = None
            load().rv_group_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def device_count() -> int:
    """HIP devices this process sees (through the library: no second HIP runtime is pulled into the process)."""
    return int(load().rv_device_count())


def shard_range(n_rows: int, world: int, rank: int):
    b, e = C.c_uint64(), C.c_uint64()
    _check(load().rv_shard_range(n_rows, world, rank, C.byref(b), C.byref(e)))
    return b.value, e.value
