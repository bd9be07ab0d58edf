"""CPU suite: the C-ABI library loads, exports every symbol include/rivulus_gpu.h declares,
and fails loudly (no fallback) when no GPU is present.  No compute calls here."""
import ctypes
import os
import re

import pytest

from rivulus_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rivulus_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rv_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert set(_declared_symbols()) == set(capi.PROTOTYPES), "capi.PROTOTYPES must list exactly the header's functions"


@pytest.mark.parametrize("symbol", _declared_symbols())
def test_library_exports(symbol):
    lib = ctypes.CDLL(capi.LIB_PATH)
    assert getattr(lib, symbol) is not None


def test_abi_version_and_status_names():
    lib = capi.load()
    assert lib.rv_abi_version() == 4
    assert lib.rv_status_name(0) == b"RV_OK"
    assert lib.rv_status_name(2) == b"RV_ERR_LENGTH_MISMATCH"


def test_shard_ranges_are_word_aligned_and_cover():
    for n in [0, 1, 63, 64, 65, 1000, 10**9, 10**10 + 7]:
        for world in [1, 2, 3, 4, 8]:
            prev = 0
            for rank in range(world):
                b, e = capi.shard_range(n, world, rank)
                assert b == prev and b <= e <= n
                assert b % 64 == 0 or b == n
                prev = e
            assert prev == n
    with pytest.raises(capi.RvError):
        capi.shard_range(10, 2, 2)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.RvError) as e:
        capi.Context(0)
    assert "no CPU fallback" in str(e.value) or "RV_ERR_DEVICE" in str(e.value)


# ---- the boundary checked independently of the hand-written ctypes mirror -------------------------------------------
C_TO_CTYPES = {
    "int": ctypes.c_int, "uint32_t": ctypes.c_uint32, "uint64_t": ctypes.c_uint64, "int64_t": ctypes.c_int64,
    "size_t": ctypes.c_size_t, "double": ctypes.c_double, "float": ctypes.c_float,
    "rv_status": ctypes.c_int, "rv_cmp": ctypes.c_int, "rv_dtype": ctypes.c_int,
}


def _is_pointer(t):
    return t is ctypes.c_void_p or t is ctypes.c_char_p or hasattr(t, "contents") or (isinstance(t, type) and issubclass(t, ctypes._Pointer))


def test_prototypes_agree_with_the_header_argument_by_argument():
    """tools/gen_rust_ffi.py parses include/rivulus_gpu.h; the ctypes table must have the same arity, the same scalar
    types and a pointer wherever the header has one (so header, ctypes mirror and rust_shim/ffi.rs cannot drift)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_ffi
    protos = gen_rust_ffi.prototypes()
    assert len(protos) == len(capi.PROTOTYPES)
    for name, ret, args in protos:
        res, argtypes = capi.PROTOTYPES[name]
        assert len(args) == len(argtypes), name
        for i, (c, t) in enumerate(zip([ret] + args, [res] + list(argtypes))):
            c = c.strip()
            if c.endswith("*"):
                assert _is_pointer(t), f"{name} arg {i - 1}: header has {c}, binding has {t}"
            else:
                assert C_TO_CTYPES[c.replace("const ", "")] is t, f"{name} arg {i - 1}: header has {c}, binding has {t}"


def test_rust_declarations_are_in_step_with_the_header():
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(os.path.join(ROOT, "rust_shim", "ffi.rs")).read()
    for symbol in _declared_symbols():
        assert f"pub fn {symbol}(" in text


def test_struct_layouts_c99_and_ctypes(tmp_path):
    """tests/c/abi_check.c pins sizeof / offsetof with _Static_assert under gcc -std=c99; the ctypes mirror must have
    the same numbers."""
    import subprocess
    src = os.path.join(ROOT, "tests", "c", "abi_check.c")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-DRV_ABI_LAYOUT_ONLY", "-c", src, "-o",
                        str(tmp_path / "abi_check.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert ctypes.sizeof(capi.RvColumn) == 56 and capi.RvColumn.offsets.offset == 40 and capi.RvColumn.data_bytes.offset == 48
    assert ctypes.sizeof(capi.RvTerm) == 32 and capi.RvTerm.lit.offset == 16
    assert ctypes.sizeof(capi.RvPredicate) == 32 and capi.RvPredicate.expr.offset == 16 and capi.RvPredicate.n_expr.offset == 24
    assert ctypes.sizeof(capi.RvSynthSpec) == 80 and capi.RvSynthSpec.validity_seed.offset == 48
    assert ctypes.sizeof(capi.RvColumnInfo) == 48 and capi.RvColumnInfo.null_count.offset == 32


def test_product_path_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under rivulus_amd/ or tools/ may import, link or open it, and bench.py
    may only do so inside cpu_baseline() (the reported CPU leg)."""
    import ast
    hits = []
    for base in ("rivulus_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dirpath.split(os.sep):
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")) or f == "host_tests":
                    continue
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("pyoracle", "liboracle", "oracle/", "oracle_"):
                    if needle in text and not (base == "rivulus_amd" and f == "Makefile" and "tests/cpp" in text):
                        hits.append((os.path.join(dirpath, f), needle))
    # the host-layer Makefile builds the TEST binary (tests/cpp/host_tests.cpp), which includes the oracle as the checker
    hits = [h for h in hits if not h[0].endswith(os.path.join("rivulus_amd", "host", "Makefile"))]
    assert not hits, hits
    header = open(os.path.join(ROOT, "include", "rivulus_gpu.h")).read()
    assert not [ln for ln in header.splitlines() if ln.lstrip().startswith("#include") and "oracle" in ln]
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name != "cpu_baseline":
            src = ast.get_source_segment(open(os.path.join(ROOT, "bench.py")).read(), node)
            assert "pyoracle" not in src and "oracle" not in src.replace("the oracle", ""), node.name


def test_every_threshold_of_the_table_is_read_by_the_launch_code():
    """rivulus_amd/csrc/thresholds.hpp is THE table of switches between kernels / paths: a constant nobody reads (or a literal that crept back
    into the launch code next to it) is drift.  Every constant must be used by a .hip unit, and tests/test_paths_gpu.py must name the ones
    that decide which kernel runs."""
    import glob
    import re
    csrc = os.path.join(ROOT, "rivulus_amd", "csrc")
    names = re.findall(r"\b(k[A-Z]\w+)\s*=", open(os.path.join(csrc, "thresholds.hpp")).read())
    assert len(names) >= 25, names
    code = "".join(open(f).read() for f in glob.glob(os.path.join(csrc, "*.hip")))
    unused = [n for n in names if f"rvt::{n}" not in code]
    assert not unused, f"constants of thresholds.hpp that no unit reads: {unused}"
    paths = open(os.path.join(ROOT, "tests", "test_paths_gpu.py")).read()
    for n in ("kDirectFromOneColumn", "kDirectFromTwoProjected", "kDirectFromThreeProjected", "kDirectFromOneProjectedOfSeveral", "kDirectFromTwoProjectedNullable",
              "kDeferPlainUpTo", "kMaskPathPlainUpTo", "kSampleFromRows", "kRangesFromRows"):
        assert n in names and n in paths, n
