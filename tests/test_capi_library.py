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
    assert lib.rv_abi_version() == 1
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
