"""The C-ABI library loads and exports every symbol include/sleap_amd.h declares (no GPU needed)."""
import os
import re

from sleap_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sleap_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", src)))


import pytest


@pytest.mark.parametrize("dtype", _lib.DTYPES)
def test_all_declared_symbols_exported_and_bound(dtype):
    """Both builds of the library (bf16 and fp16 storage, csrc/bf16.h) export the same ABI."""
    names = _declared()
    assert len(names) >= 15
    h = _lib.lib(dtype)
    for n in names:
        assert hasattr(h, n), f"{n} declared in sleap_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert h.sa_abi_version() == 1
    assert h.sa_storage_dtype().decode() == dtype


def test_argument_validation_without_gpu():
    h = _lib.lib()
    # bad shapes are rejected before any HIP call
    rc = h.sa_find_local_peaks(None, None, 0, 4, 4, 1, 0.2, 0, 5, 1.0, 16, None, None, None, None, None, None, 0, None)
    assert rc == -1
    assert b"bad shape" in h.sa_last_error()
    assert h.sa_find_local_peaks_workspace(4, 100) == 1600
