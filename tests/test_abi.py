"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/tezip_hip.h declares; without a GPU the product path fails
loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from tezip_amd import build
    build.build()
    from tezip_amd import _lib
    return _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "tezip_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tz_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 30
    dll = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), "libtezip_hip.so does not export %s" % n
    assert sorted(lib.EXPORTS) == names, "tezip_amd/_lib.py binds a different set than the header declares"


def test_version_and_strerror(lib):
    L = lib.load()
    assert L.tz_version() == 101
    # a product build names no diagnostic switch (a library built with TEZIP_DEFINES is refused by _lib.load())
    assert L.tz_build_info() == b"tezip_hip 101 gfx950 defines:" and lib.diagnostic_defines() == []
    assert L.tz_strerror(0) == b"ok" and L.tz_strerror(-2) == b"no HIP device"


def test_table_builder_is_host_only_and_matches_reference_order(lib):
    # tz_build_table needs no GPU: compress.py:352-361 ordering incl. the tie-break
    L = lib.load()
    hist = np.zeros(2111, np.uint64)
    hist[[1600, 1595, 1598, 1605]] = [4, 2, 1, 1]
    table = np.zeros(1021, np.int16)
    n = ctypes.c_int(0)
    assert L.tz_build_table(hist.ctypes.data, 2111, table.ctypes.data, ctypes.byref(n)) == 0
    assert table[: n.value].tolist() == [1600, 1595, 1598, 1605]


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.TezipError):
        lib.Context(0)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "tezip_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libtzoracle" not in src and "tz_oracle" not in src.replace("oracle/tz_oracle.c", ""), f
