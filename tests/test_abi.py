"""The C-ABI library loads and exports every symbol include/kami_hip.h declares (CPU only:
no compute calls)."""
import ctypes as C
import os
import re

import pytest

from kami_amd import _lib as L
from kami_amd import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "kami_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    lib = L.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in kami_hip.h but not exported"
    assert set(names) == set(L.SYMBOLS), "ctypes table and header disagree"


def test_search_header_symbols_all_exported():
    """include/kami_search.h (host search row) against libkamisearch.so and its ctypes table."""
    from kami_amd import search as S
    text = open(os.path.join(ROOT, "include", "kami_search.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(ks_[a-z0-9_]+)\s*\(", text)))
    lib = S.load()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in kami_search.h but not exported"
    assert set(names) == set(S.SYMBOLS), "ctypes table and header disagree"
    assert C.sizeof(S.PoolConfig) == 64 and C.sizeof(S.Record) == 80 + 8 + 96 * 2 + 96 * 4


def test_struct_layouts():
    assert C.sizeof(L.Config) == 64
    assert L.BOARD_DTYPE.itemsize == 80
    assert L.BOARD_DTYPE.fields["ply"][1] == 64 and L.BOARD_DTYPE.fields["ctm"][1] == 72


def test_weight_count_agrees():
    lib = L.load()
    for F, Cc, R in [(30, 64, 6), (119, 64, 6), (119, 128, 10), (30, 8, 0), (119, 256, 20)]:
        assert lib.kh_weight_count(F, Cc, R) == W.weight_count(F, Cc, R)


def test_version_and_error_strings():
    lib = L.load()
    assert b"gfx950" in lib.kh_version()
    assert isinstance(L.last_error(), str)


def test_no_cpu_fallback():
    """Without a GPU the engine must refuse to exist rather than compute on the host."""
    lib = L.load()
    if lib.kh_device_count() > 0:
        pytest.skip("GPU present")
    from kami_amd import NN, KamiError
    with pytest.raises(KamiError) as ei:
        NN(filters=8, residuals=1)
    assert ei.value.status == L.KH_ERR_NO_DEVICE


def test_invalid_config_rejected():
    lib = L.load()
    cfg = L.Config(width=9, height=8, features=30, psize=4672, filters=8, residuals=1)
    h = C.c_void_p()
    assert lib.kh_create(C.byref(cfg), C.byref(h)) == L.KH_ERR_INVALID
    assert "8x8" in L.last_error()


def test_product_does_not_import_oracle():
    """The product package must never route through oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "kami_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                s = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in s and "kami_oracle" not in s and "oracle/" not in s, f
