"""The drop-in boundary: libconga_hip.so loads without a GPU and exports every symbol that
include/conga_hip.h declares; with no device it refuses to create a context (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "conga_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(conga_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(capi):
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(capi.EXPORTS) == names


def test_abi_version_and_struct_sizes(capi):
    lib = capi.load()
    assert lib.conga_abi_version() == 9
    assert capi.RESULT_DTYPE.itemsize == 64
    assert ctypes.sizeof(capi.Opts) == 32
    assert ctypes.sizeof(capi.BgzfBlock) == 24 and ctypes.sizeof(capi.BamSegment) == 24


def test_strerror(capi):
    lib = capi.load()
    assert lib.conga_strerror(0) == b"ok"
    assert b"sorted" in lib.conga_strerror(capi.CONGA_ERR_UNSORTED)
    assert b"no CPU path" in lib.conga_strerror(capi.CONGA_ERR_NO_DEVICE)


def test_no_device_means_no_context(capi):
    lib = capi.load()
    if lib.conga_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.CongaError) as e:
        capi.Context(device=0)
    assert e.value.status == capi.CONGA_ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py may use oracle/."""
    dirs = [os.path.join(ROOT, "conga_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "include")]
    for base, _, files in (x for d in dirs for x in os.walk(d)):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", ".c")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "liboracle" not in text and "conga_oracle" not in text, os.path.join(base, f)
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), os.path.join(base, f)
