"""CPU: the C-ABI library builds, loads, and exports exactly the symbols include/polymath_hip.h
declares.  No compute calls (no GPU here); on a GPU-less box context creation must fail loudly."""
import ctypes as ct
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "polymath_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from polymath_amd import api, build
    build.build_library(verbose=False)
    L = api.load_library()
    syms = declared_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert sorted(api.EXPORTS) == syms


def test_no_cpu_fallback_without_gpu():
    from polymath_amd import api
    L = api.load_library()
    if L.pm_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(api.PolymathError) as e:
        api.Context(0)
    assert e.value.status == 7  # PM_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "polymath_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("import oracle", "from oracle", "oracle/", "oracle.", "libpolymath_oracle", "po_"):
                    assert needle not in src.replace("no oracle import", ""), (os.path.join(dirpath, f), needle)
