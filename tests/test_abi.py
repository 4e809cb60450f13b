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


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/polymath_hip.h must compile as C99 (what cgo / bindgen / a JNI stub would feed it to)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "polymath_hip.h"\nint main(void) { pm_ctx *c = 0; (void)c; return PM_NUM_OPTIONS > 0 && PM_OK == 0 ? 0 : 1; }\n')
    run = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                          str(tmp_path / "t.o")], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr


def test_integration_md_binds_every_symbol():
    """INTEGRATION.md's `extern "C"` block (what a Rust maintainer would paste) names exactly the header's entry points, and the
    counts quoted in its prose are the real one."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    bound = sorted(set(re.findall(r"pub fn (pm_[a-z0-9_]+)", text)))
    syms = declared_symbols()
    assert bound == syms
    for m in re.finditer(r"(\d+) (?:entry points|symbols)", text):
        assert int(m.group(1)) == len(syms), m.group(0)


def test_no_getenv_on_the_proving_path():
    """Modes are per-context options (pm_ctx_set_option): the library reads the environment only for a new context's
    defaults and three process-wide developer aids -- never per proof (at most 8 sites)."""
    sites = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "polymath_amd")):
        for f in files:
            if f.endswith((".hip", ".cuh", ".h", ".hpp")):
                for ln, line in enumerate(open(os.path.join(dirpath, f), errors="ignore"), 1):
                    if "getenv(" in line:
                        sites.append((f, ln))
    assert len(sites) <= 8, sites
    assert {f for f, _ in sites} <= {"api.hip", "comm.hip", "internal.h"}, sites


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
