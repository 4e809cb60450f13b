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


RUST_SYS = os.path.join(ROOT, "rust", "polymath-hip-sys", "src", "lib.rs")


def _split_args(argtext):
    """top-level comma split (function-pointer arguments carry commas of their own)"""
    out, depth, cur = [], 0, ""
    for ch in argtext:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _c_class(arg):
    """pointer / float / callback / the integer type's width class of one C parameter"""
    if "(*" in arg or "pm_combine_fn" in arg:
        return "callback"
    if "*" in arg or "[" in arg:
        return "ptr"
    t = " ".join(arg.split()[:-1])
    return {"double": "f64", "int": "i32", "unsigned": "u32", "uint64_t": "u64", "size_t": "usize", "long long": "longlong", "long": "long"}[t]


def _rust_class(arg):
    t = arg.split(":", 1)[1].strip()
    if t.startswith("Option<") or t == "pm_combine_fn":
        return "callback"
    if t.startswith("*"):
        return "ptr"
    return {"f64": "f64", "i32": "i32", "u32": "u32", "u64": "u64", "usize": "usize", "c_longlong": "longlong", "c_long": "long"}[t]


def c_prototypes():
    text = open(os.path.join(ROOT, "include", "polymath_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(pm_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        argv = [] if args in ("", "void") else _split_args(args)
        rc = "ptr" if "*" in ret else {"int": "i32", "void": "void", "double": "f64", "size_t": "usize"}[ret]
        protos[name] = (rc, [_c_class(a) for a in argv])
    return protos


def rust_prototypes():
    text = re.sub(r"//[^\n]*", "", open(RUST_SYS).read())
    block = text[text.index('extern "C" {'):]
    protos = {}
    for m in re.finditer(r"pub fn (pm_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2).strip(), (m.group(3) or "void").strip()
        rc = "ptr" if ret.startswith("*") else {"i32": "i32", "void": "void", "f64": "f64", "usize": "usize"}[ret]
        protos[name] = (rc, [_rust_class(a) for a in _split_args(args)])
    return protos


def test_rust_sys_crate_binds_every_symbol_with_the_headers_signature():
    """rust/polymath-hip-sys/src/lib.rs (the `extern "C"` block a maintainer of the reference links against; no cargo in this
    image, so it is checked HERE): exactly the header's entry points == the library's exports, and for each of them the same
    number of arguments, the same argument classes (pointer / callback / i32 / u32 / u64 / usize / long / long long / double)
    in the same order and the same return class.  The enum values quoted there are the header's."""
    c, r = c_prototypes(), rust_prototypes()
    assert sorted(c) == declared_symbols() == sorted(r)
    for name in sorted(c):
        assert r[name] == c[name], (name, c[name], r[name])
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "polymath_hip.h")).read(), flags=re.S)
    enums = {k: int(v) for k, v in re.findall(r"\b(PM_[A-Z0-9_]+)\s*=\s*(\d+)", header)}
    consts = {k: int(v) for k, v in re.findall(r"pub const (PM_[A-Z0-9_]+): [a-z0-9_]+ = (\d+);", open(RUST_SYS).read())}
    assert consts.pop("PM_FR_LIMBS") == 4
    assert consts == enums, sorted(set(consts.items()) ^ set(enums.items()))


def test_rust_wrapper_and_reference_patch_are_consistent():
    """The safe wrapper only calls functions the -sys crate declares; the patch to the reference only uses what the wrapper
    exports, touches create_proof_with_assignment behind `#[cfg(feature = "hip")]` and leaves `forbid(unsafe_code)` alone."""
    sysfns = set(rust_prototypes())
    wrapper = open(os.path.join(ROOT, "rust", "polymath-hip", "src", "lib.rs")).read()
    called = set(re.findall(r"sys::(pm_[a-z0-9_]+)\s*\(", wrapper))
    assert called and called <= sysfns, called - sysfns
    for needed in ("pm_pk_load", "pm_prove_phase1", "pm_prove_phase2", "pm_prove_phase3", "pm_pk_free", "pm_ctx_create", "pm_ctx_destroy"):
        assert needed in called
    patch = open(os.path.join(ROOT, "rust", "reference-patch", "sigma0-polymath-hip.patch")).read()
    added = "\n".join(l[1:] for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++"))
    assert "unsafe" not in added.replace("forbid(unsafe_code)", "")
    assert '#[cfg(feature = "hip")]' in added and "create_proof_with_assignment_hip" in added
    exported = set(re.findall(r"pub (?:fn|struct|enum) ([A-Za-z_][A-Za-z0-9_]*)", wrapper))
    used = set(re.findall(r"polymath_hip::\{([^}]*)\}", added)[0].replace(" ", "").split(",")) | set(re.findall(r"polymath_hip::([a-z_]+)", added))
    assert used <= exported, used - exported
    for fn in ("prove_phase1", "prove_phase2", "prove_phase3", "get_or_upload", "with_thread_local"):
        assert "pub fn " + fn in wrapper and fn in added


def test_integration_md_quotes_the_rust_sources():
    """INTEGRATION.md points at the Rust sources instead of carrying a second copy; the counts quoted in its prose are real."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for path in ("rust/polymath-hip-sys/src/lib.rs", "rust/polymath-hip/src/lib.rs", "rust/reference-patch/sigma0-polymath-hip.patch", "rust/check.sh"):
        assert path in text and os.path.exists(os.path.join(ROOT, path)), path
    syms = declared_symbols()
    for m in re.finditer(r"(\d+) (?:entry points|symbols)", text):
        assert int(m.group(1)) == len(syms), m.group(0)
    quoted = set(re.findall(r"pub fn (pm_[a-z0-9_]+)", text))
    assert quoted <= set(syms)


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
