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
    for fn in ("prove_phase1", "prove_phase2", "prove_phase3", "get_or_upload", "with_thread_local", "generate", "info", "export_all", "adopt"):
        assert "pub fn " + fn in wrapper and fn in added


REFERENCE = "/root/reference"


def _patch_files(patch):
    return re.findall(r"^diff --git a/(\S+) b/", patch, flags=re.M)


def test_every_snark_method_of_the_reference_has_a_route_in_the_patch():
    """`Polymath<E, T>: SNARK<F>` (src/lib.rs:63-90) has three O(n) or larger entry points.  The reference-side binding must
    cover each: circuit_specific_setup -> generate_proving_key (generator.rs:24) through pm_pk_generate + pm_pk_export_bases
    (the dense CPU loop cannot finish at any BASELINE size), prove -> create_proof_with_assignment (prover.rs:66) through the
    three phases, verify -> stays on the CPU (O(m0) + one pairing product), said so in hip.rs.  The wrapper binds every C
    entry point those routes need, and the phases are tied to the key's lifetime (no use-after-free in safe code)."""
    wrapper = open(os.path.join(ROOT, "rust", "polymath-hip", "src", "lib.rs")).read()
    called = set(re.findall(r"sys::(pm_[a-z0-9_]+)\s*\(", wrapper))
    for needed in ("pm_pk_generate", "pm_pk_export_bases", "pm_pk_info", "pm_pk_load", "pm_prove_phase1", "pm_prove_phase2", "pm_prove_phase3"):
        assert needed in called, needed
    patch = open(os.path.join(ROOT, "rust", "reference-patch", "sigma0-polymath-hip.patch")).read()
    files = _patch_files(patch)
    assert {"src/generator.rs", "src/prover.rs", "src/hip.rs", "src/lib.rs", "Cargo.toml"} == set(files), files
    added = "\n".join(l[1:] for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++"))
    assert "unsafe" not in added.replace("forbid(unsafe_code)", "")
    # the generator hook: behind the feature, AFTER both trapdoor draws (same rng stream as the CPU path), G2 on the CPU
    gen = patch[patch.index("diff --git a/src/generator.rs"):patch.index("diff --git a/src/hip.rs")]
    gen_added = [l[1:] for l in gen.splitlines() if l.startswith("+") and not l.startswith("+++")]
    assert any('#[cfg(feature = "hip")]' in l for l in gen_added)
    assert any("generate_bases_hip" in l for l in gen_added) and any("adopt_resident_key_hip" in l for l in gen_added)
    ctx_before = gen[:gen.index("generate_bases_hip")]
    assert "let z: F = domain.sample_element_outside_domain(rng);" in ctx_before       # hooked after generator.rs:77
    assert any("x_g2: (g2 * &x).into()" in l for l in gen_added) and any("z_g2: (g2 * &z).into()" in l for l in gen_added)
    assert "GpuKey::generate::<E>" in added and "export_all::<E>" in added and "global_key_cache().adopt" in added
    assert "src/verifier.rs:19" in added and "stays on the CPU" in added
    # phases 2 and 3 only exist on the guard that borrows the key
    guard = wrapper[wrapper.index("pub struct ProofInFlight<'c, 'k>"):]
    guard = guard[:guard.index("\n// ----")]
    assert "ctx: &'c mut Context" in guard and "_key: &'k GpuKey" in guard
    assert "pub fn prove_phase2" in guard and "pub fn prove_phase3" in guard
    ctx_impl = wrapper[wrapper.index("impl Context {"):wrapper.index("pub struct ProofInFlight")]
    assert "pub fn prove_phase2" not in ctx_impl and "pub fn prove_phase3" not in ctx_impl
    assert "in_flight.prove_phase2" in added and "in_flight.prove_phase3" in added


def test_reference_patch_applies_to_the_reference(tmp_path):
    """With the reference checkout present (this container; never on the GPU box): `patch -p1 --dry-run` of the committed
    patch against an untouched copy succeeds with no fuzz and no rejects."""
    import shutil
    import subprocess
    if not os.path.isdir(REFERENCE) or not shutil.which("patch"):
        pytest.skip("no reference checkout / no patch(1) here")
    work = tmp_path / "ref"
    shutil.copytree(REFERENCE, work, ignore=shutil.ignore_patterns(".git", "target"))
    run = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "rust", "reference-patch", "sigma0-polymath-hip.patch")],
                         cwd=work, capture_output=True, text=True)
    assert run.returncode == 0 and "fuzz" not in run.stdout and "FAILED" not in run.stdout, run.stdout + run.stderr


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
