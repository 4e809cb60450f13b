"""Builds libpolymath_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m polymath_amd.build [--force]

Objects go to polymath_amd/_build/, the library to polymath_amd/libpolymath_hip.so (git-ignored,
but shipped to the GPU box by gpurun)."""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libpolymath_hip.so")
SOURCES = ["api.hip", "ntt.hip", "msm.hip", "msm_reduce.hip", "prove.hip", "setup.hip", "host_prove.hip", "synth.hip", "comm.hip", "prove_sharded.hip", "selftest.hip"]
HEADERS = ["field.cuh", "ec.cuh", "fq28.cuh", "constants.cuh", "internal.h", os.path.join("..", "host", "hashes.hpp"), os.path.join("..", "host", "polymath.hpp"), os.path.join("..", "host", "pairing.hpp"), os.path.join("..", "host", "wire.hpp"), os.path.join("..", "host", "layout.hpp"), os.path.join("..", "host", "rng.hpp"), "comm.h", "prove_common.cuh",
           os.path.join("..", "..", "include", "polymath_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]
FLAGS += os.environ.get("PM_BUILD_FLAGS", "").split()     # same-box A/B builds of compile-time variants (tools/README.md)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if not _newer(obj, deps):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout[-4000:], r.stderr[-8000:]))
    return obj, True


def build_library(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with cf.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout[-4000:], r.stderr[-8000:]))
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
