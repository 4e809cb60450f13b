"""Builds the test-only librccl stand-in (tests/native/fake_rccl.hip -> tests/native/_build/fake_rccl/librccl.so.1) and
describes how rank processes find it.  TEST INFRASTRUCTURE: the product never loads it; see the header of the source."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "fake_rccl.hip")
OUT_DIR = os.path.join(HERE, "native", "_build", "fake_rccl")
LIB = os.path.join(OUT_DIR, "librccl.so.1")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SYMBOLS = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclCommGetAsyncError", "ncclAllToAll", "ncclAllGather",
           "ncclGetErrorString"]       # polymath_amd/csrc/comm.hip:198-205


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wall", SRC, "-o", LIB, "-lrt", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for the librccl stand-in:\n%s\n%s" % (r.stdout[-4000:], r.stderr[-8000:]))
    return LIB


def rank_env(log_prefix=None, extra=None):
    """Environment of a rank process: the stand-in's directory FIRST in LD_LIBRARY_PATH, so that comm.hip's
    dlopen("librccl.so.1") resolves to it (libpolymath_hip.so carries a RUNPATH, which is searched after LD_LIBRARY_PATH)."""
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = OUT_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if log_prefix:
        env["PM_FAKE_RCCL_LOG"] = log_prefix
    env.update(extra or {})
    return env


if __name__ == "__main__":
    print(build(force=True))
