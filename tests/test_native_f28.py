"""CPU: the reduced-radix (28-bit limb) field / mixed-add code of csrc/fq28.cuh, compiled for the HOST
with g++ (the templates are plain C++), against the dense 32-bit-limb path that the GPU parity tests pin
to the oracle: products, lazy add/sub chains, squares of loose operands, mixed-add chains with the
exceptional (doubling / cancellation) cases, radix conversions -- both curves."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_f28_host_selftest(tmp_path):
    exe = str(tmp_path / "f28_selftest")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "f28_selftest.cpp")])
    out = subprocess.run([exe, "1500"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bls12_381: 0 failures" in out.stdout and "bn254: 0 failures" in out.stdout
