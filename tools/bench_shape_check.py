"""Times the reference's own bench circuit shape (benches/bench.rs:38-61: every padding witness holds the same value --
the skewed worst case of SURVEY.md §8d) at 2^20 - 100 constraints on one GPU.  Product code only."""
import sys, time
sys.path.insert(0, '.')
from polymath_amd import circuits as PC
from polymath_amd.polymath import Polymath, FIELDS
r = FIELDS["bls12_381"]["r"]
nc = (1 << 20) - 100
g = PC.SplitMix64(7)
circ = PC.BenchCircuit(g.fr(r), g.fr(r), nc, nc)      # benches/bench.rs:16-17: num_variables = num_constraints
pm = Polymath("bls12_381", "merlin", device=0)
t0 = time.time()
r1cs, inst, wit = pm._synthesize(circ)
print("synth", time.time() - t0, r1cs.m0, r1cs.mw, r1cs.nr)
x, z, r_a = g.fr(r), g.fr(r), [g.fr(r), g.fr(r)]
t0 = time.time(); pk = pm.setup((r1cs, inst, wit), x, z); print("setup", time.time() - t0, pk.n)
f = pm.field
xl, wl = f.fr_limbs(inst), f.fr_limbs(wit)
pm.prove_limbs(pk, inst, xl, wl, r_a)
pm.collect_timings = True
t0 = time.perf_counter(); p = pm.prove_limbs(pk, inst, xl, wl, r_a); dt = time.perf_counter() - t0
print("prove ms", dt * 1e3)
for i, tm in enumerate(pm.phase_timings): print(i + 1, {k: round(v, 2) for k, v in tm.items() if v})
print("proof", p.to_bytes().hex())   # acceptance (pairing verifier) lives in tests/test_gpu_parity.py::test_reference_bench_circuit_skew
