//! Pin kit: the REAL reference (sigma0-polymath on arkworks) as a fixture generator and judge for this repository.
//!
//! NEVER COMPILED IN THE BUILD IMAGE (no cargo there; see Cargo.toml).  Written against the reference's public API
//! (src/lib.rs:52-91, src/data_structures.rs:9-73) and arkworks' algebra HEAD, which the reference patches in.
//!
//!   emit <dir>            reference setup + prove on the shapes of tests/dummy.rs:37-80 and tests/mimc.rs:145-227, with the
//!                         same seeds (`StdRng::seed_from_u64(test_rng().next_u64())`), plus two product chains with 11 and 0 public
//!                         inputs (m0 = 12 and 1: the reference's tests only have m0 = 2), for the three transcripts; writes
//!                         <dir>/ref_dummy.json, ref_inputs11.json, ref_inputs0.json and ref_mimc322.json in the schema of tests/golden/proofs.json:
//!                         R1CS, assignment, the trapdoors x, z and the prover's r_a (recovered by replaying the draws on a
//!                         clone of the RNG and CHECKED against [x]_2, [z]_2 and the proof itself), proof bytes, vk bytes,
//!                         and (dummy only) the whole serialised ProvingKey.
//!   verify <proofs.json>  this repository's golden proofs -> the reference's Polymath::verify (lib.rs:80-90).
//!   bench <out.json>      ark-ec msm_unchecked and ark-poly fft at 2^20 .. 2^24 (benches/bench.rs:2 thread convention).
use std::{fmt::Write as _, fs, time::Instant};

use ark_bls12_381::{Bls12_381, Fr, G1Affine, G1Projective, G2Projective};
use ark_crypto_primitives::snark::{CircuitSpecificSetupSNARK, SNARK};
use ark_ec::{AffineRepr, CurveGroup, PrimeGroup, VariableBaseMSM};
use ark_ff::{BigInteger, Field, PrimeField, UniformRand};   // Field: square()
use ark_poly::{EvaluationDomain, Radix2EvaluationDomain};
use ark_relations::{
    lc,
    r1cs::{ConstraintSynthesizer, ConstraintSystem, ConstraintSystemRef, OptimizationGoal, SynthesisError, SynthesisMode, Variable},
};
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize};
use ark_std::{
    rand::{rngs::StdRng, Rng, RngCore, SeedableRng},
    test_rng,
};
use sigma0_polymath::{
    blake3::Blake3Transcript, keccak256::Keccak256Transcript, merlin::MerlinFieldTranscript, PairingVK, Polymath, Proof, ProvingKey, Transcript,
    VerifyingKey,
};

type E = Bls12_381;

// ------------------------------------------------------------------------------------------------ circuits
/// a * b = c with c public: the shape of tests/dummy.rs:21-35.
#[derive(Clone)]
struct Product {
    a: Option<Fr>,
    b: Option<Fr>,
}
impl ConstraintSynthesizer<Fr> for Product {
    fn generate_constraints(self, cs: ConstraintSystemRef<Fr>) -> Result<(), SynthesisError> {
        let a = cs.new_witness_variable(|| self.a.ok_or(SynthesisError::AssignmentMissing))?;
        let b = cs.new_witness_variable(|| self.b.ok_or(SynthesisError::AssignmentMissing))?;
        let c = cs.new_input_variable(|| Ok(self.a.ok_or(SynthesisError::AssignmentMissing)? * self.b.ok_or(SynthesisError::AssignmentMissing)?))?;
        cs.enforce_constraint(lc!() + a, lc!() + b, lc!() + c)
    }
}

/// A chain of products v_i * v_{i+1} = p_i whose first `public` outputs are PUBLIC inputs and the rest witnesses: m0 = public + 1.
/// The reference's own tests only ever have one public input (m0 = 2); its SAP matrices have arms for every m0
/// (common.rs:77-97, 138-207) and the witness-only part of u is cut at column m0 (prover.rs:156-166).  public = 11 gives m0 = 12
/// (2 m0 > 16: this repository's HIP path then runs a fifth transform), public = 0 gives m0 = 1 (no public input at all).
#[derive(Clone)]
struct Chain {
    vals: Vec<Option<Fr>>,
    public: usize,
}
impl ConstraintSynthesizer<Fr> for Chain {
    fn generate_constraints(self, cs: ConstraintSystemRef<Fr>) -> Result<(), SynthesisError> {
        let missing = || SynthesisError::AssignmentMissing;
        let mut vars = Vec::with_capacity(self.vals.len());
        for v in &self.vals {
            let v = *v;
            vars.push(cs.new_witness_variable(|| v.ok_or(missing()))?);
        }
        for i in 0..self.vals.len() - 1 {
            let prod = match (self.vals[i], self.vals[i + 1]) {
                (Some(a), Some(b)) => Some(a * b),
                _ => None,
            };
            let out = if i < self.public { cs.new_input_variable(|| prod.ok_or(missing()))? } else { cs.new_witness_variable(|| prod.ok_or(missing()))? };
            cs.enforce_constraint(lc!() + vars[i], lc!() + vars[i + 1], lc!() + out)?;
        }
        Ok(())
    }
}

const ROUNDS: usize = 322;

/// LongsightF322p3 preimage knowledge (eprint 2016/492), the gate list of tests/mimc.rs:74-143: per round
///   t = (xL + C)^2            (xL + C) * (xL + C) = t
///   xL' = xR + t (xL + C)     t * (xL + C) = xL' - xR         xR' = xL
/// all variables witnesses except the last xL', which is the public image.
#[derive(Clone)]
struct Longsight {
    left: Option<Fr>,
    right: Option<Fr>,
    round_constants: Vec<Fr>,
}
fn longsight(mut l: Fr, mut r: Fr, k: &[Fr]) -> Fr {
    for c in k {
        let s = l + c;
        let next = r + s.square() * s;
        r = l;
        l = next;
    }
    l
}
impl ConstraintSynthesizer<Fr> for Longsight {
    fn generate_constraints(self, cs: ConstraintSystemRef<Fr>) -> Result<(), SynthesisError> {
        assert_eq!(self.round_constants.len(), ROUNDS);
        let missing = || SynthesisError::AssignmentMissing;
        let (mut lv, mut rv) = (self.left, self.right);
        let mut l = cs.new_witness_variable(|| lv.ok_or(missing()))?;
        let mut r = cs.new_witness_variable(|| rv.ok_or(missing()))?;
        for (i, k) in self.round_constants.iter().enumerate() {
            let sum = lv.map(|v| v + k);
            let tv = sum.map(|s| s.square());
            let t = cs.new_witness_variable(|| tv.ok_or(missing()))?;
            cs.enforce_constraint(lc!() + l + (*k, Variable::One), lc!() + l + (*k, Variable::One), lc!() + t)?;
            let nv = match (sum, tv, rv) {
                (Some(s), Some(t2), Some(rr)) => Some(rr + t2 * s),
                _ => None,
            };
            let next = if i + 1 == ROUNDS { cs.new_input_variable(|| nv.ok_or(missing()))? } else { cs.new_witness_variable(|| nv.ok_or(missing()))? };
            cs.enforce_constraint(lc!() + t, lc!() + l + (*k, Variable::One), lc!() + next - r)?;
            r = l;
            rv = lv;
            l = next;
            lv = nv;
        }
        Ok(())
    }
}

// ------------------------------------------------------------------------------------------------ JSON helpers
fn fq_hex<F: PrimeField>(v: &F) -> String {
    let h = hex::encode(v.into_bigint().to_bytes_be());
    let t = h.trim_start_matches('0');
    format!("\"0x{}\"", if t.is_empty() { "0" } else { t })
}
fn g1_json(p: &G1Affine) -> String {
    // `xy()` hands out references in ark-ec 0.4 and values on algebra HEAD: `to_owned()` gives a plain Fq either way
    match p.xy() {
        Some((x, y)) => {
            let (x, y): (ark_bls12_381::Fq, ark_bls12_381::Fq) = (x.to_owned(), y.to_owned());
            format!("[{}, {}]", fq_hex(&x), fq_hex(&y))
        }
        None => "null".to_string(),
    }
}
fn bytes_hex<T: CanonicalSerialize>(t: &T) -> String {
    let mut b = Vec::new();
    t.serialize_compressed(&mut b).expect("serialize_compressed");
    hex::encode(b)
}
fn matrix_json(m: &[Vec<(Fr, usize)>]) -> String {
    let rows: Vec<String> = m.iter().map(|row| format!("[{}]", row.iter().map(|(v, j)| format!("[{}, {}]", fq_hex(v), j)).collect::<Vec<_>>().join(", "))).collect();
    format!("[{}]", rows.join(", "))
}
fn list_json(v: &[Fr]) -> String {
    format!("[{}]", v.iter().map(fq_hex).collect::<Vec<_>>().join(", "))
}
fn points_json(v: &[G1Affine]) -> String {
    format!("[{}]", v.iter().map(g1_json).collect::<Vec<_>>().join(", "))
}

// ------------------------------------------------------------------------------------------------ emit
/// The assignment the prover will see (prover.rs:33-52): synthesise in Prove mode, as create_proof does.
fn assignment<C: ConstraintSynthesizer<Fr>>(c: C) -> (Vec<Fr>, Vec<Fr>) {
    let cs = ConstraintSystem::<Fr>::new_ref();
    cs.set_optimization_goal(OptimizationGoal::Constraints);
    cs.set_mode(SynthesisMode::Prove { construct_matrices: false });
    c.generate_constraints(cs.clone()).expect("synthesis");
    cs.finalize();
    let inner = cs.borrow().unwrap();
    (inner.instance_assignment.clone(), inner.witness_assignment.clone())
}

struct Run {
    transcript: &'static str,
    proof: Proof<E>,
    r_a: [Fr; 2],
}

/// One reference run with transcript T on a fresh RNG seeded like the reference's tests.  `prelude` draws whatever the
/// test draws before setup (mimc.rs:155: the round constants) and returns the circuits for setup and for prove.
fn run_one<T, C, P>(name: &'static str, prelude: P) -> (ProvingKey<E>, VerifyingKey<E>, Fr, Fr, Run, Vec<Fr>, Vec<Fr>)
where
    T: Transcript<Challenge = Fr>,
    C: ConstraintSynthesizer<Fr> + Clone,
    P: Fn(&mut StdRng) -> (C, Box<dyn Fn(&mut StdRng) -> C>),
{
    let mut rng = StdRng::seed_from_u64(test_rng().next_u64()); // dummy.rs:44, mimc.rs:152
    let (blank, make) = prelude(&mut rng);
    // the generator draws x then z (generator.rs:72,77): replay them on a clone, check them against the key below
    let mut replay = rng.clone();
    let (pk, vk) = Polymath::<E, T>::setup(blank, &mut rng).expect("setup");
    let domain = Radix2EvaluationDomain::<Fr>::new(vk.n as usize).expect("domain");
    let x: Fr = domain.sample_element_outside_domain(&mut replay);
    let z: Fr = domain.sample_element_outside_domain(&mut replay);
    assert_eq!(vk.e.x_g2, (G2Projective::generator() * x).into_affine(), "replayed x is not the key's trapdoor");
    assert_eq!(vk.e.z_g2, (G2Projective::generator() * z).into_affine(), "replayed z is not the key's trapdoor");
    let circuit = make(&mut rng); // the test's own draws for the witness (dummy.rs:56-57, mimc.rs:186-187)
    let (instance, witness) = assignment(circuit.clone());
    // the prover draws r_a's constant term, then the linear one (prover.rs:110)
    let mut replay = rng.clone();
    let r_a = [Fr::rand(&mut replay), Fr::rand(&mut replay)];
    let proof = Polymath::<E, T>::prove(&pk, circuit, &mut rng).expect("prove");
    assert!(Polymath::<E, T>::verify(&vk, &instance[1..], &proof).expect("verify"), "the reference rejects its own proof");
    (pk, vk, x, z, Run { transcript: name, proof, r_a }, instance, witness)
}

fn emit_case<C, P>(out_dir: &str, case: &str, with_pk_bytes: bool, prelude: P)
where
    C: ConstraintSynthesizer<Fr> + Clone,
    P: Fn(&mut StdRng) -> (C, Box<dyn Fn(&mut StdRng) -> C>) + Copy,
{
    let (pk, vk, x, z, m, instance, witness) = run_one::<MerlinFieldTranscript<Fr>, C, P>("merlin", prelude);
    let (_, _, x2, z2, k, i2, w2) = run_one::<Keccak256Transcript<Fr>, C, P>("keccak256", prelude);
    let (_, _, x3, z3, b, i3, w3) = run_one::<Blake3Transcript<Fr>, C, P>("blake3", prelude);
    // the seeds are equal, so everything but the challenges must be (the transcripts only enter after [a]_1, [c]_1)
    assert!(x == x2 && x == x3 && z == z2 && z == z3 && instance == i2 && instance == i3 && witness == w2 && witness == w3);
    assert!(m.r_a == k.r_a && m.r_a == b.r_a && m.proof.a_g1 == k.proof.a_g1 && m.proof.c_g1 == b.proof.c_g1);
    let sap = &pk.sap_matrices;
    let mut j = String::new();
    write!(j, "{{\"name\": \"ref_{}\", \"curve\": \"bls12_381\", \"source\": \"sigma0-polymath (reference) via tools/ark_crosscheck\", ", case).unwrap();
    write!(j, "\"r1cs\": {{\"m0\": {}, \"mw\": {}, \"a\": {}, \"b\": {}, \"c\": {}}}, ", sap.num_instance_variables, sap.num_r1cs_witness_variables,
           matrix_json(&sap.a), matrix_json(&sap.b), matrix_json(&sap.c)).unwrap();
    write!(j, "\"instance\": {}, \"witness\": {}, ", list_json(&instance), list_json(&witness)).unwrap();
    write!(j, "\"x_trapdoor\": {}, \"z_trapdoor\": {}, \"r_a\": {}, ", fq_hex(&x), fq_hex(&z), list_json(&m.r_a)).unwrap();
    write!(j, "\"n\": {}, \"sigma\": {}, \"omega\": {}, \"vk_bytes\": \"{}\", ", vk.n, vk.sigma, fq_hex(&vk.omega), bytes_hex(&vk)).unwrap();
    if with_pk_bytes {
        write!(j, "\"pk_bytes\": \"{}\", ", bytes_hex(&pk)).unwrap();
    }
    write!(j, "\"bases\": {{\"x_powers_g1\": {}, \"x_powers_y_alpha_g1\": {}, \"x_powers_y_gamma_g1\": {}, \"x_powers_y_gamma_z_g1\": {}, \
               \"x_powers_zh_by_y_alpha_g1\": {}, \"uj_wj_lcs_by_y_alpha_g1\": {}}}, ",
           points_json(&pk.x_powers_g1), points_json(&pk.x_powers_y_alpha_g1), points_json(&pk.x_powers_y_gamma_g1),
           points_json(&pk.x_powers_y_gamma_z_g1), points_json(&pk.x_powers_zh_by_y_alpha_g1), points_json(&pk.uj_wj_lcs_by_y_alpha_g1)).unwrap();
    let runs: Vec<String> = [&m, &k, &b]
        .iter()
        .map(|r| {
            format!("\"{}\": {{\"a_g1\": {}, \"c_g1\": {}, \"a_at_x1\": {}, \"d_g1\": {}, \"bytes\": \"{}\"}}", r.transcript, g1_json(&r.proof.a_g1),
                    g1_json(&r.proof.c_g1), fq_hex(&r.proof.a_at_x1), g1_json(&r.proof.d_g1), bytes_hex(&r.proof))
        })
        .collect();
    write!(j, "\"proofs\": {{{}}}}}", runs.join(", ")).unwrap();
    let path = format!("{}/ref_{}.json", out_dir, case);
    fs::write(&path, format!("[{}]\n", j)).expect("write fixture");
    println!("wrote {}  (n = {}, {} constraints)", path, vk.n, sap.num_r1cs_constraints);
}

fn emit(out_dir: &str) {
    emit_case(out_dir, "dummy", true, |_rng: &mut StdRng| {
        let make: Box<dyn Fn(&mut StdRng) -> Product> = Box::new(|rng: &mut StdRng| {
            let a = Fr::rand(rng); // dummy.rs:56-57
            let b = Fr::rand(rng);
            Product { a: Some(a), b: Some(b) }
        });
        (Product { a: None, b: None }, make)
    });
    // m0 = 12 and m0 = 1 (round 5): 40 chained values, the first 11 / 0 products public
    for (case, public) in [("inputs11", 11usize), ("inputs0", 0usize)] {
        emit_case(out_dir, case, false, move |_rng: &mut StdRng| {
            let make: Box<dyn Fn(&mut StdRng) -> Chain> = Box::new(move |rng: &mut StdRng| Chain { vals: (0..40).map(|_| Some(Fr::rand(rng))).collect(), public });
            (Chain { vals: vec![None; 40], public }, make)
        });
    }
    emit_case(out_dir, "mimc322", false, |rng: &mut StdRng| {
        let k: Vec<Fr> = (0..ROUNDS).map(|_| rng.gen()).collect(); // mimc.rs:155
        let k2 = k.clone();
        let make: Box<dyn Fn(&mut StdRng) -> Longsight> = Box::new(move |rng: &mut StdRng| {
            let l: Fr = rng.gen(); // mimc.rs:186-187 (first sample of the loop)
            let r: Fr = rng.gen();
            let _image = longsight(l, r, &k2);
            Longsight { left: Some(l), right: Some(r), round_constants: k2.clone() }
        });
        (Longsight { left: None, right: None, round_constants: k }, make)
    });
}

// ------------------------------------------------------------------------------------------------ verify
// A JSON reader for exactly the fields needed (no serde in the dependency set on purpose: fewer crates to resolve against
// arkworks HEAD).  The golden files are machine-written, one fixture object per top-level array element.
fn field_after<'a>(s: &'a str, key: &str) -> &'a str {
    let at = s.find(&format!("\"{}\"", key)).unwrap_or_else(|| panic!("missing key {}", key));
    let rest = &s[at + key.len() + 2..];
    rest[rest.find(':').unwrap() + 1..].trim_start()
}
fn scalar_str<'a>(s: &'a str, key: &str) -> &'a str {
    let v = field_after(s, key);
    if let Some(stripped) = v.strip_prefix('"') {
        &stripped[..stripped.find('"').unwrap()]
    } else {
        let end = v.find(|c: char| c == ',' || c == '}' || c == ']').unwrap();
        v[..end].trim()
    }
}
fn fr_from_hex(h: &str) -> Fr {
    let h = h.trim_start_matches("0x");
    let padded = if h.len() % 2 == 1 { format!("0{}", h) } else { h.to_string() };
    Fr::from_be_bytes_mod_order(&hex::decode(padded).expect("hex"))
}
fn hex_list(s: &str, key: &str) -> Vec<Fr> {
    let v = field_after(s, key);
    let end = v.find(']').unwrap();
    v[1..end].split(',').map(|t| t.trim().trim_matches('"')).filter(|t| !t.is_empty()).map(fr_from_hex).collect()
}

fn verify(path: &str) {
    let text = fs::read_to_string(path).expect("read golden proofs");
    // fixtures start at `{"name":` at nesting depth 1
    let mut starts: Vec<usize> = text.match_indices("{\"name\"").map(|(i, _)| i).collect();
    starts.push(text.len());
    let (mut total, mut accepted) = (0, 0);
    for w in starts.windows(2) {
        let fx = &text[w[0]..w[1]];
        let name = scalar_str(fx, "name");
        let n: u64 = scalar_str(fx, "n").parse().unwrap();
        let sigma: u64 = scalar_str(fx, "sigma").parse().unwrap();
        let r1cs = field_after(fx, "r1cs");
        let m0: u64 = scalar_str(r1cs, "m0").parse().unwrap();
        let omega = fr_from_hex(scalar_str(fx, "omega"));
        let x = fr_from_hex(scalar_str(fx, "x_trapdoor"));
        let z = fr_from_hex(scalar_str(fx, "z_trapdoor"));
        let instance = hex_list(fx, "instance");
        // the verifying key of generator.rs:139-157 from the fixture's trapdoors
        let vk = VerifyingKey::<E> {
            e: PairingVK {
                one_g1: G1Projective::generator().into_affine(),
                one_g2: G2Projective::generator().into_affine(),
                x_g2: (G2Projective::generator() * x).into_affine(),
                z_g2: (G2Projective::generator() * z).into_affine(),
            },
            n,
            m0,
            sigma,
            omega,
        };
        let proofs = field_after(fx, "proofs");
        for t in ["merlin", "keccak256", "blake3"] {
            let body = field_after(proofs, t);
            let bytes = hex::decode(scalar_str(body, "bytes")).expect("proof hex");
            let proof = Proof::<E>::deserialize_compressed(&bytes[..]).expect("the reference cannot parse this repository's proof bytes");
            let ok = match t {
                "merlin" => Polymath::<E, MerlinFieldTranscript<Fr>>::verify(&vk, &instance[1..], &proof),
                "keccak256" => Polymath::<E, Keccak256Transcript<Fr>>::verify(&vk, &instance[1..], &proof),
                _ => Polymath::<E, Blake3Transcript<Fr>>::verify(&vk, &instance[1..], &proof),
            }
            .expect("verify");
            total += 1;
            accepted += ok as usize;
            println!("{:<12} {:<10} {}", name, t, if ok { "accepted" } else { "REJECTED" });
        }
    }
    println!("{} / {} accepted", accepted, total);
    assert_eq!(accepted, total, "the reference's verifier rejects a golden proof of this repository");
}

// ------------------------------------------------------------------------------------------------ bench
fn bench(out: &str) {
    let threads = rayon::current_num_threads();
    let mut rng = StdRng::seed_from_u64(0);
    let mut rows = Vec::new();
    for log in [20usize, 22, 24] {
        let len = 1usize << log;
        // bases (i + 1) G by running addition, batch-normalised
        let g = G1Projective::generator();
        let mut acc = g;
        let mut proj = Vec::with_capacity(len);
        for _ in 0..len {
            proj.push(acc);
            acc += g;
        }
        let bases = G1Projective::normalize_batch(&proj);
        drop(proj);
        let scalars: Vec<Fr> = (0..len).map(|_| Fr::rand(&mut rng)).collect();
        let t0 = Instant::now();
        let r = G1Projective::msm_unchecked(&bases, &scalars);
        let msm_s = t0.elapsed().as_secs_f64();
        std::hint::black_box(r);
        let domain = Radix2EvaluationDomain::<Fr>::new(len).unwrap();
        let mut v = scalars.clone();
        let t0 = Instant::now();
        domain.fft_in_place(&mut v);
        let fft_s = t0.elapsed().as_secs_f64();
        std::hint::black_box(&v);
        println!("2^{}: msm_unchecked {:.3} s = {:.2} M pairs/s, fft {:.3} s  ({} threads)", log, msm_s, len as f64 / msm_s / 1e6, fft_s, threads);
        rows.push(format!("{{\"log_len\": {}, \"msm_seconds\": {}, \"msm_pairs_per_sec\": {}, \"fft_seconds\": {}}}", log, msm_s, len as f64 / msm_s, fft_s));
    }
    fs::write(out, format!("{{\"kind\": \"reference\", \"library\": \"ark-ec / ark-poly (algebra HEAD)\", \"threads\": {}, \"rows\": [{}]}}\n", threads, rows.join(", ")))
        .expect("write bench json");
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    match args.get(1).map(|s| s.as_str()) {
        Some("emit") => emit(args.get(2).expect("emit <out_dir>")),
        Some("verify") => verify(args.get(2).expect("verify <tests/golden/proofs.json>")),
        Some("bench") => bench(args.get(2).expect("bench <out.json>")),
        _ => eprintln!("usage: ark-crosscheck emit <dir> | verify <proofs.json> | bench <out.json>"),
    }
}
