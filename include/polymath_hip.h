/*
 * polymath_hip.h -- C ABI of libpolymath_hip.so, the MI355X (gfx950) implementation of
 * the Polymath prover hot path.  This is the drop-in boundary: everything the reference's
 * Rust crate would bind over FFI for `create_proof_with_assignment`
 * (/root/reference/src/prover.rs:66-237) plus the standalone MSM / NTT entry points the
 * headline metric is quoted on.  Plain pointers and sizes only; no C++/torch types.
 *
 * Data conventions (identical to arkworks' in-memory layout):
 *   Fr  : PM_FR_LIMBS  u64 little-endian limbs, MONTGOMERY form (R = 2^256).
 *   Fq  : fq_limbs     u64 little-endian limbs, MONTGOMERY form (R = 2^384 BLS12-381, 2^256 BN254).
 *   G1 affine in : x||y (2*fq_limbs u64) every `stride` bytes.  If stride > 16*fq_limbs the byte
 *                  at offset 16*fq_limbs is arkworks' `infinity: bool`; a point whose x and y are
 *                  both all-zero is also treated as the point at infinity.
 *   G1 affine out: x||y Montgomery into `out_xy` (2*fq_limbs u64) and *out_inf = 1 for infinity
 *                  (then x = y = 0).
 * All functions return PM_OK (0) or a pm_status error; none throws or aborts.  The caller owns
 * every host buffer; the library keeps no host pointer after a call returns.
 *
 * Threading: pm_pk is immutable after creation and may be shared by contexts on the same device;
 * a pm_ctx owns its HIP stream and per-proof state, so one proof in flight per ctx
 * (the reference has no global state: src/lib.rs:44-50).
 */
#ifndef POLYMATH_HIP_H
#define POLYMATH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_FR_LIMBS 4

typedef enum pm_curve {
    PM_BLS12_381 = 0, /* the reference's only instantiated curve (Cargo.toml:35) */
    PM_BN254 = 1      /* BASELINE.json configs[4] */
} pm_curve;

typedef enum pm_status {
    PM_OK = 0,
    PM_ERR_INVALID_ARG = 1,
    PM_ERR_LEN_MISMATCH = 2,       /* == assert!(scalars.len() <= g1_elems.len())  prover.rs:381 */
    PM_ERR_DOMAIN_TOO_LARGE = 3,   /* == D::new(..) None / PolynomialDegreeTooLarge prover.rs:83,317 */
    PM_ERR_REMAINDER_NONZERO = 4,  /* == assert!(rem_poly.is_zero())               prover.rs:108,221 */
    PM_ERR_DEGREE_BOUND = 5,       /* == degree asserts                            prover.rs:107,113,222 */
    PM_ERR_HIP = 6,                /* a HIP runtime call failed; see pm_last_error */
    PM_ERR_NO_DEVICE = 7,
    PM_ERR_STATE = 8,              /* phases called out of order */
    PM_ERR_COMM = 9                /* multi-GPU proofs only (no reference counterpart): a peer rank failed, or a collective
                                      did not complete within the communicator's deadline; the communicator is dead */
} pm_status;

typedef struct pm_ctx pm_ctx;
typedef struct pm_pk pm_pk;
typedef struct pm_bases pm_bases;

/* R1CS matrix in CSR form: what ark-relations `ConstraintMatrices` (generator.rs:46-54) holds as
 * Vec<Vec<(F, usize)>>, flattened.  Column 0 = One, 1..m0-1 instance, then witness. */
typedef struct pm_csr {
    uint64_t nrows;
    const uint64_t *rowptr; /* nrows+1 */
    const uint32_t *col;    /* nnz */
    const uint64_t *val;    /* nnz * PM_FR_LIMBS, Montgomery */
} pm_csr;

/* Base-vector selector, one per ProvingKey field (data_structures.rs:56-73). */
typedef enum pm_base_vec {
    PM_X_POWERS = 0,             /* x_powers_g1                  n+1 points     generator.rs:82  */
    PM_X_POWERS_Y_ALPHA = 1,     /* x_powers_y_alpha_g1          3 points       generator.rs:86  */
    PM_X_POWERS_Y_GAMMA = 2,     /* x_powers_y_gamma_g1          2 points       generator.rs:90  */
    PM_X_POWERS_Y_GAMMA_Z = 3,   /* x_powers_y_gamma_z_g1        10n+23 points  generator.rs:94  */
    PM_X_POWERS_ZH_BY_Y_ALPHA = 4,/* x_powers_zh_by_y_alpha_g1   n-1 points     generator.rs:107 */
    PM_UJ_WJ_LCS_BY_Y_ALPHA = 5, /* uj_wj_lcs_by_y_alpha_g1      M-m0 points    generator.rs:115 */
    PM_NUM_BASE_VECS = 6
} pm_base_vec;

/* How a key is spread over the ranks of a multi-GPU proof.
 *   PM_SHARD_PAIRS : only the MSM pair ranges are sharded (contiguous 1/N slices); witness map, NTTs and scans are
 *                    replicated on every rank (BASELINE.json configs[2]: "MSM buckets sharded").
 *   PM_SHARD_VECTOR: the vector phases are sharded too (configs[3]: "NTT domain + MSM both 8-way partitioned"):
 *                    evaluations cyclic, coefficients blocked (polymath_amd/host/layout.hpp), four-step NTT with one
 *                    all-to-all per transform, scans exchange per-segment values; the ranks' contexts must be joined
 *                    by a pm_comm (pm_ctx_set_comm).  Needs shard_count a power of two with shard_count^2 | n. */
typedef enum pm_shard_layout { PM_SHARD_PAIRS = 0, PM_SHARD_VECTOR = 1 } pm_shard_layout;

typedef struct pm_base_array {
    const void *points; /* host pointer, G1 affine-in convention above */
    size_t len;         /* number of points */
    size_t stride;      /* bytes between points */
} pm_base_array;

/* ---- library / context ------------------------------------------------------------------ */
int pm_device_count(void);
int pm_ctx_create(int device, pm_ctx **out);
void pm_ctx_destroy(pm_ctx *ctx);
const char *pm_last_error(const pm_ctx *ctx);
/* Wall-clock GPU milliseconds of the most recent call's kernels, by stage (hipEvent timers;
 * replaces the reference's start_timer!/end_timer! tracing, prover.rs:32-61).  After pm_host_prove[_sharded] the
 * slots cover the whole proof and the events are read HERE (a few dozen event queries, ~0.1 ms of host time), not
 * between the proof's phases: call it from the thread that proved, before the context's next call.  It MUTATES the
 * context (pending events are read and recycled, the device is made current): not thread-safe against any other call on
 * the same context -- hence no const. */
int pm_last_timings(pm_ctx *ctx, double *ms_out, int n_slots);

/* ---- options -----------------------------------------------------------------------------
 * What a host may choose, per context.  The reference keeps no global state (src/lib.rs:44-50: `Polymath<E, T>` is a
 * PhantomData), so neither does the library: every mode below is a field of the context, set through this call, and
 * contexts with different settings prove side by side.  The environment variables named in the comments only give a
 * new context its DEFAULTS -- they are read once, in pm_ctx_create, never while a proof runs.  Options marked (key) are
 * read when a key or a resident base vector is created on the context (pm_pk_generate / pm_pk_load / pm_bases_precompute)
 * and stay with that key. */
typedef enum pm_option {
    PM_OPT_MSM_OVERLAP = 0,       /* 0 / 1: the [a]_1 and [c]_1 MSMs of phase 1 (prover.rs:118-123,132) run concurrently.
                                   * Default 1 (PM_MSM_OVERLAP). */
    PM_OPT_NTT_OVERLAP = 1,       /* 0 / 1, multi-GPU proofs: w's distributed transform on a second stream beside u's chain
                                   * (prover.rs:93-96: the two are independent).  Default 1 (PM_NTT_OVERLAP). */
    PM_OPT_TABLES = 2,            /* (key) pm_tables_mode.  Default PM_TABLES_AUTO (PM_TABLES = 0 | 1 | wide, PM_WIDE = 0). */
    PM_OPT_MSM_MAX_PIECE_LOG = 3, /* one bucket pipeline covers at most 2^v pairs, 4 <= v <= 27; longer MSMs run in pieces.
                                   * Default 27 (PM_MSM_MAX_PIECE_LOG); lower values exercise the piece split at small sizes. */
    PM_OPT_MAX_SEG_LOG = 4,       /* (key, multi-GPU) sub-segments of the division scan hold at most 2^v indices; 0 = chosen
                                   * from n and the world size.  Default 0 (PM_MAX_SEG_LOG). */
    PM_OPT_INFLIGHT_CONTEXTS = 5, /* (key) how many contexts will prove on the key at once: their per-proof vectors and MSM
                                   * workspaces are left out of the HBM granted to window tables.  Default 1 (PM_INFLIGHT_CONTEXTS). */
    PM_OPT_MSM_TASK_LEN = 6,      /* entries of one bucket-accumulation task; 0 = twice the mean bucket load.  Default 0 (PM_MSM_SEG). */
    PM_OPT_TABLE_WINDOW_BITS = 7, /* (key) widest window of the table sets; 0 = the cost model of tables_plan.  Default 0 (PM_TABLE_C). */
    PM_NUM_OPTIONS = 8
} pm_option;
typedef enum pm_tables_mode {
    PM_TABLES_OFF = 0,      /* no window tables, no wide mode: every MSM on the per-window pipeline */
    PM_TABLES_AUTO = 1,     /* tables for the MSMs whose tables fit in HBM, the wide mode for the others */
    PM_TABLES_WIDE = 2,     /* the wide mode for every MSM (test / tuning) */
    PM_TABLES_NO_WIDE = 3   /* tables where they fit, the per-window pipeline for the others */
} pm_tables_mode;
/* PM_ERR_INVALID_ARG for an unknown option or a value outside its range.  A change takes effect at the next phase (next MSM,
 * next key) the context starts; call it from the thread that drives the context. */
int pm_ctx_set_option(pm_ctx *ctx, int option, long long value);
int pm_ctx_get_option(const pm_ctx *ctx, int option, long long *value);

/* ---- standalone kernels (unit parity + the "G1 MSM pairs/s" metric) ----------------------- */
/* Radix-2 NTT over Fr, natural order in and out, like ark-poly Radix2EvaluationDomain::fft /
 * ifft (prover.rs:241,319,325); inverse scales by 1/n.  `data` is a HOST buffer of 2^log_n Fr. */
int pm_ntt(pm_ctx *ctx, int curve, uint64_t *data, unsigned log_n, int inverse);
/* Same on a DEVICE buffer (hipMalloc'd by the caller, e.g. a torch tensor's data_ptr). */
int pm_ntt_device(pm_ctx *ctx, int curve, uint64_t *d_data, unsigned log_n, int inverse);

/* Variable-base MSM == E::G1::msm_unchecked(bases, scalars) (prover.rs:380-384), host buffers. */
int pm_msm_g1(pm_ctx *ctx, int curve, const void *bases, size_t base_stride, const uint64_t *scalars,
              size_t len, uint64_t *out_xy, int *out_inf);
/* Resident form: upload a base vector once (the pk's bases are fixed), then run MSMs against a
 * sub-range of it with scalars already in HBM (`d_scalars` device pointer) or on the host. */
int pm_bases_upload(pm_ctx *ctx, int curve, const void *bases, size_t base_stride, size_t len, pm_bases **out);
/* Synthetic base vector P_i = (i+1)*G built on the device (SURVEY.md §8d MSM micro-inputs). */
int pm_bases_generate_multiples(pm_ctx *ctx, int curve, size_t len, pm_bases **out);
/* Build the window tables 2^(c w) * P_i of a resident base vector (W x its memory): later resident MSMs
 * against it then need ceil(256/c) instead of 16 mixed additions per pair.  Proving keys do this
 * themselves when the tables fit in HBM (PM_TABLES=0 in the environment disables it). */
int pm_bases_precompute(pm_ctx *ctx, pm_bases *b);
int pm_bases_download(pm_ctx *ctx, const pm_bases *b, size_t offset, size_t len, uint64_t *out_xy);
size_t pm_bases_len(const pm_bases *b);
void pm_bases_free(pm_bases *b);
int pm_msm_g1_resident(pm_ctx *ctx, const pm_bases *bases, size_t base_offset, const uint64_t *scalars,
                       int scalars_on_device, size_t len, uint64_t *out_xy, int *out_inf);

/* Host-side G1 helpers used to combine per-GPU partial MSM results (RCCL has no
 * elliptic-curve reduction op, so partial points are all-gathered and summed locally). */
int pm_g1_sum(int curve, const uint64_t *points_xy, const int *infs, size_t count, uint64_t *out_xy, int *out_inf);

/* ---- proving key ------------------------------------------------------------------------ */
/* Upload an existing ProvingKey (data_structures.rs:56-73).  `shard_rank/shard_count` keep only
 * the contiguous 1/shard_count slice of every MSM's pair range on this device;
 * pass 0,1 for a whole key. */
int pm_pk_load(pm_ctx *ctx, int curve, uint64_t n, uint64_t m0, uint64_t mw, uint64_t nr, uint64_t sigma,
               const pm_csr *a, const pm_csr *b, const pm_csr *c, const pm_base_array bases[PM_NUM_BASE_VECS],
               int shard_rank, int shard_count, pm_pk **out);
/* Circuit-specific setup on the device == generate_proving_key (generator.rs:24-167) with the two
 * rng draws (x then z, generator.rs:72,77) supplied by the caller so RNG stays on the host side.
 * Sparse O(nnz) replacement of the dense uj_wj_lcs loop; base vectors never leave HBM. */
int pm_pk_generate(pm_ctx *ctx, int curve, uint64_t m0, uint64_t mw, uint64_t nr, const pm_csr *a,
                   const pm_csr *b, const pm_csr *c, const uint64_t *x_trapdoor, const uint64_t *z_trapdoor,
                   int shard_rank, int shard_count, pm_pk **out);
/* The same two with the shard layout chosen (pm_shard_layout); pm_pk_load / pm_pk_generate are layout PM_SHARD_PAIRS. */
int pm_pk_load_sharded(pm_ctx *ctx, int curve, uint64_t n, uint64_t m0, uint64_t mw, uint64_t nr, uint64_t sigma,
                       const pm_csr *a, const pm_csr *b, const pm_csr *c, const pm_base_array bases[PM_NUM_BASE_VECS],
                       int shard_rank, int shard_count, int layout, pm_pk **out);
int pm_pk_generate_sharded(pm_ctx *ctx, int curve, uint64_t m0, uint64_t mw, uint64_t nr, const pm_csr *a,
                           const pm_csr *b, const pm_csr *c, const uint64_t *x_trapdoor, const uint64_t *z_trapdoor,
                           int shard_rank, int shard_count, int layout, pm_pk **out);
/* PM_SHARD_VECTOR index maps (polymath_amd/host/layout.hpp), host arithmetic only: out[p] = the global coefficient index
 * (coefficients != 0) or evaluation row (coefficients == 0) of local position p < n / shard_count of rank shard_rank. */
int pm_layout_indices(uint64_t n, int shard_count, int shard_rank, int coefficients, uint64_t *out);
/* The resident pairs of merged MSM `which` on this key/shard, as ranges [cat_lo, cat_lo + count) of the logical base
 * concatenation [uj_wj_lcs | x_powers_zh | x_powers | y_alpha | y_gamma | y_gamma_z], in device / scalar order. */
int pm_pk_msm_pieces(const pm_pk *pk, int which, uint64_t *cat_lo, uint64_t *count, size_t capacity, size_t *n_pieces);
int pm_pk_info(const pm_pk *pk, uint64_t *n, uint64_t *m0, uint64_t *sigma, uint64_t *omega /*Fr*/,
               uint64_t base_lens[PM_NUM_BASE_VECS]);
/* How merged MSM `which` (0 = [a]_1: prover.rs:118,330-338; 1 = [c]_1: :121,340-357; 2 = [d]_1: :229) runs on
 * this key/shard: resident pairs, bucket additions per pair (windows), widest window in bits, and whether the
 * key holds window tables for it.  Measurement aid (bench.py's VALU roofline); no reference counterpart. */
int pm_pk_msm_plan(const pm_pk *pk, int which, uint64_t *pairs, unsigned *windows, unsigned *window_bits, int *tables);
/* Copy (a range of) one base vector back to the host, x||y Montgomery, 16*fq_limbs bytes apart. */
int pm_pk_export_bases(pm_ctx *ctx, const pm_pk *pk, int which, size_t offset, size_t len, uint64_t *out_xy);
void pm_pk_free(pm_pk *pk);

/* ---- prove: create_proof_with_assignment split at its two transcript calls ---------------- */
/* Fiat-Shamir choices of the reference (src/transcript/{merlin,keccak256,blake3}.rs) for pm_host_prove. */
typedef enum pm_transcript { PM_TRANSCRIPT_MERLIN = 0, PM_TRANSCRIPT_KECCAK256 = 1, PM_TRANSCRIPT_BLAKE3 = 2 } pm_transcript;
/* Phase 1 (prover.rs:75-123): witness map, iNTTs, u^2, h, then [a]_1 and [c]_1.
 *   x   : m0 Fr, instance assignment INCLUDING the leading one (prover.rs:56)
 *   w   : mw Fr, witness assignment
 *   r_a : 2 Fr, the two F::rand draws of prover.rs:110 (constant term first) -- an input so
 *         the RNG stays with the caller.
 * On a PM_SHARD_PAIRS key the outputs are this shard's PARTIAL sums; combine with pm_g1_sum (pm_comm_combine_points).
 * On a PM_SHARD_VECTOR key they are already the sums over all ranks: each phase ends with ONE small exchange that carries
 * the ranks' status flags, partial points and (phase 1) the block-boundary coefficients the division needs, so every rank
 * returns the same points and the same status.  With HOST x, w a PM_SHARD_VECTOR rank uploads only its 1/shard_count slice
 * of w and the communicator's device all-gather delivers the rest.
 * On a non-zero status the output points are UNDEFINED (the [a]_1 MSM may already have run when the witness check
 * fails: it overlaps the transforms). */
int pm_prove_phase1(pm_ctx *ctx, const pm_pk *pk, const uint64_t *x, const uint64_t *w, const uint64_t *r_a,
                    uint64_t *a_g1_xy, int *a_inf, uint64_t *c_g1_xy, int *c_inf);
/* Same with the assignment ALREADY RESIDENT in HBM: d_x (m0 Fr) and d_w (mw Fr) are device pointers (e.g. the
 * output of a GPU witness generator); r_a stays a host pointer (2 Fr). */
int pm_prove_phase1_device(pm_ctx *ctx, const pm_pk *pk, const uint64_t *d_x, const uint64_t *d_w, const uint64_t *r_a,
                           uint64_t *a_g1_xy, int *a_inf, uint64_t *c_g1_xy, int *c_inf);
/* Phase 2 (prover.rs:132): u(x1), the only O(n) part of a_at_x1; the caller adds r_a(x1)*y1^alpha. */
int pm_prove_phase2(pm_ctx *ctx, const uint64_t *x1, uint64_t *u_at_x1);
/* Phase 3 (prover.rs:142-229): assemble the Y^-gamma-scaled numerator, divide by (X - x1),
 * commit the dense quotient: [d]_1.  Returns PM_ERR_REMAINDER_NONZERO like prover.rs:221.
 * On a PM_SHARD_VECTOR key x1 must be the x1 of phase 2 (the x1-dependent sums of the division scan were exchanged there);
 * another value returns PM_ERR_INVALID_ARG. */
int pm_prove_phase3(pm_ctx *ctx, const uint64_t *x1, const uint64_t *x2, const uint64_t *a_at_x1,
                    const uint64_t *c_at_x1, uint64_t *d_g1_xy, int *d_inf);

/* Whole create_proof_with_assignment (prover.rs:66-237) for an UNSHARDED key, transcript included: the three
 * phases above plus the host glue between them (compute_x1 / compute_x2, common.rs:21-71; pi and c at x1,
 * :73-98; transcript = pm_transcript), run by the library's own C++ mirror of that glue.  For hosts without a
 * Transcript implementation of their own; a Rust host keeps calling the phases and owns T.
 * instance_host: the m0 public inputs (leading one included) as host Montgomery limbs -- they are hashed;
 * x, w: the assignment, host pointers or (assignment_on_device != 0) device pointers as in
 * pm_prove_phase1_device.  proof_bytes receives Proof::serialize_compressed (data_structures.rs:10-19):
 * 176 bytes on BLS12-381, 128 on BN254.  Status codes as for the phases. */
int pm_host_prove(pm_ctx *ctx, const pm_pk *pk, int transcript, const uint64_t *instance_host, const uint64_t *x,
                  const uint64_t *w, int assignment_on_device, const uint64_t *r_a, uint8_t *proof_bytes, size_t capacity,
                  size_t *proof_len);

/* The same on a SHARDED key (one rank of a multi-GPU proof): `combine` is called between the phases with this
 * rank's partial points -- count = 2 ([a]_1, [c]_1) after phase 1, count = 1 ([d]_1) after phase 3 -- and must
 * replace them, in place, by the sums over all ranks (all-gather over RCCL + pm_g1_sum: SURVEY.md §8e; RCCL has
 * no elliptic-curve reduction).  xy: count x (x||y Montgomery, 16*fq_limbs bytes); inf: count flags.  A non-zero
 * return aborts the proof with that status.  Every rank then hashes the same points and returns the same proof.
 * `combine` is for PM_SHARD_PAIRS keys; it is ignored on a PM_SHARD_VECTOR key, whose phases return summed points. */
typedef int (*pm_combine_fn)(void *user, int count, uint64_t *xy, int *inf);
int pm_host_prove_sharded(pm_ctx *ctx, const pm_pk *pk, int transcript, const uint64_t *instance_host, const uint64_t *x,
                          const uint64_t *w, int assignment_on_device, const uint64_t *r_a, pm_combine_fn combine, void *user,
                          uint8_t *proof_bytes, size_t capacity, size_t *proof_len);

/* Polymath::verify (lib.rs:80-90 -> verify_proof, verifier.rs:19-62) and the VerifyingKey of a key made from the trapdoors
 * (generator.rs:139-157), for hosts without a pairing implementation of their own.  HOST code (the verifier is O(1): two
 * Miller loops and a final exponentiation on the CPU, ~0.3 s): needs no GPU and no context.  Both pairing engines.
 * vk_bytes: VerifyingKey::serialize_compressed (data_structures.rs:25-52): one_g1, one_g2, x_g2, z_g2, n, m0, sigma, omega --
 * 392 bytes on BLS12-381 (zcash point encoding), 280 on BN254 (ark-serialize's default short-Weierstrass form).
 * public_inputs: n_inputs Fr, Montgomery, WITHOUT the leading one (verifier.rs:26); proof_bytes: Proof::serialize_compressed.
 * Malformed bytes return PM_ERR_INVALID_ARG; otherwise PM_OK with *accepted = 0 / 1. */
int pm_host_make_vk(int curve, uint64_t n, uint64_t m0, uint64_t sigma, const uint64_t *omega, const uint64_t *x_trapdoor,
                    const uint64_t *z_trapdoor, uint8_t *vk_bytes, size_t capacity, size_t *vk_len);
int pm_host_verify(int curve, int transcript, const uint8_t *vk_bytes, size_t vk_len, const uint64_t *public_inputs, size_t n_inputs,
                   const uint8_t *proof_bytes, size_t proof_len, int *accepted);

/* Host helper: Keccak-f[1600] on 25 little-endian lanes, shared by the host mirrors' Merlin / Keccak256
 * transcripts (the reference's transcripts are host code too: src/transcript/ *.rs). */
void pm_host_keccak_f1600(uint64_t state[25]);

/* ---- multi-GPU exchange layer (no reference counterpart: the reference is single-process CPU code) ------
 * One pm_comm per rank.  The sharded prover needs two collectives: an all-to-all of equal DEVICE blocks (the transpose of
 * the four-step NTT, prover.rs:239-243 / 315-328 split over ranks) and an all-gather of small HOST payloads (partial
 * points, status flags, scan carries).  RCCL form: rank 0 calls pm_comm_rccl_unique_id and ships the 128 bytes to the
 * other ranks by any means (the Rust host's own channel, torch.distributed's store, MPI ...); every rank then calls
 * pm_comm_rccl_create (collective: ncclCommInitRank).  librccl is loaded at first use (dlopen): no link-time dependency.
 * Contract: collectives are matched by program order on every rank.  The status checks of the prover itself (unsatisfied
 * witness, degree bounds: prover.rs:107,108,221) are exchanged first and fail on ALL ranks together, with the communicator
 * intact.  Any other failure of one rank (HIP error, out of memory, a dead process) ends the proof on every rank with
 * PM_ERR_COMM instead of a hang: no collective waits longer than the communicator's deadline (pm_comm_set_timeout_ms;
 * PM_COMM_TIMEOUT_MS in the environment; 120 s by default), a failing phase aborts its communicator (pm_comm_abort: the local
 * group wakes its peers at once, RCCL calls ncclCommAbort and the peers run into their deadline), and a failed communicator
 * stays failed -- the host tears the job down and starts again, as with any NCCL / MPI program. */
typedef struct pm_comm pm_comm;
int pm_comm_rccl_unique_id(void *out_128_bytes);
int pm_comm_rccl_create(const void *unique_id_128_bytes, int rank, int world, int device, pm_comm **out);
/* `world` ranks as threads of ONE process (one or several devices): rendezvous + device-to-device copies.  out[world]. */
int pm_comm_local_create(int world, pm_comm **out);
/* The host brings its own transport. */
typedef struct pm_comm_ops {
    void *user;
    /* block p of d_send (bytes_per_peer each) goes to rank p; block p of d_recv comes from rank p; device pointers;
     * must be ordered after the work already enqueued on hip_stream and complete (or be enqueued on it) on return */
    int (*all_to_all)(void *user, const void *d_send, void *d_recv, size_t bytes_per_peer, void *hip_stream);
    /* host pointers; recv holds world x bytes, rank r's block at r * bytes */
    int (*all_gather)(void *user, const void *send, void *recv, size_t bytes);
} pm_comm_ops;
int pm_comm_from_callbacks(const pm_comm_ops *ops, int rank, int world, pm_comm **out);
void pm_comm_destroy(pm_comm *c);
int pm_comm_rank(const pm_comm *c);
int pm_comm_world(const pm_comm *c);
const char *pm_comm_last_error(const pm_comm *c);
/* "rccl" | "local" | "callbacks": which transport a communicator runs on (bench.py reports it). */
const char *pm_comm_kind(const pm_comm *c);
/* Deadline of every collective of this communicator, in milliseconds (> 0). */
int pm_comm_set_timeout_ms(pm_comm *c, long timeout_ms);
/* This rank gives up (`why` goes to pm_comm_last_error of whoever notices): the communicator fails for good. */
int pm_comm_abort(pm_comm *c, const char *why);
/* 1 once the communicator has failed (deadline, peer abort, transport error). */
int pm_comm_failed(const pm_comm *c);
/* Local group only (measurement aid; any handle of the group, before its first collective): on = 1 makes the ranks take
 * turns between collectives, so that N ranks emulated on ONE GPU do not time-slice it -- each rank's kernels then take what
 * they would take alone (tools/shard_emulation.py).  PM_ERR_INVALID_ARG for other transports. */
int pm_comm_local_set_serialize(pm_comm *c, int on);
/* Serialised local group: the milliseconds this rank spent running, waits for its peers excluded; 0 for other communicators. */
double pm_comm_busy_ms(pm_comm *c, int reset);
int pm_comm_all_gather(pm_comm *c, const void *send, void *recv, size_t bytes);
int pm_comm_all_to_all(pm_comm *c, const void *d_send, void *d_recv, size_t bytes_per_peer, void *hip_stream);
/* All-gather of equal DEVICE blocks (rank r's `bytes` at d_recv + r * bytes), stream-ordered: how the sharded prover
 * distributes the assignment -- each rank uploads 1/world of the witness over its own PCIe link (prover.rs:75-80 needs all
 * of it on every rank: "broadcast of assignment"). */
int pm_comm_all_gather_device(pm_comm *c, const void *d_send, void *d_recv, size_t bytes, void *hip_stream);
/* Sum over ranks of `count` partial G1 points, in place (all-gather + pm_g1_sum): the native pm_combine_fn. */
int pm_comm_combine_points(pm_comm *c, int curve, int count, uint64_t *xy, int *inf);
/* Join a context to its rank's communicator.  Required before proving on a PM_SHARD_VECTOR key; with it,
 * pm_host_prove_sharded needs no `combine` callback either.  The context does not own the comm. */
int pm_ctx_set_comm(pm_ctx *ctx, pm_comm *comm);

/* Harness workload (no reference counterpart in src/; benches/bench.rs:38-61 is the reference's own): the synthetic
 * "random A*B=C gates" R1CS of BASELINE.json configs[1..4], generated natively with the same splitmix64
 * draws as polymath_amd/circuits.py: synthetic_r1cs.  m0 = 2, mw = nr + 1, one entry per row in A, B and C, so
 * rowptr = 0..nr is implied.  a_val/b_val/c_val: nr x 4 u64 Montgomery; *_col: nr u32; instance: 2 Fr (one, out);
 * witness: (nr + 1) Fr.  Host code: needs no GPU. */
int pm_synth_r1cs(int curve, uint64_t nr, uint64_t seed, uint64_t *a_val, uint32_t *a_col, uint64_t *b_val, uint32_t *b_col,
                  uint64_t *c_val, uint32_t *c_col, uint64_t *instance, uint64_t *witness);

/* Diagnostic: `products_per_field` seeded products a*b and squares a*a per field (plus p-1, 0, 1 operands), computed on the
 * device the way the kernels do -- the dense reduced-radix product of field.cuh and the internal-radix product of fq28.cuh --
 * and compared word for word with the host's 32-bit CIOS.  mismatches[k]: k = 0 BLS12-381 Fr, 1 BN254 Fr, 2 BLS12-381 Fq,
 * 3 BN254 Fq; all zero on a healthy build.  No reference counterpart: ark-ff's `Fp::mul_assign` is the thing restated. */
int pm_selftest_field(pm_ctx *ctx, size_t products_per_field, uint64_t seed, uint64_t mismatches[4]);

/* Debug / parity taps: copy an intermediate vector of the proof in flight back to the host.
 * which: 0 u_evals(n) 1 w_evals(n) 2 u coeffs(n) 3 w coeffs(n) 4 h coeffs(n) 5 witness-u coeffs(n)
 *        6 z_tail(M-m0) 7 quotient (10n+23) */
int pm_prove_tap(pm_ctx *ctx, int which, uint64_t *out, size_t max_elems, size_t *n_elems);

#ifdef __cplusplus
}
#endif
#endif /* POLYMATH_HIP_H */
