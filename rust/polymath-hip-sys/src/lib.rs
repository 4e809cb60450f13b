//! Raw bindings of `include/polymath_hip.h`, one to one: every entry point of libpolymath_hip.so (59), every struct and
//! enum value.  `tests/test_abi.py` (CPU, no cargo needed) parses THIS file and compares symbol names, argument counts and
//! argument classes (pointer / integer / float) with the header and with the library's export table.
//!
//! Conventions (the header's): field elements are arkworks' in-memory form -- `PM_FR_LIMBS` (4) little-endian u64 limbs in
//! MONTGOMERY form; G1 affine points are `x || y` every `stride` bytes, arkworks' `infinity: bool` at byte 16 * fq_limbs when
//! the stride leaves room (104-byte `G1Affine` of BLS12-381).  All functions return `PM_OK` or a `pm_status` code.
//!
//! The reference crate is `#![forbid(unsafe_code)]` (src/lib.rs:13): it never names this crate, only the safe wrapper
//! `polymath-hip`.
#![no_std]
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_long, c_longlong, c_void};

pub const PM_FR_LIMBS: usize = 4;

// pm_curve
pub const PM_BLS12_381: i32 = 0;
pub const PM_BN254: i32 = 1;

// pm_status
pub const PM_OK: i32 = 0;
pub const PM_ERR_INVALID_ARG: i32 = 1;
pub const PM_ERR_LEN_MISMATCH: i32 = 2; // assert!(scalars.len() <= g1_elems.len())   prover.rs:381
pub const PM_ERR_DOMAIN_TOO_LARGE: i32 = 3; // D::new(..) None / PolynomialDegreeTooLarge  prover.rs:83,317
pub const PM_ERR_REMAINDER_NONZERO: i32 = 4; // assert!(rem_poly.is_zero())                 prover.rs:108,221
pub const PM_ERR_DEGREE_BOUND: i32 = 5; // the degree asserts                          prover.rs:107,113,222
pub const PM_ERR_HIP: i32 = 6;
pub const PM_ERR_NO_DEVICE: i32 = 7;
pub const PM_ERR_STATE: i32 = 8;
pub const PM_ERR_COMM: i32 = 9;

// pm_base_vec: one per ProvingKey field (data_structures.rs:56-73), the order of `pm_pk_load`'s array
pub const PM_X_POWERS: i32 = 0;
pub const PM_X_POWERS_Y_ALPHA: i32 = 1;
pub const PM_X_POWERS_Y_GAMMA: i32 = 2;
pub const PM_X_POWERS_Y_GAMMA_Z: i32 = 3;
pub const PM_X_POWERS_ZH_BY_Y_ALPHA: i32 = 4;
pub const PM_UJ_WJ_LCS_BY_Y_ALPHA: i32 = 5;
pub const PM_NUM_BASE_VECS: usize = 6;

// pm_shard_layout
pub const PM_SHARD_PAIRS: i32 = 0;
pub const PM_SHARD_VECTOR: i32 = 1;

// pm_option
pub const PM_OPT_MSM_OVERLAP: i32 = 0;
pub const PM_OPT_NTT_OVERLAP: i32 = 1;
pub const PM_OPT_TABLES: i32 = 2;
pub const PM_OPT_MSM_MAX_PIECE_LOG: i32 = 3;
pub const PM_OPT_MAX_SEG_LOG: i32 = 4;
pub const PM_OPT_INFLIGHT_CONTEXTS: i32 = 5;
pub const PM_OPT_MSM_TASK_LEN: i32 = 6;
pub const PM_OPT_TABLE_WINDOW_BITS: i32 = 7;
pub const PM_NUM_OPTIONS: i32 = 8;

// pm_tables_mode
pub const PM_TABLES_OFF: c_longlong = 0;
pub const PM_TABLES_AUTO: c_longlong = 1;
pub const PM_TABLES_WIDE: c_longlong = 2;
pub const PM_TABLES_NO_WIDE: c_longlong = 3;

// pm_transcript
pub const PM_TRANSCRIPT_MERLIN: i32 = 0;
pub const PM_TRANSCRIPT_KECCAK256: i32 = 1;
pub const PM_TRANSCRIPT_BLAKE3: i32 = 2;

#[repr(C)]
pub struct pm_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct pm_pk {
    _p: [u8; 0],
}
#[repr(C)]
pub struct pm_bases {
    _p: [u8; 0],
}
#[repr(C)]
pub struct pm_comm {
    _p: [u8; 0],
}

/// R1CS matrix in CSR form: ark-relations `ConstraintMatrices` rows `Vec<Vec<(F, usize)>>` (generator.rs:46-54) flattened.
#[repr(C)]
pub struct pm_csr {
    pub nrows: u64,
    pub rowptr: *const u64, // nrows + 1
    pub col: *const u32,    // nnz
    pub val: *const u64,    // nnz * PM_FR_LIMBS, Montgomery
}

#[repr(C)]
pub struct pm_base_array {
    pub points: *const c_void,
    pub len: usize,
    pub stride: usize,
}

pub type pm_combine_fn = Option<unsafe extern "C" fn(user: *mut c_void, count: i32, xy: *mut u64, inf: *mut i32) -> i32>;

#[repr(C)]
pub struct pm_comm_ops {
    pub user: *mut c_void,
    pub all_to_all:
        Option<unsafe extern "C" fn(user: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, bytes_per_peer: usize, hip_stream: *mut c_void) -> i32>,
    pub all_gather: Option<unsafe extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes: usize) -> i32>,
}

extern "C" {
    // ---- library / context
    pub fn pm_device_count() -> i32;
    pub fn pm_ctx_create(device: i32, out: *mut *mut pm_ctx) -> i32;
    pub fn pm_ctx_destroy(ctx: *mut pm_ctx);
    pub fn pm_last_error(ctx: *const pm_ctx) -> *const c_char;
    pub fn pm_last_timings(ctx: *mut pm_ctx, ms_out: *mut f64, n_slots: i32) -> i32;
    pub fn pm_ctx_set_comm(ctx: *mut pm_ctx, comm: *mut pm_comm) -> i32;
    // ---- per-context options (pm_option / pm_tables_mode): no process-wide state -- lib.rs:44-50 has none either
    pub fn pm_ctx_set_option(ctx: *mut pm_ctx, option: i32, value: c_longlong) -> i32;
    pub fn pm_ctx_get_option(ctx: *const pm_ctx, option: i32, value: *mut c_longlong) -> i32;
    // ---- standalone kernels: Radix2EvaluationDomain::fft / ifft (prover.rs:241,319,325), msm_unchecked (prover.rs:380-384)
    pub fn pm_ntt(ctx: *mut pm_ctx, curve: i32, data: *mut u64, log_n: u32, inverse: i32) -> i32;
    pub fn pm_ntt_device(ctx: *mut pm_ctx, curve: i32, d_data: *mut u64, log_n: u32, inverse: i32) -> i32;
    pub fn pm_msm_g1(ctx: *mut pm_ctx, curve: i32, bases: *const c_void, base_stride: usize, scalars: *const u64, len: usize, out_xy: *mut u64, out_inf: *mut i32) -> i32;
    pub fn pm_bases_upload(ctx: *mut pm_ctx, curve: i32, bases: *const c_void, base_stride: usize, len: usize, out: *mut *mut pm_bases) -> i32;
    pub fn pm_bases_generate_multiples(ctx: *mut pm_ctx, curve: i32, len: usize, out: *mut *mut pm_bases) -> i32;
    pub fn pm_bases_precompute(ctx: *mut pm_ctx, b: *mut pm_bases) -> i32;
    pub fn pm_bases_download(ctx: *mut pm_ctx, b: *const pm_bases, offset: usize, len: usize, out_xy: *mut u64) -> i32;
    pub fn pm_bases_len(b: *const pm_bases) -> usize;
    pub fn pm_bases_free(b: *mut pm_bases);
    pub fn pm_msm_g1_resident(ctx: *mut pm_ctx, bases: *const pm_bases, base_offset: usize, scalars: *const u64, scalars_on_device: i32, len: usize, out_xy: *mut u64, out_inf: *mut i32) -> i32;
    pub fn pm_g1_sum(curve: i32, points_xy: *const u64, infs: *const i32, count: usize, out_xy: *mut u64, out_inf: *mut i32) -> i32;
    // ---- proving key (data_structures.rs:56-73; generator.rs:24-167)
    pub fn pm_pk_load(ctx: *mut pm_ctx, curve: i32, n: u64, m0: u64, mw: u64, nr: u64, sigma: u64, a: *const pm_csr, b: *const pm_csr, c: *const pm_csr, bases: *const pm_base_array, shard_rank: i32, shard_count: i32, out: *mut *mut pm_pk) -> i32;
    pub fn pm_pk_generate(ctx: *mut pm_ctx, curve: i32, m0: u64, mw: u64, nr: u64, a: *const pm_csr, b: *const pm_csr, c: *const pm_csr, x_trapdoor: *const u64, z_trapdoor: *const u64, shard_rank: i32, shard_count: i32, out: *mut *mut pm_pk) -> i32;
    pub fn pm_pk_load_sharded(ctx: *mut pm_ctx, curve: i32, n: u64, m0: u64, mw: u64, nr: u64, sigma: u64, a: *const pm_csr, b: *const pm_csr, c: *const pm_csr, bases: *const pm_base_array, shard_rank: i32, shard_count: i32, layout: i32, out: *mut *mut pm_pk) -> i32;
    pub fn pm_pk_generate_sharded(ctx: *mut pm_ctx, curve: i32, m0: u64, mw: u64, nr: u64, a: *const pm_csr, b: *const pm_csr, c: *const pm_csr, x_trapdoor: *const u64, z_trapdoor: *const u64, shard_rank: i32, shard_count: i32, layout: i32, out: *mut *mut pm_pk) -> i32;
    pub fn pm_layout_indices(n: u64, shard_count: i32, shard_rank: i32, coefficients: i32, out: *mut u64) -> i32;
    pub fn pm_pk_msm_pieces(pk: *const pm_pk, which: i32, cat_lo: *mut u64, count: *mut u64, capacity: usize, n_pieces: *mut usize) -> i32;
    pub fn pm_pk_info(pk: *const pm_pk, n: *mut u64, m0: *mut u64, sigma: *mut u64, omega: *mut u64, base_lens: *mut u64) -> i32;
    pub fn pm_pk_msm_plan(pk: *const pm_pk, which: i32, pairs: *mut u64, windows: *mut u32, window_bits: *mut u32, tables: *mut i32) -> i32;
    pub fn pm_pk_export_bases(ctx: *mut pm_ctx, pk: *const pm_pk, which: i32, offset: usize, len: usize, out_xy: *mut u64) -> i32;
    pub fn pm_pk_free(pk: *mut pm_pk);
    // ---- prove: create_proof_with_assignment (prover.rs:66-237) split at its two transcript calls
    pub fn pm_prove_phase1(ctx: *mut pm_ctx, pk: *const pm_pk, x: *const u64, w: *const u64, r_a: *const u64, a_g1_xy: *mut u64, a_inf: *mut i32, c_g1_xy: *mut u64, c_inf: *mut i32) -> i32;
    pub fn pm_prove_phase1_device(ctx: *mut pm_ctx, pk: *const pm_pk, d_x: *const u64, d_w: *const u64, r_a: *const u64, a_g1_xy: *mut u64, a_inf: *mut i32, c_g1_xy: *mut u64, c_inf: *mut i32) -> i32;
    pub fn pm_prove_phase2(ctx: *mut pm_ctx, x1: *const u64, u_at_x1: *mut u64) -> i32;
    pub fn pm_prove_phase3(ctx: *mut pm_ctx, x1: *const u64, x2: *const u64, a_at_x1: *const u64, c_at_x1: *const u64, d_g1_xy: *mut u64, d_inf: *mut i32) -> i32;
    pub fn pm_host_prove(ctx: *mut pm_ctx, pk: *const pm_pk, transcript: i32, instance_host: *const u64, x: *const u64, w: *const u64, assignment_on_device: i32, r_a: *const u64, proof_bytes: *mut u8, capacity: usize, proof_len: *mut usize) -> i32;
    pub fn pm_host_prove_sharded(ctx: *mut pm_ctx, pk: *const pm_pk, transcript: i32, instance_host: *const u64, x: *const u64, w: *const u64, assignment_on_device: i32, r_a: *const u64, combine: pm_combine_fn, user: *mut c_void, proof_bytes: *mut u8, capacity: usize, proof_len: *mut usize) -> i32;
    pub fn pm_host_make_vk(curve: i32, n: u64, m0: u64, sigma: u64, omega: *const u64, x_trapdoor: *const u64, z_trapdoor: *const u64, vk_bytes: *mut u8, capacity: usize, vk_len: *mut usize) -> i32;
    pub fn pm_host_verify(curve: i32, transcript: i32, vk_bytes: *const u8, vk_len: usize, public_inputs: *const u64, n_inputs: usize, proof_bytes: *const u8, proof_len: usize, accepted: *mut i32) -> i32;
    pub fn pm_host_keccak_f1600(state: *mut u64);
    // ---- multi-GPU exchange layer (no reference counterpart: the reference is single-process CPU code)
    pub fn pm_comm_rccl_unique_id(out_128_bytes: *mut c_void) -> i32;
    pub fn pm_comm_rccl_create(unique_id_128_bytes: *const c_void, rank: i32, world: i32, device: i32, out: *mut *mut pm_comm) -> i32;
    pub fn pm_comm_local_create(world: i32, out: *mut *mut pm_comm) -> i32;
    pub fn pm_comm_from_callbacks(ops: *const pm_comm_ops, rank: i32, world: i32, out: *mut *mut pm_comm) -> i32;
    pub fn pm_comm_destroy(c: *mut pm_comm);
    pub fn pm_comm_rank(c: *const pm_comm) -> i32;
    pub fn pm_comm_world(c: *const pm_comm) -> i32;
    pub fn pm_comm_last_error(c: *const pm_comm) -> *const c_char;
    pub fn pm_comm_kind(c: *const pm_comm) -> *const c_char;
    pub fn pm_comm_set_timeout_ms(c: *mut pm_comm, timeout_ms: c_long) -> i32;
    pub fn pm_comm_abort(c: *mut pm_comm, why: *const c_char) -> i32;
    pub fn pm_comm_failed(c: *const pm_comm) -> i32;
    pub fn pm_comm_local_set_serialize(c: *mut pm_comm, on: i32) -> i32;
    pub fn pm_comm_busy_ms(c: *mut pm_comm, reset: i32) -> f64;
    pub fn pm_comm_all_gather(c: *mut pm_comm, send: *const c_void, recv: *mut c_void, bytes: usize) -> i32;
    pub fn pm_comm_all_to_all(c: *mut pm_comm, d_send: *const c_void, d_recv: *mut c_void, bytes_per_peer: usize, hip_stream: *mut c_void) -> i32;
    pub fn pm_comm_all_gather_device(c: *mut pm_comm, d_send: *const c_void, d_recv: *mut c_void, bytes: usize, hip_stream: *mut c_void) -> i32;
    pub fn pm_comm_combine_points(c: *mut pm_comm, curve: i32, count: i32, xy: *mut u64, inf: *mut i32) -> i32;
    // ---- harness / diagnostics
    pub fn pm_synth_r1cs(curve: i32, nr: u64, seed: u64, a_val: *mut u64, a_col: *mut u32, b_val: *mut u64, b_col: *mut u32, c_val: *mut u64, c_col: *mut u32, instance: *mut u64, witness: *mut u64) -> i32;
    pub fn pm_selftest_field(ctx: *mut pm_ctx, products_per_field: usize, seed: u64, mismatches: *mut u64) -> i32;
    pub fn pm_prove_tap(ctx: *mut pm_ctx, which: i32, out: *mut u64, max_elems: usize, n_elems: *mut usize) -> i32;
}
