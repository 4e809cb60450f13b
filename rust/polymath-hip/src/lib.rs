//! Safe wrapper over `polymath-hip-sys` for arkworks types.
//!
//! What crosses the FFI is what arkworks already holds in memory (include/polymath_hip.h, "Data conventions"):
//! `Fp<MontBackend<_, N>, N>` = N little-endian u64 Montgomery limbs; `Affine<P>` = `x, y, infinity: bool`.  Rust does not
//! promise that layout, so it is CHECKED, once per curve and process, against values whose limbs are known
//! ([`layout_check`]): a compiler or arkworks version that lays the types out differently makes every entry point return
//! [`Status::LayoutMismatch`] instead of feeding the GPU garbage.  Inputs are passed by pointer (no copy of a 2^20-element
//! witness), outputs (three points and one scalar per proof) are rebuilt with `Fp::new_unchecked` / `Affine::new_unchecked`.
//!
//! All entry points are generic over `E: Pairing` and dispatch on `TypeId`: the reference's `Polymath<E, T>` is generic
//! (src/lib.rs:44-50) and has no curve-specific trait to hang a GPU backend on; [`supports`] tells whether `E` has one.
//!
//! NEVER COMPILED IN THE BUILD IMAGE (no cargo there) -- `rust/check.sh` is the first thing to run on a machine with one.
use core::any::{Any, TypeId};
use core::ffi::c_void;
use std::collections::HashMap;
use std::ffi::CStr;
use std::sync::{Arc, Mutex, OnceLock};

use ark_ec::pairing::Pairing;
use ark_ff::BigInt;
use polymath_hip_sys as sys;

// ------------------------------------------------------------------------------------------ status
/// `pm_status`, plus the two conditions only this wrapper can detect.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Status {
    InvalidArg,
    /// `assert!(scalars.len() <= g1_elems.len())`, prover.rs:381
    LenMismatch,
    /// `D::new(..)` is `None` / `SynthesisError::PolynomialDegreeTooLarge`, prover.rs:83,317
    DomainTooLarge,
    /// `assert!(rem_poly.is_zero())`, prover.rs:108,221
    RemainderNonzero,
    /// the degree asserts, prover.rs:107,113,222
    DegreeBound,
    Hip,
    NoDevice,
    State,
    Comm,
    /// `E` is not a curve the library implements
    UnsupportedCurve,
    /// this build of arkworks does not lay `Fp` / `Affine` out as the header's conventions assume
    LayoutMismatch,
    Unknown(i32),
}

impl Status {
    fn from_raw(rc: i32) -> Status {
        match rc {
            sys::PM_ERR_INVALID_ARG => Status::InvalidArg,
            sys::PM_ERR_LEN_MISMATCH => Status::LenMismatch,
            sys::PM_ERR_DOMAIN_TOO_LARGE => Status::DomainTooLarge,
            sys::PM_ERR_REMAINDER_NONZERO => Status::RemainderNonzero,
            sys::PM_ERR_DEGREE_BOUND => Status::DegreeBound,
            sys::PM_ERR_HIP => Status::Hip,
            sys::PM_ERR_NO_DEVICE => Status::NoDevice,
            sys::PM_ERR_STATE => Status::State,
            sys::PM_ERR_COMM => Status::Comm,
            other => Status::Unknown(other),
        }
    }
}

#[derive(Clone, Debug)]
pub struct HipError {
    pub status: Status,
    pub message: String,
}

impl core::fmt::Display for HipError {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "libpolymath_hip: {:?}: {}", self.status, self.message)
    }
}
impl std::error::Error for HipError {}

fn err(status: Status, message: &str) -> HipError {
    HipError { status, message: message.to_string() }
}

// ------------------------------------------------------------------------------------------ curves
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Curve {
    Bls12_381,
    Bn254,
}

impl Curve {
    fn id(self) -> i32 {
        match self {
            Curve::Bls12_381 => sys::PM_BLS12_381,
            Curve::Bn254 => sys::PM_BN254,
        }
    }
    /// u64 limbs of a base-field element
    fn fq_limbs(self) -> usize {
        match self {
            Curve::Bls12_381 => 6,
            Curve::Bn254 => 4,
        }
    }
}

/// The library's curve for a pairing engine, by `TypeId` (`Pairing: 'static`).
pub fn curve_of<E: Pairing>() -> Option<Curve> {
    let t = TypeId::of::<E>();
    if t == TypeId::of::<ark_bls12_381::Bls12_381>() {
        return Some(Curve::Bls12_381);
    }
    #[cfg(feature = "bn254")]
    if t == TypeId::of::<ark_bn254::Bn254>() {
        return Some(Curve::Bn254);
    }
    None
}

/// `true` when proofs over `E` can run on the GPU in this process: a known curve, the expected memory layout, a device.
pub fn supports<E: Pairing>() -> bool {
    match curve_of::<E>() {
        Some(c) => layout_check(c) && device_count() > 0,
        None => false,
    }
}

pub fn device_count() -> i32 {
    // SAFETY: no arguments, no preconditions.
    unsafe { sys::pm_device_count() }
}

/// A value of type `A` as a value of type `B` when the two are the same type (safe: `Any` downcast of an `Option`).
fn same_type<A: 'static, B: 'static>(a: A) -> Option<B> {
    let mut slot = Some(a);
    (&mut slot as &mut dyn Any).downcast_mut::<Option<B>>().and_then(Option::take)
}

/// Does this build lay the curve's `Fr` and `G1Affine` out as `x || y || infinity` of Montgomery limbs?  Checked on the
/// generator, on the identity and on `Fr::from(7)`, whose limbs are read through the public fields.  Cached.
pub fn layout_check(curve: Curve) -> bool {
    static BLS: OnceLock<bool> = OnceLock::new();
    #[cfg(feature = "bn254")]
    static BN: OnceLock<bool> = OnceLock::new();
    match curve {
        Curve::Bls12_381 => *BLS.get_or_init(|| {
            use ark_bls12_381::{Fr, G1Affine};
            use ark_ec::AffineRepr;
            let seven = Fr::from(7u64);
            let g = G1Affine::generator();
            let id = G1Affine::identity();
            core::mem::size_of::<Fr>() == 32
                && core::mem::size_of::<G1Affine>() == 104
                && raw_words(&seven, 4) == (seven.0).0.to_vec()
                && raw_words(&g, 6) == (g.x.0).0.to_vec()
                && raw_words(&g, 12)[6..] == (g.y.0).0[..]
                && raw_byte(&g, 96) == 0
                && raw_byte(&id, 96) == 1
        }),
        #[cfg(feature = "bn254")]
        Curve::Bn254 => *BN.get_or_init(|| {
            use ark_bn254::{Fr, G1Affine};
            use ark_ec::AffineRepr;
            let seven = Fr::from(7u64);
            let g = G1Affine::generator();
            let id = G1Affine::identity();
            core::mem::size_of::<Fr>() == 32
                && core::mem::size_of::<G1Affine>() == 72
                && raw_words(&seven, 4) == (seven.0).0.to_vec()
                && raw_words(&g, 4) == (g.x.0).0.to_vec()
                && raw_words(&g, 8)[4..] == (g.y.0).0[..]
                && raw_byte(&g, 64) == 0
                && raw_byte(&id, 64) == 1
        }),
        #[cfg(not(feature = "bn254"))]
        Curve::Bn254 => false,
    }
}

fn raw_words<T>(v: &T, n: usize) -> Vec<u64> {
    assert!(core::mem::size_of::<T>() >= 8 * n && core::mem::align_of::<T>() >= 8);
    // SAFETY: `v` is a live, 8-aligned value of at least 8 n bytes made of plain integers (checked above for size and
    // alignment; the types passed in are Fp / Affine, which hold arrays of u64 and one bool).
    unsafe { core::slice::from_raw_parts(v as *const T as *const u64, n) }.to_vec()
}

fn raw_byte<T>(v: &T, at: usize) -> u8 {
    assert!(core::mem::size_of::<T>() > at);
    // SAFETY: inside the value; the byte read is the `infinity: bool` (0 or 1) or a limb byte, never padding, when the layout
    // is the expected one -- and when it is not, the comparison fails on some other line first or reads an initialised limb byte.
    unsafe { *(v as *const T as *const u8).add(at) }
}

fn curve_checked<E: Pairing>() -> Result<Curve, HipError> {
    let c = curve_of::<E>().ok_or_else(|| err(Status::UnsupportedCurve, "no GPU implementation of this pairing engine"))?;
    if !layout_check(c) {
        return Err(err(Status::LayoutMismatch, "arkworks types are not laid out as x || y || infinity of Montgomery limbs in this build"));
    }
    Ok(c)
}

fn fr_ptr<F>(s: &[F]) -> *const u64 {
    s.as_ptr() as *const u64
}

fn fr_from_raw<E: Pairing>(curve: Curve, limbs: [u64; 4]) -> E::ScalarField {
    match curve {
        Curve::Bls12_381 => same_type(ark_bls12_381::Fr::new_unchecked(BigInt::new(limbs))),
        #[cfg(feature = "bn254")]
        Curve::Bn254 => same_type(ark_bn254::Fr::new_unchecked(BigInt::new(limbs))),
        #[cfg(not(feature = "bn254"))]
        Curve::Bn254 => None,
    }
    .expect("curve_of::<E>() named this curve")
}

fn g1_from_raw<E: Pairing>(curve: Curve, xy: &[u64; 12], inf: i32) -> E::G1Affine {
    match curve {
        Curve::Bls12_381 => {
            use ark_bls12_381::{Fq, G1Affine};
            let p = if inf != 0 {
                G1Affine::identity()
            } else {
                let mut x = [0u64; 6];
                let mut y = [0u64; 6];
                x.copy_from_slice(&xy[..6]);
                y.copy_from_slice(&xy[6..12]);
                G1Affine::new_unchecked(Fq::new_unchecked(BigInt::new(x)), Fq::new_unchecked(BigInt::new(y)))
            };
            same_type(p)
        },
        #[cfg(feature = "bn254")]
        Curve::Bn254 => {
            use ark_bn254::{Fq, G1Affine};
            let p = if inf != 0 {
                G1Affine::identity()
            } else {
                let mut x = [0u64; 4];
                let mut y = [0u64; 4];
                x.copy_from_slice(&xy[..4]);
                y.copy_from_slice(&xy[4..8]);
                G1Affine::new_unchecked(Fq::new_unchecked(BigInt::new(x)), Fq::new_unchecked(BigInt::new(y)))
            };
            same_type(p)
        },
        #[cfg(not(feature = "bn254"))]
        Curve::Bn254 => None,
    }
    .expect("curve_of::<E>() named this curve")
}

// ------------------------------------------------------------------------------------------ context
/// `pm_option` (include/polymath_hip.h): what a host may choose, per context; the reference keeps no global state either.
#[derive(Clone, Copy, Debug)]
#[repr(i32)]
pub enum PmOption {
    MsmOverlap = 0,
    NttOverlap = 1,
    Tables = 2,
    MsmMaxPieceLog = 3,
    MaxSegLog = 4,
    InflightContexts = 5,
    MsmTaskLen = 6,
    TableWindowBits = 7,
}

/// One `pm_ctx`: a HIP stream and the per-proof state; one proof in flight per context, one context per thread.
pub struct Context {
    raw: *mut sys::pm_ctx,
    device: i32,
}

// SAFETY: a pm_ctx may be driven from any ONE thread at a time (the header's threading rule); `&mut self` on every call
// that touches it enforces that.
unsafe impl Send for Context {}

impl Context {
    pub fn new(device: i32) -> Result<Context, HipError> {
        let mut raw = core::ptr::null_mut();
        // SAFETY: `raw` is a valid out-pointer.
        let rc = unsafe { sys::pm_ctx_create(device, &mut raw) };
        if rc != sys::PM_OK || raw.is_null() {
            return Err(err(Status::from_raw(rc), "pm_ctx_create failed"));
        }
        Ok(Context { raw, device })
    }

    pub fn device(&self) -> i32 {
        self.device
    }

    pub fn raw(&mut self) -> *mut sys::pm_ctx {
        self.raw
    }

    pub fn set_option(&mut self, option: PmOption, value: i64) -> Result<(), HipError> {
        // SAFETY: live context.
        let rc = unsafe { sys::pm_ctx_set_option(self.raw, option as i32, value as core::ffi::c_longlong) };
        self.check(rc)
    }

    pub fn last_error(&self) -> String {
        // SAFETY: live context; the library returns a NUL-terminated string it owns until the context's next call.
        unsafe { CStr::from_ptr(sys::pm_last_error(self.raw)) }.to_string_lossy().into_owned()
    }

    fn check(&self, rc: i32) -> Result<(), HipError> {
        if rc == sys::PM_OK {
            Ok(())
        } else {
            Err(HipError { status: Status::from_raw(rc), message: self.last_error() })
        }
    }

    /// Run `f` on this thread's context (created on first use on `device`).
    pub fn with_thread_local<T>(device: i32, f: impl FnOnce(&mut Context) -> Result<T, HipError>) -> Result<T, HipError> {
        use std::cell::RefCell;
        thread_local! { static CTX: RefCell<Option<Context>> = const { RefCell::new(None) }; }
        CTX.with(|slot| {
            let mut slot = slot.borrow_mut();
            if slot.as_ref().map(|c| c.device) != Some(device) {
                *slot = Some(Context::new(device)?);
            }
            f(slot.as_mut().expect("just filled"))
        })
    }

    /// Phase 1 (prover.rs:75-123): witness map, inverse transforms, u^2, h, then `[a]_1` and `[c]_1`.
    /// `instance` INCLUDES the leading one (prover.rs:56); `r_a` = the two `F::rand` draws of prover.rs:110, constant term first.
    ///
    /// The C context keeps a pointer to the key between the phases (`pm_prove_phase3` reads the key's bases), so the phases
    /// that follow live on the returned [`ProofInFlight`], which borrows BOTH the context and the key: safe code cannot drop
    /// the key, start another proof on the context, or free the context while a proof is between its phases.
    pub fn prove_phase1<'c, 'k, E: Pairing>(
        &'c mut self,
        key: &'k GpuKey,
        instance: &[E::ScalarField],
        witness: &[E::ScalarField],
        r_a: &[E::ScalarField; 2],
    ) -> Result<(ProofInFlight<'c, 'k>, E::G1Affine, E::G1Affine), HipError> {
        let curve = curve_checked::<E>()?;
        if curve != key.curve || instance.len() as u64 != key.m0 || witness.len() as u64 != key.mw {
            return Err(err(Status::InvalidArg, "assignment does not fit the key"));
        }
        let (mut a, mut c) = ([0u64; 12], [0u64; 12]);
        let (mut a_inf, mut c_inf) = (0i32, 0i32);
        // SAFETY: the slices hold m0 / mw / 2 field elements of 4 limbs each (layout checked); the outputs have room for
        // 2 * fq_limbs <= 12 words.
        let rc = unsafe {
            sys::pm_prove_phase1(self.raw, key.raw, fr_ptr(instance), fr_ptr(witness), fr_ptr(&r_a[..]), a.as_mut_ptr(), &mut a_inf, c.as_mut_ptr(), &mut c_inf)
        };
        self.check(rc)?;
        let (a_g1, c_g1) = (g1_from_raw::<E>(curve, &a, a_inf), g1_from_raw::<E>(curve, &c, c_inf));
        Ok((ProofInFlight { ctx: self, _key: key, curve }, a_g1, c_g1))
    }

    /// `E::G1::msm_unchecked(bases, scalars)` (prover.rs:380-384) on host slices, zipped to the shorter length like arkworks.
    pub fn msm_g1<E: Pairing>(&mut self, bases: &[E::G1Affine], scalars: &[E::ScalarField]) -> Result<E::G1Affine, HipError> {
        let curve = curve_checked::<E>()?;
        let len = bases.len().min(scalars.len());
        let mut out = [0u64; 12];
        let mut inf = 0i32;
        // SAFETY: `len` points of size_of::<G1Affine>() bytes each and `len` scalars (layout checked).
        let rc = unsafe {
            sys::pm_msm_g1(self.raw, curve.id(), bases.as_ptr() as *const c_void, core::mem::size_of::<E::G1Affine>(), fr_ptr(scalars), len, out.as_mut_ptr(), &mut inf)
        };
        self.check(rc)?;
        Ok(g1_from_raw::<E>(curve, &out, inf))
    }

    /// `Radix2EvaluationDomain::{fft, ifft}_in_place` (prover.rs:241,319,325): natural order in and out, 1/n on the inverse.
    pub fn ntt_in_place<E: Pairing>(&mut self, data: &mut [E::ScalarField], inverse: bool) -> Result<(), HipError> {
        let curve = curve_checked::<E>()?;
        if !data.len().is_power_of_two() {
            return Err(err(Status::InvalidArg, "length is not a power of two"));
        }
        // SAFETY: 2^log_n field elements, read and written in place.
        let rc = unsafe { sys::pm_ntt(self.raw, curve.id(), data.as_mut_ptr() as *mut u64, data.len().trailing_zeros(), inverse as i32) };
        self.check(rc)
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        // SAFETY: created by pm_ctx_create, destroyed once.
        unsafe { sys::pm_ctx_destroy(self.raw) }
    }
}

/// A proof between its phases: phase 1 has run on `ctx` against `key`; phases 2 and 3 are methods of this guard and nothing
/// else can touch the context or drop the key until it is gone (the library's `pm_ctx` holds a raw pointer to the `pm_pk`).
pub struct ProofInFlight<'c, 'k> {
    ctx: &'c mut Context,
    _key: &'k GpuKey,
    curve: Curve,
}

impl ProofInFlight<'_, '_> {
    /// Phase 2 (prover.rs:132): `u_poly.evaluate(&x1)`; the caller adds `r_a(x1) * y1^alpha`.
    pub fn prove_phase2<E: Pairing>(&mut self, x1: &E::ScalarField) -> Result<E::ScalarField, HipError> {
        if curve_checked::<E>()? != self.curve {
            return Err(err(Status::InvalidArg, "phase 2 called with another pairing engine than phase 1"));
        }
        let mut out = [0u64; 4];
        // SAFETY: one field element in, one out; the key of phase 1 is still alive (borrowed by `self`).
        let rc = unsafe { sys::pm_prove_phase2(self.ctx.raw, fr_ptr(core::slice::from_ref(x1)), out.as_mut_ptr()) };
        self.ctx.check(rc)?;
        Ok(fr_from_raw::<E>(self.curve, out))
    }

    /// Phase 3 (prover.rs:142-229): the Y^-gamma-scaled numerator, its division by (X - x1), `[d]_1`.  Consumes the guard.
    pub fn prove_phase3<E: Pairing>(
        self,
        x1: &E::ScalarField,
        x2: &E::ScalarField,
        a_at_x1: &E::ScalarField,
        c_at_x1: &E::ScalarField,
    ) -> Result<E::G1Affine, HipError> {
        if curve_checked::<E>()? != self.curve {
            return Err(err(Status::InvalidArg, "phase 3 called with another pairing engine than phase 1"));
        }
        let mut d = [0u64; 12];
        let mut d_inf = 0i32;
        let one = |v: &E::ScalarField| fr_ptr(core::slice::from_ref(v));
        // SAFETY: four field elements in, 2 * fq_limbs <= 12 words out; the key of phase 1 is still alive (borrowed by `self`).
        let rc = unsafe { sys::pm_prove_phase3(self.ctx.raw, one(x1), one(x2), one(a_at_x1), one(c_at_x1), d.as_mut_ptr(), &mut d_inf) };
        self.ctx.check(rc)?;
        Ok(g1_from_raw::<E>(self.curve, &d, d_inf))
    }
}

// ------------------------------------------------------------------------------------------ proving key
/// The fields of the reference's `ProvingKey<E>` (data_structures.rs:56-73) the GPU needs, borrowed.  The wrapper cannot
/// name that type (the reference depends on this crate, not the other way round).
pub struct KeyParts<'a, E: Pairing> {
    pub n: u64,
    pub sigma: u64,
    /// `SAPMatrices` (common.rs:113-127): num_instance_variables, num_r1cs_witness_variables, num_r1cs_constraints, a, b, c
    pub m0: u64,
    pub mw: u64,
    pub nr: u64,
    pub a: &'a [Vec<(E::ScalarField, usize)>],
    pub b: &'a [Vec<(E::ScalarField, usize)>],
    pub c: &'a [Vec<(E::ScalarField, usize)>],
    pub x_powers_g1: &'a [E::G1Affine],
    pub x_powers_y_alpha_g1: &'a [E::G1Affine],
    pub x_powers_y_gamma_g1: &'a [E::G1Affine],
    pub x_powers_y_gamma_z_g1: &'a [E::G1Affine],
    pub x_powers_zh_by_y_alpha_g1: &'a [E::G1Affine],
    pub uj_wj_lcs_by_y_alpha_g1: &'a [E::G1Affine],
}

/// `pm_base_vec` (include/polymath_hip.h): the six base vectors of `ProvingKey<E>` (data_structures.rs:56-73), in the order
/// of `pm_pk_load`'s array and of `pm_pk_info`'s lengths.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum BaseVec {
    XPowers = 0,
    XPowersYAlpha = 1,
    XPowersYGamma = 2,
    XPowersYGammaZ = 3,
    XPowersZhByYAlpha = 4,
    UjWjLcsByYAlpha = 5,
}

impl BaseVec {
    pub const ALL: [BaseVec; 6] =
        [BaseVec::XPowers, BaseVec::XPowersYAlpha, BaseVec::XPowersYGamma, BaseVec::XPowersYGammaZ, BaseVec::XPowersZhByYAlpha, BaseVec::UjWjLcsByYAlpha];
}

/// What `generate_proving_key` has in hand after synthesis (generator.rs:46-54): the R1CS matrices of `cs.to_matrices()`
/// and their shape.  `m0` counts the leading one (`num_instance_variables`).
pub struct R1csParts<'a, E: Pairing> {
    pub m0: u64,
    pub mw: u64,
    pub nr: u64,
    pub a: &'a [Vec<(E::ScalarField, usize)>],
    pub b: &'a [Vec<(E::ScalarField, usize)>],
    pub c: &'a [Vec<(E::ScalarField, usize)>],
}

/// `pm_pk_info`: the scalars of the verifying key (data_structures.rs:38-52) a resident key was made for, and the lengths of
/// its six base vectors ([`BaseVec`] order).
pub struct KeyInfo<E: Pairing> {
    pub n: u64,
    pub m0: u64,
    pub sigma: u64,
    /// `domain.group_gen()` (generator.rs:156)
    pub omega: E::ScalarField,
    pub base_lens: [u64; 6],
}

/// The six `Vec<E::G1Affine>` of `ProvingKey<E>` (data_structures.rs:60-72), copied back from the device.
pub struct ExportedBases<E: Pairing> {
    pub x_powers_g1: Vec<E::G1Affine>,
    pub x_powers_y_alpha_g1: Vec<E::G1Affine>,
    pub x_powers_y_gamma_g1: Vec<E::G1Affine>,
    pub x_powers_y_gamma_z_g1: Vec<E::G1Affine>,
    pub x_powers_zh_by_y_alpha_g1: Vec<E::G1Affine>,
    pub uj_wj_lcs_by_y_alpha_g1: Vec<E::G1Affine>,
}

struct Csr<F> {
    rowptr: Vec<u64>,
    col: Vec<u32>,
    val: Vec<F>,
}

fn flatten<F: Copy>(rows: &[Vec<(F, usize)>]) -> Result<Csr<F>, HipError> {
    let nnz: usize = rows.iter().map(Vec::len).sum();
    let mut m = Csr { rowptr: Vec::with_capacity(rows.len() + 1), col: Vec::with_capacity(nnz), val: Vec::with_capacity(nnz) };
    m.rowptr.push(0);
    for row in rows {
        for &(v, j) in row {
            m.col.push(u32::try_from(j).map_err(|_| err(Status::InvalidArg, "column index above 2^32"))?);
            m.val.push(v);
        }
        m.rowptr.push(m.col.len() as u64);
    }
    Ok(m)
}

impl<F> Csr<F> {
    fn raw(&self) -> sys::pm_csr {
        sys::pm_csr { nrows: (self.rowptr.len() - 1) as u64, rowptr: self.rowptr.as_ptr(), col: self.col.as_ptr(), val: fr_ptr(self.val.as_slice()) }
    }
}

/// A proving key resident in HBM (`pm_pk`): bases in the internal radix, window tables, CSR matrices.  Immutable after
/// creation; contexts on the same device may share it.
pub struct GpuKey {
    raw: *mut sys::pm_pk,
    curve: Curve,
    m0: u64,
    mw: u64,
}

// SAFETY: pm_pk is immutable after creation and may be shared by contexts on the same device (header, "Threading").
unsafe impl Send for GpuKey {}
unsafe impl Sync for GpuKey {}

impl GpuKey {
    /// `pm_pk_load`: upload an existing key.  The library copies everything it needs; nothing is borrowed after the call.
    pub fn upload<E: Pairing>(ctx: &mut Context, k: &KeyParts<'_, E>) -> Result<GpuKey, HipError> {
        let curve = curve_checked::<E>()?;
        let (a, b, c) = (flatten(k.a)?, flatten(k.b)?, flatten(k.c)?);
        let stride = core::mem::size_of::<E::G1Affine>();
        let arr = |s: &[E::G1Affine]| sys::pm_base_array { points: s.as_ptr() as *const c_void, len: s.len(), stride };
        // pm_base_vec order
        let bases = [
            arr(k.x_powers_g1),
            arr(k.x_powers_y_alpha_g1),
            arr(k.x_powers_y_gamma_g1),
            arr(k.x_powers_y_gamma_z_g1),
            arr(k.x_powers_zh_by_y_alpha_g1),
            arr(k.uj_wj_lcs_by_y_alpha_g1),
        ];
        let mut raw = core::ptr::null_mut();
        // SAFETY: the CSR vectors and base slices outlive the call; strides and lengths are the slices' own.
        let rc = unsafe {
            sys::pm_pk_load(ctx.raw, curve.id(), k.n, k.m0, k.mw, k.nr, k.sigma, &a.raw(), &b.raw(), &c.raw(), bases.as_ptr(), 0, 1, &mut raw)
        };
        ctx.check(rc)?;
        Ok(GpuKey { raw, curve, m0: k.m0, mw: k.mw })
    }

    /// `pm_pk_load_sharded`: this rank's share of a key for ONE proof over `shard_count` GPUs (one process per GPU).  `layout`:
    /// [`ShardLayout::Vector`] shards witness map, transforms, scans and MSM pairs (the context must have been joined to its
    /// rank's [`Comm`] first); [`ShardLayout::Pairs`] shards the MSM pair ranges only and the phases return PARTIAL points
    /// (sum them with [`Comm::combine_points`]).  Every rank then runs exactly the three phases of a single-GPU proof.
    pub fn upload_sharded<E: Pairing>(ctx: &mut Context, k: &KeyParts<'_, E>, shard_rank: i32, shard_count: i32, layout: ShardLayout) -> Result<GpuKey, HipError> {
        let curve = curve_checked::<E>()?;
        let (a, b, c) = (flatten(k.a)?, flatten(k.b)?, flatten(k.c)?);
        let stride = core::mem::size_of::<E::G1Affine>();
        let arr = |s: &[E::G1Affine]| sys::pm_base_array { points: s.as_ptr() as *const c_void, len: s.len(), stride };
        let bases = [
            arr(k.x_powers_g1),
            arr(k.x_powers_y_alpha_g1),
            arr(k.x_powers_y_gamma_g1),
            arr(k.x_powers_y_gamma_z_g1),
            arr(k.x_powers_zh_by_y_alpha_g1),
            arr(k.uj_wj_lcs_by_y_alpha_g1),
        ];
        let mut raw = core::ptr::null_mut();
        // SAFETY: as in `upload`; a collective only in the sense that every rank must make the same call for its own share.
        let rc = unsafe {
            sys::pm_pk_load_sharded(ctx.raw, curve.id(), k.n, k.m0, k.mw, k.nr, k.sigma, &a.raw(), &b.raw(), &c.raw(), bases.as_ptr(), shard_rank, shard_count,
                                    layout as i32, &mut raw)
        };
        ctx.check(rc)?;
        Ok(GpuKey { raw, curve, m0: k.m0, mw: k.mw })
    }

    /// `pm_pk_generate` == `generate_proving_key` (generator.rs:24-167) from the point where it holds the matrices and the two
    /// trapdoor draws: `x` then `z`, both `domain.sample_element_outside_domain(rng)` (generator.rs:72,77) -- the RNG stays with
    /// the caller.  The dense, serial `uj_wj_lcs` loop (generator.rs:112-136: n * (m - m0) calls of `SAPMatrices::u / w`) is
    /// replaced by an O(nnz) pass and the ~14 n scalar multiplications of `Self::generate` (generator.rs:169-177) by a
    /// fixed-base batch on the device; the base vectors are born resident, so the first `prove` uploads nothing.
    /// The library derives `n` and `sigma = n + 3` itself (`SAPMatrices::size`, common.rs:130-134); read them with [`GpuKey::info`].
    pub fn generate<E: Pairing>(ctx: &mut Context, r1cs: &R1csParts<'_, E>, x: &E::ScalarField, z: &E::ScalarField) -> Result<GpuKey, HipError> {
        let curve = curve_checked::<E>()?;
        if r1cs.a.len() as u64 != r1cs.nr || r1cs.b.len() as u64 != r1cs.nr || r1cs.c.len() as u64 != r1cs.nr {
            return Err(err(Status::InvalidArg, "matrices do not have num_constraints rows"));
        }
        let (a, b, c) = (flatten(r1cs.a)?, flatten(r1cs.b)?, flatten(r1cs.c)?);
        let one = |v: &E::ScalarField| fr_ptr(core::slice::from_ref(v));
        let mut raw = core::ptr::null_mut();
        // SAFETY: the CSR vectors outlive the call; `x` and `z` are one field element (4 limbs, layout checked) each.
        let rc = unsafe { sys::pm_pk_generate(ctx.raw, curve.id(), r1cs.m0, r1cs.mw, r1cs.nr, &a.raw(), &b.raw(), &c.raw(), one(x), one(z), 0, 1, &mut raw) };
        ctx.check(rc)?;
        Ok(GpuKey { raw, curve, m0: r1cs.m0, mw: r1cs.mw })
    }

    /// `pm_pk_info`: n, m0, sigma, omega and the six base-vector lengths of this key.
    pub fn info<E: Pairing>(&self) -> Result<KeyInfo<E>, HipError> {
        let curve = curve_checked::<E>()?;
        if curve != self.curve {
            return Err(err(Status::InvalidArg, "key of another curve"));
        }
        let (mut n, mut m0, mut sigma) = (0u64, 0u64, 0u64);
        let mut omega = [0u64; 4];
        let mut base_lens = [0u64; 6];
        // SAFETY: live key; five valid out-pointers (4 limbs for omega, PM_NUM_BASE_VECS = 6 lengths).
        let rc = unsafe { sys::pm_pk_info(self.raw, &mut n, &mut m0, &mut sigma, omega.as_mut_ptr(), base_lens.as_mut_ptr()) };
        if rc != sys::PM_OK {
            return Err(err(Status::from_raw(rc), "pm_pk_info failed"));
        }
        Ok(KeyInfo { n, m0, sigma, omega: fr_from_raw::<E>(curve, omega), base_lens })
    }

    /// `pm_pk_export_bases`: one base vector back on the host as `Vec<E::G1Affine>`, ready for a `ProvingKey<E>` field.  The
    /// device hands over `x || y` in Montgomery form (16 * fq_limbs bytes per point, the identity as all-zero x, y); records are
    /// rebuilt with `Affine::new_unchecked` / `Affine::identity()` -- no decompression, no subgroup check (the points were
    /// computed as multiples of the generator).  Copied in chunks of 2^18 points so the staging buffer stays small.
    pub fn export_bases<E: Pairing>(&self, ctx: &mut Context, which: BaseVec) -> Result<Vec<E::G1Affine>, HipError> {
        let info = self.info::<E>()?;
        let curve = self.curve;
        let words = 2 * curve.fq_limbs();
        let total = info.base_lens[which as usize] as usize;
        let mut out = Vec::with_capacity(total);
        const CHUNK: usize = 1 << 18;
        let mut staging = vec![0u64; words * CHUNK.min(total.max(1))];
        let mut offset = 0usize;
        while offset < total {
            let len = CHUNK.min(total - offset);
            // SAFETY: live key and context; `staging` has room for `len` points of `words` u64 each.
            let rc = unsafe { sys::pm_pk_export_bases(ctx.raw, self.raw, which as i32, offset, len, staging.as_mut_ptr()) };
            ctx.check(rc)?;
            for p in staging[..words * len].chunks_exact(words) {
                let mut xy = [0u64; 12];
                xy[..words].copy_from_slice(p);
                let inf = p.iter().all(|&w| w == 0) as i32;
                out.push(g1_from_raw::<E>(curve, &xy, inf));
            }
            offset += len;
        }
        Ok(out)
    }

    /// All six vectors, in the shape of `ProvingKey<E>`'s fields.
    pub fn export_all<E: Pairing>(&self, ctx: &mut Context) -> Result<ExportedBases<E>, HipError> {
        Ok(ExportedBases {
            x_powers_g1: self.export_bases::<E>(ctx, BaseVec::XPowers)?,
            x_powers_y_alpha_g1: self.export_bases::<E>(ctx, BaseVec::XPowersYAlpha)?,
            x_powers_y_gamma_g1: self.export_bases::<E>(ctx, BaseVec::XPowersYGamma)?,
            x_powers_y_gamma_z_g1: self.export_bases::<E>(ctx, BaseVec::XPowersYGammaZ)?,
            x_powers_zh_by_y_alpha_g1: self.export_bases::<E>(ctx, BaseVec::XPowersZhByYAlpha)?,
            uj_wj_lcs_by_y_alpha_g1: self.export_bases::<E>(ctx, BaseVec::UjWjLcsByYAlpha)?,
        })
    }

    pub fn raw(&self) -> *const sys::pm_pk {
        self.raw
    }
}

/// `pm_shard_layout`
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(i32)]
pub enum ShardLayout {
    Pairs = 0,
    Vector = 1,
}

impl Drop for GpuKey {
    fn drop(&mut self) {
        // SAFETY: created by pm_pk_load / pm_pk_generate, freed once.
        unsafe { sys::pm_pk_free(self.raw) }
    }
}

/// Keys already uploaded, by an identity the caller chooses (the reference patch uses blake3 of the verifying key's bytes):
/// `ProvingKey<E>` is a plain struct of `Vec`s and cannot carry a device handle itself.
#[derive(Default)]
pub struct GpuKeyCache {
    map: Mutex<HashMap<[u8; 32], Arc<GpuKey>>>,
}

impl GpuKeyCache {
    pub fn get_or_upload(&self, id: [u8; 32], upload: impl FnOnce() -> Result<GpuKey, HipError>) -> Result<Arc<GpuKey>, HipError> {
        if let Some(k) = self.map.lock().expect("key cache poisoned").get(&id) {
            return Ok(k.clone());
        }
        let k = Arc::new(upload()?); // outside the lock: an upload takes seconds (window tables are built on the device)
        Ok(self.map.lock().expect("key cache poisoned").entry(id).or_insert(k).clone())
    }

    /// Register a key that is ALREADY resident (made by [`GpuKey::generate`]) under `id`, so the first proof with the
    /// `ProvingKey<E>` built from its exported vectors neither re-uploads ~3 GB of bases nor rebuilds the window tables.
    /// An entry already present under `id` wins (and `key` is freed).
    pub fn adopt(&self, id: [u8; 32], key: GpuKey) -> Arc<GpuKey> {
        self.map.lock().expect("key cache poisoned").entry(id).or_insert_with(|| Arc::new(key)).clone()
    }

    pub fn forget(&self, id: &[u8; 32]) {
        self.map.lock().expect("key cache poisoned").remove(id);
    }
}

pub fn global_key_cache() -> &'static GpuKeyCache {
    static CACHE: OnceLock<GpuKeyCache> = OnceLock::new();
    CACHE.get_or_init(GpuKeyCache::default)
}

// ------------------------------------------------------------------------------------------ multi-GPU exchange layer
/// One rank's `pm_comm` (include/polymath_hip.h, "multi-GPU exchange layer"): RCCL inside the library, one process per GPU.
/// Rank 0 makes the 128-byte id ([`Comm::rccl_unique_id`]) and ships it to the other ranks over the job's own channel (MPI, a
/// TCP store, a file); every rank then calls [`Comm::rccl`] (collective: ncclCommInitRank) and joins its context with
/// [`Context::set_comm`].  Collectives have a deadline; a failed communicator stays failed ([`Comm::failed`]).
pub struct Comm {
    raw: *mut sys::pm_comm,
}

// SAFETY: a pm_comm is used by the one thread that drives its rank's context (plus the library's own watchdog thread).
unsafe impl Send for Comm {}

impl Comm {
    pub fn rccl_unique_id() -> Result<[u8; 128], HipError> {
        let mut id = [0u8; 128];
        // SAFETY: 128 writable bytes.
        let rc = unsafe { sys::pm_comm_rccl_unique_id(id.as_mut_ptr() as *mut c_void) };
        if rc != sys::PM_OK {
            return Err(err(Status::from_raw(rc), "pm_comm_rccl_unique_id failed (librccl could not be loaded?)"));
        }
        Ok(id)
    }

    pub fn rccl(unique_id: &[u8; 128], rank: i32, world: i32, device: i32) -> Result<Comm, HipError> {
        let mut raw = core::ptr::null_mut();
        // SAFETY: 128 readable bytes, a valid out-pointer.
        let rc = unsafe { sys::pm_comm_rccl_create(unique_id.as_ptr() as *const c_void, rank, world, device, &mut raw) };
        if rc != sys::PM_OK || raw.is_null() {
            return Err(err(Status::from_raw(rc), "pm_comm_rccl_create failed"));
        }
        Ok(Comm { raw })
    }

    pub fn set_timeout_ms(&mut self, ms: i64) -> Result<(), HipError> {
        // SAFETY: live communicator.
        let rc = unsafe { sys::pm_comm_set_timeout_ms(self.raw, ms as core::ffi::c_long) };
        if rc == sys::PM_OK { Ok(()) } else { Err(err(Status::from_raw(rc), "pm_comm_set_timeout_ms")) }
    }

    pub fn failed(&self) -> bool {
        // SAFETY: live communicator.
        unsafe { sys::pm_comm_failed(self.raw) != 0 }
    }

    pub fn last_error(&self) -> String {
        // SAFETY: live communicator; the library returns a NUL-terminated copy it owns.
        unsafe { CStr::from_ptr(sys::pm_comm_last_error(self.raw)) }.to_string_lossy().into_owned()
    }

    /// Sum over the ranks of `points` (a [`ShardLayout::Pairs`] key's partial results), in place: all-gather + `pm_g1_sum`
    /// (RCCL has no elliptic-curve reduction).
    pub fn combine_points<E: Pairing>(&mut self, points: &mut [E::G1Affine]) -> Result<(), HipError> {
        let curve = curve_checked::<E>()?;
        let words = 2 * curve.fq_limbs();
        let mut xy = vec![0u64; words * points.len()];
        let mut inf = vec![0i32; points.len()];
        for (i, p) in points.iter().enumerate() {
            let raw = raw_words(p, words);
            inf[i] = raw_byte(p, 8 * words) as i32;
            if inf[i] == 0 {
                xy[i * words..(i + 1) * words].copy_from_slice(&raw);
            }
        }
        // SAFETY: `count` points of `words` u64 each and as many flags.
        let rc = unsafe { sys::pm_comm_combine_points(self.raw, curve.id(), points.len() as i32, xy.as_mut_ptr(), inf.as_mut_ptr()) };
        if rc != sys::PM_OK {
            return Err(HipError { status: Status::from_raw(rc), message: self.last_error() });
        }
        for (i, p) in points.iter_mut().enumerate() {
            let mut buf = [0u64; 12];
            buf[..words].copy_from_slice(&xy[i * words..(i + 1) * words]);
            *p = g1_from_raw::<E>(curve, &buf, inf[i]);
        }
        Ok(())
    }
}

impl Drop for Comm {
    fn drop(&mut self) {
        // SAFETY: created by pm_comm_rccl_create, destroyed once; detach it from its context first (`Context::set_comm(None)`).
        unsafe { sys::pm_comm_destroy(self.raw) }
    }
}

impl Context {
    /// Join this context to its rank's communicator (required before proving on a [`ShardLayout::Vector`] key), or detach it.
    /// The context does not own the communicator: keep the `Comm` alive for as long as it is attached.
    pub fn set_comm(&mut self, comm: Option<&mut Comm>) -> Result<(), HipError> {
        let raw = comm.map_or(core::ptr::null_mut(), |c| c.raw);
        // SAFETY: live context; a null communicator detaches.
        let rc = unsafe { sys::pm_ctx_set_comm(self.raw, raw) };
        self.check(rc)
    }
}

/// The device this process proves on: `POLYMATH_HIP_DEVICE`, else `LOCAL_RANK` (one process per GPU), else 0.
pub fn default_device() -> i32 {
    for var in ["POLYMATH_HIP_DEVICE", "LOCAL_RANK"] {
        if let Some(d) = std::env::var(var).ok().and_then(|v| v.parse().ok()) {
            return d;
        }
    }
    0
}
