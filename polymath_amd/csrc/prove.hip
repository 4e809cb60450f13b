// The three GPU phases of create_proof_with_assignment (/root/reference/src/prover.rs:66-237),
// split at its two Fiat-Shamir calls (:126, :189).  Everything between the FFI and the three
// output points stays in HBM.
//
// The reference materialises U and W densely (prover.rs:87-96, O(n*M)); here the same u_evals /
// w_evals come from the closed form of SURVEY.md App. A in O(nnz): one lane per R1CS row.
#include <cstring>

#include <thread>

#include "internal.h"
#include "fq28.cuh"
#include "prove_common.cuh"

namespace pm {

template <class P>
__global__ void k_witness_rows(CsrDev A, CsrDev B, CsrDev Cm, const Fp<P> *xw, Fp<P> *ue, Fp<P> *we, Fp<P> *y,
                               uint64_t m0, uint64_t nr) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nr) return;
    Fp<P> az = csr_row_dot<P>(A.rowptr, A.col, A.val, xw, r);
    Fp<P> bz = csr_row_dot<P>(B.rowptr, B.col, B.val, xw, r);
    Fp<P> cz = csr_row_dot<P>(Cm.rowptr, Cm.col, Cm.val, xw, r);
    Fp<P> d = sub<P>(az, bz), d2 = sqr<P>(d);
    Fp<P> c4 = dbl<P>(dbl<P>(cz));
    y[m0 + r] = d2;
    ue[2 * m0 + r] = add<P>(az, bz);
    we[2 * m0 + r] = add<P>(c4, d2);
    ue[2 * m0 + nr + r] = d;
    we[2 * m0 + nr + r] = d2;
}

// rows < 2 m0 (public-input rows), the x||w prefix of z_tail, and zero padding rows >= 2(m0+nr)
template <class P>
__global__ void k_witness_head(const Fp<P> *xw, Fp<P> *ue, Fp<P> *we, Fp<P> *ztail, uint64_t m0, uint64_t mw,
                               uint64_t nr, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Fp<P> one = Fp<P>::one();
    if (i < m0 + mw) ztail[i] = xw[i];
    Fp<P> *y = ztail + m0 + mw;
    if (i < m0) {
        Fp<P> xi = xw[i];
        Fp<P> omx = sub<P>(one, xi), yi = i ? sqr<P>(omx) : Fp<P>::zero();
        y[i] = yi;
        if (i == 0) {
            ue[0] = dbl<P>(one);
            we[0] = dbl<P>(dbl<P>(one));
            ue[m0] = Fp<P>::zero();
            we[m0] = Fp<P>::zero();
        } else {
            ue[i] = add<P>(one, xi);
            we[i] = add<P>(dbl<P>(dbl<P>(xi)), yi);
            ue[m0 + i] = omx;
            we[m0 + i] = yi;
        }
    }
    uint64_t rows = 2 * (m0 + nr);
    if (i >= rows && i < n) {
        ue[i] = Fp<P>::zero();
        we[i] = Fp<P>::zero();
    }
}

// flags: bit0 = (Uz)^2 != Wz somewhere (== rem != 0, prover.rs:108), bit1 = h[n-1] != 0 (deg h > n-2),
// bit2 = h has a non-zero coefficient (cleared means h == 0, prover.rs:107), bit3 = division remainder != 0
template <class P>
__global__ void k_check_sap(const Fp<P> *ue, const Fp<P> *we, uint64_t n, unsigned *flags) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!sqr<P>(ue[i]).eq(we[i])) atomicOr(flags, 1u);
}

template <class P>
__global__ void k_copy_zero_head(const Fp<P> *src, Fp<P> *dst, uint64_t n, uint64_t zero_rows) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i < zero_rows ? Fp<P>::zero() : src[i];
}

// Coefficients of the witness-only part of u (N5, prover.rs:160-162) without a transform: its evaluations are
// u's with the first `head` rows zeroed, so  wit_u = u - iNTT(head rows)  and the iNTT of a `head`-sparse
// vector is a direct sum:  wit_u[k] = u[k] - n^-1 sum_{j < head} ue[j] w^(-jk)   (head = 2 m0, Horner in w^-k).
// winv[k] = w^-k for k < n/2 (the inverse twiddle table); w^-(k + n/2) = -w^-k.
template <class P>
__global__ void k_wit_u_sparse(const Fp<P> *u, const Fp<P> *ue, const Fp<P> *winv, Fp<P> ninv, uint64_t n, unsigned head,
                               Fp<P> *wit_u) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t half = n >> 1;
    Fp<P> wk = winv[k < half ? k : k - half];
    if (k >= half) wk = neg<P>(wk);
    Fp<P> s = ue[head - 1];
    for (int j = (int)head - 2; j >= 0; --j) s = add<P>(mul<P>(s, wk), ue[j]);
    wit_u[k] = sub<P>(u[k], mul<P>(s, ninv));
}

template <class P>
__global__ void k_pad_copy(const Fp<P> *src, Fp<P> *dst, uint64_t n_src, uint64_t n_dst) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_dst) dst[i] = i < n_src ? src[i] : Fp<P>::zero();
}

// u^2 without a size-2n transform.  Once (Uz)^2 == Wz holds on the domain (k_check_sap), u^2 = w  (mod X^n - 1),
// i.e. lo + hi = w for u^2 = lo + X^n hi.  The negacyclic product neg = u^2 mod (X^n + 1) = lo - hi comes from ONE
// size-n transform pair on the twisted input u_k psi^k (psi = omega_2n):  lo = (w + neg) / 2, hi = (w - neg) / 2.
// Same coefficients as square_polynomial (prover.rs:315-328) at half the NTT work.
template <class P>
__global__ void k_twist(const Fp<P> *u, const Fp<P> *psi_pow, Fp<P> *out, uint64_t n) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = mul<P>(u[k], psi_pow[k]);
}
template <class P>
__global__ void k_untwist_combine(const Fp<P> *neg_tw, const Fp<P> *psi_inv_pow, const Fp<P> *w, Fp<P> *u2, uint64_t n, Fp<P> half) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Fp<P> neg = mul<P>(neg_tw[k], psi_inv_pow[k]), wk = w[k];
    u2[k] = mul<P>(add<P>(wk, neg), half);
    u2[n + k] = mul<P>(sub<P>(wk, neg), half);
}

template <class P>
__global__ void k_square(Fp<P> *a, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = sqr<P>(a[i]);
}

// Scalar vectors of the two phase-1 MSMs, laid out to match the pk's base concatenation:
//   sc_c = [ z_tail (Lz) | h (n-1) | 2 r_a(X) u(X) (n+1) | r_a^2 (3) | r_a (2) ]     prover.rs:118-123,340-357
//   sc_a = [ u (n) | 0 | r_a (2) ]                                                   prover.rs:330-338
// h = u2[n .. 2n-1)  (divide_by_vanishing_poly, prover.rs:105); also the degree checks.
template <class P>
__global__ void k_phase1_scalars(const Fp<P> *u, const Fp<P> *u2, const Fp<P> *ra /*r0,r1*/, Fp<P> *sc_c_after_z,
                                 Fp<P> *sc_a, uint64_t n, unsigned *flags) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Fp<P> r0 = ra[0], r1 = ra[1];
    if (i < n) {
        Fp<P> hi = u2[n + i];
        if (i < n - 1) {
            sc_c_after_z[i] = hi;
            // "h is not identically zero": nearly every lane sees it, and one atomic per wave on ONE word (the compiler already folds the lanes)
            // is 131 K serialised L2 operations -- 0.3 of this kernel's 0.39 ms.  A wave that reads the bit as set has nothing to add.
            if (!hi.is_zero() && !(*(const volatile unsigned *)flags & 4u)) atomicOr(flags, 4u);
        } else if (!hi.is_zero()) {
            atomicOr(flags, 2u);
        }
        if (sc_a) sc_a[i] = u[i];
    }
    if (i <= n) {  // coefficient i of 2 r_a(X) u(X) = 2 (r0 u_i + r1 u_{i-1})
        Fp<P> t = Fp<P>::zero();
        if (i < n) t = mul<P>(r0, u[i]);
        if (i > 0) t = add<P>(t, mul<P>(r1, u[i - 1]));
        sc_c_after_z[(n - 1) + i] = dbl<P>(t);
    }
    if (i == 0) {
        Fp<P> *tail = sc_c_after_z + (n - 1) + (n + 1);
        tail[0] = sqr<P>(r0);
        tail[1] = dbl<P>(mul<P>(r0, r1));
        tail[2] = sqr<P>(r1);
        tail[3] = r0;
        tail[4] = r1;
        if (sc_a) {
            sc_a[n] = Fp<P>::zero();
            sc_a[n + 1] = r0;
            sc_a[n + 2] = r1;
        }
    }
}

// sc_a alone, as soon as u is known: lets the [a]_1 MSM start while the rest of phase 1 still runs
template <class P>
__global__ void k_sc_a(const Fp<P> *u, const Fp<P> *ra, Fp<P> *sc_a, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sc_a[i] = u[i];
    if (i == 0) {
        sc_a[n] = Fp<P>::zero();
        sc_a[n + 1] = ra[0];
        sc_a[n + 2] = ra[1];
    }
}

// ------------------------------------------------------------------------ Horner (phase 2)
// u(x1) = sum_k u_k x1^k.  Lane t owns L consecutive coefficients: local Horner, times x1^(tL),
// workgroup LDS tree sum; one partial per workgroup, summed by the last tiny launch.
template <class P>
__global__ __launch_bounds__(256) void k_horner_partial(const Fp<P> *u, uint64_t n, Fp<P> x1, unsigned L, Fp<P> *partials) {
    __shared__ Fp<P> sh[256];
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = t * L, hi = lo + L;
    if (hi > n) hi = n;
    Fp<P> acc = Fp<P>::zero();
    if (lo < n) {
        for (uint64_t k = hi; k-- > lo;) acc = add<P>(mul<P>(acc, x1), u[k]);
        acc = mul<P>(acc, pow_u64<P>(x1, lo));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = add<P>(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

template <class P>
__global__ __launch_bounds__(256) void k_sum_small(const Fp<P> *in, unsigned count, Fp<P> *out) {   // one workgroup
    __shared__ Fp<P> sh[256];
    Fp<P> acc = Fp<P>::zero();
    for (unsigned i = threadIdx.x; i < count; i += 256) acc = add<P>(acc, in[i]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = add<P>(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// ------------------------------------------------------------- numerator + division (phase 3)
// Numerator of prover.rs:211-216, multiplied through by Y^-gamma = X^(5 sigma), as a function of
// the coefficient index k (never materialised):
//   [0,2)            x2 r_a
//   2s + [0,3)       r_a + x2 r_a^2
//   3s + [0,n)       x2 * witness_u                          (prover.rs:168-171)
//   5s + [0,n+1)     u + x2 * 2 r_a u  - (a + x2 c) at +0    (prover.rs:145-152,366-368,196-197)
//   8s + [0,2n-1)    x2 * u^2     (witness_w + (u^2 - w) = u^2: N6 == N2)
template <class P>
__device__ __forceinline__ Fp<P> numerator_at(uint64_t k, const NumParams &np, const NumConsts<P> &nc, const Fp<P> *u,
                                              const Fp<P> *wit_u, const Fp<P> *u2) {
    const uint64_t s = np.sigma, n = np.n;
    if (k >= 8 * s) {
        uint64_t i = k - 8 * s;
        return i < 2 * n - 1 ? mul<P>(nc.x2, u2[i]) : Fp<P>::zero();
    }
    if (k >= 5 * s) {
        uint64_t i = k - 5 * s;
        if (i > n) return Fp<P>::zero();
        Fp<P> t = Fp<P>::zero();
        if (i < n) t = add<P>(u[i], mul<P>(nc.two_x2_r0, u[i]));
        if (i > 0) t = add<P>(t, mul<P>(nc.two_x2_r1, u[i - 1]));
        if (i == 0) t = add<P>(t, nc.minus_const);
        return t;
    }
    if (k >= 3 * s) {
        uint64_t i = k - 3 * s;
        return i < n ? mul<P>(nc.x2, wit_u[i]) : Fp<P>::zero();
    }
    if (k >= 2 * s) {
        uint64_t i = k - 2 * s;
        return i < 3 ? nc.b2[i] : Fp<P>::zero();
    }
    if (k == 0) return nc.x2r0;
    if (k == 1) return nc.x2r1;
    return Fp<P>::zero();
}

// Round 5: the two kernels that walk the numerator -- k_div_level0 and k_div_expand0, 0.77 of the scan's 0.89 ms -- run their Horner
// chains in REDUCED RADIX (fq28.cuh: 9 limbs of 29 bits), as the transform tiles do: the multipliers x1, x2, 2 x2 r_a are handed over
// in the internal Montgomery form (value 2^261 mod p), so that f28_mul(standard-form value, multiplier) is again a standard-form
// value, sums are limb-wise, and a dense canonical element is only rebuilt where one is STORED.  The dense product mul<P> unpacks both
// operands and shifts / reduces / packs its result every time: ~370 instructions for 162 multiplier operations against ~200.
// Values are the same residues, the stored elements the same canonical words.
template <class RR>
struct NumMul28 {
    F28<RR> x1, x2, two_x2_r0, two_x2_r1;      // internal form, canonical, tight limbs
};
template <class P>
static inline NumMul28<typename Radix28<P>::RR> make_num_mul28(const Fp<P> &x1, const NumConsts<P> &nc) {
    typedef typename Radix28<P>::RR RR;
    Fp<P> k;
    for (int i = 0; i < P::N; ++i) k.l[i] = RR::STD2INT[i];
    NumMul28<RR> m;
    m.x1 = f28_unpack<RR>(mul<P>(x1, k).l);
    m.x2 = f28_unpack<RR>(mul<P>(nc.x2, k).l);
    m.two_x2_r0 = f28_unpack<RR>(mul<P>(nc.two_x2_r0, k).l);
    m.two_x2_r1 = f28_unpack<RR>(mul<P>(nc.two_x2_r1, k).l);
    return m;
}
// numerator_at on reduced-radix limbs: a LAZY standard-form value, limbs < 4 * 2^29, value < 7p (u + two products + a constant)
template <class P, class RR>
__device__ __forceinline__ F28<RR> numerator28_at(uint64_t k, const NumParams &np, const NumConsts<P> &nc, const NumMul28<RR> &m, const Fp<P> *u,
                                                  const Fp<P> *wit_u, const Fp<P> *u2) {
    const uint64_t s = np.sigma, n = np.n;
    if (k >= 8 * s) {
        const uint64_t i = k - 8 * s;
        return i < 2 * n - 1 ? f28_mul<RR>(f28_unpack<RR>(u2[i].l), m.x2) : f28_zero<RR>();
    }
    if (k >= 5 * s) {
        const uint64_t i = k - 5 * s;
        if (i > n) return f28_zero<RR>();
        F28<RR> t = f28_zero<RR>();
        if (i < n) {
            const F28<RR> ui = f28_unpack<RR>(u[i].l);
            t = f28_add<RR>(ui, f28_mul<RR>(ui, m.two_x2_r0));
        }
        if (i > 0) t = f28_add<RR>(t, f28_mul<RR>(f28_unpack<RR>(u[i - 1].l), m.two_x2_r1));
        if (i == 0) t = f28_add<RR>(t, f28_unpack<RR>(nc.minus_const.l));
        return t;
    }
    if (k >= 3 * s) {
        const uint64_t i = k - 3 * s;
        return i < n ? f28_mul<RR>(f28_unpack<RR>(wit_u[i].l), m.x2) : f28_zero<RR>();
    }
    if (k >= 2 * s) {
        const uint64_t i = k - 2 * s;
        return i < 3 ? f28_unpack<RR>(nc.b2[i].l) : f28_zero<RR>();
    }
    if (k == 0) return f28_unpack<RR>(nc.x2r0.l);
    if (k == 1) return f28_unpack<RR>(nc.x2r1.l);
    return f28_zero<RR>();
}
// one Horner step: acc (tight limbs, value < 9p) -> acc x1 + N_k: the product is < 2p, the sum < 9p; carries propagated so that the
// next product's columns stay inside 64 bits (9 * 2^29 * 2^29 * 2 < 2^63)
template <class P, class RR>
__device__ __forceinline__ F28<RR> horner28_step(const F28<RR> &acc, uint64_t k, const NumParams &np, const NumConsts<P> &nc, const NumMul28<RR> &m,
                                                 const Fp<P> *u, const Fp<P> *wit_u, const Fp<P> *u2) {
    return f28_weak_norm<RR>(f28_add<RR>(f28_mul<RR>(acc, m.x1), numerator28_at<P, RR>(k, np, nc, m, u, wit_u, u2)));
}

// Synthetic division by (X - x1): H_k = N_k + x1 H_{k+1}, quotient q_{k-1} = H_k, remainder H_0.
// Level 0: lane t owns coefficients [tL, tL+L): V_t = local Horner value (carry-in 0).
// Then carry_t = V_t + x1^L carry_{t+1} is the same recurrence on V with multiplier x1^L: recurse.
template <class P>
__global__ void k_div_level0(NumParams np, NumConsts<P> nc, NumMul28<typename Radix28<P>::RR> m28, const Fp<P> *u, const Fp<P> *wit_u, const Fp<P> *u2,
                             unsigned L, uint64_t nchunks, Fp<P> *V) {
    typedef typename Radix28<P>::RR RR;
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    uint64_t lo = t * L, hi = lo + L;
    if (hi > np.len) hi = np.len;
    // 6n of the 10n indices lie in the zero stretches between the blocks (numerator_at's table): a chunk inside one has V = 0
    {
        const uint64_t s = np.sigma, n = np.n;
        const bool zeros = (lo >= 2 && hi <= 2 * s) || (lo >= 2 * s + 3 && hi <= 3 * s) || (lo >= 3 * s + n && hi <= 5 * s) ||
                           (lo >= 5 * s + n + 1 && hi <= 8 * s);
        if (zeros) { V[t] = Fp<P>::zero(); return; }
    }
    F28<RR> acc = f28_zero<RR>();
    for (uint64_t k = hi; k-- > lo;) acc = horner28_step<P, RR>(acc, k, np, nc, m28, u, wit_u, u2);
    Fp<P> out;
    f28_pack_canonical<RR>(f28_canonical_lazy<RR, 3>(acc), out.l);       // < 9p < 16p
    V[t] = out;
}

template <class P>
__global__ void k_div_levelN(const Fp<P> *in, uint64_t count, Fp<P> xp, unsigned L, uint64_t nchunks, Fp<P> *V) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    uint64_t lo = t * L, hi = lo + L;
    if (hi > count) hi = count;
    Fp<P> acc = Fp<P>::zero();
    for (uint64_t k = hi; k-- > lo;) acc = add<P>(mul<P>(acc, xp), in[k]);
    V[t] = acc;
}

// top level: sequential over <= L values; out[k] = H_k (suffix value INCLUDING element k), out[count] = 0
template <class P>
__global__ void k_div_top(const Fp<P> *in, uint64_t count, Fp<P> xp, Fp<P> *H) {
    if (threadIdx.x || blockIdx.x) return;
    Fp<P> acc = Fp<P>::zero();
    H[count] = acc;
    for (uint64_t k = count; k-- > 0;) {
        acc = add<P>(mul<P>(acc, xp), in[k]);
        H[k] = acc;
    }
}

// Expand one level down: given Hup[t] = true suffix value at the START of chunk t (and Hup[nchunks] = 0),
// recompute chunk t of `in` with carry-in Hup[t+1] and write H[k] for every k in the chunk.
template <class P>
__global__ void k_div_expandN(const Fp<P> *in, uint64_t count, Fp<P> xp, unsigned L, uint64_t nchunks, const Fp<P> *Hup,
                              Fp<P> *H) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    uint64_t lo = t * L, hi = lo + L;
    if (hi > count) hi = count;
    Fp<P> acc = Hup[t + 1];
    for (uint64_t k = hi; k-- > lo;) {
        acc = add<P>(mul<P>(acc, xp), in[k]);
        H[k] = acc;
    }
    if (t == nchunks - 1) H[count] = Fp<P>::zero();
}

// Level 0 expansion writes the quotient: q_{k-1} = H_k for k >= 1; H_0 is the remainder.
template <class P>
__global__ void k_div_expand0(NumParams np, NumConsts<P> nc, NumMul28<typename Radix28<P>::RR> m28, const Fp<P> *u, const Fp<P> *wit_u, const Fp<P> *u2,
                              unsigned L, uint64_t nchunks, const Fp<P> *Hup, Fp<P> *q, unsigned *flags) {
    typedef typename Radix28<P>::RR RR;
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    uint64_t lo = t * L, hi = lo + L;
    if (hi > np.len) hi = np.len;
    F28<RR> acc = f28_unpack<RR>(Hup[t + 1].l);
    {   // inside a zero stretch (k_div_level0) the quotient is a geometric tail: one product per coefficient, no table walk
        const uint64_t s = np.sigma, n = np.n;
        const bool zeros = (lo >= 2 && hi <= 2 * s) || (lo >= 2 * s + 3 && hi <= 3 * s) || (lo >= 3 * s + n && hi <= 5 * s) ||
                           (lo >= 5 * s + n + 1 && hi <= 8 * s);
        if (zeros) {
            for (uint64_t k = hi; k-- > lo;) {
                acc = f28_mul<RR>(acc, m28.x1);                     // < 2p, tight: one conditional subtraction on the way out
                Fp<P> out;
                f28_pack_reduced<RR>(acc, out.l);
                q[k - 1] = out;        // lo >= 2 here
            }
            return;
        }
    }
    for (uint64_t k = hi; k-- > lo;) {
        acc = f28_canonical_lazy<RR, 3>(horner28_step<P, RR>(acc, k, np, nc, m28, u, wit_u, u2));   // the stored element: canonical
        Fp<P> out;
        f28_pack_canonical<RR>(acc, out.l);
        if (k > 0) q[k - 1] = out;
        else if (!out.is_zero()) atomicOr(flags, 8u);  // rem != 0, prover.rs:221
    }
}

// MSM `which` over THIS rank's resident pairs.  d_scalars: the rank's scalars in the order of pk->pieces[which].
template <class C>
int msm_resident(pm_ctx *ctx, const pm_pk *pk, int which, const Fp<typename C::FrP> *d_scalars, uint64_t *out_xy, int *out_inf) {
    const Affine<C> *bases = (const Affine<C> *)pk->d_bases + pk->res_dev_off[which];
    Affine<C> r;
    int inf = 1;
    if (pk->tables[which].c) {   // this MSM's own window tables (window 0 = its resident pairs), or the wide mode on the plain array
        MsmTables tb = pk->tables[which];
        tb.base_index = 0;
        PM_TRY(msm_run<C>(ctx, tb.wide ? bases : (const Affine<C> *)nullptr, d_scalars, (size_t)pk->res_cnt[which], &r, &inf, &tb));
    } else {
        PM_TRY(msm_run<C>(ctx, bases, d_scalars, (size_t)pk->res_cnt[which], &r, &inf));
    }
    store_affine_host<C>(r, inf, out_xy, out_inf);
    return PM_OK;
}

// pairs [lo, lo + count) of MSM `which` (count = 0: all of them); d_scalars points at the MSM's FIRST scalar
template <class C>
int msm_resident_begin(pm_ctx *ctx, const pm_pk *pk, int which, const Fp<typename C::FrP> *d_scalars, uint64_t lo, uint64_t count) {
    const Affine<C> *bases = (const Affine<C> *)pk->d_bases + pk->res_dev_off[which];
    if (count == 0) { lo = 0; count = pk->res_cnt[which]; }
    if (lo + count > pk->res_cnt[which]) return PM_ERR_INVALID_ARG;
    if (pk->tables[which].c) {
        MsmTables tb = pk->tables[which];
        tb.base_index = (size_t)lo;
        return msm_begin<C>(ctx, tb.wide ? bases : (const Affine<C> *)nullptr, d_scalars + lo, (size_t)count, &tb);
    }
    return msm_begin<C>(ctx, bases + lo, d_scalars + lo, (size_t)count, (const MsmTables *)nullptr);
}
template <class C>
int msm_resident_end(pm_ctx *ctx, uint64_t *out_xy, int *out_inf) {
    Affine<C> r;
    int inf = 1;
    PM_TRY(msm_end<C>(ctx, &r, &inf));
    store_affine_host<C>(r, inf, out_xy, out_inf);
    return PM_OK;
}

// PM_SHARD_PAIRS: the scalar vector is the whole logical one; this rank's pairs are the contiguous range at res_lo
template <class C>
static int msm_shard(pm_ctx *ctx, const pm_pk *pk, int which, const Fp<typename C::FrP> *d_scalars, uint64_t *out_xy,
                     int *out_inf) {
    return msm_resident<C>(ctx, pk, which, d_scalars + pk->res_lo[which], out_xy, out_inf);
}

// ------------------------------------------------------------------------------- phase 1
template <class C>
int prove_phase1_impl(pm_ctx *ctx, const pm_pk *pk, const uint64_t *x, const uint64_t *w, const uint64_t *r_a,
                      uint64_t *a_xy, int *a_inf, uint64_t *c_xy, int *c_inf, bool assignment_on_device) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    const uint64_t n = pk->n, m0 = pk->m0, mw = pk->mw, nr = pk->nr;
    const uint64_t Lz = 2 * m0 + mw + nr;  // |z_tail| = M - m0
    if (pk->log_n + 1 > (unsigned)C::TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;  // prover.rs:317
    hipStream_t st = ctx->stream;
    if (!ctx->keep_timings) timing_reset(ctx);
    ctx->pk = pk;
    ctx->phase = 0;
    TimingGuard timing_guard{ctx};
    StageTimer t_phase(ctx, T_PHASE);
    const uint64_t len_c = Lz + (n - 1) + (n + 1) + 3 + 2, len_a = n + 3;
    PM_HIP(ctx, ctx->xw.reserve((m0 + mw) * sizeof(Fr)));
    PM_HIP(ctx, ctx->ue.reserve(n * sizeof(Fr)));
    PM_HIP(ctx, ctx->we.reserve(n * sizeof(Fr)));
    PM_HIP(ctx, ctx->u.reserve(n * sizeof(Fr)));
    PM_HIP(ctx, ctx->w.reserve(n * sizeof(Fr)));
    PM_HIP(ctx, ctx->wit_u.reserve(n * sizeof(Fr)));
    PM_HIP(ctx, ctx->u2.reserve(2 * n * sizeof(Fr)));
    PM_HIP(ctx, ctx->sc_c.reserve(len_c * sizeof(Fr)));
    PM_HIP(ctx, ctx->sc_a.reserve(len_a * sizeof(Fr)));
    PM_HIP(ctx, ctx->ra.reserve(2 * sizeof(Fr)));
    PM_HIP(ctx, ctx->flags.reserve(64));
    Fr *xw = ctx->xw.as<Fr>(), *ue = ctx->ue.as<Fr>(), *we = ctx->we.as<Fr>(), *u = ctx->u.as<Fr>(), *wv = ctx->w.as<Fr>();
    Fr *wit_u = ctx->wit_u.as<Fr>(), *u2 = ctx->u2.as<Fr>(), *sc_c = ctx->sc_c.as<Fr>(), *sc_a = ctx->sc_a.as<Fr>();
    Fr *ra = ctx->ra.as<Fr>();
    unsigned *flags = ctx->flags.as<unsigned>();
    PM_HIP(ctx, hipMemsetAsync(flags, 0, 64, st));
    const hipMemcpyKind kind = assignment_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    PM_HIP(ctx, hipMemcpyAsync(xw, x, m0 * sizeof(Fr), kind, st));
    if (mw) PM_HIP(ctx, hipMemcpyAsync(xw + m0, w, mw * sizeof(Fr), kind, st));
    PM_HIP(ctx, hipMemcpyAsync(ra, r_a, 2 * sizeof(Fr), hipMemcpyHostToDevice, st));
    memcpy(ctx->ra_host, r_a, 2 * sizeof(Fr));
    {
        StageTimer t(ctx, T_WITNESS_MAP);
        CsrDev A{pk->d_rowptr[0], pk->d_col[0], pk->d_val[0]}, B{pk->d_rowptr[1], pk->d_col[1], pk->d_val[1]},
            Cm{pk->d_rowptr[2], pk->d_col[2], pk->d_val[2]};
        Fr *ztail = sc_c;  // z_tail is the head of the c-MSM scalar vector
        uint64_t head = n > m0 + mw ? n : m0 + mw;
        hipLaunchKernelGGL(k_witness_head<P>, dim3(nblk(head)), dim3(256), 0, st, xw, ue, we, ztail, m0, mw, nr, n);
        PM_HIP(ctx, hipGetLastError());
        if (nr) {
            hipLaunchKernelGGL(k_witness_rows<P>, dim3(nblk(nr)), dim3(256), 0, st, A, B, Cm, xw, ue, we, ztail + m0 + mw,
                               m0, nr);
            PM_HIP(ctx, hipGetLastError());
        }
        // rem == 0 of prover.rs:108  <=>  (Uz)^2 == Wz on the whole domain
        hipLaunchKernelGGL(k_check_sap<P>, dim3(nblk(n)), dim3(256), 0, st, ue, we, n, flags);
        PM_HIP(ctx, hipGetLastError());
    }
    // N1, N2, N5 (prover.rs:94,96,160-162): coefficients of u, w and of the witness-only U part
    PM_HIP(ctx, hipMemcpyAsync(u, ue, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    PM_HIP(ctx, hipMemcpyAsync(wv, we, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    PM_TRY(ntt_run<C>(ctx, u, pk->log_n, true));
    // [a]_1 = M1 + M2 needs only u and r_a, and is independent of [c]_1 = M7 + M6 + M3 + M4 + M5: it runs on a helper
    // context (own stream and workspace) from a second host thread, started HERE -- its sort, bucket reduction and
    // host finish are dependent chains that leave the chip mostly idle, and hide under the remaining transforms and
    // the larger MSM's accumulation.  The helper stream waits on an event, the host does not.  PM_OPT_MSM_OVERLAP = 0
    // runs the two MSMs back to back after the checks.
    const bool overlap = ctx->opt.v[PM_OPT_MSM_OVERLAP] != 0;
    pm_ctx *const aux_ctx = overlap ? ctx_aux(ctx) : nullptr;
    int st_a = PM_OK;   // written by the helper thread: declared before the joiner so that it outlives the join
    struct Joiner {     // an early error return must not leave the helper job running into freed stack variables
        pm_worker *w;
        bool pending;
        void join() { if (pending) { w->wait(); pending = false; } }
        ~Joiner() { join(); }
    } helper{&ctx->worker, false};
    // (Enqueueing [a]_1 late, next to [c]_1 after the transforms, measured neutral on one GPU: profiles/r03_a_late_msm_single_gpu_ab.txt.)
    bool a_early = aux_ctx != nullptr;
    if (a_early) {
        pm_ctx *aux = ctx->aux;
        hipLaunchKernelGGL(k_sc_a<P>, dim3(nblk(n)), dim3(256), 0, st, u, ra, sc_a, n);
        PM_HIP(ctx, hipGetLastError());
        PM_HIP(ctx, hipEventRecord(ctx->ev_sc_a, st));
        timing_reset_aux(ctx, aux);
        {
            helper.pending = ctx->worker.submit([&, aux] {
                if (hipSetDevice(aux->device) != hipSuccess || hipStreamWaitEvent(aux->stream, ctx->ev_sc_a, 0) != hipSuccess) {
                    st_a = PM_ERR_HIP;
                    aux->err = "helper stream setup failed";
                    return;
                }
                try {
                    st_a = msm_shard<C>(aux, pk, 0, sc_a, a_xy, a_inf);
                } catch (const std::exception &e) {   // bad_alloc in the MSM's host vectors: a status, never a terminate
                    st_a = PM_ERR_STATE;
                    aux->err = e.what();
                }
                timing_flush(aux);
            });
            if (!helper.pending) a_early = false;   // no helper thread: the two MSMs run back to back below (sc_a is already filled)
        }
    }
    PM_TRY(ntt_run<C>(ctx, wv, pk->log_n, true));
    if (2 * m0 <= 16 && pk->log_n >= 1) {   // few public inputs: the sparse sum beats a fifth transform
        const Fr *winv = nullptr;
        PM_TRY(twiddles_get<C>(ctx, pk->log_n, true, &winv));
        StageTimer t(ctx, T_NTT);
        hipLaunchKernelGGL(k_wit_u_sparse<P>, dim3(nblk(n)), dim3(256), 0, st, u, ue, winv, inverse<P>(from_u64<P>(n)), n,
                           (unsigned)(2 * m0), wit_u);
        PM_HIP(ctx, hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_copy_zero_head<P>, dim3(nblk(n)), dim3(256), 0, st, ue, wit_u, n, 2 * m0);
        PM_HIP(ctx, hipGetLastError());
        PM_TRY(ntt_run<C>(ctx, wit_u, pk->log_n, true));
    }
    // square_polynomial (prover.rs:315-328) via the negacyclic half (see k_twist); the 2n-point domain of
    // the reference must still exist (checked above), its root psi = omega_2n is the twist.
    {
        const Fr *psi = nullptr, *psi_inv = nullptr;
        PM_TRY(twiddles_get<C>(ctx, pk->log_n + 1, false, &psi));
        PM_TRY(twiddles_get<C>(ctx, pk->log_n + 1, true, &psi_inv));
        PM_HIP(ctx, ctx->scratch.reserve(n * sizeof(Fr)));
        Fr *tmp = ctx->scratch.as<Fr>();
        hipLaunchKernelGGL(k_twist<P>, dim3(nblk(n)), dim3(256), 0, st, u, psi, tmp, n);
        PM_HIP(ctx, hipGetLastError());
        PM_TRY(ntt_run<C>(ctx, tmp, pk->log_n, false));
        hipLaunchKernelGGL(k_square<P>, dim3(nblk(n)), dim3(256), 0, st, tmp, n);
        PM_HIP(ctx, hipGetLastError());
        PM_TRY(ntt_run<C>(ctx, tmp, pk->log_n, true));
        Fr half = inverse<P>(from_u64<P>(2));
        hipLaunchKernelGGL(k_untwist_combine<P>, dim3(nblk(n)), dim3(256), 0, st, tmp, psi_inv, wv, u2, n, half);
        PM_HIP(ctx, hipGetLastError());
    }
    {
        StageTimer t(ctx, T_POLY);
        hipLaunchKernelGGL(k_phase1_scalars<P>, dim3(nblk(n + 1)), dim3(256), 0, st, u, u2, ra, sc_c + Lz, a_early ? (Fr *)nullptr : sc_a, n,
                           flags);
        PM_HIP(ctx, hipGetLastError());
    }
    // The status flags ride behind the MSMs (round 4): they land in pinned staging and are read after the MSM's own final
    // synchronisation, so the host does not wait here and the sort's fifteen launches are enqueued while the transforms still run.
    // An unsatisfied witness (prover.rs:107-108) is reported after the MSMs it no longer stops -- the rare path pays, not the proof.
    if (!ctx_pinned(ctx)) { ctx->err = "pinned staging allocation failed"; return PM_ERR_HIP; }
    volatile unsigned *hflags_p = (volatile unsigned *)((uint8_t *)ctx->h_pinned + PINNED_SLOTS_BYTES);
    *hflags_p = 0;
    PM_HIP(ctx, hipMemcpyAsync((void *)hflags_p, flags, 4, hipMemcpyDeviceToHost, st));
    if (a_early) {
        const int st_c = msm_shard<C>(ctx, pk, 1, sc_c, c_xy, c_inf);
        helper.join();
        timing_absorb_aux(ctx, ctx->aux);
        // always synchronise before the staged flags are read: with both MSMs failed the copy may still be in flight
        // and the host-written 0 would read as a degree-bound violation instead of the MSMs' own status
        const hipError_t e_sync = hipStreamSynchronize(st);
        if (st_a != PM_OK && st_c != PM_OK) { ctx->err = ctx->aux->err; return st_a; }
        PM_HIP(ctx, e_sync);
        const unsigned hf = *hflags_p;
        if (hf & 1u) return PM_ERR_REMAINDER_NONZERO;                 // prover.rs:108
        if ((hf & 2u) || !(hf & 4u)) return PM_ERR_DEGREE_BOUND;       // prover.rs:107
        if (st_a != PM_OK) { ctx->err = ctx->aux->err; return st_a; }
        PM_TRY(st_c);
    } else {
        const int s_a = msm_shard<C>(ctx, pk, 0, sc_a, a_xy, a_inf);
        const int s_c = s_a == PM_OK ? msm_shard<C>(ctx, pk, 1, sc_c, c_xy, c_inf) : (int)PM_OK;
        const hipError_t e_sync = hipStreamSynchronize(st);
        PM_TRY(s_a);                                                  // an MSM's own failure first: the flags may not have landed
        PM_HIP(ctx, e_sync);
        const unsigned hf = *hflags_p;
        if (hf & 1u) return PM_ERR_REMAINDER_NONZERO;                 // prover.rs:108
        if ((hf & 2u) || !(hf & 4u)) return PM_ERR_DEGREE_BOUND;       // prover.rs:107
        PM_TRY(s_c);
    }
    t_phase.stop();
    timing_flush(ctx);
    ctx->phase = 1;
    return PM_OK;
}

// ------------------------------------------------------------------------------- phase 2
template <class C>
int prove_phase2_impl(pm_ctx *ctx, const uint64_t *x1_in, uint64_t *u_at_x1) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    if (ctx->phase < 1 || !ctx->pk) return PM_ERR_STATE;
    const uint64_t n = ctx->pk->n;
    hipStream_t st = ctx->stream;
    Fr x1 = load_fr<P>(x1_in);
    const unsigned L = 16;
    uint64_t lanes = (n + L - 1) / L;
    unsigned blocks = nblk(lanes);
    PM_HIP(ctx, ctx->scratch.reserve(((size_t)blocks + 1) * sizeof(Fr)));
    Fr *part = ctx->scratch.as<Fr>();
    hipLaunchKernelGGL(k_horner_partial<P>, dim3(blocks), dim3(256), 0, st, ctx->u.as<Fr>(), n, x1, L, part);
    PM_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_sum_small<P>, dim3(1), dim3(256), 0, st, part, blocks, part + blocks);
    PM_HIP(ctx, hipGetLastError());
    PM_HIP(ctx, hipMemcpyAsync(u_at_x1, part + blocks, sizeof(Fr), hipMemcpyDeviceToHost, st));
    PM_HIP(ctx, hipStreamSynchronize(st));
    ctx->phase = 2;
    return PM_OK;
}

// ------------------------------------------------------------------------------- phase 3
template <class C>
int prove_phase3_impl(pm_ctx *ctx, const uint64_t *x1_in, const uint64_t *x2_in, const uint64_t *a_in,
                      const uint64_t *c_in, uint64_t *d_xy, int *d_inf) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    if (ctx->phase < 1 || !ctx->pk) return PM_ERR_STATE;
    const pm_pk *pk = ctx->pk;
    hipStream_t st = ctx->stream;
    if (!ctx->keep_timings) timing_reset(ctx);
    TimingGuard timing_guard{ctx};
    StageTimer t_phase(ctx, T_PHASE);
    const uint64_t n = pk->n, sigma = pk->sigma;
    Fr x1 = load_fr<P>(x1_in), x2 = load_fr<P>(x2_in), a_at = load_fr<P>(a_in), c_at = load_fr<P>(c_in);
    Fr rah[2];
    memcpy(rah, ctx->ra_host, sizeof(rah));          // phase 1 kept the host copy of r_a: no device read-back, no synchronisation here
    NumParams np{n, sigma, 8 * sigma + 2 * n - 1};
    const NumConsts<P> nc = make_num_consts<P>(x2, rah, a_at, c_at);
    const NumMul28<typename Radix28<P>::RR> m28 = make_num_mul28<P>(x1, nc);     // the chains' multipliers in reduced radix, internal form
    // levels of the chunked recurrence
    const unsigned L = 16;   // coefficients per lane and level: 16 puts 20 K waves on the chip (32: 10 K, half of its wave slots idle): 0.98 -> 0.90 ms
    uint64_t cnt[8];
    cnt[0] = np.len;
    int levels = 0;
    while (cnt[levels] > 64 && levels < 6) {   // the top level is one lane: keep it to <= 64 values (16^6 covers every domain of both curves; lvl[6], cnt[8], xp[8] have room)
        cnt[levels + 1] = (cnt[levels] + L - 1) / L;
        ++levels;
    }
    // buffers: V[l] (values of level l, l >= 1) and H[l] (suffix values of level l, l >= 1)
    Fr *V[8] = {nullptr}, *H[8] = {nullptr};
    for (int l = 1; l <= levels; ++l) {
        PM_HIP(ctx, ctx->lvl[l - 1].reserve((2 * cnt[l] + 2) * sizeof(Fr)));
        V[l] = ctx->lvl[l - 1].as<Fr>();
        H[l] = V[l] + cnt[l];
    }
    PM_HIP(ctx, ctx->quotient.reserve((np.len + 1) * sizeof(Fr)));
    Fr *q = ctx->quotient.as<Fr>();
    unsigned *flags = ctx->flags.as<unsigned>();
    const Fr *u = ctx->u.as<Fr>(), *wit_u = ctx->wit_u.as<Fr>(), *u2 = ctx->u2.as<Fr>();
    {
        StageTimer t(ctx, T_POLY);
        PM_HIP(ctx, hipMemsetAsync(flags, 0, 64, st));
        Fr xp[8];
        xp[0] = x1;
        for (int l = 1; l <= levels; ++l) xp[l] = pow_u64<P>(xp[l - 1], L);
        if (levels == 0) {
            // small: one lane does the whole division (q written directly)
            PM_HIP(ctx, ctx->lvl[0].reserve(2 * sizeof(Fr)));
            Fr *zero = ctx->lvl[0].as<Fr>();
            PM_HIP(ctx, hipMemsetAsync(zero, 0, 2 * sizeof(Fr), st));
            hipLaunchKernelGGL(k_div_expand0<P>, dim3(1), dim3(64), 0, st, np, nc, m28, u, wit_u, u2, (unsigned)np.len, (uint64_t)1,
                               zero, q, flags);
            PM_HIP(ctx, hipGetLastError());
        } else {
            hipLaunchKernelGGL(k_div_level0<P>, dim3(nblk(cnt[1])), dim3(256), 0, st, np, nc, m28, u, wit_u, u2, L, cnt[1], V[1]);
            PM_HIP(ctx, hipGetLastError());
            for (int l = 1; l < levels; ++l) {
                hipLaunchKernelGGL(k_div_levelN<P>, dim3(nblk(cnt[l + 1])), dim3(256), 0, st, V[l], cnt[l], xp[l], L,
                                   cnt[l + 1], V[l + 1]);
                PM_HIP(ctx, hipGetLastError());
            }
            hipLaunchKernelGGL(k_div_top<P>, dim3(1), dim3(64), 0, st, V[levels], cnt[levels], xp[levels], H[levels]);
            PM_HIP(ctx, hipGetLastError());
            for (int l = levels - 1; l >= 1; --l) {
                hipLaunchKernelGGL(k_div_expandN<P>, dim3(nblk(cnt[l + 1])), dim3(256), 0, st, V[l], cnt[l], xp[l], L,
                                   cnt[l + 1], H[l + 1], H[l]);
                PM_HIP(ctx, hipGetLastError());
            }
            hipLaunchKernelGGL(k_div_expand0<P>, dim3(nblk(cnt[1])), dim3(256), 0, st, np, nc, m28, u, wit_u, u2, L, cnt[1],
                               H[1], q, flags);
            PM_HIP(ctx, hipGetLastError());
        }
    }
    // rem == 0 (prover.rs:221) is read after the MSM's final synchronisation: no host wait between the division and the sort
    if (!ctx_pinned(ctx)) { ctx->err = "pinned staging allocation failed"; return PM_ERR_HIP; }
    volatile unsigned *hflags_p = (volatile unsigned *)((uint8_t *)ctx->h_pinned + PINNED_SLOTS_BYTES);
    *hflags_p = 0;
    PM_HIP(ctx, hipMemcpyAsync((void *)hflags_p, flags, 4, hipMemcpyDeviceToHost, st));
    const int st_d = msm_shard<C>(ctx, pk, 2, q, d_xy, d_inf);   // [d]_1 = M8, prover.rs:229
    PM_HIP(ctx, hipStreamSynchronize(st));
    if (*hflags_p & 8u) return PM_ERR_REMAINDER_NONZERO;  // prover.rs:221
    PM_TRY(st_d);
    t_phase.stop();
    timing_flush(ctx);
    ctx->phase = 3;
    return PM_OK;
}

#define PM_INST(C)                                                                                                     \
    template int msm_resident<C>(pm_ctx *, const pm_pk *, int, const Fp<typename C::FrP> *, uint64_t *, int *);        \
    template int msm_resident_begin<C>(pm_ctx *, const pm_pk *, int, const Fp<typename C::FrP> *, uint64_t, uint64_t);  \
    template int msm_resident_end<C>(pm_ctx *, uint64_t *, int *);                                                     \
    template int prove_phase1_impl<C>(pm_ctx *, const pm_pk *, const uint64_t *, const uint64_t *, const uint64_t *,   \
                                      uint64_t *, int *, uint64_t *, int *, bool);                                         \
    template int prove_phase2_impl<C>(pm_ctx *, const uint64_t *, uint64_t *);                                         \
    template int prove_phase3_impl<C>(pm_ctx *, const uint64_t *, const uint64_t *, const uint64_t *,                  \
                                      const uint64_t *, uint64_t *, int *);
PM_INST(BlsCurve)
PM_INST(BnCurve)

}  // namespace pm
