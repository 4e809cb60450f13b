// create_proof_with_assignment (/root/reference/src/prover.rs:66-237) with the VECTOR phases sharded over N GPUs
// (BASELINE.json configs[3]; layout: polymath_amd/host/layout.hpp):
//   witness map       each rank evaluates the rows i = N j + q it owns (prover.rs:245-302, common.rs:138-207)   row 3
//   iNTT / NTT        N local size-n/N transforms + ONE all-to-all + a size-N butterfly (prover.rs:239-243,:315-328) row 2
//   pointwise, h      local on the blocked coefficient layout (prover.rs:104-108, 321-323)                          row 4
//   u(x1)             per-rank partial Horner sums, all-gather of N values (prover.rs:132)                          row 5
//   division scan     per-segment values, all-gather, carries folded on every rank, local expansion (:211-225)     row 6
//   MSMs              the pairs follow the coefficients: no scalar moves between GPUs (prover.rs:118-123, 229)     row 1
// Same three phases, same outputs (partial points) as prove.hip; every rank returns the same status because the
// check flags are exchanged before any rank decides.
#include <thread>

#include "comm.h"
#include "internal.h"
#include "prove_common.cuh"

namespace pm {

using pmlayout::Layout;
using pmlayout::Segment;

// a collective's status -> the phase's status (the communicator's message goes to pm_last_error)
static int comm_status(pm_ctx *ctx, int st, const char *what) {
    if (st) ctx->err = std::string(what) + ": " + ctx->comm->error();
    return st;
}
// end of a phase: whatever the watchdog cut short must not be taken for a result
static int comm_alive(pm_ctx *ctx) {
    if (!ctx->comm->failed) return PM_OK;
    ctx->err = "communicator failed: " + ctx->comm->error();
    return PM_ERR_COMM;
}

// omega^e from the half-size table tw[j] = omega^j, j < n/2 (omega^(n/2) = -1)
template <class P>
__device__ __forceinline__ Fp<P> tw_pow(const Fp<P> *tw, uint64_t n, uint64_t e) {
    e &= n - 1;
    const uint64_t half = n >> 1;
    return e < half ? tw[e] : neg<P>(tw[e - half]);
}

template <class P>
__global__ void k_check_sap_L(const Fp<P> *ue, const Fp<P> *we, uint64_t count, unsigned *flags) {   // (Uz)^2 == Wz, prover.rs:108
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && !sqr<P>(ue[i]).eq(we[i])) atomicOr(flags, 1u);
}
template <class P>
__global__ void k_square_L(Fp<P> *a, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) a[i] = sqr<P>(a[i]);
}

// ------------------------------------------------------------------------------------------------ witness map
// Row i = N j + q of (U z, W z): the closed form of SURVEY.md App. A (k_witness_head / k_witness_rows of prove.hip, one
// row at a time).  Second-half rows recompute (A - B) z: 2 dot products more per rank instead of an exchange.
template <class P>
__device__ __forceinline__ void sap_row(const CsrDev &A, const CsrDev &B, const CsrDev &Cm, const Fp<P> *xw, uint64_t m0, uint64_t nr,
                                        uint64_t i, Fp<P> &u, Fp<P> &w) {
    const Fp<P> one = Fp<P>::one();
    u = Fp<P>::zero();
    w = Fp<P>::zero();
    if (i < 2 * m0) {
        if (i == 0) {
            u = dbl<P>(one);
            w = dbl<P>(u);
        } else if (i < m0) {
            const Fp<P> xi = xw[i], omx = sub<P>(one, xi);
            u = add<P>(one, xi);
            w = add<P>(dbl<P>(dbl<P>(xi)), sqr<P>(omx));
        } else if (i > m0) {
            const Fp<P> omx = sub<P>(one, xw[i - m0]);
            u = omx;
            w = sqr<P>(omx);
        }
    } else if (i < 2 * m0 + nr) {
        const uint64_t r = i - 2 * m0;
        const Fp<P> az = csr_row_dot<P>(A.rowptr, A.col, A.val, xw, r), bz = csr_row_dot<P>(B.rowptr, B.col, B.val, xw, r);
        const Fp<P> cz = csr_row_dot<P>(Cm.rowptr, Cm.col, Cm.val, xw, r), d = sub<P>(az, bz);
        u = add<P>(az, bz);
        w = add<P>(dbl<P>(dbl<P>(cz)), sqr<P>(d));
    } else if (i < 2 * m0 + 2 * nr) {
        const uint64_t r = i - 2 * m0 - nr;
        const Fp<P> d = sub<P>(csr_row_dot<P>(A.rowptr, A.col, A.val, xw, r), csr_row_dot<P>(B.rowptr, B.col, B.val, xw, r));
        u = d;
        w = sqr<P>(d);
    }
}

template <class P>
__global__ void k_witness_cyclic(CsrDev A, CsrDev B, CsrDev Cm, const Fp<P> *xw, Fp<P> *ue, Fp<P> *we, uint64_t m0, uint64_t nr, Layout L) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L.m) return;
    Fp<P> u, w;
    sap_row<P>(A, B, Cm, xw, m0, nr, pmlayout::eval_global(L, j), u, w);
    ue[j] = u;
    we[j] = w;
}

// the first `count` rows of U z (public-input rows): every rank needs them for the witness-only part of u
template <class P>
__global__ void k_ue_head(CsrDev A, CsrDev B, CsrDev Cm, const Fp<P> *xw, Fp<P> *head, uint64_t m0, uint64_t nr, unsigned count) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fp<P> u, w;
    sap_row<P>(A, B, Cm, xw, m0, nr, i, u, w);
    head[i] = u;
}

// z_tail[lo .. hi) = (x || w || y)[lo .. hi)  (prover.rs:75-80, 279-302): this rank's slice of the [c]_1 scalars
template <class P>
__global__ void k_ztail_slice(CsrDev A, CsrDev B, const Fp<P> *xw, Fp<P> *out, uint64_t m0, uint64_t mw, uint64_t lo, uint64_t hi) {
    const uint64_t t = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= hi) return;
    Fp<P> v;
    if (t < m0 + mw) {
        v = xw[t];
    } else {
        const uint64_t i = t - (m0 + mw);
        if (i < m0) {
            v = i ? sqr<P>(sub<P>(Fp<P>::one(), xw[i])) : Fp<P>::zero();
        } else {
            const uint64_t r = i - m0;
            v = sqr<P>(sub<P>(csr_row_dot<P>(A.rowptr, A.col, A.val, xw, r), csr_row_dot<P>(B.rowptr, B.col, B.val, xw, r)));
        }
    }
    out[t - lo] = v;
}

// --------------------------------------------------------------------------------------- distributed transform
// The size-N butterfly across what used to be ranks.  in[i][b], out[o][b]  (i, o < N; b < B):
//   out[o][b] = scale * omega^(o k2) * sum_i roots[i o mod N] * (omega^(i k2) in[i][b]),   k2 = k2_base + b,
// the omega^(..) factors taken from `tw` (forward or inverse table of the size-n domain) when the flag is set:
//   inverse transform: twiddle on the INPUT side (i = sending rank), roots = (omega^-m)^t, scale = 1/N;
//   forward transform: twiddle on the OUTPUT side (o = receiving rank), roots = (omega^m)^t.
template <class P>
__global__ void k_cross_dft(const Fp<P> *in, Fp<P> *out, const Fp<P> *roots, uint32_t N, uint64_t B, const Fp<P> *tw, uint64_t n,
                            uint64_t k2_base, int in_twiddle, int out_twiddle, Fp<P> scale, int use_scale) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)N * B) return;
    const uint64_t o = g / B, b = g % B, k2 = k2_base + b;
    Fp<P> acc = in[b];
    for (uint32_t i = 1; i < N; ++i) {
        Fp<P> v = in[(uint64_t)i * B + b];
        if (in_twiddle) v = mul<P>(v, tw_pow<P>(tw, n, (uint64_t)i * k2));
        const uint32_t e = (uint32_t)((i * o) & (N - 1));
        acc = add<P>(acc, e ? mul<P>(v, roots[e]) : v);
    }
    if (use_scale) acc = mul<P>(acc, scale);
    if (out_twiddle && o) acc = mul<P>(acc, tw_pow<P>(tw, n, o * k2));
    out[o * B + b] = acc;
}

constexpr unsigned CROSS_COL_THREADS = 128;   // x N x 32 B of LDS per workgroup: 32 KiB at N = 8, 64 KiB at N = 16

// The same butterfly, one lane per COLUMN b for the usual rank counts: the N inputs of a column are loaded and twiddled once
// (k_cross_dft redoes that for every output: N (N - 1) twiddle products per column), then a radix-2 decimation-in-frequency network
// -- N/2 log2 N root products instead of N (N - 1) -- whose bit-reversed positions are undone when storing.  N = 8: 7 + 12 products
// per column against 112.  The column's N values live in LDS, limb-major (one bank per lane for every access): as a private array
// hipcc (ROCm 7.2) keeps 8 x 32 bytes per lane in SCRATCH whatever the unrolling (measured: 272 bytes of private segment).
template <class P, unsigned N>
__global__ __launch_bounds__(CROSS_COL_THREADS) void k_cross_dft_col(const Fp<P> *in, Fp<P> *out, const Fp<P> *roots, uint64_t B, const Fp<P> *tw, uint64_t n,
                                                                     uint64_t k2_base, int in_twiddle, int out_twiddle, Fp<P> scale, int use_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *sh = (uint32_t *)smem_raw;                              // [N][P::N limbs][CROSS_COL_THREADS]
    const unsigned t = threadIdx.x;
    auto put = [&](unsigned i, const Fp<P> &v) {
#pragma unroll
        for (int l = 0; l < P::N; ++l) sh[(i * P::N + l) * CROSS_COL_THREADS + t] = v.l[l];
    };
    auto get = [&](unsigned i) {
        Fp<P> v;
#pragma unroll
        for (int l = 0; l < P::N; ++l) v.l[l] = sh[(i * P::N + l) * CROSS_COL_THREADS + t];
        return v;
    };
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + t;
    if (b >= B) return;                                               // no barrier below: every lane owns its slice of the LDS
    const uint64_t k2 = k2_base + b;
    for (unsigned i = 0; i < N; ++i) {
        Fp<P> v = in[(uint64_t)i * B + b];
        if (in_twiddle && i) v = mul<P>(v, tw_pow<P>(tw, n, (uint64_t)i * k2));
        put(i, v);
    }
    for (unsigned s = N / 2; s >= 1; s >>= 1) {                      // DIF stages: blocks of 2 s, partner distance s
        for (unsigned j = 0; j < N; ++j) {
            if (j & s) continue;
            const Fp<P> a = get(j), c = get(j + s);
            put(j, add<P>(a, c));
            const unsigned e = (j & (s - 1)) * (N / (2 * s));
            const Fp<P> d = sub<P>(a, c);
            put(j + s, e ? mul<P>(d, roots[e]) : d);
        }
    }
    constexpr int LOGN = N == 2 ? 1 : N == 4 ? 2 : N == 8 ? 3 : 4;
    static_assert((1u << LOGN) == N, "N in {2, 4, 8, 16}");
    for (unsigned p = 0; p < N; ++p) {                                // position p holds output index bitrev(p)
        const unsigned o = __brev(p) >> (32 - LOGN);
        Fp<P> v = get(p);
        if (use_scale) v = mul<P>(v, scale);
        if (out_twiddle && o) v = mul<P>(v, tw_pow<P>(tw, n, (uint64_t)o * k2));
        out[(uint64_t)o * B + b] = v;
    }
}

// one launch of the cross-rank butterfly: the column kernel for N = 2, 4, 8, 16, the general one otherwise
template <class P>
static void launch_cross_dft(hipStream_t st, const Fp<P> *in, Fp<P> *out, const Fp<P> *roots, const Layout &L, const Fp<P> *tw, int in_twiddle,
                             int out_twiddle, Fp<P> scale, int use_scale) {
    const uint64_t k2_base = (uint64_t)L.q * L.B;
#define PM_COL(NN)                                                                                                                          \
    case NN:                                                                                                                                \
        hipLaunchKernelGGL((k_cross_dft_col<P, NN>), dim3(nblk(L.B, CROSS_COL_THREADS)), dim3(CROSS_COL_THREADS),                           \
                           (size_t)NN * sizeof(Fp<P>) * CROSS_COL_THREADS, st, in, out, roots, L.B, tw, L.n, k2_base, in_twiddle,            \
                           out_twiddle, scale, use_scale);                                                                                  \
        return;
    switch (L.N) {
        PM_COL(2) PM_COL(4) PM_COL(8) PM_COL(16)
        default: break;
    }
#undef PM_COL
    hipLaunchKernelGGL(k_cross_dft<P>, dim3(nblk(L.m)), dim3(256), 0, st, in, out, roots, L.N, L.B, tw, L.n, k2_base, in_twiddle, out_twiddle, scale,
                       use_scale);
}

template <class C>
static int shard_roots(pm_ctx *ctx, const pm_pk *pk, const Fp<typename C::FrP> **fwd, const Fp<typename C::FrP> **inv) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    const uint32_t N = (uint32_t)pk->shard_count;
    if (ctx->shard_roots_n != pk->n || ctx->shard_roots_N != N || ctx->shard_roots_curve != C::ID) {
        Fr omega;
        memcpy(omega.l, pk->omega, 32);
        const Fr wm = pow_u64<P>(omega, pk->n / N), wmi = inverse<P>(wm);
        std::vector<Fr> h(2 * (size_t)N);
        h[0] = Fr::one();
        h[N] = Fr::one();
        for (uint32_t t = 1; t < N; ++t) {
            h[t] = mul<P>(h[t - 1], wm);
            h[N + t] = mul<P>(h[N + t - 1], wmi);
        }
        PM_HIP(ctx, ctx->shard_roots.reserve(h.size() * sizeof(Fr)));
        PM_HIP(ctx, hipMemcpyAsync(ctx->shard_roots.p, h.data(), h.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->shard_roots_n = pk->n;
        ctx->shard_roots_N = N;
        ctx->shard_roots_curve = C::ID;
    }
    *fwd = ctx->shard_roots.as<Fr>();
    *inv = *fwd + N;
    return PM_OK;
}

// evaluations (cyclic, `x`, destroyed) -> coefficients (blocked, `y`); `tmp`: m elements of scratch.
// `e` = the context whose stream, transform workspace and twiddle cache carry the transform: ctx itself, or its helper context
// when w's transform runs beside u's chain (PM_NTT_OVERLAP=1).  The communicator and the error slot are always ctx's.
template <class C>
static int dist_intt(pm_ctx *ctx, pm_ctx *e, const pm_pk *pk, const Layout &L, Fp<typename C::FrP> *x, Fp<typename C::FrP> *tmp, Fp<typename C::FrP> *y) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    unsigned log_m = 0;
    while (((uint64_t)1 << log_m) < L.m) ++log_m;
    auto on_e = [&](int st) { if (st != PM_OK && e != ctx) ctx->err = e->err; return st; };
    PM_TRY(on_e(ntt_run<C>(e, x, log_m, true)));                                // N local transforms (this rank's), scaled by 1/m
    StageTimer t(e, T_NTT);
    PM_TRY(comm_status(ctx, ctx->comm->all_to_all(x, tmp, (size_t)L.B * sizeof(Fr), e->stream), "all_to_all"));   // block p -> rank p
    const Fr *rf = nullptr, *ri = nullptr, *tw = nullptr;
    PM_TRY(shard_roots<C>(ctx, pk, &rf, &ri));
    PM_TRY(on_e(twiddles_get<C>(e, pk->log_n, true, &tw)));
    const Fr ninv = inverse<P>(from_u64<P>(L.N));
    launch_cross_dft<P>(e->stream, (const Fr *)tmp, y, ri, L, tw, 1, 0, ninv, 1);
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
}
template <class C>
static int dist_intt(pm_ctx *ctx, const pm_pk *pk, const Layout &L, Fp<typename C::FrP> *x, Fp<typename C::FrP> *tmp, Fp<typename C::FrP> *y) {
    return dist_intt<C>(ctx, ctx, pk, L, x, tmp, y);
}

// coefficients (blocked, `y`, kept) -> evaluations (cyclic, `x`); `tmp`: m elements of scratch
template <class C>
static int dist_ntt(pm_ctx *ctx, const pm_pk *pk, const Layout &L, const Fp<typename C::FrP> *y, Fp<typename C::FrP> *tmp, Fp<typename C::FrP> *x) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    unsigned log_m = 0;
    while (((uint64_t)1 << log_m) < L.m) ++log_m;
    {
        StageTimer t(ctx, T_NTT);
        const Fr *rf = nullptr, *ri = nullptr, *tw = nullptr;
        PM_TRY(shard_roots<C>(ctx, pk, &rf, &ri));
        PM_TRY(twiddles_get<C>(ctx, pk->log_n, false, &tw));
        launch_cross_dft<P>(ctx->stream, y, tmp, rf, L, tw, 0, 1, Fr::one(), 0);
        PM_HIP(ctx, hipGetLastError());
        PM_TRY(comm_status(ctx, ctx->comm->all_to_all(tmp, x, (size_t)L.B * sizeof(Fr), ctx->stream), "all_to_all"));   // block r -> rank r
    }
    return ntt_run<C>(ctx, x, log_m, false);
}

// ------------------------------------------------------------------------------------- pointwise, blocked layout
template <class P>
__global__ void k_twist_L(const Fp<P> *u, const Fp<P> *psi_pow, Fp<P> *out, Layout L) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < L.m) out[p] = mul<P>(u[p], psi_pow[pmlayout::coeff_global(L, p)]);
}
template <class P>
__global__ void k_untwist_combine_L(const Fp<P> *neg_tw, const Fp<P> *psi_inv_pow, const Fp<P> *w, Fp<P> *lo, Fp<P> *hi, Layout L, Fp<P> half) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= L.m) return;
    const Fp<P> neg = mul<P>(neg_tw[p], psi_inv_pow[pmlayout::coeff_global(L, p)]), wk = w[p];
    lo[p] = mul<P>(add<P>(wk, neg), half);
    hi[p] = mul<P>(sub<P>(wk, neg), half);
}
// wit_u[k] = u[k] - n^-1 sum_{j < head} ue[j] w^(-jk)  (k_wit_u_sparse of prove.hip at the global index of p)
template <class P>
__global__ void k_wit_u_sparse_L(const Fp<P> *u, const Fp<P> *ue_head, const Fp<P> *winv, Fp<P> ninv, unsigned head, Fp<P> *wit_u, Layout L) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= L.m) return;
    const Fp<P> wk = tw_pow<P>(winv, L.n, pmlayout::coeff_global(L, p));
    Fp<P> s = ue_head[head - 1];
    for (int j = (int)head - 2; j >= 0; --j) s = add<P>(mul<P>(s, wk), ue_head[j]);
    wit_u[p] = sub<P>(u[p], mul<P>(s, ninv));
}
template <class P>
__global__ void k_zero_head_cyclic(Fp<P> *ue, uint64_t zero_rows, Layout L) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < L.m && pmlayout::eval_global(L, j) < zero_rows) ue[j] = Fp<P>::zero();
}

// This rank's [c]_1 scalars after the z_tail slice (layout.hpp: pieces_c):  h blocks | 2 r_a u blocks of B + 1 | constants
template <class P>
__global__ void k_phase1_scalars_L(const Fp<P> *u, const Fp<P> *u2hi, const Fp<P> *ra, Fp<P> *sc_h, Fp<P> *sc_ru, Fp<P> *sc_tail, Layout L,
                                   unsigned *flags) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Fp<P> r0 = ra[0], r1 = ra[1];
    if (p < L.m) {
        const uint64_t k = pmlayout::coeff_global(L, p), k1 = p / L.B, b = p % L.B;
        const Fp<P> hi = u2hi[p];
        if (k < L.n - 1) {
            sc_h[p] = hi;
            // "h is not identically zero": nearly every lane sees it, and one atomic per wave on ONE word (the compiler already folds the lanes)
            // is 131 K serialised L2 operations -- 0.3 of this kernel's 0.39 ms.  A wave that reads the bit as set has nothing to add.
            if (!hi.is_zero() && !(*(const volatile unsigned *)flags & 4u)) atomicOr(flags, 4u);
        } else if (!hi.is_zero()) {
            atomicOr(flags, 2u);                       // deg h > n - 2  (prover.rs:107)
        }
        Fp<P> t = mul<P>(r0, u[p]);
        if (b) t = add<P>(t, mul<P>(r1, u[p - 1]));
        Fp<P> *blk = sc_ru + k1 * (L.B + 1);
        blk[b] = dbl<P>(t);
        if (b == L.B - 1) blk[L.B] = dbl<P>(mul<P>(r1, u[p]));   // the pair (2 r1 u_{e-1}, X_e) of the next block's first base
    }
    if (p == 0 && sc_tail) {
        sc_tail[0] = sqr<P>(r0);
        sc_tail[1] = dbl<P>(mul<P>(r0, r1));
        sc_tail[2] = sqr<P>(r1);
        sc_tail[3] = r0;
        sc_tail[4] = r1;
    }
}

// ------------------------------------------------------------------------------------------ Horner (phase 2)
// lane t owns Lh consecutive LOCAL coefficients (Lh divides B: a lane never straddles a block)
template <class P>
__global__ __launch_bounds__(256) void k_horner_partial_L(const Fp<P> *u, Layout L, Fp<P> x1, unsigned Lh, Fp<P> *partials) {
    __shared__ Fp<P> sh[256];
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = t * Lh;
    Fp<P> acc = Fp<P>::zero();
    if (lo < L.m) {
        for (uint64_t k = lo + Lh; k-- > lo;) acc = add<P>(mul<P>(acc, x1), u[k]);
        acc = mul<P>(acc, pow_u64<P>(x1, pmlayout::coeff_global(L, lo)));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = add<P>(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}
// out[k1] = the last coefficient of this rank's block k1: the halo the next block's owner needs in the division (phase 3)
template <class P>
__global__ void k_halo_pack(const Fp<P> *u, Layout L, Fp<P> *out) {
    const unsigned k1 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k1 < L.N) out[k1] = u[(uint64_t)k1 * L.B + L.B - 1];
}
// out[0] = sum of `count` partials
template <class P>
__global__ __launch_bounds__(256) void k_phase2_pack(const Fp<P> *partials, unsigned count, Fp<P> *out) {
    __shared__ Fp<P> sh[256];
    Fp<P> acc = Fp<P>::zero();
    for (unsigned i = threadIdx.x; i < count; i += 256) acc = add<P>(acc, partials[i]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = add<P>(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// ----------------------------------------------------------------------------------- division scan (phase 3)
struct SegData {   // what numerator_seg reads
    uint64_t n, sigma;
};

// numerator coefficient at index a + j of segment g (numerator_at of prove.hip on this rank's local arrays)
template <class P>
__device__ __forceinline__ Fp<P> numerator_seg(const Segment &g, uint64_t j, const SegData &sd, const NumConsts<P> &nc, const Fp<P> *u,
                                               const Fp<P> *wit_u, const Fp<P> *u2lo, const Fp<P> *u2hi, const Fp<P> *halo) {
    switch (g.kind) {
        case pmlayout::SEG_WITU: return mul<P>(nc.x2, wit_u[g.loc0 + j]);
        case pmlayout::SEG_U2LO: return mul<P>(nc.x2, u2lo[g.loc0 + j]);
        case pmlayout::SEG_U2HI: return mul<P>(nc.x2, u2hi[g.loc0 + j]);
        case pmlayout::SEG_U: {
            const uint64_t i = g.a + j - 5 * sd.sigma;
            Fp<P> t = Fp<P>::zero();
            if (i < sd.n) {
                const Fp<P> ui = u[g.loc0 + j];
                t = add<P>(ui, mul<P>(nc.two_x2_r0, ui));
            }
            if (i > 0) {
                const Fp<P> prev = (j == 0 && g.halo) ? halo[g.halo - 1] : u[g.loc0 + j - 1];
                t = add<P>(t, mul<P>(nc.two_x2_r1, prev));
            } else {
                t = add<P>(t, nc.minus_const);
            }
            return t;
        }
        default: {
            const uint64_t k = g.a + j;
            if (k == 0) return nc.x2r0;
            if (k == 1) return nc.x2r1;
            if (k >= 2 * sd.sigma && k < 2 * sd.sigma + 3) return nc.b2[k - 2 * sd.sigma];
            return Fp<P>::zero();
        }
    }
}

constexpr unsigned SEG_THREADS_MAX = 1024;   // lane values are stored with this stride whatever the workgroup size

// The division scan in two halves that straddle the second challenge.  The numerator's coefficients depend on x2 (and on a, c at
// x1, which come after it) only through a handful of constants: on a data segment N_k = c1 D1_k + c2 D2_k (+ one constant at the
// first index of the u region), with D1, D2 this rank's coefficient arrays -- wit_u, (u^2)_lo, (u^2)_hi, or (u_i, u_{i-1}).  So
// the Horner sums of the DATA at x1 can be taken as soon as x1 is known (phase 2) and travel with the u(x1) partials; phase 3 then
// needs no exchange before its MSM: every rank derives all segment values V_s = c1 P_s + c2 Q_s + constants on the host.
//
// k_seg_base (phase 2): one workgroup per segment; lane t owns the `span` consecutive indices from a + t span:
//   laneP[t] = sum_j D1_{lo + j} x1^j,  laneQ[t] likewise for D2 (u region only);  P_s = sum_t laneP[t] x1^(t span), Q_s likewise.
template <class P>
__device__ __forceinline__ void seg_data(const Segment &g, uint64_t j, const SegData &sd, const Fp<P> *u, const Fp<P> *wit_u, const Fp<P> *u2lo,
                                         const Fp<P> *u2hi, const Fp<P> *halo, Fp<P> &d1, Fp<P> &d2) {
    d1 = Fp<P>::zero();
    d2 = Fp<P>::zero();
    switch (g.kind) {
        case pmlayout::SEG_WITU: d1 = wit_u[g.loc0 + j]; break;
        case pmlayout::SEG_U2LO: d1 = u2lo[g.loc0 + j]; break;
        case pmlayout::SEG_U2HI: d1 = u2hi[g.loc0 + j]; break;
        case pmlayout::SEG_U: {
            const uint64_t i = g.a + j - 5 * sd.sigma;
            if (i < sd.n) d1 = u[g.loc0 + j];
            if (i > 0) d2 = (j == 0 && g.halo) ? halo[g.halo - 1] : u[g.loc0 + j - 1];
            break;
        }
        default: break;
    }
}

template <class P, unsigned SEG_THREADS>
__global__ __launch_bounds__(SEG_THREADS) void k_seg_base(const Segment *segs, SegData sd, const Fp<P> *u, const Fp<P> *wit_u, const Fp<P> *u2lo,
                                                          const Fp<P> *u2hi, const Fp<P> *halo, Fp<P> x1, Fp<P> *laneP, Fp<P> *laneQ, Fp<P> *Pseg,
                                                          Fp<P> *Qseg) {
    __shared__ Fp<P> sh[SEG_THREADS];
    const Segment g = segs[blockIdx.x];
    const size_t at = (size_t)blockIdx.x * SEG_THREADS_MAX + threadIdx.x;
    if (g.kind == pmlayout::SEG_FILLER) {      // no data: zeros and a few constants that only exist once x2 is known (phase 3)
        laneP[at] = Fp<P>::zero();
        laneQ[at] = Fp<P>::zero();
        if (threadIdx.x == 0) { Pseg[blockIdx.x] = Fp<P>::zero(); Qseg[blockIdx.x] = Fp<P>::zero(); }
        return;
    }
    const bool two = g.kind == pmlayout::SEG_U;
    const uint64_t len = g.b - g.a, span = (len + SEG_THREADS - 1) / SEG_THREADS;
    const uint64_t lo = (uint64_t)threadIdx.x * span;
    uint64_t hi = lo + span;
    if (hi > len) hi = len;
    Fp<P> accP = Fp<P>::zero(), accQ = Fp<P>::zero();
    if (lo < len) {
        for (uint64_t j = hi; j-- > lo;) {
            Fp<P> d1, d2;
            seg_data<P>(g, j, sd, u, wit_u, u2lo, u2hi, halo, d1, d2);
            accP = add<P>(mul<P>(accP, x1), d1);
            if (two) accQ = add<P>(mul<P>(accQ, x1), d2);
        }
    }
    laneP[at] = accP;
    laneQ[at] = accQ;
    const Fp<P> xl = lo < len ? pow_u64<P>(x1, lo) : Fp<P>::zero();
    sh[threadIdx.x] = lo < len ? mul<P>(accP, xl) : accP;
    __syncthreads();
    for (unsigned off = SEG_THREADS / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = add<P>(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) Pseg[blockIdx.x] = sh[0];
    __syncthreads();
    sh[threadIdx.x] = two && lo < len ? mul<P>(accQ, xl) : Fp<P>::zero();
    __syncthreads();
    for (unsigned off = SEG_THREADS / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = add<P>(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) Qseg[blockIdx.x] = sh[0];
}

// 512 lanes per segment: with ~256-400 segments per rank every CU holds one or two workgroups at 2-4 waves per SIMD (1024 lanes and
// half as many segments left a third of the chip idle and four waves queueing on each busy SIMD: profiles/r03_k_*)
constexpr unsigned SEG_LANES = 512;

// does [a, b) contain one of the numerator's constants (indices 0, 1, 2 sigma .. 2 sigma + 2)?
__host__ __device__ inline bool filler_has_const(uint64_t a, uint64_t b, uint64_t sigma) {
    const uint64_t s2 = 2 * sigma;
    return a < 2 || (a < s2 + 3 && b > s2);
}

// k_seg_chain (phase 3): the carry chain over ALL ranks' segments, on the device (round 2 and the first half of round 3 folded it on
// the host: ~2 600 segments x a few Fr products = 0.16 ms of host time per proof at N = 8).  One workgroup.  In increasing index
// order the segments satisfy  H_a(i) = V_i + M_i H_a(i + 1),  M_i = x1^(b_i - a_i),  V_i = c1 P_i + c2 Q_i + constants (from phase
// 2's exchanged data sums).  Lane t composes the affine maps of its run of segments, a suffix scan over the lanes gives every
// lane its carry-in, and a second walk writes, for THIS rank's segments, carry[idx] = H at the segment's upper end.  rem[0] = H_0.
struct SegRefDev { uint64_t a, b; uint32_t rank, idx; };   // == pm_pk::SegRef
constexpr unsigned CHAIN_THREADS = 512;

template <class P>
__device__ __forceinline__ Fp<P> seg_value(const SegRefDev &e, const Fp<P> *hall, size_t rec, size_t SS, const NumConsts<P> &nc, const Fp<P> &c1u,
                                           const Fp<P> &x1, uint64_t n, uint64_t sigma) {
    const uint64_t s2 = 2 * sigma, s3 = 3 * sigma, s5 = 5 * sigma, s8 = 8 * sigma, a = e.a, b = e.b;
    const Fp<P> Ps = hall[(size_t)e.rank * rec + 1 + e.idx];
    if ((a >= s3 && a < s3 + n) || a >= s8) return mul<P>(nc.x2, Ps);                           // witness_u, (u^2)_lo, (u^2)_hi
    if (a >= s5 && a <= s5 + n) {                                                                  // the u region (n + 1 entries)
        Fp<P> v = add<P>(mul<P>(c1u, Ps), mul<P>(nc.two_x2_r1, hall[(size_t)e.rank * rec + 1 + SS + e.idx]));
        if (a == s5) v = add<P>(v, nc.minus_const);                                                // its first index, offset 0
        return v;
    }
    Fp<P> v = Fp<P>::zero();                                                                       // fillers: the constants
    if (filler_has_const(a, b, sigma)) {
        const uint64_t pos[5] = {0, 1, s2, s2 + 1, s2 + 2};
        const Fp<P> val[5] = {nc.x2r0, nc.x2r1, nc.b2[0], nc.b2[1], nc.b2[2]};
        for (int i = 0; i < 5; ++i)
            if (pos[i] >= a && pos[i] < b) v = add<P>(v, mul<P>(val[i], pow_u64<P>(x1, pos[i] - a)));
    }
    return v;
}

template <class P>
__global__ __launch_bounds__(CHAIN_THREADS) void k_seg_chain(const SegRefDev *segs, unsigned T, const Fp<P> *hall, size_t rec, size_t SS, NumConsts<P> nc,
                                                             Fp<P> x1, Fp<P> x1_pow_common, uint64_t common_len, uint64_t n, uint64_t sigma, uint32_t my_rank,
                                                             Fp<P> *carry, Fp<P> *rem) {
    __shared__ Fp<P> shA[CHAIN_THREADS], shB[CHAIN_THREADS];
    const unsigned t = threadIdx.x, per = (T + CHAIN_THREADS - 1) / CHAIN_THREADS;
    const unsigned lo = t * per < T ? t * per : T, hi = lo + per < T ? lo + per : T;
    const Fp<P> c1u = add<P>(Fp<P>::one(), nc.two_x2_r0);
    auto mult = [&](const SegRefDev &e) { const uint64_t len = e.b - e.a; return len == common_len ? x1_pow_common : pow_u64<P>(x1, len); };
    // this lane's run as ONE affine map h -> A + B h (h = H at the run's upper end)
    Fp<P> A = Fp<P>::zero(), B = Fp<P>::one();
    for (unsigned i = hi; i-- > lo;) {
        const SegRefDev e = segs[i];
        const Fp<P> M = mult(e);
        A = add<P>(seg_value<P>(e, hall, rec, SS, nc, c1u, x1, n, sigma), mul<P>(M, A));
        B = mul<P>(M, B);
    }
    shA[t] = A;
    shB[t] = B;
    __syncthreads();
    // suffix scan of the maps: after it (shA[t], shB[t]) maps the carry-in of the LAST lane (0) to H at the start of lane t's run
    for (unsigned d = 1; d < CHAIN_THREADS; d <<= 1) {
        Fp<P> a2 = Fp<P>::zero(), b2 = Fp<P>::one();
        const bool take = t + d < CHAIN_THREADS;
        if (take) { a2 = shA[t + d]; b2 = shB[t + d]; }
        __syncthreads();
        if (take) {                                    // f_t o f_(t+d):  A_t + B_t (a2 + b2 h)
            shA[t] = add<P>(shA[t], mul<P>(shB[t], a2));
            shB[t] = mul<P>(shB[t], b2);
        }
        __syncthreads();
    }
    // carry-in of lane t = H at the start of lane t + 1's run (0 beyond the last lane: nothing lies above the top index)
    Fp<P> h = t + 1 < CHAIN_THREADS ? shA[t + 1] : Fp<P>::zero();
    for (unsigned i = hi; i-- > lo;) {
        const SegRefDev e = segs[i];
        if (e.rank == my_rank) carry[e.idx] = h;
        h = add<P>(seg_value<P>(e, hall, rec, SS, nc, c1u, x1, n, sigma), mul<P>(mult(e), h));
    }
    if (t == 0) rem[0] = h;                            // H_0: the remainder of the division (prover.rs:221)
}

// k_seg_expand (phase 3).  carry[s] = H_b of segment s (from the chain over ALL ranks' segment values).  Suffix scan of the lane values with the
// constant multiplier X = x1^span (every lane but the last active one owns a full span; the carry-in is folded into the
// last lane's value), then every lane re-walks its span and writes q_{k-1} = H_k.
template <class P, unsigned SEG_THREADS>
__global__ __launch_bounds__(SEG_THREADS) void k_seg_expand(const Segment *segs, SegData sd, NumConsts<P> nc, const Fp<P> *u, const Fp<P> *wit_u,
                                                            const Fp<P> *u2lo, const Fp<P> *u2hi, const Fp<P> *halo, Fp<P> x1,
                                                            const Fp<P> *laneP, const Fp<P> *laneQ, const Fp<P> *carry, Fp<P> *q) {
    __shared__ Fp<P> sh[SEG_THREADS];
    const Segment g = segs[blockIdx.x];
    const uint64_t len = g.b - g.a, span = (len + SEG_THREADS - 1) / SEG_THREADS;
    const unsigned t = threadIdx.x, active = (unsigned)((len + span - 1) / span);
    const uint64_t lo = (uint64_t)t * span;
    uint64_t hi = lo + span;
    if (hi > len) hi = len;
    const Fp<P> cin = carry[blockIdx.x];
    Fp<P> w = Fp<P>::zero();
    if (t < active) {
        // this lane's Horner value of the numerator (carry-in 0) from phase 2's data sums and the constants phase 3 brought
        const size_t at = (size_t)blockIdx.x * SEG_THREADS_MAX + t;
        switch (g.kind) {
            case pmlayout::SEG_WITU:
            case pmlayout::SEG_U2LO:
            case pmlayout::SEG_U2HI: w = mul<P>(nc.x2, laneP[at]); break;
            case pmlayout::SEG_U: {
                const Fp<P> p = laneP[at];
                w = add<P>(add<P>(p, mul<P>(nc.two_x2_r0, p)), mul<P>(nc.two_x2_r1, laneQ[at]));
                const uint64_t k0 = 5 * sd.sigma;                        // the constant term sits at the region's first index
                if (g.a + lo <= k0 && k0 < g.a + hi) w = add<P>(w, mul<P>(nc.minus_const, pow_u64<P>(x1, k0 - g.a - lo)));
                break;
            }
            default:
                if (filler_has_const(g.a + lo, g.a + hi, sd.sigma))      // a handful of lanes in the whole proof
                    for (uint64_t j = hi; j-- > lo;) w = add<P>(mul<P>(w, x1), numerator_seg<P>(g, j, sd, nc, u, wit_u, u2lo, u2hi, halo));
                break;
        }
        if (t == active - 1) w = add<P>(w, mul<P>(cin, pow_u64<P>(x1, hi - lo)));
    }
    sh[t] = w;
    __syncthreads();
    Fp<P> Xd = pow_u64<P>(x1, span);
    for (unsigned d = 1; d < SEG_THREADS; d <<= 1) {      // W_t = sum_{t' >= t} w_t' X^(t' - t)
        Fp<P> add_in = Fp<P>::zero();
        const bool take = t + d < active;
        if (take) add_in = mul<P>(sh[t + d], Xd);
        __syncthreads();
        if (take) sh[t] = add<P>(sh[t], add_in);
        __syncthreads();
        Xd = sqr<P>(Xd);
    }
    if (t >= active) return;
    Fp<P> acc = t + 1 < active ? sh[t + 1] : cin;          // H at this lane's upper end
    const uint64_t skip = g.a == 0 ? 1 : 0;                // index 0 is the remainder, not a quotient coefficient
    for (uint64_t j = hi; j-- > lo;) {
        acc = add<P>(mul<P>(acc, x1), numerator_seg<P>(g, j, sd, nc, u, wit_u, u2lo, u2hi, halo));
        if (g.a + j > 0) q[g.qoff + j - skip] = acc;
    }
}

// ------------------------------------------------------------------------------------------------- helpers
static int require_comm(pm_ctx *ctx, const pm_pk *pk) {
    if (!ctx->comm || ctx->comm->world != pk->shard_count || ctx->comm->rank != pk->shard_rank) {
        ctx->err = "a PM_SHARD_VECTOR key needs pm_ctx_set_comm with this rank's communicator";
        return PM_ERR_STATE;
    }
    return PM_OK;
}

// One per phase.  (i) A phase runs holding the turn and every return path hands it on (local serialised emulation; no-op
// for RCCL).  (ii) Fail-fast: a return that was not marked `agreed` -- i.e. anything but success or a status every rank
// derives from the same exchanged flags -- is a failure the peers cannot know of (HIP error, allocation, layout mismatch):
// the communicator is aborted so that they leave their next collective with PM_ERR_COMM instead of waiting for ever.
struct PhaseEnd {
    pm_ctx *ctx;
    pm_comm *c;
    const char *name;
    bool agreed = false;
    PhaseEnd(pm_ctx *x, const char *nm) : ctx(x), c(x->comm), name(nm) { if (c) c->phase_begin(); }
    int ok(int status) { agreed = true; return status; }
    ~PhaseEnd() {
        if (!c) return;
        if (!agreed && !c->failed) c->abort((std::string(name) + " failed locally: " + ctx->err).c_str());
        c->phase_end();
    }
};


// ------------------------------------------------------------------------------------------------- phase 1
template <class C>
int prove_phase1_sharded(pm_ctx *ctx, const pm_pk *pk, const uint64_t *x, const uint64_t *w, const uint64_t *r_a, uint64_t *a_xy, int *a_inf,
                         uint64_t *c_xy, int *c_inf, bool assignment_on_device) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    PM_TRY(require_comm(ctx, pk));
    PhaseEnd phase_end(ctx, "phase 1");
    HostProfile hp("phase 1", pk->shard_rank);
    const uint64_t n = pk->n, m0 = pk->m0, mw = pk->mw, nr = pk->nr, Lz = 2 * m0 + mw + nr;
    if (pk->log_n + 1 > (unsigned)C::TWO_ADICITY) return phase_end.ok(PM_ERR_DOMAIN_TOO_LARGE);   // prover.rs:317 -- the key's shape: all ranks
    const Layout L = pmlayout::make_layout(n, (uint32_t)pk->shard_count, (uint32_t)pk->shard_rank);
    const uint64_t m = L.m, B = L.B;
    const uint32_t N = L.N, q = L.q;
    hipStream_t st = ctx->stream;
    if (!ctx->keep_timings) timing_reset(ctx);
    ctx->pk = pk;
    ctx->phase = 0;
    TimingGuard timing_guard{ctx};
    StageTimer t_phase(ctx, T_PHASE);
    const uint64_t zl = pmlayout::ztail_lo(Lz, N, q), zh = pmlayout::ztail_lo(Lz, N, q + 1), zcnt = zh - zl;
    const uint64_t hcnt = m - (q == N - 1 ? 1 : 0), rucnt = (uint64_t)N * (B + 1), tail = q == 0 ? 5 : 0;
    const uint64_t len_c = zcnt + hcnt + rucnt + tail;
    if (len_c != pk->res_cnt[1] || m + (q == 0 ? 2 : 0) != pk->res_cnt[0]) return PM_ERR_STATE;   // key and prover disagree on the layout
    // assignment: [x (m0) | w padded to N slices of wblk | staging of this rank's slice]
    const uint64_t wblk = (mw + N - 1) / N;
    PM_HIP(ctx, ctx->xw.reserve((m0 + ((uint64_t)N + 1) * wblk) * sizeof(Fr)));
    PM_HIP(ctx, ctx->ue.reserve(m * sizeof(Fr)));
    PM_HIP(ctx, ctx->we.reserve(m * sizeof(Fr)));
    PM_HIP(ctx, ctx->u.reserve((m + 2) * sizeof(Fr)));
    PM_HIP(ctx, ctx->w.reserve(m * sizeof(Fr)));
    PM_HIP(ctx, ctx->wit_u.reserve(m * sizeof(Fr)));
    PM_HIP(ctx, ctx->u2.reserve(2 * m * sizeof(Fr)));
    PM_HIP(ctx, ctx->sc_c.reserve((len_c + 1) * sizeof(Fr)));
    PM_HIP(ctx, ctx->ra.reserve(2 * sizeof(Fr)));
    PM_HIP(ctx, ctx->flags.reserve(64));
    PM_HIP(ctx, ctx->sh_a.reserve(m * sizeof(Fr)));
    PM_HIP(ctx, ctx->sh_b.reserve(m * sizeof(Fr)));
    PM_HIP(ctx, ctx->sh_c.reserve((m > 2 * m0 ? m : 2 * m0) * sizeof(Fr)));
    Fr *xw = ctx->xw.as<Fr>(), *ue = ctx->ue.as<Fr>(), *we = ctx->we.as<Fr>(), *u = ctx->u.as<Fr>(), *wv = ctx->w.as<Fr>();
    Fr *wit_u = ctx->wit_u.as<Fr>(), *u2lo = ctx->u2.as<Fr>(), *u2hi = u2lo + m, *sc_c = ctx->sc_c.as<Fr>(), *ra = ctx->ra.as<Fr>();
    Fr *ta = ctx->sh_a.as<Fr>(), *tb = ctx->sh_b.as<Fr>(), *tc = ctx->sh_c.as<Fr>();
    unsigned *flags = ctx->flags.as<unsigned>();
    PM_HIP(ctx, hipMemsetAsync(flags, 0, 64, st));
    // Every rank needs the whole assignment (the rows it owns read arbitrary columns: prover.rs:279-302).  From HOST buffers
    // each rank uploads only its 1/N slice of w over its own PCIe link and the fabric delivers the rest (all-gather of
    // device blocks: "broadcast of assignment") -- N times less PCIe traffic per rank than N full copies.
    if (assignment_on_device) {
        PM_HIP(ctx, hipMemcpyAsync(xw, x, m0 * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        if (mw) PM_HIP(ctx, hipMemcpyAsync(xw + m0, w, mw * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    } else {
        PM_HIP(ctx, hipMemcpyAsync(xw, x, m0 * sizeof(Fr), hipMemcpyHostToDevice, st));
        if (mw) {
            Fr *stage = xw + m0 + (uint64_t)N * wblk;
            const uint64_t lo = std::min<uint64_t>(mw, (uint64_t)q * wblk), hi = std::min<uint64_t>(mw, (uint64_t)(q + 1) * wblk);
            if (hi - lo < wblk) PM_HIP(ctx, hipMemsetAsync(stage + (hi - lo), 0, (wblk - (hi - lo)) * sizeof(Fr), st));
            if (hi > lo) PM_HIP(ctx, hipMemcpyAsync(stage, w + 4 * lo, (hi - lo) * sizeof(Fr), hipMemcpyHostToDevice, st));
            PM_TRY(comm_status(ctx, ctx->comm->all_gather_device(stage, xw + m0, wblk * sizeof(Fr), st), "all_gather_device"));
        }
    }
    PM_HIP(ctx, hipMemcpyAsync(ra, r_a, 2 * sizeof(Fr), hipMemcpyHostToDevice, st));
    memcpy(ctx->ra_host, r_a, 2 * sizeof(Fr));
    CsrDev A{pk->d_rowptr[0], pk->d_col[0], pk->d_val[0]}, Bm{pk->d_rowptr[1], pk->d_col[1], pk->d_val[1]},
        Cm{pk->d_rowptr[2], pk->d_col[2], pk->d_val[2]};
    const bool sparse_head = 2 * m0 <= 16;
    {
        StageTimer t(ctx, T_WITNESS_MAP);
        hipLaunchKernelGGL(k_witness_cyclic<P>, dim3(nblk(m)), dim3(256), 0, st, A, Bm, Cm, xw, ue, we, m0, nr, L);
        PM_HIP(ctx, hipGetLastError());
        if (zcnt) {
            hipLaunchKernelGGL(k_ztail_slice<P>, dim3(nblk(zcnt)), dim3(256), 0, st, A, Bm, xw, sc_c, m0, mw, zl, zh);
            PM_HIP(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(k_check_sap_L<P>, dim3(nblk(m)), dim3(256), 0, st, ue, we, m, flags);   // rem == 0 of prover.rs:108
        PM_HIP(ctx, hipGetLastError());
    }
    hp.mark("upload+witness");
    // N1, N2 (prover.rs:94,96): the transforms consume their input, so the witness-only U part (N5) is prepared first
    if (!sparse_head) {
        PM_HIP(ctx, hipMemcpyAsync(tc, ue, m * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_zero_head_cyclic<P>, dim3(nblk(m)), dim3(256), 0, st, tc, 2 * m0, L);
        PM_HIP(ctx, hipGetLastError());
        PM_TRY(dist_intt<C>(ctx, pk, L, tc, ta, wit_u));
    }
    // PM_OPT_NTT_OVERLAP: w's transform (N2) runs on the helper context's stream beside u's chain
    // (N1, then the three transforms of the square) and joins before k_untwist_combine_L, so that on a fabric its all-to-all
    // travels under u's kernels and u's under w's local passes.  Every rank issues the exchanges in the same order
    // (u, w, square forward, square inverse); they are on two streams of ONE communicator.
    const bool ntt_beside = ctx->opt.v[PM_OPT_NTT_OVERLAP] != 0;
    pm_ctx *wctx = ctx;
    if (ntt_beside) {
        pm_ctx *aux = ctx_aux(ctx);
        if (aux && aux->sh_a.reserve(m * sizeof(Fr)) == hipSuccess) wctx = aux;
    }
    if (wctx != ctx) {   // `we` is complete (and k_check_sap_L has read it) at this point of the stream
        PM_HIP(ctx, hipEventRecord(ctx->ev_sc_a, st));
        PM_HIP(ctx, hipStreamWaitEvent(wctx->stream, ctx->ev_sc_a, 0));
    }
    PM_TRY(dist_intt<C>(ctx, pk, L, ue, ta, u));
    if (q == 0) PM_HIP(ctx, hipMemcpyAsync(u + m, ra, 2 * sizeof(Fr), hipMemcpyDeviceToDevice, st));   // sc_a = u || r_a on rank 0
    PM_TRY(dist_intt<C>(ctx, wctx, pk, L, we, wctx != ctx ? wctx->sh_a.as<Fr>() : ta, wv));
    if (wctx != ctx) PM_HIP(ctx, hipEventRecord(wctx->ev_sc_a, wctx->stream));
    if (sparse_head) {
        const Fr *winv = nullptr;
        PM_TRY(twiddles_get<C>(ctx, pk->log_n, true, &winv));
        StageTimer t(ctx, T_NTT);
        hipLaunchKernelGGL(k_ue_head<P>, dim3(1), dim3(64), 0, st, A, Bm, Cm, xw, tc, m0, nr, (unsigned)(2 * m0));
        PM_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_wit_u_sparse_L<P>, dim3(nblk(m)), dim3(256), 0, st, u, tc, winv, inverse<P>(from_u64<P>(n)), (unsigned)(2 * m0), wit_u, L);
        PM_HIP(ctx, hipGetLastError());
    }
    {   // square_polynomial (prover.rs:315-328) through the negacyclic half (prove.hip: k_twist): u^2 = lo + X^n hi
        const Fr *psi = nullptr, *psi_inv = nullptr;
        PM_TRY(twiddles_get<C>(ctx, pk->log_n + 1, false, &psi));
        PM_TRY(twiddles_get<C>(ctx, pk->log_n + 1, true, &psi_inv));
        hipLaunchKernelGGL(k_twist_L<P>, dim3(nblk(m)), dim3(256), 0, st, u, psi, ta, L);
        PM_HIP(ctx, hipGetLastError());
        PM_TRY(dist_ntt<C>(ctx, pk, L, ta, tb, tc));
        hipLaunchKernelGGL(k_square_L<P>, dim3(nblk(m)), dim3(256), 0, st, tc, m);
        PM_HIP(ctx, hipGetLastError());
        PM_TRY(dist_intt<C>(ctx, pk, L, tc, tb, ta));
        if (wctx != ctx) PM_HIP(ctx, hipStreamWaitEvent(st, wctx->ev_sc_a, 0));   // w's coefficients
        hipLaunchKernelGGL(k_untwist_combine_L<P>, dim3(nblk(m)), dim3(256), 0, st, ta, psi_inv, wv, u2lo, u2hi, L, inverse<P>(from_u64<P>(2)));
        PM_HIP(ctx, hipGetLastError());
    }
    {
        StageTimer t(ctx, T_POLY);
        hipLaunchKernelGGL(k_phase1_scalars_L<P>, dim3(nblk(m)), dim3(256), 0, st, u, u2hi, ra, sc_c + zcnt, sc_c + zcnt + hcnt,
                           tail ? sc_c + zcnt + hcnt + rucnt : (Fr *)nullptr, L, flags);
        PM_HIP(ctx, hipGetLastError());
    }
    hp.mark("transforms");
    // ONE exchange for the whole phase, after the MSMs: the status flags, the two partial points and the halo coefficients
    // phase 3 needs (the last coefficient of each of this rank's blocks).  The flags travel with the points instead of
    // ahead of them: an unsatisfied witness costs the MSMs before every rank reports it -- the rare path pays, not the proof.
    struct Rec1 {
        uint32_t flags, a_inf, c_inf, pad;
        uint64_t a_xy[sizeof(Affine<C>) / 8], c_xy[sizeof(Affine<C>) / 8];
    };
    const size_t rec_bytes = sizeof(Rec1) + (size_t)N * sizeof(Fr);
    std::vector<uint8_t> mine(rec_bytes), all(rec_bytes * N);
    Rec1 *r1 = (Rec1 *)mine.data();
    Fr *my_halo = (Fr *)(mine.data() + sizeof(Rec1));
    PM_HIP(ctx, ctx->halo.reserve(2 * (size_t)N * sizeof(Fr)));            // [halo of my blocks (phase 3) | my blocks' last coefficients]
    Fr *d_last = ctx->halo.as<Fr>() + N;
    hipLaunchKernelGGL(k_halo_pack<P>, dim3((N + 63) / 64), dim3(64), 0, st, (const Fr *)u, L, d_last);
    PM_HIP(ctx, hipGetLastError());
    // flags and halo go to PINNED staging (a pageable destination would make these two copies wait for the stream -- all transforms --
    // before the MSM pipelines below could even be enqueued) and into the record after the MSMs' final synchronisation
    uint8_t *stage = nullptr;
    const size_t halo_bytes = (size_t)N * sizeof(Fr);
    if (ctx_pinned(ctx) && 16 + halo_bytes <= PINNED_STAGE_BYTES) stage = (uint8_t *)ctx->h_pinned + PINNED_SLOTS_BYTES;
    struct DrainOnExit {      // without pinned staging the copies target `mine`: no return path may free it while they are pending
        hipStream_t st;
        bool armed;
        ~DrainOnExit() { if (armed) (void)hipStreamSynchronize(st); }
    } drain{st, stage == nullptr};
    PM_HIP(ctx, hipMemcpyAsync(stage ? (void *)stage : (void *)&r1->flags, flags, 4, hipMemcpyDeviceToHost, st));
    PM_HIP(ctx, hipMemcpyAsync(stage ? (void *)(stage + 16) : (void *)my_halo, d_last, halo_bytes, hipMemcpyDeviceToHost, st));   // both land before the MSM's final sync
    // [a]_1 and [c]_1 are independent MSMs: both pipelines are ENQUEUED before either is waited for -- [a]_1 on the helper
    // context's stream and workspace, behind an event on this stream; [c]_1 here -- so the smaller one's latency-bound sort
    // front end and bucket reduction run under the larger one's accumulation.  One host thread, no collective in between
    // (the single-GPU prover starts [a]_1 earlier, from a second thread: prove.hip).  PM_OPT_MSM_OVERLAP = 0: one after the other.
    int a_inf_l = 1, c_inf_l = 1;
    const bool overlap = ctx->opt.v[PM_OPT_MSM_OVERLAP] != 0;
    if (pm_ctx *aux = overlap ? ctx_aux(ctx) : nullptr) {
        timing_reset_aux(ctx, aux);
        PM_HIP(ctx, hipEventRecord(ctx->ev_sc_a, st));
        PM_HIP(ctx, hipStreamWaitEvent(aux->stream, ctx->ev_sc_a, 0));
        int st_a = msm_resident_begin<C>(aux, pk, 0, u);
        const int st_c = st_a == PM_OK ? msm_resident_begin<C>(ctx, pk, 1, sc_c) : (int)PM_OK;
        const int en_c = st_a == PM_OK && st_c == PM_OK ? msm_resident_end<C>(ctx, r1->c_xy, &c_inf_l) : (int)PM_OK;
        if (st_a == PM_OK) st_a = msm_resident_end<C>(aux, r1->a_xy, &a_inf_l);      // always drained: its scalars live in this context
        timing_flush(aux);
        timing_absorb_aux(ctx, aux);
        if (st_a != PM_OK) { ctx->err = aux->err; return st_a; }
        PM_TRY(st_c);
        PM_TRY(en_c);
    } else {
        PM_TRY(msm_resident<C>(ctx, pk, 0, u, r1->a_xy, &a_inf_l));
        PM_TRY(msm_resident<C>(ctx, pk, 1, sc_c, r1->c_xy, &c_inf_l));
    }
    hp.mark("msm_a+c");
    if (stage) {              // the MSMs' final synchronisation is behind us: the staged words have landed
        memcpy(&r1->flags, stage, 4);
        memcpy(my_halo, stage + 16, halo_bytes);
    }
    drain.armed = false;
    r1->a_inf = (uint32_t)a_inf_l;
    r1->c_inf = (uint32_t)c_inf_l;
    r1->pad = 0;
    PM_TRY(comm_status(ctx, ctx->comm->all_gather(mine.data(), all.data(), rec_bytes, st), "all_gather"));
    hp.mark("exchange");
    unsigned hflags = 0;
    std::vector<uint64_t> pa((size_t)N * (sizeof(Affine<C>) / 8)), pc(pa.size());
    std::vector<int> ia(N), ic(N);
    for (uint32_t r = 0; r < N; ++r) {
        const Rec1 *rr = (const Rec1 *)(all.data() + (size_t)r * rec_bytes);
        hflags |= rr->flags;
        memcpy(&pa[r * (sizeof(Affine<C>) / 8)], rr->a_xy, sizeof(Affine<C>));
        memcpy(&pc[r * (sizeof(Affine<C>) / 8)], rr->c_xy, sizeof(Affine<C>));
        ia[r] = (int)rr->a_inf;
        ic[r] = (int)rr->c_inf;
    }
    if (hflags & 1u) return phase_end.ok(PM_ERR_REMAINDER_NONZERO);                 // prover.rs:108 -- the same flags on every rank
    if ((hflags & 2u) || !(hflags & 4u)) return phase_end.ok(PM_ERR_DEGREE_BOUND);   // prover.rs:107
    // [a]_1, [c]_1 = the sums over the ranks (RCCL has no elliptic-curve reduction: all-gather + local adds)
    PM_TRY(pm_g1_sum(C::ID, pa.data(), ia.data(), N, a_xy, a_inf));
    PM_TRY(pm_g1_sum(C::ID, pc.data(), ic.data(), N, c_xy, c_inf));
    // halo of my block k1 = the coefficient just below its first one: the last of rank q-1's block k1, or (q = 0) of
    // rank N-1's block k1-1; block (0, 0) starts at coefficient 0 and has none
    {
        std::vector<Fr> halo(N, Fr::zero());
        auto last_of = [&](uint32_t r, uint32_t k1) { return ((const Fr *)(all.data() + (size_t)r * rec_bytes + sizeof(Rec1)))[k1]; };
        for (uint32_t k1 = 0; k1 < N; ++k1) {
            if (q > 0) halo[k1] = last_of(q - 1, k1);
            else if (k1 > 0) halo[k1] = last_of(N - 1, k1 - 1);
        }
        PM_HIP(ctx, hipMemcpyAsync(ctx->halo.p, halo.data(), (size_t)N * sizeof(Fr), hipMemcpyHostToDevice, st));
        PM_HIP(ctx, hipStreamSynchronize(st));     // `halo` is a stack vector
    }
    hp.mark("sums+halo");
    t_phase.stop();
    timing_flush(ctx);
    hp.mark("timers");
    PM_TRY(comm_alive(ctx));
    ctx->phase = 1;
    return phase_end.ok(PM_OK);
}

// ------------------------------------------------------------------------------------------------- phase 2
template <class C>
int prove_phase2_sharded(pm_ctx *ctx, const uint64_t *x1_in, uint64_t *u_at_x1) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    if (ctx->phase < 1 || !ctx->pk) return PM_ERR_STATE;
    const pm_pk *pk = ctx->pk;
    PM_TRY(require_comm(ctx, pk));
    PhaseEnd phase_end(ctx, "phase 2");
    HostProfile hp("phase 2", pk->shard_rank);
    const Layout L = pmlayout::make_layout(pk->n, (uint32_t)pk->shard_count, (uint32_t)pk->shard_rank);
    hipStream_t st = ctx->stream;
    const Fr x1 = load_fr<P>(x1_in);
    unsigned Lh = 16;
    while (Lh > L.B) Lh >>= 1;
    const uint64_t lanes = L.m / Lh;
    const unsigned blocks = nblk(lanes);
    const size_t S = pk->segs.size(), SS = pk->seg_slots;   // SS = the longest segment list of any rank (the exchanged record size)
    const size_t rec = 1 + 2 * SS;                          // [ u(x1) partial | P_s (SS) | Q_s (SS) ]
    PM_HIP(ctx, ctx->scratch.reserve(((size_t)blocks + rec) * sizeof(Fr)));
    PM_HIP(ctx, ctx->lvl[0].reserve(2 * S * SEG_THREADS_MAX * sizeof(Fr)));    // lane values of the division scan: laneP | laneQ
    Fr *part = ctx->scratch.as<Fr>(), *out = part + blocks;
    Fr *laneP = ctx->lvl[0].as<Fr>(), *laneQ = laneP + S * SEG_THREADS_MAX;
    hipLaunchKernelGGL(k_horner_partial_L<P>, dim3(blocks), dim3(256), 0, st, ctx->u.as<Fr>(), L, x1, Lh, part);
    PM_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_phase2_pack<P>, dim3(1), dim3(256), 0, st, (const Fr *)part, blocks, out);
    PM_HIP(ctx, hipGetLastError());
    {   // the x1-dependent half of phase 3's division scan (k_seg_base): its segment sums ride on this phase's exchange
        StageTimer t(ctx, T_POLY);
        const SegData sd{pk->n, pk->sigma};
        const uint64_t m = pk->n / (uint64_t)pk->shard_count;
        const Fr *u = ctx->u.as<Fr>(), *wit_u = ctx->wit_u.as<Fr>(), *u2lo = ctx->u2.as<Fr>(), *u2hi = u2lo + m, *halo = ctx->halo.as<Fr>();
        const Segment *d_segs = (const Segment *)pk->d_segs;
        PM_HIP(ctx, hipMemsetAsync(out + 1, 0, 2 * SS * sizeof(Fr), st));
        hipLaunchKernelGGL((k_seg_base<P, SEG_LANES>), dim3((unsigned)S), dim3(SEG_LANES), 0, st, d_segs, sd, u, wit_u, u2lo, u2hi, halo, x1, laneP, laneQ,
                           out + 1, out + 1 + SS);
        PM_HIP(ctx, hipGetLastError());
    }
    std::vector<Fr> mine(rec);
    ctx->seg_all.resize(rec * L.N * sizeof(Fr));
    Fr *all = (Fr *)ctx->seg_all.data();
    PM_HIP(ctx, hipMemcpyAsync(mine.data(), out, rec * sizeof(Fr), hipMemcpyDeviceToHost, st));
    PM_HIP(ctx, hipStreamSynchronize(st));
    hp.mark("horner+seg_base");
    PM_TRY(comm_status(ctx, ctx->comm->all_gather(mine.data(), all, rec * sizeof(Fr), st), "all_gather"));
    hp.mark("exchange");
    Fr sum = Fr::zero();
    for (uint32_t r = 0; r < L.N; ++r) sum = add<P>(sum, all[r * rec]);
    memcpy(u_at_x1, sum.l, sizeof(Fr));
    memcpy(ctx->x1_host, x1_in, sizeof(Fr));                // phase 3 must be called with the same x1: its data sums are in ctx->seg_all
    PM_TRY(comm_alive(ctx));
    ctx->phase = 2;
    return phase_end.ok(PM_OK);
}

// ------------------------------------------------------------------------------------------------- phase 3
template <class C>
int prove_phase3_sharded(pm_ctx *ctx, const uint64_t *x1_in, const uint64_t *x2_in, const uint64_t *a_in, const uint64_t *c_in, uint64_t *d_xy,
                         int *d_inf) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    if (ctx->phase < 2 || !ctx->pk) return PM_ERR_STATE;            // x1 is phase 2's; the halo coefficients came with phase 1's exchange
    const pm_pk *pk = ctx->pk;
    PM_TRY(require_comm(ctx, pk));
    PhaseEnd phase_end(ctx, "phase 3");
    HostProfile hp("phase 3", pk->shard_rank);
    hipStream_t st = ctx->stream;
    if (!ctx->keep_timings) timing_reset(ctx);
    TimingGuard timing_guard{ctx};
    StageTimer t_phase(ctx, T_PHASE);
    const uint64_t n = pk->n, sigma = pk->sigma, m = n / (uint64_t)pk->shard_count;
    const uint32_t N = (uint32_t)pk->shard_count, q = (uint32_t)pk->shard_rank;
    const Fr x1 = load_fr<P>(x1_in), x2 = load_fr<P>(x2_in), a_at = load_fr<P>(a_in), c_at = load_fr<P>(c_in);
    Fr rah[2];
    memcpy(rah, ctx->ra_host, sizeof(rah));          // phase 1 kept the host copy of r_a: no device read-back, no synchronisation here
    const NumConsts<P> nc = make_num_consts<P>(x2, rah, a_at, c_at);
    const SegData sd{n, sigma};
    const size_t S = pk->segs.size(), SS = pk->seg_slots, rec = 1 + 2 * SS;
    if (memcmp(ctx->x1_host, x1_in, sizeof(Fr)) != 0 || ctx->seg_all.size() != rec * N * sizeof(Fr)) {
        ctx->err = "phase 3 needs the x1 of phase 2 (its data sums of the division scan were taken at that point)";
        return phase_end.ok(PM_ERR_INVALID_ARG);   // the same call on every rank: a verdict they share, the communicator stays usable
    }
    PM_HIP(ctx, ctx->quotient.reserve((pk->res_cnt[2] + 1) * sizeof(Fr)));
    PM_HIP(ctx, ctx->lvl[1].reserve((S + 1) * sizeof(Fr)));            // carries
    Fr *qv = ctx->quotient.as<Fr>(), *laneP = ctx->lvl[0].as<Fr>(), *laneQ = laneP + S * SEG_THREADS_MAX, *carry = ctx->lvl[1].as<Fr>();
    const Fr *u = ctx->u.as<Fr>(), *wit_u = ctx->wit_u.as<Fr>(), *u2lo = ctx->u2.as<Fr>(), *u2hi = u2lo + m, *halo = ctx->halo.as<Fr>();
    const Segment *d_segs = (const Segment *)pk->d_segs;
    // The carry chain over ALL segments (k_seg_chain, one workgroup), then the expansion of this rank's segments; phase 2's
    // exchanged data sums go up once (N x (1 + 2 SS) Fr, ~80 KB at N = 8), the remainder H_0 comes back with the MSM's result.
    const size_t T = pk->all_segs.size();
    PM_HIP(ctx, ctx->lvl[2].reserve(rec * N * sizeof(Fr)));
    Fr *d_hall = ctx->lvl[2].as<Fr>(), *d_rem = carry + S;
    if (!ctx_pinned(ctx)) {
        ctx->err = "pinned result slot allocation failed";
        return PM_ERR_HIP;
    }
    Fr *h_rem = (Fr *)((uint8_t *)ctx->h_pinned + 3072);     // pinned: the copy below must not block the host (msm.hip uses [0, 2052))
    *h_rem = Fr::one();
    {
        StageTimer t(ctx, T_POLY);
        PM_HIP(ctx, hipMemcpyAsync(d_hall, ctx->seg_all.data(), rec * N * sizeof(Fr), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_seg_chain<P>, dim3(1), dim3(CHAIN_THREADS), 0, st, (const SegRefDev *)pk->d_all_segs, (unsigned)T, (const Fr *)d_hall, rec, SS,
                           nc, x1, pow_u64<P>(x1, pk->max_seg), pk->max_seg, n, sigma, q, carry, d_rem);
        PM_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL((k_seg_expand<P, SEG_LANES>), dim3((unsigned)S), dim3(SEG_LANES), 0, st, d_segs, sd, nc, u, wit_u, u2lo, u2hi, halo, x1,
                           (const Fr *)laneP, (const Fr *)laneQ, (const Fr *)carry, qv);
        PM_HIP(ctx, hipGetLastError());
        PM_HIP(ctx, hipMemcpyAsync(h_rem, d_rem, sizeof(Fr), hipMemcpyDeviceToHost, st));      // lands before the MSM's final synchronisation
    }
    hp.mark("chain+expand enqueue");
    // [d]_1 = M8, prover.rs:229: this rank's partial sum, then the sum over the ranks (all-gather + local adds, like phase 1)
    {
        struct Rec3 {
            uint64_t xy[sizeof(Affine<C>) / 8];
            uint64_t inf;
        } mine3;
        int inf_l = 1;
        // (Two half-length pipelines with a bucket set each lost 0.6 ms per rank at N = 8: profiles/r03_n_*; chunks of pairs sorted
        // under the accumulation into ONE bucket set lost on one GPU: profiles/r04_chunked_sort_overlap_negative.txt.)
        PM_TRY(msm_resident<C>(ctx, pk, 2, qv, mine3.xy, &inf_l));
        hp.mark("expand+msm_d");
        if (!h_rem->is_zero()) return phase_end.ok(PM_ERR_REMAINDER_NONZERO);   // prover.rs:221 -- H_0 is the same value on every rank
        mine3.inf = (uint64_t)inf_l;
        std::vector<Rec3> all3(N);
        PM_TRY(comm_status(ctx, ctx->comm->all_gather(&mine3, all3.data(), sizeof(Rec3), st), "all_gather"));
        std::vector<uint64_t> pts((size_t)N * (sizeof(Affine<C>) / 8));
        std::vector<int> infs(N);
        for (uint32_t r = 0; r < N; ++r) {
            memcpy(&pts[r * (sizeof(Affine<C>) / 8)], all3[r].xy, sizeof(Affine<C>));
            infs[r] = (int)all3[r].inf;
        }
        hp.mark("exchange_d");
        PM_TRY(pm_g1_sum(C::ID, pts.data(), infs.data(), N, d_xy, d_inf));
        hp.mark("sum");
    }
    t_phase.stop();
    timing_flush(ctx);
    PM_TRY(comm_alive(ctx));
    ctx->phase = 3;
    return phase_end.ok(PM_OK);
}

#define PM_INST_SH(C)                                                                                                               \
    template int prove_phase1_sharded<C>(pm_ctx *, const pm_pk *, const uint64_t *, const uint64_t *, const uint64_t *, uint64_t *, \
                                         int *, uint64_t *, int *, bool);                                                           \
    template int prove_phase2_sharded<C>(pm_ctx *, const uint64_t *, uint64_t *);                                                   \
    template int prove_phase3_sharded<C>(pm_ctx *, const uint64_t *, const uint64_t *, const uint64_t *, const uint64_t *, uint64_t *, int *);
PM_INST_SH(BlsCurve)
PM_INST_SH(BnCurve)

}  // namespace pm
