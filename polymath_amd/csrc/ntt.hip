// Radix-2 NTT / iNTT over the scalar field Fr for gfx950.
//
// Replaces ark-poly Radix2EvaluationDomain::{fft_in_place, ifft_in_place} as the reference calls
// them (/root/reference/src/prover.rs:241,319,325): natural order in, natural order out, omega =
// two_adic_root^(2^(s - log n)), inverse scaled by n^-1.
//
// Kernel plan: the transform is split into passes of up to LOG_TILE = 8 butterfly
// stages.  In one pass a 256-thread workgroup owns a tile of 2^8 rows x 8 columns of Fr
// (64 KiB of LDS out of CDNA4's 160 KiB), loads it with 256-byte-contiguous segments (8 x 32 B),
// runs its 8 stages out of LDS, and stores it back, so a 2^22-point transform touches HBM 3 times
// instead of 22.  Twiddles come from one n/2-entry table per (curve, log n, direction), built
// once per context and L2/MALL resident across passes.
#include <cstring>
#include <type_traits>

#include "internal.h"
#include "fq28.cuh"

namespace pm {

template <class P>
struct PowTable {
    Fp<P> w[33];  // w[k] = omega^(2^k)
};

// out[j] = omega^j in the standard Montgomery form; out_int[j] = the same power as the plain integer
// omega^j 2^(28 N) mod r ("internal" form of fq28.cuh): a reduced-radix product of a standard-form value with
// it stays in the standard form, so the butterflies need no conversion of the data.
// The internal form is kept either dense (out_int, 8 x u32: the 32-bit-limb tile kernels unpack it per butterfly) or already on
// 28-bit limbs (out28, Tw28 records of 10 x u32: the 28-bit tile kernels load it ready for the product).
template <class RR>
struct alignas(8) Tw28 {
    uint32_t l[RR::N];
};

template <class P, class RR>
__global__ void k_twiddles(Fp<P> *out, Fp<P> *out_int, Tw28<RR> *out28, size_t count, PowTable<P> tab, unsigned nbits) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fp<P> acc = Fp<P>::one();
    for (unsigned k = 0; k < nbits; ++k)
        if ((j >> k) & 1) acc = mul<P>(acc, tab.w[k]);
    out[j] = acc;
    Fp<P> c;
    for (int i = 0; i < P::N; ++i) c.l[i] = RR::STD2INT[i];
    const Fp<P> wi = mul<P>(acc, c);
    if (out_int) out_int[j] = wi;
    if (out28) {
        const F28<RR> u = f28_unpack<RR>(wi.l);
        Tw28<RR> r;
#pragma unroll
        for (int i = 0; i < RR::N; ++i) r.l[i] = u.l[i];
        out28[j] = r;
    }
}

template <class P>
__global__ void k_bitrev(Fp<P> *a, unsigned log_n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)1 << log_n;
    if (i >= n) return;
    size_t j = __brevll((unsigned long long)i) >> (64 - log_n);
    if (i < j) {
        Fp<P> x = a[i], y = a[j];
        a[i] = y;
        a[j] = x;
    }
}

// One pass = stages [s0, s0+ns) of the DIT network on bit-reversed input.  Stage s (1-based)
// pairs indices differing in bit s-1; the twiddle of butterfly (blk, j) is tw[j << (log_n - s)].
// Decompose index i = (hi, mid, lo): lo = s0 bits (already-processed strides), mid = ns bits
// (this pass), hi = the rest.  A workgroup takes one `hi`, 2^log_cols consecutive `lo`, all 2^ns `mid`.
constexpr int LOG_TILE = 8;

// The butterfly product runs on 28-bit limbs (fq28.cuh: 200 carry-free mads + 70 instead of the dense
// 32-bit-limb Montgomery product the compiler lowers to ~700 instructions, half of them register moves);
// `tw` holds INTERNAL-form twiddles (k_twiddles), the tile stays canonical standard-form Fr.
// `dst` may be `a` (in place: a workgroup reads and writes the same positions) or another buffer; `do_scale`: multiply the
// outputs by `scale` on the way out (the n^-1 of an inverse transform, folded into its last pass).
template <class P, class RR>
__global__ __launch_bounds__(256) void k_ntt_pass(const Fp<P> *a, Fp<P> *dst, const Fp<P> *tw, unsigned log_n, unsigned s0, unsigned ns,
                                                   unsigned log_cols, Fp<P> scale, int do_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Fp<P> *tile = (Fp<P> *)smem_raw;  // [2^ns][cols]
    const unsigned cols = 1u << log_cols, rows = 1u << ns;
    const size_t lo_groups = ((size_t)1 << s0) >> log_cols;  // groups of `cols` consecutive lo values
    const size_t g = blockIdx.x;
    const size_t hi = g / lo_groups, lo0 = (g % lo_groups) << log_cols;
    const size_t base = (hi << (s0 + ns)) + lo0;
    const unsigned tid = threadIdx.x;
    // load: element (r, c) lives at base + (r << s0) + c
    for (unsigned e = tid; e < rows * cols; e += blockDim.x) {
        unsigned r = e >> log_cols, c = e & (cols - 1);
        tile[e] = a[base + ((size_t)r << s0) + c];
    }
    __syncthreads();
    const unsigned nbf = (rows >> 1) * cols;    // butterflies per stage
    for (unsigned t = 0; t < ns; ++t) {
        const unsigned s = s0 + t + 1;          // global stage, 1-based
        const unsigned half = 1u << t;          // distance in rows
        auto bfly = [&](unsigned r0, unsigned r1, unsigned c, const Fp<P> &w) {
            const Fp<P> x = tile[r0 * cols + c], y0 = tile[r1 * cols + c];
            Fp<P> y;
            f28_pack_reduced<RR>(f28_mul<RR>(f28_unpack<RR>(y0.l), f28_unpack<RR>(w.l)), y.l);
            tile[r0 * cols + c] = add<P>(x, y);
            tile[r1 * cols + c] = sub<P>(x, y);
        };
        // position of the butterfly inside its size-2^s block: j = (r0 mod 2^(t+1)) * 2^s0 + lo
        if (nbf == 4 * blockDim.x) {
            // full tile: issue the four twiddle gathers of this lane's butterflies before the arithmetic
            // (2 waves/SIMD cannot hide a dependent global load per butterfly)
            Fp<P> w[4];
            unsigned r0v[4], cv[4];
#pragma unroll
            for (unsigned q = 0; q < 4; ++q) {
                const unsigned e = tid + q * blockDim.x, c = e & (cols - 1), k = e >> log_cols;
                r0v[q] = ((k >> t) << (t + 1)) | (k & (half - 1));
                cv[q] = c;
                const size_t j = ((size_t)(r0v[q] & (half - 1)) << s0) + lo0 + c;
                w[q] = tw[j << (log_n - s)];
            }
#pragma unroll
            for (unsigned q = 0; q < 4; ++q) bfly(r0v[q], r0v[q] + half, cv[q], w[q]);
        } else {
            for (unsigned e = tid; e < nbf; e += blockDim.x) {
                const unsigned c = e & (cols - 1), k = e >> log_cols;
                const unsigned r0 = ((k >> t) << (t + 1)) | (k & (half - 1));
                const size_t j = ((size_t)(r0 & (half - 1)) << s0) + lo0 + c;
                bfly(r0, r0 + half, c, tw[j << (log_n - s)]);
            }
        }
        __syncthreads();
    }
    for (unsigned e = tid; e < rows * cols; e += blockDim.x) {
        unsigned r = e >> log_cols, c = e & (cols - 1);
        dst[base + ((size_t)r << s0) + c] = do_scale ? mul<P>(tile[e], scale) : tile[e];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round-2 pass kernels: the tile lives in LDS in REDUCED RADIX (C::FrNttRR: 9 x u32 limbs of 29 bits per element, limb-major
// planes -- one bank per lane for every access), so a butterfly is one carry-free product (fq28.cuh) plus lazy limb-wise add / sub:
// no dense <-> reduced-radix conversion of the data and no modular correction per stage.  What the kernels are bound by is the
// multiplier: A/B runs on one box showed that removing 10 % of the instructions -- all of them cheap ones: masks,
// shifts, adds, the twiddle unpack -- or batching the tile's loads changes nothing, so the lever is the number of v_mad_u64_u32:
// 2 N^2 per product, 162 on 9 limbs of 29 bits against 200 on 10 limbs of 28.  A 255-bit field leaves 261 - 255 = 6 spare bits:
// values grow by at most 4p per stage (x + yw < V + 2, x + 4p - yw < V + 4), 37p after nine stages, below the 64p the radix holds
// and the 2^6 p the product tolerates; they are brought back to the canonical range once, on the way out (conditional subtractions
// of 32p ... p; in the last pass of an inverse transform the n^-1 product does it).  Elements enter and leave in the dense
// canonical form: the results are bit-identical.
// 512 threads and one 72 KiB tile per workgroup: two workgroups share a CU's 160 KiB (tools/lds_occupancy.hip) = 4 waves
// per SIMD, and one workgroup's barriers and loads hide under the other's butterflies (round 5: 256 threads on 36 KiB tiles, four
// workgroups per CU, wherever a pass has at most 8 stages -- see TH below).
// Round 5: the workgroup size is a template parameter TH; a full tile is 4 TH elements (TH = 512: 2^11 elements, 72 KiB, two
// workgroups per CU; TH = 256: 2^10 elements, 36 KiB, four per CU -- the same 4 waves per SIMD, but a barrier holds 4 waves
// instead of 8 and a wave's stall costs its SIMD less).
constexpr unsigned L28_BPL = 2;   // butterflies per lane and stage of a full tile
constexpr unsigned L28_EPL = 4;   // elements per lane of a full tile

// TILE (elements per limb plane) is a compile-time constant: the nine accesses of an element are then ONE address register plus the
// immediate offsets l * TILE * 4 of ds_read_b32 / ds_write_b32 (round 6; with a run-time plane stride the compiler kept one address
// VGPR per limb plane and element -- 36 live registers and as many adds in the two-stage body, at the 128-VGPR cap)
template <class RR, unsigned TILE>
__device__ __forceinline__ F28<RR> l28_load(const uint32_t *t, unsigned e) {
    F28<RR> r;
    const uint32_t *q = t + e;
#pragma unroll
    for (int l = 0; l < RR::N; ++l) r.l[l] = q[l * TILE];
    return r;
}
template <class RR, unsigned TILE>
__device__ __forceinline__ void l28_store(uint32_t *t, unsigned e, const F28<RR> &v) {
    uint32_t *q = t + e;
#pragma unroll
    for (int l = 0; l < RR::N; ++l) q[l * TILE] = v.l[l];
}
// canonicalisation of lazily grown tile values: fq28.cuh (f28_canonical / f28_pack_canonical; shared with the division scan since round 5)
#ifndef PM_NTT_CANONICAL_QUOT
#define PM_NTT_CANONICAL_QUOT 1
#endif
template <class RR, int JMAX = 5>
__device__ __forceinline__ F28<RR> l28_canonical(const F28<RR> &x) {
    if (JMAX >= 2 && PM_NTT_CANONICAL_QUOT) return f28_canonical_quot<RR, JMAX>(x);      // one quotient step + one conditional subtraction
    return f28_canonical_lazy<RR, JMAX>(x);
}
template <class RR>
__device__ __forceinline__ void l28_pack_canonical(const F28<RR> &c, uint32_t *d) { f28_pack_canonical<RR>(c, d); }

// the ns butterfly stages on the tile; element (r, c) sits at slot r * cols + (FIRST ? (c + r) & (cols - 1) : c).
// Carries are propagated after every second stage only: a stage adds at most 2^(W+1) to a limb (x + yw: + 2^W; x + K4 - yw: K4's
// limbs are < 2^(W+1)), so a limb entering a product is below 2^W + 2^(W+1) and the column sums stay in 64 bits
// (W = 29, N = 9: 9 * 1.5 * 2^30 * 2^29 + 9 * 2^58 < 2^63 against a tight twiddle; the n^-1 product of l28_emit sees at most
// 2.5 * 2^30: < 2^63.8).  The stage that ends the pass leaves its carries to l28_emit.
template <class P, class RR, bool FIRST, unsigned TH>
__device__ __forceinline__ void l28_stages(uint32_t *t, const Tw28<RR> *tw, unsigned log_n, unsigned s0, unsigned ns, unsigned log_cols,
                                           size_t lo0, bool pair_stages) {
    constexpr unsigned tile = L28_EPL * TH;      // every workgroup of every pass holds a full tile (checked by the host: ntt_run)
    const unsigned cols = 1u << log_cols, cm = cols - 1, rows = 1u << ns, nbf = (rows >> 1) * cols, tid = threadIdx.x;
    // Two stages per LDS round trip (round 3): a lane takes the four rows r0 + {0, h, 2h, 3h} of one column, runs stage st on
    // (r0, r0 + h), (r0 + 2h, r0 + 3h) and stage st + 1 on (r0, r0 + 2h), (r0 + h, r0 + 3h) in registers -- the same four products,
    // additions and (after the odd stage) weak normalisations in the same order, so the values are bit-identical to the
    // stage-at-a-time loop below; three twiddle records instead of four (stage st's is shared), half the barriers and half the
    // LDS traffic.  An odd stage count ends with one ordinary stage.  PM_NTT_RADIX4=0: one stage at a time (round 2).
    auto two_stages = [&](unsigned st, auto norm_tag) {
        constexpr bool NORM = decltype(norm_tag)::value;
        const unsigned half = 1u << st, sA = s0 + st + 1, sB = sA + 1, ngroups = (rows >> 2) * cols;
        for (unsigned e = tid; e < ngroups; e += TH) {
            const unsigned c = e & cm, k = e >> log_cols;
            const unsigned r0 = ((k >> st) << (st + 2)) | (k & (half - 1)), r1 = r0 + half, r2 = r1 + half, r3 = r2 + half;
            const unsigned low = r0 & (half - 1);
            const size_t jA = FIRST ? (size_t)low : ((size_t)low << s0) + lo0 + c;
            const size_t jB0 = jA, jB1 = FIRST ? (size_t)(low + half) : ((size_t)(low + half) << s0) + lo0 + c;
            const unsigned e0 = r0 * cols + (FIRST ? ((c + r0) & cm) : c), e1 = r1 * cols + (FIRST ? ((c + r1) & cm) : c);
            const unsigned e2 = r2 * cols + (FIRST ? ((c + r2) & cm) : c), e3 = r3 * cols + (FIRST ? ((c + r3) & cm) : c);
            // register budget (4 waves per SIMD = 128 VGPRs, a 64-bit column accumulator of 18 inside every product): one
            // butterfly's operands at a time -- an element is loaded right before its product, a twiddle right before its stage
            F28<RR> w, a0, a1, a2, a3;
            {
                const Tw28<RR> wA = tw[jA << (log_n - sA)];
#pragma unroll
                for (int i = 0; i < RR::N; ++i) w.l[i] = wA.l[i];
            }
            {
                const F28<RR> p1 = f28_mul<RR>(l28_load<RR, tile>(t, e1), w), x0 = l28_load<RR, tile>(t, e0);
                a0 = f28_add<RR>(x0, p1);
                a1 = f28_sub_k4<RR>(x0, p1);
            }
            {
                const F28<RR> p3 = f28_mul<RR>(l28_load<RR, tile>(t, e3), w), x2 = l28_load<RR, tile>(t, e2);
                a2 = f28_add<RR>(x2, p3);
                a3 = f28_sub_k4<RR>(x2, p3);
            }
            {
                const Tw28<RR> wB0 = tw[jB0 << (log_n - sB)];
#pragma unroll
                for (int i = 0; i < RR::N; ++i) w.l[i] = wB0.l[i];
            }
            const F28<RR> q2 = f28_mul<RR>(a2, w);
            const F28<RR> b0 = f28_add<RR>(a0, q2), b2 = f28_sub_k4<RR>(a0, q2);
            if (NORM) {
                l28_store<RR, tile>(t, e0, f28_weak_norm<RR>(b0));
                l28_store<RR, tile>(t, e2, f28_weak_norm<RR>(b2));
            } else {
                l28_store<RR, tile>(t, e0, b0);
                l28_store<RR, tile>(t, e2, b2);
            }
            {
                const Tw28<RR> wB1 = tw[jB1 << (log_n - sB)];
#pragma unroll
                for (int i = 0; i < RR::N; ++i) w.l[i] = wB1.l[i];
            }
            const F28<RR> q3 = f28_mul<RR>(a3, w);
            const F28<RR> b1 = f28_add<RR>(a1, q3), b3 = f28_sub_k4<RR>(a1, q3);
            if (NORM) {
                l28_store<RR, tile>(t, e1, f28_weak_norm<RR>(b1));
                l28_store<RR, tile>(t, e3, f28_weak_norm<RR>(b3));
            } else {
                l28_store<RR, tile>(t, e1, b1);
                l28_store<RR, tile>(t, e3, b3);
            }
        }
    };
    for (unsigned st = 0; st < ns; ++st) {
        if (pair_stages && (st & 1) == 0 && st + 1 < ns) {          // stages st (even: no normalisation) and st + 1 (odd) together
            if (st + 2 != ns)
                two_stages(st, std::true_type{});
            else
                two_stages(st, std::false_type{});                    // the stage that ends the pass leaves its carries to l28_emit
            __syncthreads();
            ++st;
            continue;
        }
        const unsigned s = s0 + st + 1, half = 1u << st;
        const bool norm = (st & 1) == 1 && st + 1 != ns;
        auto slots = [&](unsigned e, unsigned &e0, unsigned &e1, size_t &j) {
            const unsigned c = e & cm, k = e >> log_cols;
            const unsigned r0 = ((k >> st) << (st + 1)) | (k & (half - 1)), r1 = r0 + half;
            j = FIRST ? (size_t)(r0 & (half - 1)) : ((size_t)(r0 & (half - 1)) << s0) + lo0 + c;
            e0 = r0 * cols + (FIRST ? ((c + r0) & cm) : c);
            e1 = r1 * cols + (FIRST ? ((c + r1) & cm) : c);
        };
        // NORM is a compile-time flag of the stage body (two instantiations behind one uniform branch): as a run-time select the
        // compiler computes the carry chain in every stage and picks afterwards
        auto stage = [&](auto norm_tag) {
            constexpr bool NORM = decltype(norm_tag)::value;
            auto bfly = [&](unsigned e0, unsigned e1, const Tw28<RR> &wd) {
                const F28<RR> x = l28_load<RR, tile>(t, e0), y = l28_load<RR, tile>(t, e1);
                F28<RR> w;
#pragma unroll
                for (int i = 0; i < RR::N; ++i) w.l[i] = wd.l[i];
                const F28<RR> yw = f28_mul<RR>(y, w);                                    // tight, < 2p
                const F28<RR> lo = f28_add<RR>(x, yw), hi = f28_sub_k4<RR>(x, yw);       // values < V + 2, < V + 4
                if (NORM) {
                    l28_store<RR, tile>(t, e0, f28_weak_norm<RR>(lo));
                    l28_store<RR, tile>(t, e1, f28_weak_norm<RR>(hi));
                } else {
                    l28_store<RR, tile>(t, e0, lo);
                    l28_store<RR, tile>(t, e1, hi);
                }
            };
            if (nbf == L28_BPL * TH) {   // full tile: the twiddle loads of this lane's butterflies are issued before the arithmetic
                Tw28<RR> w[L28_BPL];
                unsigned e0v[L28_BPL], e1v[L28_BPL];
#pragma unroll
                for (unsigned q = 0; q < L28_BPL; ++q) {
                    size_t j;
                    slots(tid + q * TH, e0v[q], e1v[q], j);
                    w[q] = tw[j << (log_n - s)];
                }
#pragma unroll
                for (unsigned q = 0; q < L28_BPL; ++q) bfly(e0v[q], e1v[q], w[q]);
            } else {
                for (unsigned e = tid; e < nbf; e += TH) {
                    unsigned e0, e1;
                    size_t j;
                    slots(e, e0, e1, j);
                    bfly(e0, e1, tw[j << (log_n - s)]);
                }
            }
        };
        if (norm)
            stage(std::true_type{});
        else
            stage(std::false_type{});
        __syncthreads();
    }
}

template <class P, class RR>
__device__ __forceinline__ void l28_emit(const F28<RR> &v, Fp<P> *dst, const F28<RR> &scale28, int do_scale, bool short_pass) {
    // v: limbs < 2^32 (the last stages' carries are still pending), value < (1 + 4 ns) p.  The product takes it as it is and
    // returns < 2p: one conditional subtraction; otherwise the carries are propagated here, once per element and pass, and the
    // chain starts at 16p after a pass of <= 7 stages (short_pass: < 30p), at 32p after 8 or 9 (< 38p).  All three branches are
    // uniform over the workgroup.
    Fp<P> out;
    if (do_scale)
        l28_pack_canonical<RR>(l28_canonical<RR, 0>(f28_mul<RR>(v, scale28)), out.l);
    else if (short_pass)
        l28_pack_canonical<RR>(l28_canonical<RR, 4>(f28_weak_norm<RR>(v)), out.l);
    else
        l28_pack_canonical<RR>(l28_canonical<RR, 5>(f28_weak_norm<RR>(v)), out.l);
    *dst = out;
}

// general pass: stages [s0, s0 + ns), tile of 2^ns rows x 2^log_cols contiguous columns, src -> dst at the same positions
template <class P, class RR, unsigned TH>
__global__ __launch_bounds__(TH) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ntt_pass28(const Fp<P> *a, Fp<P> *dst, const Tw28<RR> *tw, unsigned log_n, unsigned s0, unsigned ns,
                                                             unsigned log_cols, Fp<P> scale_int, int do_scale, int pair_stages) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *t = (uint32_t *)smem_raw;
    constexpr unsigned tile = L28_EPL * TH;      // == rows * cols: full tiles only (ntt_run)
    const unsigned cols = 1u << log_cols;
    const size_t lo_groups = ((size_t)1 << s0) >> log_cols, g = blockIdx.x;
    const size_t hi = g / lo_groups, lo0 = (g % lo_groups) << log_cols, base = (hi << (s0 + ns)) + lo0;
    // every load of the tile is in flight before the first one is unpacked (a 2^11-element tile is 8 per lane = 64 VGPRs):
    // with one load per loop trip the workgroup paid eight HBM round trips in sequence, a quarter of its time
    for (unsigned e8 = threadIdx.x; e8 < tile; e8 += L28_EPL * TH) {
        Fp<P> v[L28_EPL];
#pragma unroll
        for (unsigned q = 0; q < L28_EPL; ++q) {
            const unsigned e = e8 + q * TH, r = e >> log_cols, c = e & (cols - 1);
            if (e < tile) v[q] = a[base + ((size_t)r << s0) + c];
        }
#pragma unroll
        for (unsigned q = 0; q < L28_EPL; ++q) {
            const unsigned e = e8 + q * TH;
            if (e < tile) l28_store<RR, tile>(t, e, f28_unpack<RR>(v[q].l));
        }
    }
    __syncthreads();
    l28_stages<P, RR, false, TH>(t, tw, log_n, s0, ns, log_cols, lo0, pair_stages != 0);
    const F28<RR> sc = f28_unpack<RR>(scale_int.l);
    for (unsigned e = threadIdx.x; e < tile; e += TH) {
        const unsigned r = e >> log_cols, c = e & (cols - 1);
        l28_emit<P, RR>(l28_load<RR, tile>(t, e), &dst[base + ((size_t)r << s0) + c], sc, do_scale, ns <= 7);
    }
}

// FIRST pass with the bit reversal folded into its loads (no k_bitrev round trip): stages [0, ns) of the DIT network
// need, for every value `hi` of the upper H = log_n - ns index bits, the 2^ns elements src[brev(hi 2^ns + r)] =
// src[(brev_ns(r) << H) + brev_H(hi)].  A workgroup takes the 2^log_cols values of hi whose brev_H are CONSECUTIVE
// (8g .. 8g + 7): its loads are then 256-byte-contiguous segments like every other pass's, and it writes, for each of
// its hi, 2^ns contiguous outputs.  Reads and writes touch different positions, so the pass goes src -> dst.
// (src -> dst.)  The tile is swizzled (column (c + r) & (cols - 1)) so that the column-major store phase is bank-conflict free.
template <class P, class RR, unsigned TH>
__global__ __launch_bounds__(TH) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ntt_first_pass28(const Fp<P> *src, Fp<P> *dst, const Tw28<RR> *tw, unsigned log_n, unsigned ns,
                                                                   unsigned log_cols, int pair_stages) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *t = (uint32_t *)smem_raw;
    constexpr unsigned tile = L28_EPL * TH;      // == rows * cols: full tiles only (ntt_run)
    const unsigned cols = 1u << log_cols, rows = 1u << ns, H = log_n - ns, cm = cols - 1;
    const size_t g = blockIdx.x;
    for (unsigned e8 = threadIdx.x; e8 < tile; e8 += L28_EPL * TH) {      // loads batched as in k_ntt_pass28
        Fp<P> v[L28_EPL];
#pragma unroll
        for (unsigned q = 0; q < L28_EPL; ++q) {
            const unsigned e = e8 + q * TH, rb = e >> log_cols, c = e & cm;
            if (e < tile) v[q] = src[((size_t)rb << H) + (g << log_cols) + c];
        }
#pragma unroll
        for (unsigned q = 0; q < L28_EPL; ++q) {
            const unsigned e = e8 + q * TH, rb = e >> log_cols, c = e & cm, r = __brev(rb) >> (32 - ns);
            if (e < tile) l28_store<RR, tile>(t, r * cols + ((c + r) & cm), f28_unpack<RR>(v[q].l));
        }
    }
    __syncthreads();
    l28_stages<P, RR, true, TH>(t, tw, log_n, 0, ns, log_cols, 0, pair_stages != 0);
    const F28<RR> none = f28_zero<RR>();
    for (unsigned e = threadIdx.x; e < tile; e += TH) {
        const unsigned c = e >> ns, r = e & (rows - 1);
        const size_t hi = __brevll((unsigned long long)((g << log_cols) + c)) >> (64 - H);
        l28_emit<P, RR>(l28_load<RR, tile>(t, r * cols + ((c + r) & cm)), &dst[(hi << ns) + r], none, 0, ns <= 7);
    }
}

template <class P>
__global__ void k_scale(Fp<P> *a, size_t n, Fp<P> s) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = mul<P>(a[i], s);
}

// domains transformed by the reduced-radix tile kernels (their twiddle records are then kept on reduced-radix limbs too); smaller
// ones run k_bitrev + k_ntt_pass (+ k_scale)
static bool ntt_l28_domain(unsigned log_n) {
    const unsigned tile_log = log_n > 24 ? (unsigned)LOG_TILE + 1 : (unsigned)LOG_TILE;
    return log_n >= tile_log + 3;
}

template <class C>
static int twiddles_slot(pm_ctx *ctx, unsigned log_n, TwiddleCache **out) {
    typedef typename C::FrP P;
    typedef typename C::FrRR RR;
    typedef typename C::FrNttRR NttRR;
    typedef Fp<P> Fr;
    const int cid = C::ID;
    TwiddleCache *slot = nullptr;
    for (auto &t : ctx->tw)
        if (t.curve == cid && t.log_n == log_n) slot = &t;
    if (slot) slot->stamp = ++ctx->tw_clock;
    if (!slot) {
        // evict the least recently used table
        slot = &ctx->tw[0];
        for (auto &t : ctx->tw)
            if (t.stamp < slot->stamp) slot = &t;
        slot->stamp = ++ctx->tw_clock;
        slot->curve = cid;
        slot->log_n = log_n;
        const bool l28 = ntt_l28_domain(log_n);
        size_t half = log_n ? ((size_t)1 << (log_n - 1)) : 1;
        const size_t int_bytes = half * (l28 ? sizeof(Tw28<NttRR>) : sizeof(Fr));
        PM_HIP(ctx, slot->fwd.reserve(half * sizeof(Fr)));
        PM_HIP(ctx, slot->inv.reserve(half * sizeof(Fr)));
        PM_HIP(ctx, slot->fwd_int.reserve(int_bytes));
        PM_HIP(ctx, slot->inv_int.reserve(int_bytes));
        Fr root;
        for (int i = 0; i < P::N; ++i) root.l[i] = C::ROOT_MONT[i];
        for (unsigned i = log_n; i < (unsigned)C::TWO_ADICITY; ++i) root = sqr<P>(root);
        Fr rinv = inverse<P>(root);
        for (int dir = 0; dir < 2; ++dir) {
            PowTable<P> tab;
            Fr w = dir ? rinv : root;
            for (unsigned k = 0; k < 33; ++k) {
                tab.w[k] = w;
                w = sqr<P>(w);
            }
            Fr *dst = dir ? slot->inv.as<Fr>() : slot->fwd.as<Fr>();
            DevBuf &ib = dir ? slot->inv_int : slot->fwd_int;
            unsigned blocks = (unsigned)((half + 255) / 256);
            if (l28)
                hipLaunchKernelGGL((k_twiddles<P, NttRR>), dim3(blocks), dim3(256), 0, ctx->stream, dst, (Fr *)nullptr, ib.as<Tw28<NttRR>>(), half,
                                   tab, log_n ? log_n - 1 : 0);
            else
                hipLaunchKernelGGL((k_twiddles<P, RR>), dim3(blocks), dim3(256), 0, ctx->stream, dst, ib.as<Fr>(), (Tw28<RR> *)nullptr, half, tab,
                                   log_n ? log_n - 1 : 0);
            PM_HIP(ctx, hipGetLastError());
        }
    }
    *out = slot;
    return PM_OK;
}

template <class C>
int twiddles_get(pm_ctx *ctx, unsigned log_n, bool inv_dir, const Fp<typename C::FrP> **out) {
    TwiddleCache *slot = nullptr;
    PM_TRY(twiddles_slot<C>(ctx, log_n, &slot));
    *out = inv_dir ? slot->inv.as<Fp<typename C::FrP>>() : slot->fwd.as<Fp<typename C::FrP>>();
    return PM_OK;
}

template <class C>
int ntt_run(pm_ctx *ctx, Fp<typename C::FrP> *d, unsigned log_n, bool inv_dir) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    if (log_n > (unsigned)C::TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;  // D::new(..) None, prover.rs:83,317
    if (log_n == 0) return PM_OK;
    StageTimer timer(ctx, T_NTT);
    const size_t n = (size_t)1 << log_n;
    TwiddleCache *slot = nullptr;
    PM_TRY(twiddles_slot<C>(ctx, log_n, &slot));
    const DevBuf &twb = inv_dir ? slot->inv_int : slot->fwd_int;
    const Fr *tw = twb.as<Fr>();                                  // dense internal form, unless ntt_l28_domain(log_n)
    // passes of LOG_TILE = 8 stages on 256-row x 8-column tiles; domains above 2^24 take 9 stages per pass
    // (512 rows x 4 columns, the same 64 KiB of LDS) so that 2^25..2^27 points still need only three passes
    const unsigned tile_log = log_n > 24 ? (unsigned)LOG_TILE + 1 : (unsigned)LOG_TILE;
    const Fr ninv = inv_dir ? inverse<P>(from_u64<P>((uint64_t)n)) : Fr::one();
    if (ntt_l28_domain(log_n)) {
        // the same pass structure on the reduced-radix tiles (k_ntt_pass28, 9 limbs of 29 bits): 2^11 elements x 36 B = 72 KiB of LDS per workgroup
        PM_HIP(ctx, ctx->ntt_tmp.reserve(n * sizeof(Fr)));
        Fr *tmp = ctx->ntt_tmp.as<Fr>();
        typedef typename C::FrNttRR RR;
        const Tw28<RR> *tw28 = twb.as<Tw28<RR>>();
        // workgroup shape (round 5): TH = 256 lanes on 2^10-element tiles where no pass has more than 8 stages -- an 8-stage pass
        // reads segments of 2^(10 - 8) = 4 elements (128 B), shorter passes longer ones; TH = 512 on 2^11-element tiles otherwise
        // (9-stage passes, domains above 2^24), where the small tile's segments would shrink to 64 B
        const unsigned p8 = (log_n + 7) / 8, p9 = (log_n + 8) / 9, npass = p9 < p8 ? p9 : p8;
        const unsigned ns_max = log_n / npass + (log_n % npass ? 1 : 0);
        // same-box A/B (profiles/r05_ntt_workgroup_shape_ab.txt): 2^21 forward 0.289 -> 0.276 ms, 2^22 inverse 0.562 -> 0.533 ms; 9-stage
        // passes would read 64-byte segments on the small tile and stay on the large one
#ifndef PM_NTT_TH_SMALL_MAX_NS
#define PM_NTT_TH_SMALL_MAX_NS 8
#endif
        const bool small = ns_max <= PM_NTT_TH_SMALL_MAX_NS;
        const unsigned tile_log28 = small ? 10 : 11;
        // the raised dynamic-LDS limit is a per-DEVICE function attribute: remembered per context (a context lives on one
        // device and runs one proof at a time), not in a process-wide flag -- a local group may span several devices and its
        // rank threads call this concurrently
        if (!ctx->ntt_lds_attr[C::ID]) {
            PM_HIP(ctx, hipFuncSetAttribute((const void *)k_ntt_pass28<P, RR, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)1 << 11) * RR::N * 4)));
            PM_HIP(ctx, hipFuncSetAttribute((const void *)k_ntt_first_pass28<P, RR, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)1 << 11) * RR::N * 4)));
            ctx->ntt_lds_attr[C::ID] = true;
        }
        Fr scale_int;
        for (int i = 0; i < P::N; ++i) scale_int.l[i] = RR::STD2INT[i];
        scale_int = mul<P>(ninv, scale_int);                      // n^-1 2^(W N) as a plain integer: the product with it scales AND reduces
        // passes balanced over the stages, every tile 2^11 (2^10) elements (21 = 7 + 7 + 7 with 16-column tiles, not 8 + 8 + 5 with a
        // last pass whose workgroups hold 2^8 elements; 17 and 18 take two 9-stage passes instead of three)
        const int pair_stages = 1;   // two stages per LDS round trip (profiles/r03_ntt_two_stage_ab.txt: -7 ... -13 % per transform)
        unsigned s0 = 0;
        for (unsigned k = 0; k < npass; ++k) {
            const unsigned ns = log_n / npass + (k < log_n % npass ? 1 : 0);
            unsigned log_cols = tile_log28 - ns;
            if (k && s0 < log_cols) return PM_ERR_STATE;          // never for log_n >= 11: the kernels' limb-plane stride is the FULL tile
            const bool last = s0 + ns == log_n;
            if (k == 0) {
                const dim3 grid((unsigned)(n >> (ns + log_cols)));
                const size_t lds = ((size_t)1 << (ns + log_cols)) * RR::N * 4;
                if (small)
                    hipLaunchKernelGGL((k_ntt_first_pass28<P, RR, 256>), grid, dim3(256), lds, ctx->stream, (const Fr *)d, tmp, tw28, log_n, ns, log_cols, pair_stages);
                else
                    hipLaunchKernelGGL((k_ntt_first_pass28<P, RR, 512>), grid, dim3(512), lds, ctx->stream, (const Fr *)d, tmp, tw28, log_n, ns, log_cols, pair_stages);
            } else {
                if (s0 < log_cols) log_cols = s0;
                const dim3 grid((unsigned)(n >> (ns + log_cols)));
                const size_t lds = ((size_t)1 << (ns + log_cols)) * RR::N * 4;
                if (small)
                    hipLaunchKernelGGL((k_ntt_pass28<P, RR, 256>), grid, dim3(256), lds, ctx->stream, (const Fr *)tmp, last ? d : tmp, tw28, log_n, s0, ns,
                                       log_cols, scale_int, last && inv_dir ? 1 : 0, pair_stages);
                else
                    hipLaunchKernelGGL((k_ntt_pass28<P, RR, 512>), grid, dim3(512), lds, ctx->stream, (const Fr *)tmp, last ? d : tmp, tw28, log_n, s0, ns,
                                       log_cols, scale_int, last && inv_dir ? 1 : 0, pair_stages);
            }
            PM_HIP(ctx, hipGetLastError());
            s0 += ns;
        }
        return PM_OK;
    }
    hipLaunchKernelGGL(k_bitrev<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d, log_n);
    PM_HIP(ctx, hipGetLastError());
    unsigned s0 = 0;
    while (s0 < log_n) {
        unsigned ns = log_n - s0 < tile_log ? log_n - s0 : tile_log;
        // columns: consecutive `lo` values (contiguous in memory); needs s0 >= log_cols; tile <= 2^11 elements
        unsigned log_cols = 11 - ns < 3 ? 11 - ns : 3;
        if (s0 < log_cols) log_cols = s0;
        size_t tiles = n >> (ns + log_cols);
        size_t lds = ((size_t)1 << (ns + log_cols)) * sizeof(Fr);
        hipLaunchKernelGGL((k_ntt_pass<P, typename C::FrRR>), dim3((unsigned)tiles), dim3(256), lds, ctx->stream, (const Fr *)d, d, tw, log_n, s0, ns,
                           log_cols, ninv, 0);
        PM_HIP(ctx, hipGetLastError());
        s0 += ns;
    }
    if (inv_dir) {
        hipLaunchKernelGGL(k_scale<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d, n, ninv);
        PM_HIP(ctx, hipGetLastError());
    }
    return PM_OK;
}

template int twiddles_get<BlsCurve>(pm_ctx *, unsigned, bool, const Fp<BlsFrP> **);
template int twiddles_get<BnCurve>(pm_ctx *, unsigned, bool, const Fp<BnFrP> **);
template int ntt_run<BlsCurve>(pm_ctx *, Fp<BlsFrP> *, unsigned, bool);
template int ntt_run<BnCurve>(pm_ctx *, Fp<BnFrP> *, unsigned, bool);

}  // namespace pm
