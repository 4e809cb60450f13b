// Table-mode bucket reduction (the window-table MSM of msm.hip: ONE set of NB >= 4096 buckets): levels 0 / 1 / final and the
// lane-group point operations they run on.  A translation unit of its own: it is a third of msm.hip's compile time.
#include "internal.h"
#include "fq28.cuh"

namespace pm {

// Table-mode bucket reduction  S = sum_b (b + 1) B_b  over ONE set of NB buckets.  Every step below is a
// chain of DEPENDENT point additions (~21 us each on a lone wave), so the layout minimises chain length,
// not work:
//   level 0  lane t owns buckets [t K0, t K0 + K0):  A_t = sum B_b,  acc_t = sum (b - t K0 + 1) B_b
//            (2 K0 adds; K0 is chosen so that NB / K0 ~ 2^17 lanes = one full round of the chip);
//   level 1  lane j owns R level-0 outputs: running sums, (jR) * sum A by double-and-add, + acc_t; LDS tree
//            per workgroup (2^15 lanes: a short chain on a quarter-full chip beats a full chip of scalar muls);
//   final    one workgroup sums the level-1 partials.
//   S = sum_t acc_t + sum_t (t K0) A_t.
template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_reduce_level0(const XYZZ<C> *partials, const uint32_t *task_off, const uint32_t *task_cnt, size_t nbuckets, size_t lanes,
                                                       unsigned K0, XYZZ<C> *outA, XYZZ<C> *outAcc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XYZZ28<C> *sh = (XYZZ28<C> *)smem_raw;
    typedef typename C::FqRR RR;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    XYZZ28<C> run;
    run.X = run.Y = run.ZZ = run.ZZZ = f28_zero<RR>();
    XYZZ28<C> *acc = &sh[threadIdx.x];     // the weighted accumulator lives in LDS (register pressure)
    *acc = run;
    for (int j = (int)K0 - 1; j >= 0; --j) {
        const size_t g = t * K0 + (size_t)j;
        if (g >= nbuckets) continue;
        for (uint32_t q = task_off[g], qe = q + task_cnt[g]; q < qe; ++q) xyzz28_add_full<C>(run, xyzz28_load<C>(partials[q]));
        xyzz28_add_into_full<C>(acc, run);
    }
    outA[t] = xyzz28_store<C>(run);
    outAcc[t] = xyzz28_store<C>(*acc);
}

// Several bucket SETS in one launch (the wide-window MSM without tables reduces all of its windows at once): the level-0 outputs
// of set s are lanes [s set_lanes, (s + 1) set_lanes) and the weight of lane t is its index INSIDE its set; set_lanes is a multiple
// of the lanes a level-1 workgroup covers, so no workgroup straddles two sets.

// ------------------------------------------------------------------------------ lane groups
// Levels 1 and the final sum are chains of DEPENDENT point operations on a chip that is at most a quarter full, so what counts is
// the latency of one operation, and a point addition is 12 products + 2 squares of which only the critical path has to be
// sequential.  The four lanes of a group hold the SAME point and each computes one product per step, exchanging field elements
// through DPP quad permutations (register moves, no LDS).  The four lanes run ONE instruction stream: operands are picked per role
// with v_cndmask.  Same formulas, bounds and invariants as xyzz28_add / xyzz28_dbl (X: W < 14p, Y: W < 6p, ZZ, ZZZ: T); Y3 is the
// sum of two products here (tight + tight, then a carry propagation) instead of one fused one.  (One lane and two lanes per point,
// round 2's PM_RED_PAIR = 0 / 2, measured 1.70 / 1.48 ms against 1.39 at 2^21 buckets and 1.01 / 0.77 against 0.66 at 2^19:
// profiles/r02_m_reduce_pair_sweep.txt; removed in round 4.)
template <class RR>
__device__ __forceinline__ F28<RR> f28_pick(bool c, const F28<RR> &a, const F28<RR> &b) {   // c ? a : b
    F28<RR> r;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}

// ---- four lanes per point: the critical path of an addition is 3M + 1S (U/S products | P^2, R^2 | PPP, Q, ZZ1 ZZ2, ZZZ1 ZZZ2 |
// R QX, S1 (4p - PPP), ZZ3, ZZZ3), one product per lane and step; results travel by DPP quad broadcasts.  The cheap limb arithmetic
// between the steps (X3, Q - X3, 4p - PPP) is done by all four lanes on broadcast values, so X3 needs no trip back.
template <int CTRL, class RR>
__device__ __forceinline__ F28<RR> f28_dpp(const F28<RR> &a) {
    F28<RR> r;
#pragma unroll
    for (int i = 0; i < RR::N; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], CTRL, 0xF, 0xF, true);
    return r;
}
constexpr int QP_SWAP = 0xB1, QP_B0 = 0x00, QP_B1 = 0x55, QP_B2 = 0xAA, QP_B3 = 0xFF;   // quad_perm [1,0,3,2]; broadcast lane k

template <class C>
__device__ __forceinline__ bool xyzz28_add_quad(XYZZ28<C> &a, const XYZZ28<C> &b, unsigned role) {
    typedef typename C::FqRR RR;
    typedef F28<RR> F;
    if (f28_all_zero<RR>(b.ZZ)) return true;
    if (f28_all_zero<RR>(a.ZZ)) { a = b; return true; }
    const bool b0 = role & 1, b1 = role & 2;
    // lane 0: U1 = X1 ZZ2, lane 1: U2 = X2 ZZ1, lane 2: S1 = Y1 ZZZ2, lane 3: S2 = Y2 ZZZ1
    const F T = f28_mul<RR>(f28_pick<RR>(b1, f28_pick<RR>(b0, b.Y, a.Y), f28_pick<RR>(b0, b.X, a.X)),
                            f28_pick<RR>(b1, f28_pick<RR>(b0, a.ZZZ, b.ZZZ), f28_pick<RR>(b0, a.ZZ, b.ZZ)));
    const F xT = f28_dpp<QP_SWAP, RR>(T);
    const F lo = f28_pick<RR>(b0, xT, T);                                   // U1 (lanes 0, 1) | S1 (lanes 2, 3)
    const F D = f28_sub_k4<RR>(f28_pick<RR>(b0, T, xT), lo);               // P | R                  L e<=1.6, < 6p
    const F DD = f28_sqr<RR>(D);                                            // PP | RR
    if (__builtin_amdgcn_update_dpp(0, (int)f28_is_zero_mod_p<RR>(DD), QP_B0, 0xF, 0xF, true)) return false;
    // lane 0: PPP = P PP, lane 1: Q = U1 PP, lane 2: ZZ1 ZZ2, lane 3: ZZZ1 ZZZ2
    const F V = f28_mul<RR>(f28_pick<RR>(b1, f28_pick<RR>(b0, a.ZZZ, a.ZZ), f28_pick<RR>(b0, lo, D)),
                            f28_pick<RR>(b1, f28_pick<RR>(b0, b.ZZZ, b.ZZ), DD));
    const F PPP = f28_dpp<QP_B0, RR>(V), Q = f28_dpp<QP_B1, RR>(V), PP = f28_dpp<QP_B0, RR>(DD), RR2 = f28_dpp<QP_B2, RR>(DD);
    const F Rv = f28_dpp<QP_B2, RR>(D), S1 = f28_dpp<QP_B2, RR>(lo);
    F X3 = f28_sub_k4<RR>(RR2, PPP);
    X3 = f28_weak_norm<RR>(f28_sub_k8<RR>(X3, f28_add<RR>(Q, Q)));          // W, < 14p
    const F QX = f28_sub_k16<RR>(Q, X3);                                    // L e<=1.6, < 18p
    const F nPPP = f28_sub_k4<RR>(f28_zero<RR>(), PPP);                     // limbs < 2^29, < 4p
    // lane 0: R QX, lane 1: S1 (4p - PPP), lane 2: ZZ3 = (ZZ1 ZZ2) PP, lane 3: ZZZ3 = (ZZZ1 ZZZ2) PPP
    const F W = f28_mul<RR>(f28_pick<RR>(b1, V, f28_pick<RR>(b0, S1, Rv)),
                            f28_pick<RR>(b1, f28_pick<RR>(b0, PPP, PP), f28_pick<RR>(b0, nPPP, QX)));
    a.X = X3;
    a.Y = f28_weak_norm<RR>(f28_add<RR>(f28_dpp<QP_B0, RR>(W), f28_dpp<QP_B1, RR>(W)));   // W, < 4p
    a.ZZ = f28_dpp<QP_B2, RR>(W);
    a.ZZZ = f28_dpp<QP_B3, RR>(W);
    return true;
}
template <class C>
__device__ __forceinline__ void xyzz28_dbl_quad(XYZZ28<C> &a, unsigned role) {
    typedef typename C::FqRR RR;
    typedef F28<RR> F;
    if (f28_all_zero<RR>(a.ZZ)) return;
    const bool b0 = role & 1, b1 = role & 2;
    const F U = f28_add<RR>(a.Y, a.Y);                                      // limbs < 2^29, < 12p
    const F A = f28_sqr<RR>(f28_pick<RR>(b0, a.X, U));                      // lanes 0, 2: V = U^2 | lanes 1, 3: X^2
    const F M = f28_add<RR>(f28_add<RR>(A, A), A);                          // lanes 1, 3: 3 X^2, limbs < 3 2^28, < 6p
    const F V = f28_dpp<QP_B0, RR>(A);
    // lane 0: W = U V, lane 1: M^2, lane 2: S = X V, lane 3: M^2 (again)
    const F B = f28_mul<RR>(f28_pick<RR>(b0, M, f28_pick<RR>(b1, a.X, U)), f28_pick<RR>(b0, M, V));
    const F Wv = f28_dpp<QP_B0, RR>(B), MM = f28_dpp<QP_B1, RR>(B), S = f28_dpp<QP_B2, RR>(B);
    const F X3 = f28_weak_norm<RR>(f28_sub_k8<RR>(MM, f28_add<RR>(S, S)));  // W, < 10p
    const F SX = f28_sub_k16<RR>(S, X3);                                    // L e<=1.6, < 18p
    const F nW = f28_sub_k4<RR>(f28_zero<RR>(), Wv);
    // lane 0: ZZ3 = V ZZ, lane 1: M (S - X3), lane 2: ZZZ3 = W ZZZ, lane 3: Y (4p - W)
    const F Cc = f28_mul<RR>(f28_pick<RR>(b1, f28_pick<RR>(b0, a.Y, Wv), f28_pick<RR>(b0, M, V)),
                             f28_pick<RR>(b1, f28_pick<RR>(b0, nW, a.ZZZ), f28_pick<RR>(b0, SX, a.ZZ)));
    a.X = X3;
    a.Y = f28_weak_norm<RR>(f28_add<RR>(f28_dpp<QP_B1, RR>(Cc), f28_dpp<QP_B3, RR>(Cc)));   // W, < 4p
    a.ZZ = f28_dpp<QP_B0, RR>(Cc);
    a.ZZZ = f28_dpp<QP_B2, RR>(Cc);
}

// one interface for the kernels: LP = 4 lanes per point, role = lane % LP
template <class C, unsigned LP>
__device__ __forceinline__ void xyzz28_add_coop(XYZZ28<C> &a, const XYZZ28<C> &b, unsigned role) {
    static_assert(LP == 4, "four lanes per point");
    if (!xyzz28_add_quad<C>(a, b, role)) a = xyzz28_add_exceptional<C>(a, b);     // every lane of the group, the same result
}
template <class C, unsigned LP>
__device__ __forceinline__ void xyzz28_dbl_coop(XYZZ28<C> &a, unsigned role) {
    static_assert(LP == 4, "four lanes per point");
    xyzz28_dbl_quad<C>(a, role);
}

// sh[k], k < blockDim.x / LP: the value of lane group k on entry (written by its role-0 lane); on exit sh[0] = the workgroup sum
template <class C, unsigned LP>
__device__ __forceinline__ void lds_tree_sum_coop(XYZZ28<C> *sh) {
    const unsigned k = threadIdx.x / LP, role = threadIdx.x % LP;
    __syncthreads();
    for (unsigned off = blockDim.x / LP / 2; off > 0; off >>= 1) {
        if (k < off) {
            XYZZ28<C> a = sh[k];
            xyzz28_add_coop<C, LP>(a, sh[k + off], role);
            if (!role) sh[k] = a;
        }
        __syncthreads();
    }
}

// k_reduce_level1 on lane groups: group j owns R level-0 outputs; t0 * run goes through the non-adjacent form of t0 (digits
// +-1, no two adjacent: the lanes of a wave cover every pattern of the low bits, so what a wave pays is the weight of the HIGH
// bits it shares -- at most half of them in this form -- plus one addition per low position).
template <class C, unsigned LP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_reduce_level1_coop(const XYZZ<C> *A, const XYZZ<C> *Acc, size_t lanes0,
                                                                                                         unsigned K0, unsigned R, XYZZ<C> *out, size_t set_lanes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XYZZ28<C> *sh = (XYZZ28<C> *)smem_raw;
    typedef typename C::FqRR RR;
    const size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LP, t0 = j * R;
    const unsigned role = threadIdx.x % LP;
    XYZZ28<C> acc;
    acc.X = acc.Y = acc.ZZ = acc.ZZZ = f28_zero<RR>();
    if (t0 < lanes0) {
        XYZZ28<C> run = acc;
        for (int i = (int)R - 1; i >= 0; --i) {
            if (t0 + i >= lanes0) continue;
            xyzz28_add_coop<C, LP>(run, xyzz28_load<C>(A[t0 + i]), role);
            if (i > 0) xyzz28_add_coop<C, LP>(acc, run, role);          // weight i
        }
        const size_t tw = t0 % set_lanes;                              // position inside the bucket set (k_reduce_level1)
        if (tw && !f28_all_zero<RR>(run.ZZ)) {                         // acc += tw * run
            uint32_t pos = 0, neg = 0;
            {
                uint64_t x = tw;
                for (unsigned b = 0; x; ++b, x >>= 1)
                    if (x & 1) {
                        if ((x & 3) == 3) { neg |= 1u << b; x += 1; } else { pos |= 1u << b; x -= 1; }
                    }
            }
            // -run: Y back below 2p (one product by the radix)
            const F28<RR> nY = f28_mul<RR>(f28_sub_k16<RR>(f28_zero<RR>(), run.Y), f28_one<RR>());
            XYZZ28<C> m = run;                                         // the leading digit of a positive number is +1
            for (int b = 30 - __clz((int)(pos | neg)); b >= 0; --b) {
                xyzz28_dbl_coop<C, LP>(m, role);
                if (((pos | neg) >> b) & 1) {
                    XYZZ28<C> t = run;
                    t.Y = f28_pick<RR>((neg >> b) & 1, nY, run.Y);
                    xyzz28_add_coop<C, LP>(m, t, role);
                }
            }
            xyzz28_add_coop<C, LP>(acc, m, role);
        }
        for (unsigned k = 1; k < K0; k <<= 1) xyzz28_dbl_coop<C, LP>(acc, role);   // K0 is a power of two
        for (unsigned i = 0; i < R; ++i)
            if (t0 + i < lanes0) xyzz28_add_coop<C, LP>(acc, xyzz28_load<C>(Acc[t0 + i]), role);
    }
    if (!role) sh[threadIdx.x / LP] = acc;
    lds_tree_sum_coop<C, LP>(sh);
    if (threadIdx.x == 0) out[blockIdx.x] = xyzz28_store<C>(sh[0]);
}

// out[0] = sum of parts[0 .. count) by one workgroup of 256 / LP lane groups
template <class C, unsigned LP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_sum_final_coop(const XYZZ<C> *parts, unsigned count, XYZZ<C> *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XYZZ28<C> *sh = (XYZZ28<C> *)smem_raw;
    typedef typename C::FqRR RR;
    const unsigned role = threadIdx.x % LP;
    XYZZ28<C> acc;
    acc.X = acc.Y = acc.ZZ = acc.ZZZ = f28_zero<RR>();
    parts += (size_t)blockIdx.x * count;     // workgroup s sums the level-1 partials of bucket set s
    for (unsigned i = threadIdx.x / LP; i < count; i += 256 / LP) xyzz28_add_coop<C, LP>(acc, xyzz28_load<C>(parts[i]), role);
    if (!role) sh[threadIdx.x / LP] = acc;
    lds_tree_sum_coop<C, LP>(sh);
    if (threadIdx.x == 0) out[blockIdx.x] = xyzz28_store<C>(sh[0]);
}

// sum_b (b + 1) B_b over each of `nsets` sets of NB buckets (set s = buckets [s NB, (s + 1) NB)) whose (folded) task partials sit in
// ctx->msm: three launches on ctx->stream, the nsets results (internal form) at (*out)[0 .. nsets) in the workspace.  nsets = 1:
// the shared bucket set of the window-table MSM; nsets = windows: the wide-window MSM without tables.
template <class C>
int reduce_two_level(pm_ctx *ctx, size_t NB, XYZZ<C> **out, unsigned nsets, size_t bucket0) {
    MsmWorkspace &ws = ctx->msm;
    const MsmSet *S = &ws.set;
    if (nsets == 0 || (NB & (NB - 1)) != 0) return PM_ERR_INVALID_ARG;
    const size_t total = NB * nsets;
    unsigned K0 = 4;                                   // level-0 fan-in: <= 2^17 lanes = 2 waves per SIMD, one round of the chip
    while (total / K0 > ((size_t)1 << 17) && K0 < 64) K0 <<= 1;   // (swept in profiles/r02_levers.jsonl: K0 = 2 doubles level 1's work and loses)
    // levels 1 and final on lane GROUPS (4 lanes per point, a share of the products each): 256 / LP points per workgroup, and
    // R = 2 LP keeps level 1 at 2^16 lanes = one wave per SIMD (its register budget); K0 and R swept on MI355X in round 2
    // (profiles/r02_levers.jsonl, r02_m_reduce_pair_sweep.txt)
    constexpr unsigned coop = 4;
    unsigned R1 = 2 * coop;
    const unsigned per_block1 = 256 / coop;                   // points per level-1 workgroup
    const size_t set_lanes = (NB + K0 - 1) / K0;              // level-0 outputs per bucket set
    if (nsets > 1) {   // no level-1 workgroup may straddle two sets
        while (set_lanes % ((size_t)R1 * per_block1) != 0 && R1 > 1) R1 >>= 1;
        if (NB % K0 != 0 || set_lanes % ((size_t)R1 * per_block1) != 0) return PM_ERR_INVALID_ARG;
    }
    const size_t lanes0 = set_lanes * nsets, lanes1 = (lanes0 + R1 - 1) / R1, blocks0 = (lanes0 + 255) / 256,
                 blocks1 = (lanes1 + per_block1 - 1) / per_block1;
    PM_HIP(ctx, ws.wsum.reserve((2 * lanes0 + (blocks0 > blocks1 ? blocks0 : blocks1) + 4 + nsets) * sizeof(XYZZ<C>)));
    XYZZ<C> *A = ws.wsum.as<XYZZ<C>>(), *Acc = A + lanes0, *parts = Acc + lanes0, *dres = parts + (blocks0 > blocks1 ? blocks0 : blocks1);
    hipLaunchKernelGGL(k_reduce_level0<C>, dim3((unsigned)blocks0), dim3(256), 256 * sizeof(XYZZ28<C>), ctx->stream,
                       S->partials.as<XYZZ<C>>(), S->task_off.as<uint32_t>() + bucket0, S->task_cnt.as<uint32_t>() + bucket0, total, lanes0, K0, A, Acc);
    PM_HIP(ctx, hipGetLastError());
    const unsigned per_set = (unsigned)(blocks1 / nsets);     // level-1 workgroups (= partials) per set; exact when nsets > 1
    hipLaunchKernelGGL((k_reduce_level1_coop<C, 4>), dim3((unsigned)blocks1), dim3(256), 64 * sizeof(XYZZ28<C>), ctx->stream, A, Acc,
                       lanes0, K0, R1, parts, set_lanes);
    PM_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL((k_sum_final_coop<C, 4>), dim3(nsets), dim3(256), 64 * sizeof(XYZZ28<C>), ctx->stream, parts, per_set, dres);
    PM_HIP(ctx, hipGetLastError());
    *out = dres;
    return PM_OK;
}

template int reduce_two_level<BlsCurve>(pm_ctx *, size_t, XYZZ<BlsCurve> **, unsigned, size_t);
template int reduce_two_level<BnCurve>(pm_ctx *, size_t, XYZZ<BnCurve> **, unsigned, size_t);

}  // namespace pm
