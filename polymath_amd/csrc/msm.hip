// G1 variable-base multi-scalar multiplication for gfx950 (Pippenger bucket method).
//
// Replaces E::G1::msm_unchecked as the reference calls it (/root/reference/src/prover.rs:380-384,
// call sites :118,:121,:229,:335-354): result = sum_i scalar_i * base_i.  The result is a canonical
// group element, so the schedule below is free to differ from ark-ec's.
//
// Two pipelines share the task / accumulate / reduce kernels: (A) below, per-window Pippenger
// over a plain base array (one-shot pm_msm_g1, keys whose tables do not fit); (B) "Table mode" further down,
// the path every resident key takes.
// Pipeline (A), all on one HIP stream, no host round trip until the final point:
//   k_digits      scalar (Montgomery) -> canonical -> W signed c-bit digits, one u32 per
//                 (window, scalar): (bucket << 1 | negate), NONE for zero digits / infinity bases.
//   k_hist        LDS-staged histogram: a workgroup owns one (chunk, window) and counts into a
//                 2^(c-1)-entry LDS table (128 KiB at c = 16 -- this is what CDNA4's 160 KiB LDS
//                 buys), then flushes its non-zero bins with coalesced global atomics.
//   k_scan        exclusive scans: bucket offsets, and task offsets after splitting buckets
//                 longer than SEG entries into SEG-sized tasks (skewed scalars -> hot buckets).
//   k_scatter     same LDS staging: claim a range per (workgroup, bucket) with one global atomic,
//                 then place entries with LDS atomics -> base indices grouped by bucket.
//   k_task_bins   tasks ordered by descending length, so the lanes of a wave carry equal loads.
//   k_accumulate  one lane per task: XYZZ mixed adds (8M+2S each) over its <= SEG entries;
//                 bases are gathered from HBM by index (96 B affine points, or 128 B TablePoint records).
//   k_task_fold   buckets that own many tasks (skewed scalars) have their partials summed in parallel.
//   k_bucket_reduce  per window sum_b (b+1) * B_b by per-lane running sums over K buckets, a
//                 small scalar multiple, and an LDS tree reduction per workgroup.
//   host_finish   Horner over the W window sums (c doublings each) and one inversion to affine, on the
//                 host: an O(W) dependent chain (see host_finish).
#include <cstdlib>
#include <cstring>

#include "internal.h"
#include "fq28.cuh"

namespace pm {

constexpr uint32_t DIGIT_NONE = 0xFFFFFFFFu;

struct MsmPlan {
    unsigned c, nwin, nbuckets;  // nbuckets = 2^(c-1) per window
    unsigned seg;                // max entries per accumulate task
    size_t len;
    unsigned chunk, nchunks;     // scalars per histogram workgroup
    size_t max_tasks;
};

static MsmPlan make_plan(size_t len, unsigned scalar_bits, unsigned task_len = 0) {
    MsmPlan p;
    p.len = len;
    // choose c minimising  W(c) * (len + 6 * 2^(c-1))   (6 ~ cost of the reduce per bucket in madds)
    double best = 1e300;
    unsigned bc = 4;
    for (unsigned c = 4; c <= 16; ++c) {
        unsigned w = (scalar_bits + 1 + c - 1) / c;
        double cost = (double)w * ((double)len + 6.0 * (double)(1u << (c - 1)));
        if (cost < best) { best = cost; bc = c; }
    }
    p.c = bc;
    p.nwin = (scalar_bits + 1 + bc - 1) / bc;
    p.nbuckets = 1u << (bc - 1);
    size_t avg = len / p.nbuckets + 1;
    size_t seg = 2 * avg;
    if (seg < 64) seg = 64;
    p.seg = (unsigned)seg;
    p.chunk = 1u << 15;   // scalars per (chunk, window) histogram / scatter workgroup (tools/msm_bench.py sweep)
    if (len < p.chunk) p.chunk = (unsigned)(len ? len : 1);
    p.nchunks = (unsigned)((len + p.chunk - 1) / p.chunk);
    if (task_len) p.seg = task_len;   // PM_OPT_MSM_TASK_LEN (tuning sweeps)
    p.max_tasks = (size_t)p.nwin * p.nbuckets + ((size_t)p.nwin * len) / p.seg + 1;
    return p;
}

// windows (= bucket additions per pair) and window width the per-window pipeline picks for `len` pairs
void msm_plan_query(size_t len, unsigned scalar_bits, unsigned *nwin, unsigned *c) {
    MsmPlan p = make_plan(len ? len : 1, scalar_bits);
    *nwin = p.nwin;
    *c = p.c;
}

// ------------------------------------------------------------------------------ digits
template <class C>
__global__ void k_digits(const Fp<typename C::FrP> *scalars, const Affine<C> *bases, uint32_t *digits, size_t len,
                         unsigned c, unsigned nwin) {
    typedef typename C::FrP P;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    Fp<P> k = from_mont<P>(scalars[i]);
    // a base at infinity contributes nothing (ark: msm adds the identity); drop it here so it
    // never reaches a bucket
    const uint32_t *bx = (const uint32_t *)&bases[i];
    uint32_t any = 0;
#pragma unroll
    for (int t = 0; t < 2 * C::FqP::N; ++t) any |= bx[t];
    const bool skip = (any == 0);
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1), full = 1u << c, mask = full - 1;
    for (unsigned w = 0; w < nwin; ++w) {
        unsigned lo = w * c, limb = lo >> 5, off = lo & 31;
        uint32_t v = 0;
        if (limb < (unsigned)P::N) {
            uint64_t two = k.l[limb];
            if (limb + 1 < (unsigned)P::N) two |= (uint64_t)k.l[limb + 1] << 32;
            v = (uint32_t)(two >> off) & mask;
        }
        uint32_t d = v + carry;
        uint32_t out;
        if (d > half) {  // negative digit d - 2^c, magnitude m = 2^c - d in [0, 2^(c-1))
            uint32_t m = full - d;
            out = m ? (((m - 1) << 1) | 1u) : DIGIT_NONE;
            carry = 1;
        } else {
            out = d ? ((d - 1) << 1) : DIGIT_NONE;
            carry = 0;
        }
        digits[(size_t)w * len + i] = skip ? DIGIT_NONE : out;
    }
}

// --------------------------------------------------------------------- LDS-staged counting sort
__global__ __launch_bounds__(1024) void k_hist(const uint32_t *digits, uint32_t *counts, size_t len, unsigned chunk,
                                               unsigned nbuckets) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *h = (uint32_t *)smem_raw;
    const unsigned w = blockIdx.y;
    for (unsigned b = threadIdx.x; b < nbuckets; b += blockDim.x) h[b] = 0;
    __syncthreads();
    size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk;
    if (hi > len) hi = len;
    const uint32_t *d = digits + (size_t)w * len;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t v = d[i];
        if (v != DIGIT_NONE) atomicAdd(&h[v >> 1], 1u);
    }
    __syncthreads();
    uint32_t *cw = counts + (size_t)w * nbuckets;
    for (unsigned b = threadIdx.x; b < nbuckets; b += blockDim.x) {
        uint32_t v = h[b];
        if (v) atomicAdd(&cw[b], v);
    }
}

// Exclusive scans over all G = nwin*nbuckets buckets, three small launches (tile scan, scan of the
// tile totals, offset add):
//   bucket_off[g] = sum_{g' < g} counts[g'],   task_off[g] = sum_{g' < g} ceil(counts[g'] / seg)
// (+ totals at index G).  A lane owns SCAN_PER contiguous counters (64 B), a workgroup SCAN_TILE.
constexpr unsigned SCAN_PER = 16, SCAN_TILE = 256 * SCAN_PER;

__global__ __launch_bounds__(256) void k_scan_tiles(const uint32_t *counts, uint32_t *bucket_off, uint32_t *task_off,
                                                    uint32_t *tile_tot, size_t G, unsigned seg) {
    __shared__ uint32_t s_a[256], s_b[256];
    const unsigned tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)tid * SCAN_PER;
    uint32_t c[SCAN_PER], nt[SCAN_PER];
    uint32_t sa = 0, sb = 0;
#pragma unroll
    for (unsigned k = 0; k < SCAN_PER; ++k) {
        c[k] = base + k < G ? counts[base + k] : 0u;
        nt[k] = (c[k] + seg - 1) / seg;
        sa += c[k];
        sb += nt[k];
    }
    s_a[tid] = sa;
    s_b[tid] = sb;
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {
        uint32_t va = 0, vb = 0;
        if (tid >= off) { va = s_a[tid - off]; vb = s_b[tid - off]; }
        __syncthreads();
        s_a[tid] += va;
        s_b[tid] += vb;
        __syncthreads();
    }
    uint32_t ra = s_a[tid] - sa, rb = s_b[tid] - sb;
#pragma unroll
    for (unsigned k = 0; k < SCAN_PER; ++k) {
        if (base + k < G) { bucket_off[base + k] = ra; task_off[base + k] = rb; }
        ra += c[k];
        rb += nt[k];
    }
    if (tid == 255) { tile_tot[2 * blockIdx.x] = s_a[255]; tile_tot[2 * blockIdx.x + 1] = s_b[255]; }
}

__global__ __launch_bounds__(1024) void k_scan_totals(uint32_t *tile_tot, unsigned ntiles, uint32_t *bucket_off,
                                                      uint32_t *task_off, size_t G) {
    __shared__ uint32_t s_a[1024], s_b[1024];
    const unsigned tid = threadIdx.x;
    uint32_t carry_a = 0, carry_b = 0;
    for (unsigned base = 0; base < ntiles; base += 1024) {
        unsigned i = base + tid;
        uint32_t a = i < ntiles ? tile_tot[2 * i] : 0u, b = i < ntiles ? tile_tot[2 * i + 1] : 0u;
        s_a[tid] = a;
        s_b[tid] = b;
        __syncthreads();
        for (unsigned off = 1; off < 1024; off <<= 1) {
            uint32_t va = 0, vb = 0;
            if (tid >= off) { va = s_a[tid - off]; vb = s_b[tid - off]; }
            __syncthreads();
            s_a[tid] += va;
            s_b[tid] += vb;
            __syncthreads();
        }
        if (i < ntiles) { tile_tot[2 * i] = carry_a + s_a[tid] - a; tile_tot[2 * i + 1] = carry_b + s_b[tid] - b; }
        carry_a += s_a[1023];
        carry_b += s_b[1023];
        __syncthreads();
    }
    if (tid == 0) { bucket_off[G] = carry_a; task_off[G] = carry_b; }
}

__global__ __launch_bounds__(256) void k_scan_add(uint32_t *bucket_off, uint32_t *task_off, const uint32_t *tile_tot, size_t G) {
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_PER;
    const uint32_t oa = tile_tot[2 * blockIdx.x], ob = tile_tot[2 * blockIdx.x + 1];
#pragma unroll
    for (unsigned k = 0; k < SCAN_PER; ++k)
        if (base + k < G) { bucket_off[base + k] += oa; task_off[base + k] += ob; }
}

__global__ __launch_bounds__(1024) void k_scatter(const uint32_t *digits, const uint32_t *bucket_off, uint32_t *cursor,
                                                  uint32_t *sorted, size_t len, unsigned chunk, unsigned nbuckets) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *h = (uint32_t *)smem_raw;
    const unsigned w = blockIdx.y;
    for (unsigned b = threadIdx.x; b < nbuckets; b += blockDim.x) h[b] = 0;
    __syncthreads();
    size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk;
    if (hi > len) hi = len;
    const uint32_t *d = digits + (size_t)w * len;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t v = d[i];
        if (v != DIGIT_NONE) atomicAdd(&h[v >> 1], 1u);
    }
    __syncthreads();
    const size_t gbase = (size_t)w * nbuckets;
    for (unsigned b = threadIdx.x; b < nbuckets; b += blockDim.x) {
        uint32_t v = h[b];
        if (v) h[b] = bucket_off[gbase + b] + atomicAdd(&cursor[gbase + b], v);
    }
    __syncthreads();
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t v = d[i];
        if (v != DIGIT_NONE) {
            uint32_t pos = atomicAdd(&h[v >> 1], 1u);
            sorted[pos] = ((uint32_t)i << 1) | (v & 1u);
        }
    }
}

// ------------------------------------------------------------------------------ task order
// k_accumulate gives every lane one task; a wave runs as long as its LONGEST task.  Bucket loads are
// Poisson (mean m, sigma sqrt(m)), so in launch order the 64 lanes of a wave idle ~2.5 sigma / m of the
// time (15 % at m = 168).  order[] lists the tasks by DESCENDING length (counting sort over the length,
// LDS histogram per workgroup): the lanes of a wave then carry (almost) equal loads and the longest
// tasks start first.  bin = (seg - L) >> lshift, L in [1, seg].
constexpr unsigned TASK_MAX_BINS = 8192;

// One lane per BUCKET g (its tasks are task_off[g] .. task_off[g + 1] - 1: all but the last are full, L = seg, bin 0; the last
// one has the remainder), so a lane issues at most two LDS atomics and no search.  (Round 2 ran one lane per TASK and found
// the task's bucket by a 21-step binary search over task_off: 0.19 ms for the two launches at 2^21 buckets.)
template <bool SCATTER>
__global__ __launch_bounds__(1024) void k_task_bins(const uint32_t *counts, const uint32_t *task_off, size_t G, unsigned seg,
                                                    unsigned lshift, unsigned nbins, uint32_t *len_cnt, const uint32_t *len_off,
                                                    uint32_t *len_cursor, uint32_t *order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *h = (uint32_t *)smem_raw;
    for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) h[b] = 0;
    __syncthreads();
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t t0 = 0, n = 0, rank_full = 0, rank_last = 0;
    unsigned bin = 0;
    if (g < G) {
        t0 = task_off[g];
        n = task_off[g + 1] - t0;
        if (n) {
            const uint32_t L = counts[g] - (n - 1) * seg;           // 1 .. seg
            bin = (seg - L) >> lshift;
            if (n > 1) rank_full = atomicAdd(&h[0], n - 1);
            rank_last = atomicAdd(&h[bin], 1u);
        }
    }
    __syncthreads();
    if (!SCATTER) {
        for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) {
            const uint32_t v = h[b];
            if (v) atomicAdd(&len_cnt[b], v);
        }
    } else {
        for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) {
            const uint32_t v = h[b];
            if (v) h[b] = len_off[b] + atomicAdd(&len_cursor[b], v);
        }
        __syncthreads();
        if (n) {
            const uint32_t at = h[0] + rank_full;
            for (uint32_t k = 0; k + 1 < n; ++k) order[at + k] = t0 + k;       // (a hot bucket: thousands; the rest: none or one)
            order[h[bin] + rank_last] = t0 + n - 1;
        }
    }
}

// ---------------------------------------------------------------------------- accumulate
// One lane per task (taken in order[]); TABLE selects the point record (TablePoint of a window table, or the
// dense affine array of the per-window pipeline).  The accumulator lives in reduced-radix registers (fq28.cuh): 10 carry-free
// Montgomery products and 7 lazy add/sub per mixed add.  `bases` are in INTERNAL Montgomery form.
// The exceptional case acc == +-point (doubling / cancellation) is resolved on the dense path.
template <class C, bool TABLE>
__global__ __launch_bounds__(128) void k_accumulate(const uint32_t *sorted, const uint32_t *counts,
                                                    const uint32_t *bucket_off, const uint32_t *task_off,
                                                    const uint32_t *order, const void *points, XYZZ<C> *partials, size_t G,
                                                    unsigned seg) {
    typedef typename C::FqRR RR;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = task_off[G];
    if (t >= total) return;
    t = order[t];   // tasks by descending length: equal loads inside a wave
    // bucket of task t: largest g with task_off[g] <= t  (empty buckets have zero tasks)
    size_t lo = 0, hi = G;
    while (hi - lo > 1) {
        size_t mid = (lo + hi) >> 1;
        if (task_off[mid] <= t) lo = mid; else hi = mid;
    }
    const size_t g = lo;
    const uint32_t k = (uint32_t)t - task_off[g];
    const uint32_t start = bucket_off[g] + k * seg;
    uint32_t end = bucket_off[g] + counts[g];
    if (start + seg < end) end = start + seg;
    XYZZ28<C> acc;
    acc.X = acc.Y = acc.ZZ = acc.ZZZ = f28_zero<RR>();
#ifndef PM_ACC_IDX_VEC
#define PM_ACC_IDX_VEC 4
#endif
#if PM_ACC_IDX_VEC == 4
    // The task's indices are read four at a time (one aligned 16-byte load per four additions): a lane's 4-byte load used to pull a
    // whole line of sorted[] through L2 for one entry, and by the lane's next addition (~4 500 instructions and 3 MB of gathered
    // points per XCD later) the line was gone again.  sorted[] carries 16 bytes of padding for the last quad.
    uint4 quad = make_uint4(0u, 0u, 0u, 0u);
    uint32_t quad_at = 0xffffffffu;
#endif
    for (uint32_t e = start; e < end; ++e) {
#if PM_ACC_IDX_VEC == 4
        if ((e & ~3u) != quad_at) {
            quad_at = e & ~3u;
            quad = *(const uint4 *)(sorted + quad_at);
        }
        const unsigned q = e & 3u;
        const uint32_t v = q == 0 ? quad.x : q == 1 ? quad.y : q == 2 ? quad.z : quad.w;
#else
        const uint32_t v = sorted[e];
#endif
        const bool neg = (v & 1u) != 0;
        if (TABLE) {   // window tables: one aligned 128-byte record, already on 28-bit limbs
            const TablePoint<C> tp = ((const TablePoint<C> *)points)[v >> 1];
            F28<RR> x2, y2;
#pragma unroll
            for (int i = 0; i < RR::N; ++i) { x2.l[i] = tp.x[i]; y2.l[i] = tp.y[i]; }
            if (!xyzz28_madd_limbs<C>(acc, x2, y2, neg)) acc = xyzz28_madd_exceptional<C>(acc, table_point_to_affine<C>(tp), neg);
        } else {
            const Affine<C> p = ((const Affine<C> *)points)[v >> 1];
            if (!xyzz28_madd<C>(acc, p, neg)) acc = xyzz28_madd_exceptional<C>(acc, p, neg);
        }
    }
    partials[t] = xyzz28_store<C>(acc);   // INTERNAL form: the bucket reduction stays on reduced-radix arithmetic
}

// ------------------------------------------------------------------------- bucket reduce
// Window sum S_w = sum_b (b+1) * B_b, B_b = sum of the bucket's task partials.  Lane j of the window
// owns RED_K consecutive buckets [jK, jK+K):  sum (b+1) B_b = jK * A + sum_i (i+1) B_{jK+i}  with
// A = running sum (descending), second term = sum of the running sums; jK * A by double-and-add
// (jK < 2^c).  Everything on F28 registers; workgroup LDS tree, then k_sum_parts per window.
// RED_K = 4 keeps the dependent chain short (8 adds + <= 15 doublings) and puts 8192 lanes on a window.
constexpr unsigned RED_K = 4;

// sh[threadIdx.x] holds each lane's value on entry; on exit sh[0] holds the workgroup sum
template <class C>
__device__ __forceinline__ void lds_tree_sum(XYZZ28<C> *sh) {
    __syncthreads();
    for (unsigned off = blockDim.x >> 1; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const XYZZ28<C> b = sh[threadIdx.x + off];
            xyzz28_add_into_full<C>(&sh[threadIdx.x], b);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------- hot buckets
// Skewed scalars (the reference's own bench circuit repeats one witness value 2^20 times, benches/bench.rs:
// 49-51) put ~len entries into one bucket per window: the task split keeps k_accumulate balanced, but the
// bucket then owns thousands of task partials, and the reduction's lane would add them one after the other
// (measured: 268 ms).  k_task_counts lists the buckets with many tasks; k_task_fold sums such a bucket's
// partials in parallel into its first slot and sets its effective task count to 1.
//   tier A: more than FOLD_BLOCK_MIN tasks -> one workgroup per bucket;  tier B: more than FOLD_MIN -> one wave.
constexpr uint32_t FOLD_MIN = 8, FOLD_BLOCK_MIN = 1024;

__global__ void k_task_counts(const uint32_t *task_off, size_t G, uint32_t *task_cnt, uint32_t *hot_a, uint32_t *hot_b,
                              uint32_t *hot_counts /*[2], zeroed*/, uint32_t cap) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint32_t n = task_off[g + 1] - task_off[g];
    task_cnt[g] = n;
    if (n > FOLD_BLOCK_MIN) {
        const uint32_t i = atomicAdd(&hot_counts[0], 1u);
        if (i < cap) hot_a[i] = (uint32_t)g;   // an unlisted bucket keeps its sequential sum
    } else if (n > FOLD_MIN) {
        const uint32_t i = atomicAdd(&hot_counts[1], 1u);
        if (i < cap) hot_b[i] = (uint32_t)g;
    }
}

// LANES = 256: one workgroup per listed bucket (tier A);  LANES = 64: one wave per listed bucket (tier B)
template <class C, unsigned LANES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_task_fold(XYZZ<C> *partials, const uint32_t *task_off, uint32_t *task_cnt,
                                                   const uint32_t *hot, const uint32_t *hot_count, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XYZZ28<C> *sh = (XYZZ28<C> *)smem_raw;
    typedef typename C::FqRR RR;
    uint32_t listed = *hot_count;
    if (listed > cap) listed = cap;
    constexpr unsigned GROUPS = 256 / LANES;                       // buckets per workgroup per round
    const unsigned group = threadIdx.x / LANES, lane = threadIdx.x % LANES;
    for (uint32_t h0 = blockIdx.x * GROUPS; h0 < listed; h0 += gridDim.x * GROUPS) {   // uniform trip count per workgroup
        const uint32_t h = h0 + group;
        XYZZ28<C> *acc = &sh[threadIdx.x];
        acc->X = acc->Y = acc->ZZ = acc->ZZZ = f28_zero<RR>();
        uint32_t first = 0, n = 0;
        if (h < listed) {
            first = task_off[hot[h]];
            n = task_off[hot[h] + 1] - first;
            for (uint32_t i = lane; i < n; i += LANES) xyzz28_add_into_full<C>(acc, xyzz28_load<C>(partials[first + i]));
        }
        __syncthreads();
        for (unsigned off = LANES >> 1; off > 0; off >>= 1) {
            if (lane < off) {
                const XYZZ28<C> b = sh[threadIdx.x + off];
                xyzz28_add_into_full<C>(&sh[threadIdx.x], b);
            }
            __syncthreads();
        }
        if (h < listed && lane == 0) {
            partials[first] = xyzz28_store<C>(sh[threadIdx.x]);
            task_cnt[hot[h]] = 1;
        }
        __syncthreads();
    }
}

// Input bucket b = sum of its task partials, weight b + 1 (a Pippenger window).
template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_bucket_reduce(const XYZZ<C> *partials, const uint32_t *task_off, const uint32_t *task_cnt, unsigned nbuckets,
                                                       unsigned lanes_per_window, unsigned bpw, XYZZ<C> *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XYZZ28<C> *sh = (XYZZ28<C> *)smem_raw;
    typedef typename C::FqRR RR;
    const unsigned w = blockIdx.x / bpw, bw = blockIdx.x % bpw;
    const unsigned j = bw * blockDim.x + threadIdx.x;
    XYZZ28<C> run;
    run.X = run.Y = run.ZZ = run.ZZZ = f28_zero<RR>();
    XYZZ28<C> *acc = &sh[threadIdx.x];     // the weighted accumulator lives in LDS (register pressure)
    *acc = run;
    if (j < lanes_per_window) {
        const size_t gbase = (size_t)w * nbuckets + (size_t)j * RED_K;
        for (int i = (int)RED_K - 1; i >= 0; --i) {
            if ((size_t)j * RED_K + i >= nbuckets) continue;
            for (uint32_t q = task_off[gbase + i], qe = q + task_cnt[gbase + i]; q < qe; ++q) xyzz28_add_full<C>(run, xyzz28_load<C>(partials[q]));
            xyzz28_add_into_full<C>(acc, run);                 // weight i + 1
        }
        const uint32_t s = j * RED_K;          // acc += s * run
        if (s) {
            XYZZ28<C> m = run;
            for (int b = 30 - __clz((int)s); b >= 0; --b) {
                xyzz28_dbl<C>(m);
                if ((s >> b) & 1) xyzz28_add_full<C>(m, run);
            }
            xyzz28_add_into_full<C>(acc, m);
        }
    }
    lds_tree_sum<C>(sh);
    if (threadIdx.x == 0) out[blockIdx.x] = xyzz28_store<C>(sh[0]);
}

// out[w] = sum of parts[w * count .. + count)
template <class C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sum_parts(const XYZZ<C> *parts, unsigned count, XYZZ<C> *out) {
    __shared__ XYZZ28<C> sh[64];
    typedef typename C::FqRR RR;
    XYZZ28<C> *acc = &sh[threadIdx.x];
    acc->X = acc->Y = acc->ZZ = acc->ZZZ = f28_zero<RR>();
    for (unsigned i = threadIdx.x; i < count; i += 64) xyzz28_add_into_full<C>(acc, xyzz28_load<C>(parts[(size_t)blockIdx.x * count + i]));
    lds_tree_sum<C>(sh);
    if (threadIdx.x == 0) out[blockIdx.x] = xyzz28_store<C>(sh[0]);
}

// =====================================================================================================
// Table mode: the base vector is resident together with its window tables T_w[i] = 2^(c w) P_i
// (setup.hip: tables_build), so every window's digits share ONE set of 2^(c-1) buckets: c = 22 gives
// 12 mixed adds per pair instead of 16, no per-window reduction and no doublings.  2^21 buckets do not
// fit an LDS histogram, so the sort is a three-level MSD radix over (u16 key, u32 table index) entries:
// k_tbl_count / k_block_sums + k_block_offsets / k_tbl_partition (64 regions of 2^15 buckets, atomic-free offsets),
// k_region_pass<HIST> + k_region_pass_staged<MID> (128 sub-regions of 256 buckets), <HIST> + the bucket
// scan + k_region_pass_staged<FINAL>; every scatter is staged through LDS and written out coalesced.
// Reduction: k_reduce_level0 / k_reduce_level1 / k_sum_final (chains of dependent point additions).
// =====================================================================================================
constexpr unsigned LO_BITS = 15;

// Inclusive scan of one value per lane across the workgroup (blockDim <= 1024): wave shuffles + one LDS hop
// (2 barriers; the LDS Hillis-Steele it replaces needed 2 log2(n)).  wt: 64 words of LDS scratch.
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t *wt) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (unsigned o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    if (lane == 63) wt[wave] = v;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < (blockDim.x >> 6) ? wt[lane] : 0u;
#pragma unroll
        for (unsigned o = 1; o < 16; o <<= 1) {
            const uint32_t n = __shfl_up(w, o, 64);
            if (lane >= o) w += n;
        }
        wt[lane] = w;
    }
    __syncthreads();
    if (wave) v += wt[wave - 1];
    return v;
}

// canonical scalar (zero when the base is the point at infinity -- `inf` holds one flag byte per point of
// window 0, setup.hip: tables_build): digits are then pure bit extraction
template <class P>
__device__ __forceinline__ Fp<P> canon_scalar(const Fp<P> *scalars, const unsigned char *inf, size_t i) {
    Fp<P> k = from_mont<P>(scalars[i]);
    if (inf[i]) k = Fp<P>::zero();
    return k;
}

// signed digit w of canonical scalar k: returns false for a zero digit; bucket = |d| - 1
// Balanced window layout as a function of the window count (the same formula as setup.hip: tables_layout).
// The kernels are instantiated per NWIN so that every bit offset, shift and limb index is an immediate:
// with a run-time layout the compiler parks the scalar in LDS and fetches offsets with vector loads.
__host__ __device__ constexpr unsigned win_width(unsigned nwin, unsigned w) { return 256 / nwin + (w < 256 % nwin ? 1u : 0u); }
__host__ __device__ constexpr unsigned win_off(unsigned nwin, unsigned w) { return w * (256 / nwin) + (w < 256 % nwin ? w : 256 % nwin); }
// Wide mode (every window has its own bucket set): first bucket of window w's set.  The 256 % nwin windows that are one bit wider come
// first (win_width) and own `wide_b` buckets each, the others `narrow_b` (round 6: half as many, where that is still whole sort
// regions; until then every set had wide_b buckets: 12 windows of 22 / 21 bits held 12 x 2^21 = 25.2 M buckets instead of 16.8 M).
// Table mode passes 0 for both: one shared set.
__host__ __device__ constexpr uint32_t win_base(unsigned nwin, unsigned w, uint32_t wide_b, uint32_t narrow_b) {
    return w <= 256 % nwin ? w * wide_b : (256 % nwin) * wide_b + (w - 256 % nwin) * narrow_b;
}

template <class P>
__device__ __forceinline__ bool digit_at(const Fp<P> &k, unsigned lo, unsigned c, uint32_t &carry, uint32_t &bucket, uint32_t &neg) {
    const uint32_t half = 1u << (c - 1), full = 1u << c, mask = full - 1;
    const unsigned limb = lo >> 5, off = lo & 31;
    uint32_t v = 0;
    if (limb < (unsigned)P::N) {
        uint64_t two = k.l[limb];
        if (limb + 1 < (unsigned)P::N) two |= (uint64_t)k.l[limb + 1] << 32;
        v = (uint32_t)(two >> off) & mask;
    }
    // branch-free: the callers unroll this over compile-time (lo, c), lane divergence here would cost
    // an exec-mask region per window
    const uint32_t d = v + carry;
    const bool over = d > half;
    const uint32_t m = over ? full - d : d;
    carry = over ? 1u : 0u;
    neg = carry;
    bucket = m - 1;
    return m != 0;
}

// Region populations PER WORKGROUP: block_cnt[workgroup][region], with the same workgroup -> scalar mapping
// as k_tbl_partition (one scalar per lane).  A column-wise scan (k_block_sums, k_block_offsets) turns them into each
// workgroup's offset inside each region, so neither kernel touches a global atomic: 41 K workgroups x 64
// regions hammering 64 addresses serialised in L2 and cost more than the rest of the kernel.
template <class P, unsigned NWIN>
__global__ __launch_bounds__(1024) void k_tbl_count(const Fp<P> *scalars, const unsigned char *inf, size_t len, unsigned regions,
                                                   uint32_t *block_cnt, uint32_t win_buckets, uint32_t narrow_buckets) {
    __shared__ uint32_t cnt[1024];
    for (unsigned r = threadIdx.x; r < regions; r += blockDim.x) cnt[r] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) {
        const Fp<P> k = canon_scalar<P>(scalars, inf, i);
        uint32_t carry = 0, b, neg;
#pragma unroll
        for (unsigned w = 0; w < NWIN; ++w)
            if (digit_at<P>(k, win_off(NWIN, w), win_width(NWIN, w), carry, b, neg)) atomicAdd(&cnt[(b + win_base(NWIN, w, win_buckets, narrow_buckets)) >> LO_BITS], 1u);
    }
    __syncthreads();
    for (unsigned r = threadIdx.x; r < regions; r += blockDim.x) block_cnt[(size_t)blockIdx.x * regions + r] = cnt[r];
}

// block_cnt[nblocks][regions] -> per region the exclusive prefix over the workgroups (in place), totals -> region_count.
// Two launches of G workgroups.  Workgroup g owns the block rows [g span, (g + 1) span); lane (j, r) = (t / regions, t % regions),
// j < rows <= 16, owns `per` consecutive rows of region r, so a wave reads 64 consecutive regions of one row: whole cache lines.
// (Round 2's form -- one workgroup per REGION, every lane striding through the rows of its column -- kept 64 CUs busy with
// uncoalesced 4-byte loads: 0.16 ms at 41 K rows x 64 regions, profiles/r03_final_rocprofv3_kernel_stats_bench_2p20.csv.)
struct BlockScanShape { unsigned G, span, rows, per; };
static BlockScanShape block_scan_shape(unsigned nblocks, unsigned regions) {
    BlockScanShape h;
    h.rows = 1024 / regions < 16 ? 1024 / regions : 16;
    if (h.rows < 1) h.rows = 1;
    h.G = (nblocks + 15) / 16 < 256 ? (nblocks + 15) / 16 : 256;
    if (h.G < 1) h.G = 1;
    h.span = (nblocks + h.G - 1) / h.G;
    h.per = (h.span + h.rows - 1) / h.rows;
    return h;
}
__device__ __forceinline__ void block_scan_range(unsigned g, unsigned j, BlockScanShape h, unsigned nblocks, unsigned &lo, unsigned &hi) {
    const unsigned end = (g + 1) * h.span < nblocks ? (g + 1) * h.span : nblocks;
    lo = g * h.span + j * h.per;
    hi = lo + h.per < end ? lo + h.per : end;
}
__global__ __launch_bounds__(1024) void k_block_sums(const uint32_t *block_cnt, unsigned nblocks, unsigned regions, BlockScanShape h, uint32_t *partial) {
    __shared__ uint32_t s[1024];
    const unsigned t = threadIdx.x, j = t / regions, r = t % regions;
    if (j < h.rows) {
        unsigned lo, hi;
        block_scan_range(blockIdx.x, j, h, nblocks, lo, hi);
        uint32_t sum = 0;
        for (unsigned b = lo; b < hi; ++b) sum += block_cnt[(size_t)b * regions + r];
        s[t] = sum;
    }
    __syncthreads();
    if (t < regions) {
        uint32_t tot = 0;
        for (unsigned jj = 0; jj < h.rows; ++jj) tot += s[jj * regions + t];
        partial[(size_t)blockIdx.x * regions + t] = tot;
    }
}
__global__ __launch_bounds__(1024) void k_block_offsets(uint32_t *block_cnt, unsigned nblocks, unsigned regions, BlockScanShape h, const uint32_t *partial,
                                                        uint32_t *region_count) {
    __shared__ uint32_t s[1024], base[1024];
    const unsigned t = threadIdx.x, j = t / regions, r = t % regions, g = blockIdx.x;
    if (j < h.rows) {                                   // the earlier workgroups' totals, rows-way split
        uint32_t acc = 0;
        for (unsigned q = j; q < g; q += h.rows) acc += partial[(size_t)q * regions + r];
        s[t] = acc;
    }
    __syncthreads();
    if (t < regions) {
        uint32_t tot = 0;
        for (unsigned jj = 0; jj < h.rows; ++jj) tot += s[jj * regions + t];
        base[t] = tot;
    }
    __syncthreads();
    unsigned lo = 0, hi = 0;
    if (j < h.rows) {
        block_scan_range(g, j, h, nblocks, lo, hi);
        uint32_t sum = 0;
        for (unsigned b = lo; b < hi; ++b) sum += block_cnt[(size_t)b * regions + r];
        s[t] = sum;
    }
    __syncthreads();
    if (j < h.rows) {
        uint32_t run = base[r];
        for (unsigned jj = 0; jj < j; ++jj) run += s[jj * regions + r];
        for (unsigned b = lo; b < hi; ++b) {
            const uint32_t v = block_cnt[(size_t)b * regions + r];
            block_cnt[(size_t)b * regions + r] = run;
            run += v;
        }
    }
    if (g == gridDim.x - 1 && t < regions) {
        uint32_t tot = base[t];
        for (unsigned jj = 0; jj < h.rows; ++jj) tot += s[jj * regions + t];
        region_count[t] = tot;
    }
}

// region_off = exclusive scan of region_count (regions <= 1024); also clears the claim cursors
__global__ __launch_bounds__(1024) void k_region_offsets(const uint32_t *region_count, uint32_t *region_off, uint32_t *region_cursor,
                                                         unsigned regions) {
    __shared__ uint32_t s[1024];
    const unsigned t = threadIdx.x;
    uint32_t v = t < regions ? region_count[t] : 0u;
    s[t] = v;
    __syncthreads();
    for (unsigned off = 1; off < 1024; off <<= 1) {
        uint32_t a = t >= off ? s[t - off] : 0u;
        __syncthreads();
        s[t] += a;
        __syncthreads();
    }
    if (t < regions) { region_off[t] = s[t] - v; region_cursor[t] = 0; }
    if (t == 1023) region_off[regions] = s[1023];
}

// entry (i, w) -> region of its bucket: key = low LO_BITS of the bucket, val = table index << 1 | negate.
// One scalar per lane.  The workgroup's entries are staged through LDS in region order so that the global
// stores are coalesced (consecutive lanes -> consecutive addresses of a region's run).
//   LDS: cnt[1024] | delta[1024] | staged vals (u32 x blockDim nwin) | staged region<<16|key (u32 x same)
template <class P, unsigned NWIN>
__global__ __launch_bounds__(1024) void k_tbl_partition(const Fp<P> *scalars, const unsigned char *inf, size_t len,
                                                       unsigned regions, const uint32_t *region_off, const uint32_t *block_off, size_t tbl_stride, size_t base_index, uint16_t *keys,
                                                       uint32_t *vals, uint32_t win_buckets, uint32_t narrow_buckets) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *cnt = (uint32_t *)smem_raw, *delta = cnt + 1024;
    uint32_t *st_val = delta + 1024, *st_key = st_val + (size_t)blockDim.x * NWIN;
    __shared__ uint32_t tot, wt[64];
    const unsigned t = threadIdx.x, BD = blockDim.x;
    for (unsigned r = t; r < regions; r += BD) cnt[r] = 0;
    __syncthreads();
    // pass 1: count per region -- the counting atomic returns the entry's rank inside this workgroup, kept in
    // registers (NWIN per lane) so that pass 2 needs no second LDS atomic
    const size_t i = (size_t)blockIdx.x * BD + t;
    uint32_t rank[NWIN];
    Fp<P> k = Fp<P>::zero();
    if (i < len) {
        k = canon_scalar<P>(scalars, inf, i);
        uint32_t carry = 0, b, neg;
#pragma unroll
        for (unsigned w = 0; w < NWIN; ++w)
            if (digit_at<P>(k, win_off(NWIN, w), win_width(NWIN, w), carry, b, neg)) rank[w] = atomicAdd(&cnt[(b + win_base(NWIN, w, win_buckets, narrow_buckets)) >> LO_BITS], 1u);
    }
    __syncthreads();
    {   // exclusive scan over the regions: lane t owns regions [t rpl, (t + 1) rpl), rpl = 1 or 2 (regions <= 2 blockDim, checked by
        // the host: the 768 regions of a 12-window wide plan on 512 lanes)
        const unsigned rpl = (regions + BD - 1) / BD;
        uint32_t c[2] = {0u, 0u};
        for (unsigned j = 0; j < rpl; ++j) {
            const unsigned r = t * rpl + j;
            c[j] = r < regions ? cnt[r] : 0u;
        }
        const uint32_t incl = block_inclusive_scan(c[0] + c[1], wt);
        uint32_t ex = incl - (c[0] + c[1]);
        for (unsigned j = 0; j < rpl; ++j) {
            const unsigned r = t * rpl + j;
            if (r < regions) {
                cnt[r] = ex;                                                                  // first staged slot of the region
                delta[r] = region_off[r] + block_off[(size_t)blockIdx.x * regions + r] - ex;  // global = delta[region] + slot
                ex += c[j];
            }
        }
        if (t == BD - 1) tot = incl;
    }
    __syncthreads();
    if (i < len) {
        uint32_t carry = 0, b, neg;
#pragma unroll
        for (unsigned w = 0; w < NWIN; ++w)
            if (digit_at<P>(k, win_off(NWIN, w), win_width(NWIN, w), carry, b, neg)) {
                b += win_base(NWIN, w, win_buckets, narrow_buckets);
                const uint32_t rg = b >> LO_BITS, slot = cnt[rg] + rank[w];
                st_key[slot] = (rg << 16) | (b & ((1u << LO_BITS) - 1));
                st_val[slot] = (uint32_t)(((size_t)w * tbl_stride + base_index + i) << 1) | neg;
            }
    }
    __syncthreads();
    const uint32_t total = tot;
    for (uint32_t s = t; s < total; s += BD) {
        const uint32_t kv = st_key[s], pos = delta[kv >> 16] + s;
        keys[pos] = (uint16_t)kv;
        vals[pos] = st_val[s];
    }
}

// Levels 2 and 3: one generic LDS-staged pass over a segment-partitioned (key, val) array.  Workgroup
// `blockIdx.x` owns entries [x CH, (x+1) CH) and handles each segment piece inside it: bin = key >> bin_shift
// (nbins per segment, global bin id = segment * nbins + bin).
//   RS_HIST : counts[global bin] += occurrences
//   RS_MID  : entries move to (keys_out, vals_out) at out_off[global bin] + rank, key keeps its low bin_shift bits
//   RS_FINAL: sorted[out_off[global bin] + rank] = val
// With 64 regions -> 128 sub-regions -> 256 buckets every pass writes runs of ~100+ entries per bin instead
// of single scattered words (the two-level version spent most of its time on 4-byte scattered stores).
enum { RS_HIST = 0, RS_MID = 1, RS_FINAL = 2 };
constexpr unsigned ST_MAX_BINS_HIST = 256;   // k_hist_small: bins per segment
constexpr unsigned RS_CHUNK_LOG = 15, RS_PER_LANE = (1u << RS_CHUNK_LOG) / 1024;   // entries per lane of a 1024-lane workgroup

template <int MODE>
__global__ __launch_bounds__(1024) void k_region_pass(const uint16_t *keys, const uint32_t *vals, const uint32_t *seg_off,
                                                      unsigned nseg, unsigned bin_shift, unsigned nbins, unsigned chunk,
                                                      uint32_t *counts, const uint32_t *out_off, uint32_t *cursor,
                                                      uint32_t *sorted, uint16_t *keys_out, uint32_t *vals_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *h = (uint32_t *)smem_raw;
    const uint32_t total = seg_off[nseg];
    const uint32_t lo = blockIdx.x * chunk;
    if (lo >= total) return;
    uint32_t hi = lo + chunk;
    if (hi > total) hi = total;
    unsigned ra = 0, rb = nseg;   // segment containing `lo`: largest r with seg_off[r] <= lo
    while (rb - ra > 1) {
        unsigned mid = (ra + rb) >> 1;
        if (seg_off[mid] <= lo) ra = mid; else rb = mid;
    }
    const uint16_t low_mask = (uint16_t)((1u << bin_shift) - 1);
    for (unsigned r = ra; r < nseg; ++r) {
        const uint32_t s0 = seg_off[r] > lo ? seg_off[r] : lo;
        const uint32_t s1 = seg_off[r + 1] < hi ? seg_off[r + 1] : hi;
        if (s0 >= hi) break;
        if (s0 >= s1) continue;
        for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) h[b] = 0;
        __syncthreads();
        const size_t gbase = (size_t)r * nbins;
        if (MODE == RS_HIST) {
            for (uint32_t e = s0 + threadIdx.x; e < s1; e += blockDim.x) atomicAdd(&h[keys[e] >> bin_shift], 1u);
            __syncthreads();
            for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) {
                uint32_t v = h[b];
                if (v) atomicAdd(&counts[gbase + b], v);
            }
        } else {
            // the counting atomic already returns the entry's rank inside this workgroup: keep it in a
            // register (<= RS_PER_LANE entries per lane) instead of a second LDS atomic per entry
            uint32_t rank[RS_PER_LANE];
            uint16_t key[RS_PER_LANE];
#pragma unroll
            for (unsigned q = 0; q < RS_PER_LANE; ++q) {
                const uint32_t e = s0 + q * blockDim.x + threadIdx.x;
                if (e < s1) {
                    key[q] = keys[e];
                    rank[q] = atomicAdd(&h[key[q] >> bin_shift], 1u);
                }
            }
            __syncthreads();
            for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) {
                uint32_t v = h[b];
                if (v) h[b] = out_off[gbase + b] + atomicAdd(&cursor[gbase + b], v);
            }
            __syncthreads();
#pragma unroll
            for (unsigned q = 0; q < RS_PER_LANE; ++q) {
                const uint32_t e = s0 + q * blockDim.x + threadIdx.x;
                if (e < s1) {
                    const uint32_t pos = h[key[q] >> bin_shift] + rank[q];
                    if (MODE == RS_FINAL) {
                        sorted[pos] = vals[e];
                    } else {
                        keys_out[pos] = key[q] & low_mask;
                        vals_out[pos] = vals[e];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// Histogram for SMALL bin counts (<= 256): counts[segment * nbins + (key >> bin_shift)] += 1.  Every wave counts
// into its own LDS copy: with one shared copy the 1024 lanes of a workgroup pile onto 128-256 addresses and the
// LDS serialises them (0.30 ms per pass over 252 M keys; the data is only 0.5 GB).
__global__ __launch_bounds__(1024) void k_hist_small(const uint16_t *keys, const uint32_t *seg_off, unsigned nseg, unsigned bin_shift,
                                                     unsigned nbins, unsigned chunk, uint32_t *counts) {
    __shared__ uint32_t h[16][ST_MAX_BINS_HIST];
    const uint32_t total = seg_off[nseg];
    const uint32_t lo = blockIdx.x * chunk;
    if (lo >= total) return;
    uint32_t hi = lo + chunk;
    if (hi > total) hi = total;
    unsigned ra = 0, rb = nseg;
    while (rb - ra > 1) {
        unsigned mid = (ra + rb) >> 1;
        if (seg_off[mid] <= lo) ra = mid; else rb = mid;
    }
    const unsigned wave = threadIdx.x >> 6;
    for (unsigned r = ra; r < nseg; ++r) {
        const uint32_t s0 = seg_off[r] > lo ? seg_off[r] : lo;
        const uint32_t s1 = seg_off[r + 1] < hi ? seg_off[r + 1] : hi;
        if (s0 >= hi) break;
        if (s0 >= s1) continue;
        for (unsigned b = threadIdx.x; b < 16 * ST_MAX_BINS_HIST; b += blockDim.x) (&h[0][0])[b] = 0;
        __syncthreads();
        // 8 keys per 16-byte load over the aligned body of the piece; scalar head and tail
        const uint32_t b0 = (s0 + 7u) & ~7u, b1 = s1 & ~7u;
        if (b0 < b1) {
            for (uint32_t e = s0 + threadIdx.x; e < b0; e += blockDim.x) atomicAdd(&h[wave][keys[e] >> bin_shift], 1u);
            for (uint32_t e = b0 + threadIdx.x * 8u; e < b1; e += blockDim.x * 8u) {
                const uint4 v = *(const uint4 *)(keys + e);
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    atomicAdd(&h[wave][(w4[q] & 0xffffu) >> bin_shift], 1u);
                    atomicAdd(&h[wave][(w4[q] >> 16) >> bin_shift], 1u);
                }
            }
            for (uint32_t e = b1 + threadIdx.x; e < s1; e += blockDim.x) atomicAdd(&h[wave][keys[e] >> bin_shift], 1u);
        } else {
            for (uint32_t e = s0 + threadIdx.x; e < s1; e += blockDim.x) atomicAdd(&h[wave][keys[e] >> bin_shift], 1u);
        }
        __syncthreads();
        for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) {
            uint32_t v = 0;
#pragma unroll
            for (unsigned w = 0; w < 16; ++w) v += h[w][b];
            if (v) atomicAdd(&counts[(size_t)r * nbins + b], v);
        }
        __syncthreads();
    }
}

// The same pass for SMALL bin counts (<= 256), with the chunk staged through LDS in bin order so that
// the global stores are coalesced: consecutive lanes write consecutive addresses of a bin's run
// (the direct version above issues 64 scattered 4-byte stores per wave-instruction).
//   LDS: h[nbins] | start[nbins] | delta[nbins] | staged vals (u32 x ST_CHUNK) | staged keys (u16 x ST_CHUNK)
#ifndef PM_ST_THREADS
#define PM_ST_THREADS 1024
#endif
#ifndef PM_ST_PER_LANE
#define PM_ST_PER_LANE 8
#endif
constexpr unsigned ST_THREADS = PM_ST_THREADS, ST_PER_LANE = PM_ST_PER_LANE, ST_CHUNK = ST_PER_LANE * ST_THREADS, ST_MAX_BINS = 256;

template <int MODE>
__global__ __launch_bounds__(ST_THREADS) void k_region_pass_staged(const uint16_t *keys, const uint32_t *vals, const uint32_t *seg_off,
                                                             unsigned nseg, unsigned bin_shift, unsigned nbins,
                                                             const uint32_t *out_off, uint32_t *cursor, uint32_t *sorted,
                                                             uint16_t *keys_out, uint32_t *vals_out) {
    __shared__ uint32_t h[ST_MAX_BINS], delta[ST_MAX_BINS], wt[64];
    __shared__ uint32_t st_val[ST_CHUNK];
    __shared__ uint16_t st_key[ST_CHUNK];
    const uint32_t total = seg_off[nseg];
    const uint32_t lo = blockIdx.x * ST_CHUNK;
    if (lo >= total) return;
    uint32_t hi = lo + ST_CHUNK;
    if (hi > total) hi = total;
    unsigned ra = 0, rb = nseg;
    while (rb - ra > 1) {
        unsigned mid = (ra + rb) >> 1;
        if (seg_off[mid] <= lo) ra = mid; else rb = mid;
    }
    const uint16_t low_mask = (uint16_t)((1u << bin_shift) - 1);
    const unsigned t = threadIdx.x;
    for (unsigned r = ra; r < nseg; ++r) {
        const uint32_t s0 = seg_off[r] > lo ? seg_off[r] : lo;
        const uint32_t s1 = seg_off[r + 1] < hi ? seg_off[r + 1] : hi;
        if (s0 >= hi) break;
        if (s0 >= s1) continue;
        const uint32_t cnt = s1 - s0;
        if (t < nbins) h[t] = 0;
        __syncthreads();
        // Round 5: a lane takes EIGHT CONSECUTIVE entries with three 16-byte loads (8 keys, 2 x 4 values) instead of eight strided
        // 2-byte + eight 4-byte ones.  The passes are bound by the number of vector-memory instructions, not by bytes (13.4 M wave
        // instructions for 2.4 GB per pass at ~3.3 TB/s; the address coalescer takes a wave's 64 lanes in 16 clocks whatever their
        // width): 16 -> 3 load instructions per lane.  The piece [s0, s1) is widened down to a multiple of 8 entries so that the loads are
        // aligned (the buffers are; a piece starts inside the chunk, so the workgroup's lanes still cover it); `vm` masks the entries outside it.
        uint32_t rank[ST_PER_LANE], val[ST_PER_LANE];
        uint16_t key[ST_PER_LANE];
        static_assert(ST_PER_LANE % 8 == 0, "whole uint4s of keys per lane");
        const uint32_t e0 = (s0 & ~(ST_PER_LANE - 1u)) + ST_PER_LANE * t;
        uint32_t vm = 0;
        if (e0 < s1 && e0 + ST_PER_LANE > s0) {
            uint32_t kw[ST_PER_LANE / 2], vw[ST_PER_LANE];
#pragma unroll
            for (unsigned g = 0; g < ST_PER_LANE / 8; ++g) {
                const uint4 kv = *(const uint4 *)(keys + e0 + 8 * g);
                kw[4 * g] = kv.x; kw[4 * g + 1] = kv.y; kw[4 * g + 2] = kv.z; kw[4 * g + 3] = kv.w;
            }
#pragma unroll
            for (unsigned g = 0; g < ST_PER_LANE / 4; ++g) {
                const uint4 va = *(const uint4 *)(vals + e0 + 4 * g);
                vw[4 * g] = va.x; vw[4 * g + 1] = va.y; vw[4 * g + 2] = va.z; vw[4 * g + 3] = va.w;
            }
#pragma unroll
            for (unsigned q = 0; q < ST_PER_LANE; ++q) {
                const uint32_t e = e0 + q;
                key[q] = (uint16_t)(kw[q >> 1] >> (16 * (q & 1)));
                val[q] = vw[q];
                if (e >= s0 && e < s1) {
                    vm |= 1u << q;
                    rank[q] = atomicAdd(&h[key[q] >> bin_shift], 1u);
                }
            }
        }
        __syncthreads();
        {   // exclusive scan of h over the bins (lane b owns bin b); claim the global runs
            const uint32_t c = t < nbins ? h[t] : 0u;
            const uint32_t incl = block_inclusive_scan(c, wt);
            if (t < nbins) {
                const uint32_t ex = incl - c;
                const size_t g = (size_t)r * nbins + t;
                const uint32_t gpos = c ? out_off[g] + atomicAdd(&cursor[g], c) : 0u;
                delta[t] = gpos - ex;                          // global position = delta[bin] + staged slot
                h[t] = ex;                                     // h now = local start of the bin
            }
        }
        __syncthreads();
#pragma unroll
        for (unsigned q = 0; q < ST_PER_LANE; ++q) {
            if ((vm >> q) & 1u) {
                const uint32_t slot = h[key[q] >> bin_shift] + rank[q];
                st_key[slot] = key[q];
                st_val[slot] = val[q];
            }
        }
        __syncthreads();
        // (Pair stores -- every bin's staged run at its destination's parity inside an even-sized area, two keys in one 4-byte and two
        // values in one 8-byte store, the <= 2 singles per bin in a second sweep -- were built and measured: parity-green, 2.43 -> 2.48 ms
        // for the sort of a 2^24-pair MSM.  Stores do not wait for anything; the loads did.  profiles/r05_sort_pair_stores_negative.txt)
        for (uint32_t i = t; i < cnt; i += ST_THREADS) {
            const uint16_t k = st_key[i];
            const uint32_t pos = delta[k >> bin_shift] + i;
            if (MODE == RS_FINAL) {
                sorted[pos] = st_val[i];
            } else {
                keys_out[pos] = k & low_mask;
                vals_out[pos] = st_val[i];
            }
        }
        __syncthreads();
    }
}

// exclusive scan of n <= 2^20 counters by one workgroup (sub-region offsets); off[n] = total; clears `zero`
__global__ __launch_bounds__(1024) void k_scan_small(const uint32_t *cnt, uint32_t *off, uint32_t *zero, unsigned n) {
    __shared__ uint32_t s[1024];
    __shared__ uint32_t carry;
    const unsigned t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (unsigned base = 0; base < n; base += 1024) {
        const unsigned i = base + t;
        const uint32_t v = i < n ? cnt[i] : 0u;
        s[t] = v;
        __syncthreads();
        for (unsigned o = 1; o < 1024; o <<= 1) {
            uint32_t a = t >= o ? s[t - o] : 0u;
            __syncthreads();
            s[t] += a;
            __syncthreads();
        }
        if (i < n) { off[i] = carry + s[t] - v; zero[i] = 0; }
        __syncthreads();
        if (t == 1023) carry += s[1023];
        __syncthreads();
    }
    if (t == 0) off[n] = carry;
}

// Final combine on the host: S = sum_w 2^(c w) S_w by Horner (c doublings per window) and one inversion
// to affine -- an O(W c) dependent chain on W points (9 ms on one GPU lane, ~0.3 ms here).
template <class C>
static void host_finish(const XYZZ<C> *S /*[nwin], internal form*/, unsigned nwin, unsigned c, Affine<C> *out, int *inf) {
    XYZZ<C> acc = XYZZ<C>::identity();
    for (int w = (int)nwin - 1; w >= 0; --w) {
        for (unsigned k = 0; k < c; ++k) acc = xyzz_dbl<C>(acc);
        acc = xyzz_add<C>(acc, xyzz_internal_to_std<C>(S[w]));
    }
    *inf = acc.is_identity() ? 1 : 0;
    *out = xyzz_to_affine<C>(acc);
}

// effective task counts for the reduction + parallel fold of hot buckets (k_task_counts / k_task_fold);
// enqueued after k_accumulate, before the bucket reduction
template <class C>
static int fold_hot_buckets(pm_ctx *ctx, MsmSet &S, size_t G, size_t max_tasks) {
    MsmWorkspace &ws = ctx->msm;
    size_t cap = max_tasks > G ? max_tasks - G + 1 : 1;          // a listed bucket has > FOLD_MIN tasks
    cap = cap / FOLD_MIN + 1;
    PM_HIP(ctx, S.task_cnt.reserve(G * 4));
    PM_HIP(ctx, ws.hot.reserve((2 * cap + 2) * 4));
    uint32_t *hot_counts = ws.hot.as<uint32_t>(), *hot_a = hot_counts + 2, *hot_b = hot_a + cap;
    PM_HIP(ctx, hipMemsetAsync(hot_counts, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_task_counts, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, ctx->stream, S.task_off.as<uint32_t>(), G,
                       S.task_cnt.as<uint32_t>(), hot_a, hot_b, hot_counts, (uint32_t)cap);
    PM_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL((k_task_fold<C, 256>), dim3(256), dim3(256), 256 * sizeof(XYZZ28<C>), ctx->stream, S.partials.as<XYZZ<C>>(),
                       S.task_off.as<uint32_t>(), S.task_cnt.as<uint32_t>(), hot_a, hot_counts, (uint32_t)cap);
    PM_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL((k_task_fold<C, 64>), dim3(1024), dim3(256), 256 * sizeof(XYZZ28<C>), ctx->stream, S.partials.as<XYZZ<C>>(),
                       S.task_off.as<uint32_t>(), S.task_cnt.as<uint32_t>(), hot_b, hot_counts + 1, (uint32_t)cap);
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
}

// order[] for k_accumulate (see k_task_bins); enqueued on the context's stream after the bucket scan.
static int task_order(pm_ctx *ctx, MsmSet &S, const uint32_t *counts, size_t G, size_t seg, size_t max_tasks) {
    MsmWorkspace &ws = ctx->msm;
    unsigned lshift = 0;
    while (((seg - 1) >> lshift) + 1 > TASK_MAX_BINS) ++lshift;
    const unsigned nbins = (unsigned)(((seg - 1) >> lshift) + 1);
    PM_HIP(ctx, S.order.reserve(max_tasks * 4));
    PM_HIP(ctx, ws.len_bins.reserve((3 * (size_t)nbins + 4) * 4));
    uint32_t *len_cnt = ws.len_bins.as<uint32_t>(), *len_off = len_cnt + nbins, *len_cursor = len_off + nbins + 1;
    PM_HIP(ctx, hipMemsetAsync(len_cnt, 0, (size_t)nbins * 4, ctx->stream));
    const unsigned blocks = (unsigned)((G + 1023) / 1024);      // one lane per bucket
    hipLaunchKernelGGL(k_task_bins<false>, dim3(blocks), dim3(1024), nbins * 4, ctx->stream, counts, S.task_off.as<uint32_t>(), G,
                       (unsigned)seg, lshift, nbins, len_cnt, (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
    PM_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, ctx->stream, len_cnt, len_off, len_cursor, nbins);
    PM_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_task_bins<true>, dim3(blocks), dim3(1024), nbins * 4, ctx->stream, counts, S.task_off.as<uint32_t>(), G,
                       (unsigned)seg, lshift, nbins, (uint32_t *)nullptr, len_off, len_cursor, S.order.as<uint32_t>());
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
}

// ------------------------------------------------------------------------------- driver
// One bucket pipeline over at most msm_max_piece() pairs (internal.h; sorted-entry positions are u32: W * len < 2^32).

template <class C>
static int msm_piece(pm_ctx *ctx, const Affine<C> *d_bases, const Fp<typename C::FrP> *d_scalars, size_t len, Affine<C> *h_out,
                     int *h_inf) {
    typedef typename C::FrP FrP;
    StageTimer t_total(ctx, T_MSM_TOTAL);
    MsmPlan p = make_plan(len, (unsigned)FrP::BITS, (unsigned)ctx->opt.v[PM_OPT_MSM_TASK_LEN]);
    MsmWorkspace &ws = ctx->msm;
    MsmSet &S = ws.set;
    const size_t G = (size_t)p.nwin * p.nbuckets;
    PM_HIP(ctx, ws.digits.reserve((size_t)p.nwin * len * 4));
    PM_HIP(ctx, S.sorted.reserve((size_t)p.nwin * len * 4 + 16));   // + 16: k_accumulate reads aligned quads of indices
    PM_HIP(ctx, S.counts.reserve(2 * G * 4));  // counts | cursor, one memset
    PM_HIP(ctx, S.bucket_off.reserve((G + 1) * 4));
    PM_HIP(ctx, S.task_off.reserve((G + 1) * 4));
    PM_HIP(ctx, ws.cursor.reserve(((G + SCAN_TILE - 1) / SCAN_TILE + 1) * 8));  // scan tile totals
    PM_HIP(ctx, S.partials.reserve(p.max_tasks * sizeof(XYZZ<C>)));
    const unsigned red_lanes = (p.nbuckets + RED_K - 1) / RED_K;      // lanes per window
    unsigned red_block = 64;
    while (red_block < red_lanes && red_block < 256) red_block <<= 1;
    const unsigned bpw = (red_lanes + red_block - 1) / red_block;
    PM_HIP(ctx, ws.wsum.reserve(((size_t)p.nwin * bpw + p.nwin) * sizeof(XYZZ<C>)));
    uint32_t *counts = S.counts.as<uint32_t>(), *cursor = counts + G;
    {
        StageTimer t(ctx, T_MSM_SORT);
        PM_HIP(ctx, hipMemsetAsync(counts, 0, 2 * G * 4, ctx->stream));
        hipLaunchKernelGGL(k_digits<C>, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream, d_scalars, d_bases,
                           ws.digits.as<uint32_t>(), len, p.c, p.nwin);
        PM_HIP(ctx, hipGetLastError());
        size_t lds = (size_t)p.nbuckets * 4;
        if (lds > 48 * 1024) {  // CDNA4: up to 160 KiB of LDS per workgroup, opt in above the default cap
            PM_HIP(ctx, hipFuncSetAttribute((const void *)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            PM_HIP(ctx, hipFuncSetAttribute((const void *)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        hipLaunchKernelGGL(k_hist, dim3(p.nchunks, p.nwin), dim3(1024), lds, ctx->stream, ws.digits.as<uint32_t>(), counts,
                           len, p.chunk, p.nbuckets);
        PM_HIP(ctx, hipGetLastError());
        const unsigned ntiles = (unsigned)((G + SCAN_TILE - 1) / SCAN_TILE);
        hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(256), 0, ctx->stream, counts, S.bucket_off.as<uint32_t>(),
                           S.task_off.as<uint32_t>(), ws.cursor.as<uint32_t>(), G, p.seg);
        PM_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, ctx->stream, ws.cursor.as<uint32_t>(), ntiles,
                           S.bucket_off.as<uint32_t>(), S.task_off.as<uint32_t>(), G);
        PM_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(256), 0, ctx->stream, S.bucket_off.as<uint32_t>(),
                           S.task_off.as<uint32_t>(), ws.cursor.as<uint32_t>(), G);
        PM_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_scatter, dim3(p.nchunks, p.nwin), dim3(1024), lds, ctx->stream, ws.digits.as<uint32_t>(),
                           S.bucket_off.as<uint32_t>(), cursor, S.sorted.as<uint32_t>(), len, p.chunk, p.nbuckets);
        PM_HIP(ctx, hipGetLastError());
        PM_TRY(task_order(ctx, S, counts, G, p.seg, p.max_tasks));
    }
    {
        StageTimer t(ctx, T_MSM_ACCUMULATE);
        size_t blocks = (p.max_tasks + 127) / 128;
        hipLaunchKernelGGL((k_accumulate<C, false>), dim3((unsigned)blocks), dim3(128), 0, ctx->stream, S.sorted.as<uint32_t>(),
                           counts, S.bucket_off.as<uint32_t>(), S.task_off.as<uint32_t>(), S.order.as<uint32_t>(), (const void *)d_bases,
                           S.partials.as<XYZZ<C>>(), G, p.seg);
        PM_HIP(ctx, hipGetLastError());
    }
    std::vector<XYZZ<C>> hS(p.nwin);
    {
        StageTimer t(ctx, T_MSM_REDUCE);
        PM_TRY(fold_hot_buckets<C>(ctx, S, G, p.max_tasks));
        XYZZ<C> *parts = ws.wsum.as<XYZZ<C>>(), *dS = parts + (size_t)p.nwin * bpw;
        hipLaunchKernelGGL(k_bucket_reduce<C>, dim3(p.nwin * bpw), dim3(red_block), red_block * sizeof(XYZZ28<C>), ctx->stream,
                           S.partials.as<XYZZ<C>>(), S.task_off.as<uint32_t>(), S.task_cnt.as<uint32_t>(), p.nbuckets, red_lanes, bpw, parts);
        PM_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_sum_parts<C>, dim3(p.nwin), dim3(64), 0, ctx->stream, parts, bpw, dS);
        PM_HIP(ctx, hipGetLastError());
        PM_HIP(ctx, hipMemcpyAsync(hS.data(), dS, hS.size() * sizeof(XYZZ<C>), hipMemcpyDeviceToHost, ctx->stream));
    }
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    host_finish<C>(hS.data(), p.nwin, p.c, h_out, h_inf);
    return PM_OK;
}

// ------------------------------------------------------------------------- table-mode driver
// tb.wide: `plain` = the MSM's first base (internal form); the piece's pairs are plain[tb.base_index ...], every window has its own
// bucket set and the window sums are combined on the host.  Otherwise the window tables of tb.
// async_res != nullptr (table mode only): everything is ENQUEUED on ctx->stream, the reduced point (internal form) is copied to
// *async_res -- pinned host memory -- and the call returns without waiting; the caller synchronises and finishes (msm_end).
//
// (Round 4 measured the sort in CHUNKS of pairs -- chunk k + 1 sorted on a second stream under chunk k's accumulation, one bucket
// set, the next chunk's tasks continuing from the previous partials -- and lost: k_accumulate holds 2 x 248 of a SIMD's 512 registers,
// so no other wave can be resident beside it; the sort's 1024-lane workgroups wait for whole CUs to drain and the two kernels
// time-slice instead of overlapping.  +1.8 ms per proof at two chunks, +5 at three: profiles/r04_chunked_sort_overlap_negative.txt;
// the implementation is commit 5de5c70.)
template <class C>
static int msm_piece_tables(pm_ctx *ctx, const MsmTables &tb, const Fp<typename C::FrP> *d_scalars,
                            size_t len, Affine<C> *h_out, int *h_inf, const Affine<C> *plain = nullptr, XYZZ<C> *async_res = nullptr) {
    typedef typename C::FrP FrP;
    StageTimer t_total(ctx, T_MSM_TOTAL);
    MsmWorkspace &ws = ctx->msm;
    const unsigned c = tb.c, nwin = tb.nwin;
    const bool wide = tb.wide;
    if ((wide && !plain) || (wide && async_res)) return PM_ERR_INVALID_ARG;
    const size_t NB1 = (size_t)1 << (c - 1);               // buckets of one window
    // buckets of the pipeline: one shared set, or one set per window -- 2^(c-1) for the 256 % nwin windows of c bits, half of that for
    // the narrower ones (round 6; wide_narrow_buckets)
    const size_t NBn = wide ? wide_narrow_buckets(nwin, c) : 0;
    const unsigned n_wide_sets = !wide ? 1u : (NBn == NB1 ? nwin : 256 % nwin), n_narrow_sets = wide ? nwin - n_wide_sets : 0u;
    const size_t NB = (size_t)n_wide_sets * NB1 + (size_t)n_narrow_sets * NBn;
    const uint32_t win_buckets = wide ? (uint32_t)NB1 : 0u, narrow_buckets = (uint32_t)NBn;
    const unsigned lo_buckets = (unsigned)(NB < ((size_t)1 << LO_BITS) ? NB : ((size_t)1 << LO_BITS));
    const unsigned regions = (unsigned)(NB / lo_buckets);
    if (regions > 1024 || (size_t)regions * lo_buckets != NB) return PM_ERR_INVALID_ARG;   // whole regions only (wide mode: nwin 2^(c-1) buckets)
    for (unsigned w = 0; w < nwin; ++w)   // the kernels derive the layout from nwin alone
        if (tb.off[w] != win_off(nwin, w) || tb.width[w] != win_width(nwin, w)) return PM_ERR_INVALID_ARG;
    const bool two_level = NB1 >= 4096;                // msm_reduce.hip: k_reduce_level0 / level1 / final
    if (wide && !two_level) return PM_ERR_INVALID_ARG;  // wide plans have c >= 16 (setup.hip: wide_plan)
    const size_t Emax = (size_t)nwin * len;
    size_t seg = 2 * (Emax / NB + 1);
    if (seg < 64) seg = 64;
    if (ctx->opt.v[PM_OPT_MSM_TASK_LEN] > 0) seg = (size_t)ctx->opt.v[PM_OPT_MSM_TASK_LEN];
    const size_t max_tasks = NB + Emax / seg + 1;
    const unsigned chunk = 1u << RS_CHUNK_LOG;
    const size_t keys_bytes = (Emax * 2 + 15) & ~(size_t)15;
    PM_HIP(ctx, ws.digits.reserve(keys_bytes + Emax * 4 + 128));     // + 128: the staged passes read whole groups of ST_PER_LANE entries
    PM_HIP(ctx, ws.region.reserve((3 * (size_t)regions + 4) * 4));
    PM_HIP(ctx, ws.cursor.reserve(((NB + SCAN_TILE - 1) / SCAN_TILE + 1) * 8));
    MsmSet &S = ws.set;
    {
        PM_HIP(ctx, S.sorted.reserve(Emax * 4 + 16));      // + 16: k_accumulate reads aligned quads of indices
        PM_HIP(ctx, S.counts.reserve(2 * NB * 4));
        PM_HIP(ctx, S.bucket_off.reserve((NB + 1) * 4));
        PM_HIP(ctx, S.task_off.reserve((NB + 1) * 4));
        PM_HIP(ctx, S.partials.reserve(max_tasks * sizeof(XYZZ<C>)));
        PM_HIP(ctx, S.task_cnt.reserve(NB * 4));
    }
    const unsigned red_lanes = (unsigned)((NB + RED_K - 1) / RED_K);              // single-level path (small NB)
    unsigned red_block = 64;
    while (red_block < red_lanes && red_block < 256) red_block <<= 1;
    const unsigned bpw = (red_lanes + red_block - 1) / red_block;
    if (!two_level) PM_HIP(ctx, ws.wsum.reserve(((size_t)bpw + 4) * sizeof(XYZZ<C>)));
    uint16_t *keys = (uint16_t *)ws.digits.p;
    uint32_t *vals = (uint32_t *)((uint8_t *)ws.digits.p + keys_bytes);
    uint32_t *region_count = ws.region.as<uint32_t>(), *region_off = region_count + regions, *region_cursor = region_off + regions + 1;
#ifndef PM_PARTITION_LANES
#define PM_PARTITION_LANES 512
#endif
#ifndef PM_PARTITION_WIDE_LANES
#define PM_PARTITION_WIDE_LANES 1024
#endif
    // scalars per partition workgroup.  From 512 regions up (the 12-window wide plan of a 2^24-gate key: 4 x 2^21 + 8 x 2^20 buckets):
    // 1024, so that a workgroup's run inside a region is 24 entries (96 B of values), not 12 -- what the first level pays for is
    // the length of that run, not the number of regions: same-box A/B at 2^24 gates in profiles/r06_wide_ragged_sets_ab.txt (512
    // regions on 512 lanes: sort + 5 ... 8 ms against 768 regions on 1024) and r06_wide_12_windows_ab.txt
#ifndef PM_PARTITION_WIDE_FROM
#define PM_PARTITION_WIDE_FROM 512
#endif
    const unsigned pbd = regions >= PM_PARTITION_WIDE_FROM ? PM_PARTITION_WIDE_LANES : nwin <= 16 ? PM_PARTITION_LANES : 256;
    if (regions > 2 * pbd) return PM_ERR_INVALID_ARG;                  // at most two regions per scan lane (k_tbl_partition)
    const size_t plds = 2 * 1024 * 4 + (size_t)pbd * nwin * 8;
    {
        const unsigned pblocks_max = (unsigned)((len + pbd - 1) / pbd);
        const BlockScanShape bsh_max = block_scan_shape(pblocks_max, regions);
        PM_HIP(ctx, ws.block_cnt.reserve(((size_t)pblocks_max + bsh_max.G) * regions * 4));
    }
    const unsigned SUB_BINS = 128, FIN_BINS = 256, FIN_BITS = 8;
    const unsigned nsub = regions * SUB_BINS;
    if (lo_buckets == (1u << LO_BITS)) {
        PM_HIP(ctx, ws.sub.reserve((3 * (size_t)nsub + 4) * 4));
        PM_HIP(ctx, ws.digits2.reserve(keys_bytes + Emax * 4 + 128));
    }

    // ---- the sort: (scalar, window) entries -> table indices grouped by bucket, task order
    auto sort_all = [&]() -> int {
        StageTimer t(ctx, T_MSM_SORT);
        hipStream_t st = ctx->stream;
        const size_t lo = 0, cnt = len;
        const size_t E = (size_t)nwin * cnt;
        const unsigned char *inf = tb.inf + tb.base_index + lo;
        const Fp<FrP> *sc = d_scalars + lo;
        uint32_t *counts = S.counts.as<uint32_t>(), *cursor = counts + NB;
        PM_HIP(ctx, hipMemsetAsync(counts, 0, 2 * NB * 4, st));
        PM_HIP(ctx, hipMemsetAsync(region_count, 0, (size_t)regions * 4, st));
        const unsigned pblocks = (unsigned)((cnt + pbd - 1) / pbd);
        const BlockScanShape bsh = block_scan_shape(pblocks, regions);
        uint32_t *block_cnt = ws.block_cnt.as<uint32_t>(), *block_partial = block_cnt + (size_t)pblocks * regions;
        int launched = 0;
#define PM_TBL_CASE(NW)                                                                                                     \
        case NW:                                                                                                                \
            hipLaunchKernelGGL((k_tbl_count<FrP, NW>), dim3(pblocks), dim3(pbd), 0, st, sc, inf, cnt, regions,                  \
                               block_cnt, win_buckets, narrow_buckets);                                                         \
            hipLaunchKernelGGL(k_block_sums, dim3(bsh.G), dim3(1024), 0, st, block_cnt, pblocks, regions, bsh, block_partial);  \
            hipLaunchKernelGGL(k_block_offsets, dim3(bsh.G), dim3(1024), 0, st, block_cnt, pblocks, regions, bsh, block_partial, \
                               region_count);                                                                                  \
            hipLaunchKernelGGL(k_region_offsets, dim3(1), dim3(1024), 0, st, region_count, region_off, region_cursor,           \
                               regions);                                                                                        \
            if (hipFuncSetAttribute((const void *)k_tbl_partition<FrP, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                    (int)plds) != hipSuccess) break;                                                            \
            hipLaunchKernelGGL((k_tbl_partition<FrP, NW>), dim3(pblocks), dim3(pbd), plds, st, sc, inf, cnt,                    \
                               regions, region_off, block_cnt, wide ? (size_t)0 : tb.stride, tb.base_index + lo, keys, vals,    \
                               win_buckets, narrow_buckets);                                                                    \
            launched = 1;                                                                                                       \
            break;
        switch (nwin) {
            PM_TBL_CASE(10) PM_TBL_CASE(11) PM_TBL_CASE(12) PM_TBL_CASE(13) PM_TBL_CASE(14) PM_TBL_CASE(15) PM_TBL_CASE(16)
            PM_TBL_CASE(17) PM_TBL_CASE(18) PM_TBL_CASE(19) PM_TBL_CASE(20) PM_TBL_CASE(21) PM_TBL_CASE(22) PM_TBL_CASE(23)
            PM_TBL_CASE(24) PM_TBL_CASE(25) PM_TBL_CASE(26) PM_TBL_CASE(27) PM_TBL_CASE(28) PM_TBL_CASE(29) PM_TBL_CASE(30)
            PM_TBL_CASE(31) PM_TBL_CASE(32)
            default: break;
        }
#undef PM_TBL_CASE
        if (!launched) return PM_ERR_INVALID_ARG;
        PM_HIP(ctx, hipGetLastError());
        const unsigned sblocks = (unsigned)((E + chunk - 1) / chunk);
        const unsigned ntiles = (unsigned)((NB + SCAN_TILE - 1) / SCAN_TILE);
        auto bucket_scan = [&]() -> int {
            hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(256), 0, st, counts, S.bucket_off.as<uint32_t>(),
                               S.task_off.as<uint32_t>(), ws.cursor.as<uint32_t>(), NB, (unsigned)seg);
            PM_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, st, ws.cursor.as<uint32_t>(), ntiles,
                               S.bucket_off.as<uint32_t>(), S.task_off.as<uint32_t>(), NB);
            PM_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(256), 0, st, S.bucket_off.as<uint32_t>(),
                               S.task_off.as<uint32_t>(), ws.cursor.as<uint32_t>(), NB);
            PM_HIP(ctx, hipGetLastError());
            return PM_OK;
        };
        if (lo_buckets == (1u << LO_BITS)) {
            // three levels: regions (2^15 buckets) -> 128 sub-regions of 256 buckets -> buckets
            uint32_t *sub_count = ws.sub.as<uint32_t>(), *sub_off = sub_count + nsub, *sub_cursor = sub_off + nsub + 1;
            uint16_t *keys2 = (uint16_t *)ws.digits2.p;
            uint32_t *vals2 = (uint32_t *)((uint8_t *)ws.digits2.p + keys_bytes);
            PM_HIP(ctx, hipMemsetAsync(sub_count, 0, (size_t)nsub * 4, st));
            hipLaunchKernelGGL(k_hist_small, dim3(sblocks), dim3(1024), 0, st, keys, region_off, regions, FIN_BITS, SUB_BINS, chunk,
                               sub_count);
            PM_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, st, sub_count, sub_off, sub_cursor, nsub);
            PM_HIP(ctx, hipGetLastError());
            const unsigned stblocks = (unsigned)((E + ST_CHUNK - 1) / ST_CHUNK);
            hipLaunchKernelGGL(k_region_pass_staged<RS_MID>, dim3(stblocks), dim3(ST_THREADS), 0, st, keys, vals, region_off, regions,
                               FIN_BITS, SUB_BINS, sub_off, sub_cursor, (uint32_t *)nullptr, keys2, vals2);
            PM_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL(k_hist_small, dim3(sblocks), dim3(1024), 0, st, keys2, sub_off, nsub, 0u, FIN_BINS, chunk, counts);
            PM_HIP(ctx, hipGetLastError());
            PM_TRY(bucket_scan());
            // (round 5 measured the last level WITHOUT LDS staging -- lanes storing their 4-byte indices straight into the sub-region's
            // 120 KB output window, which stays in L2: the sort of a 2^24-pair MSM 2.66 -> 3.70 ms, a proof +2.2 ms; 64 partial-line
            // stores per wave instruction cost more than the staging saves: profiles/r05_sort_final_direct_negative.txt)
            hipLaunchKernelGGL(k_region_pass_staged<RS_FINAL>, dim3(stblocks), dim3(ST_THREADS), 0, st, keys2, vals2, sub_off, nsub, 0u,
                               FIN_BINS, S.bucket_off.as<uint32_t>(), cursor, S.sorted.as<uint32_t>(), (uint16_t *)nullptr,
                               (uint32_t *)nullptr);
            PM_HIP(ctx, hipGetLastError());
        } else {
            // small bucket sets (< 2^15): one region, sorted directly with an nbuckets-entry LDS table
            const size_t lds = (size_t)lo_buckets * 4;
            hipLaunchKernelGGL(k_region_pass<RS_HIST>, dim3(sblocks), dim3(1024), lds, st, keys, vals, region_off, regions, 0u,
                               lo_buckets, chunk, counts, (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                               (uint16_t *)nullptr, (uint32_t *)nullptr);
            PM_HIP(ctx, hipGetLastError());
            PM_TRY(bucket_scan());
            hipLaunchKernelGGL(k_region_pass<RS_FINAL>, dim3(sblocks), dim3(1024), lds, st, keys, vals, region_off, regions, 0u,
                               lo_buckets, chunk, (uint32_t *)nullptr, S.bucket_off.as<uint32_t>(), cursor, S.sorted.as<uint32_t>(),
                               (uint16_t *)nullptr, (uint32_t *)nullptr);
            PM_HIP(ctx, hipGetLastError());
        }
        PM_TRY(task_order(ctx, S, counts, NB, seg, max_tasks));
        return PM_OK;
    };
    PM_TRY(sort_all());
    {
        StageTimer t(ctx, T_MSM_ACCUMULATE);
        const size_t blocks = (max_tasks + 127) / 128;
        if (wide)
            hipLaunchKernelGGL((k_accumulate<C, false>), dim3((unsigned)blocks), dim3(128), 0, ctx->stream, S.sorted.as<uint32_t>(), S.counts.as<uint32_t>(),
                               S.bucket_off.as<uint32_t>(), S.task_off.as<uint32_t>(), S.order.as<uint32_t>(), (const void *)plain,
                               S.partials.as<XYZZ<C>>(), NB, (unsigned)seg);
        else
            hipLaunchKernelGGL((k_accumulate<C, true>), dim3((unsigned)blocks), dim3(128), 0, ctx->stream, S.sorted.as<uint32_t>(), S.counts.as<uint32_t>(),
                               S.bucket_off.as<uint32_t>(), S.task_off.as<uint32_t>(), S.order.as<uint32_t>(), tb.table,
                               S.partials.as<XYZZ<C>>(), NB, (unsigned)seg);
        PM_HIP(ctx, hipGetLastError());
    }
    if (wide) {
        // all windows' bucket sets reduced by ONE set of launches; then sum_w 2^(off_w) S_w by Horner from the top window: a chain
        // of 256 dependent doublings -- a few hundred microseconds on the host, milliseconds on one GPU lane
        std::vector<XYZZ<C>> hS(nwin);
        {
            StageTimer t(ctx, T_MSM_REDUCE);
            PM_TRY(fold_hot_buckets<C>(ctx, S, NB, max_tasks));
            XYZZ<C> *dres = nullptr;
            PM_TRY(reduce_two_level<C>(ctx, NB1, &dres, n_wide_sets));                       // the sets of the c-bit windows ...
            PM_HIP(ctx, hipMemcpyAsync(hS.data(), dres, n_wide_sets * sizeof(XYZZ<C>), hipMemcpyDeviceToHost, ctx->stream));
            if (n_narrow_sets) {                                                             // ... then those of the (c - 1)-bit ones (stream order:
                PM_TRY(reduce_two_level<C>(ctx, NBn, &dres, n_narrow_sets, (size_t)n_wide_sets * NB1));   // the workspace is free again)
                PM_HIP(ctx, hipMemcpyAsync(hS.data() + n_wide_sets, dres, n_narrow_sets * sizeof(XYZZ<C>), hipMemcpyDeviceToHost, ctx->stream));
            }
        }
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        XYZZ<C> acc = XYZZ<C>::identity();
        for (int w = (int)nwin - 1; w >= 0; --w) {
            for (unsigned b = 0; b < tb.width[w]; ++b) acc = xyzz_dbl<C>(acc);
            acc = xyzz_add<C>(acc, xyzz_internal_to_std<C>(hS[w]));
        }
        *h_inf = acc.is_identity() ? 1 : 0;
        *h_out = xyzz_to_affine<C>(acc);
        return PM_OK;
    }
    XYZZ<C> hres;
    {
        StageTimer t(ctx, T_MSM_REDUCE);
        PM_TRY(fold_hot_buckets<C>(ctx, S, NB, max_tasks));
        XYZZ<C> *dres = nullptr;
        if (two_level) {
            PM_TRY(reduce_two_level<C>(ctx, NB, &dres));
        } else {
            XYZZ<C> *parts = ws.wsum.as<XYZZ<C>>();
            dres = parts + bpw;
            hipLaunchKernelGGL(k_bucket_reduce<C>, dim3(bpw), dim3(red_block), red_block * sizeof(XYZZ28<C>), ctx->stream,
                               S.partials.as<XYZZ<C>>(), S.task_off.as<uint32_t>(), S.task_cnt.as<uint32_t>(), (unsigned)NB, red_lanes, bpw, parts);
            PM_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL(k_sum_parts<C>, dim3(1), dim3(64), 0, ctx->stream, parts, bpw, dres);
            PM_HIP(ctx, hipGetLastError());
        }
        PM_HIP(ctx, hipMemcpyAsync(async_res ? async_res : &hres, dres, sizeof(hres), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (async_res) return PM_OK;                       // msm_end: stream sync, then the two lines below
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    XYZZ<C> acc = xyzz_internal_to_std<C>(hres);
    *h_inf = acc.is_identity() ? 1 : 0;
    *h_out = xyzz_to_affine<C>(acc);
    return PM_OK;
}

template <class C>
int msm_run(pm_ctx *ctx, const Affine<C> *d_bases, const Fp<typename C::FrP> *d_scalars, size_t len, Affine<C> *h_out,
            int *h_inf, const MsmTables *tables) {
    if (len == 0) {
        *h_out = Affine<C>::infinity();
        *h_inf = 1;
        return PM_OK;
    }
    const bool tbl = tables && tables->c;
    const size_t MSM_MAX_PIECE = msm_max_piece(ctx);
    const bool wide = tbl && tables->wide;
    if (len <= MSM_MAX_PIECE)
        return tbl ? msm_piece_tables<C>(ctx, *tables, d_scalars, len, h_out, h_inf, wide ? d_bases : nullptr)
                   : msm_piece<C>(ctx, d_bases, d_scalars, len, h_out, h_inf);
    // very long MSMs (the 10n-pair quotient commitment at n >= 2^24 on one GPU): pieces, summed on the host
    XYZZ<C> acc = XYZZ<C>::identity();
    for (size_t off = 0; off < len; off += MSM_MAX_PIECE) {
        size_t cnt = len - off < MSM_MAX_PIECE ? len - off : MSM_MAX_PIECE;
        Affine<C> part;
        int inf = 1;
        if (tbl) {
            MsmTables tb = *tables;
            tb.base_index += off;
            PM_TRY(msm_piece_tables<C>(ctx, tb, d_scalars + off, cnt, &part, &inf, wide ? d_bases : nullptr));
        } else {
            PM_TRY(msm_piece<C>(ctx, d_bases + off, d_scalars + off, cnt, &part, &inf));
        }
        if (!inf) xyzz_madd<C>(acc, part, false);
    }
    *h_inf = acc.is_identity() ? 1 : 0;
    *h_out = xyzz_to_affine<C>(acc);
    return PM_OK;
}

// Asynchronous pair for a single-piece table-mode MSM: msm_begin enqueues the whole pipeline on ctx->stream and returns at once
// (the result travels to the context's pinned slot), msm_end waits for the stream and converts the point.  Two contexts (ctx and
// its helper ctx->aux: own streams, own workspaces) can so run two MSMs concurrently from ONE host thread -- the latency-bound
// sort front end and bucket reduction of one hide under the accumulation of the other.  MSMs that need the host in the middle
// (several pieces, wide mode, no tables) run synchronously inside msm_begin; msm_end then just hands the result over.
template <class C>
int msm_begin(pm_ctx *ctx, const Affine<C> *d_bases, const Fp<typename C::FrP> *d_scalars, size_t len, const MsmTables *tables) {
    ctx->msm_async = 0;
    const bool tbl = tables && tables->c && !tables->wide;
    if (!ctx_pinned(ctx)) { ctx->err = "pinned result slot allocation failed"; return PM_ERR_HIP; }
    Affine<C> *slot_pt = (Affine<C> *)((uint8_t *)ctx->h_pinned + 1024);
    int *slot_inf = (int *)((uint8_t *)ctx->h_pinned + 2048);
    if (len == 0 || !tbl || len > msm_max_piece(ctx)) {           // synchronous: result parked in the slot
        PM_TRY(msm_run<C>(ctx, d_bases, d_scalars, len, slot_pt, slot_inf, tables));
        ctx->msm_async = 2;
        return PM_OK;
    }
    PM_TRY(msm_piece_tables<C>(ctx, *tables, d_scalars, len, (Affine<C> *)nullptr, (int *)nullptr, (const Affine<C> *)nullptr, (XYZZ<C> *)ctx->h_pinned));
    ctx->msm_async = 1;
    return PM_OK;
}

template <class C>
int msm_end(pm_ctx *ctx, Affine<C> *h_out, int *h_inf) {
    if (ctx->msm_async == 2) {
        *h_out = *(const Affine<C> *)((const uint8_t *)ctx->h_pinned + 1024);
        *h_inf = *(const int *)((const uint8_t *)ctx->h_pinned + 2048);
    } else if (ctx->msm_async == 1) {
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const XYZZ<C> acc = xyzz_internal_to_std<C>(*(const XYZZ<C> *)ctx->h_pinned);
        *h_inf = acc.is_identity() ? 1 : 0;
        *h_out = xyzz_to_affine<C>(acc);
    } else {
        return PM_ERR_STATE;
    }
    ctx->msm_async = 0;
    return PM_OK;
}

template int msm_begin<BlsCurve>(pm_ctx *, const Affine<BlsCurve> *, const Fp<BlsFrP> *, size_t, const MsmTables *);
template int msm_begin<BnCurve>(pm_ctx *, const Affine<BnCurve> *, const Fp<BnFrP> *, size_t, const MsmTables *);
template int msm_end<BlsCurve>(pm_ctx *, Affine<BlsCurve> *, int *);
template int msm_end<BnCurve>(pm_ctx *, Affine<BnCurve> *, int *);
template int msm_run<BlsCurve>(pm_ctx *, const Affine<BlsCurve> *, const Fp<BlsFrP> *, size_t, Affine<BlsCurve> *, int *, const MsmTables *);
template int msm_run<BnCurve>(pm_ctx *, const Affine<BnCurve> *, const Fp<BnFrP> *, size_t, Affine<BnCurve> *, int *, const MsmTables *);

}  // namespace pm
