// Vector-sharded layout of ONE proof over N GPUs: who owns which
// evaluation row, which coefficient, which MSM pair and which stretch of the quotient.  Pure index arithmetic,
// host and device; the only file that knows the layout -- the key builder (api.hip), the sharded prover
// (prove_sharded.hip), the C ABI's pm_layout_* helpers and the tests all go through it.
//
// n = domain size, N = ranks (power of two, N^2 <= n), m = n / N, B = m / N.
//   evaluations  (prover.rs:87-96 u_evals / w_evals): CYCLIC   -- rank r owns rows i = N j + r, local index j < m;
//   coefficients (prover.rs:239-243 poly_coeffs)    : BLOCKED  -- rank q owns k = k1 m + q B + b  (k1 < N, b < B),
//                                                                 local index p = k1 B + b.
// A size-n (i)NTT between the two is N local size-m transforms, one twiddle, ONE all-to-all of B-element blocks and a
// size-N butterfly across what the exchange brought together (the four-step NTT with the transpose being the
// exchange).  Everything pointwise in k stays local; the scans (Horner, synthetic division) exchange per-segment
// values only.  MSM pairs follow the coefficients they multiply, so no scalar ever moves between GPUs.
#pragma once
#include <stdint.h>

#include <vector>

#ifndef PM_LAYOUT_HD
#ifdef __HIPCC__
#define PM_LAYOUT_HD __host__ __device__ inline
#else
#define PM_LAYOUT_HD inline
#endif
#endif

namespace pmlayout {

struct Layout {
    uint64_t n, m, B;
    uint32_t N, q;   // ranks, this rank
};

inline Layout make_layout(uint64_t n, uint32_t N, uint32_t q) {
    Layout L;
    L.n = n; L.N = N; L.q = q;
    L.m = n / N;
    L.B = L.m / N;
    return L;
}
inline bool layout_ok(uint64_t n, uint32_t N) { return N >= 1 && (N & (N - 1)) == 0 && (uint64_t)N * N <= n && n % ((uint64_t)N * N) == 0; }

// coefficient index k <-> (owner, local position)
PM_LAYOUT_HD uint32_t coeff_owner(const Layout &L, uint64_t k) { return (uint32_t)((k % L.m) / L.B); }
PM_LAYOUT_HD uint64_t coeff_local(const Layout &L, uint64_t k) { return (k / L.m) * L.B + (k % L.B); }
PM_LAYOUT_HD uint64_t coeff_global(const Layout &L, uint64_t p) { return (p / L.B) * L.m + (uint64_t)L.q * L.B + (p % L.B); }
// evaluation row i <-> (owner, local position)
PM_LAYOUT_HD uint32_t eval_owner(const Layout &L, uint64_t i) { return (uint32_t)(i % L.N); }
PM_LAYOUT_HD uint64_t eval_local(const Layout &L, uint64_t i) { return i / L.N; }
PM_LAYOUT_HD uint64_t eval_global(const Layout &L, uint64_t j) { return j * L.N + L.q; }

// ---------------------------------------------------------------------------------------------- MSM pair lists
// A rank's share of a merged MSM is a list of pieces of the key's LOGICAL base concatenation
//   [uj_wj_lcs (Lz) | x_powers_zh (n-1) | x_powers (n+1) | y_alpha (3) | y_gamma (2) | y_gamma_z (10n+23)]
// stored back to back in HBM in list order; the scalar vector the prover builds has the same order.
struct Piece {
    uint64_t cat_lo, count;
};

struct KeyShape {
    uint64_t n, m0, mw, nr, sigma, Lz;
    uint64_t off_lcs, off_zh, off_xp, off_ya, off_yg, off_ygz;   // segment offsets inside the concatenation
};

inline KeyShape key_shape(uint64_t n, uint64_t m0, uint64_t mw, uint64_t nr) {
    KeyShape s;
    s.n = n; s.m0 = m0; s.mw = mw; s.nr = nr; s.sigma = n + 3;
    s.Lz = 2 * m0 + mw + nr;
    s.off_lcs = 0;
    s.off_zh = s.Lz;
    s.off_xp = s.off_zh + (n - 1);
    s.off_ya = s.off_xp + (n + 1);
    s.off_yg = s.off_ya + 3;
    s.off_ygz = s.off_yg + 2;
    return s;
}

// [a]_1 = sum u_k X_k + r_a(X) Y^alpha  (prover.rs:330-338): the rank's u blocks; r_a's two pairs on rank 0
inline std::vector<Piece> pieces_a(const KeyShape &s, const Layout &L) {
    std::vector<Piece> v;
    for (uint32_t k1 = 0; k1 < L.N; ++k1) v.push_back({s.off_xp + k1 * L.m + (uint64_t)L.q * L.B, L.B});
    if (L.q == 0) v.push_back({s.off_ya, 2});
    return v;
}
// z_tail slice of rank q: contiguous [lo, hi) of the Lz entries
PM_LAYOUT_HD uint64_t ztail_lo(uint64_t Lz, uint32_t N, uint32_t q) { return Lz * q / N; }

// [c]_1 (prover.rs:118-123, 340-357): z_tail slice | h blocks (index n-1 does not exist) | 2 r_a(X) u(X) blocks, each with
// ONE extra base X_e (e = block end): coefficient i of 2 r_a u is 2 (r0 u_i + r1 u_{i-1}), and u_{e-1} belongs to this
// rank while X_e's other term belongs to the next block's owner -- the MSM is linear, so the pair (2 r1 u_{e-1}, X_e) is
// simply added here | r_a^2 (3) and r_a (2) on rank 0.
inline std::vector<Piece> pieces_c(const KeyShape &s, const Layout &L) {
    std::vector<Piece> v;
    const uint64_t zl = ztail_lo(s.Lz, L.N, L.q), zh = ztail_lo(s.Lz, L.N, L.q + 1);
    v.push_back({s.off_lcs + zl, zh - zl});
    for (uint32_t k1 = 0; k1 < L.N; ++k1) {
        const uint64_t k = k1 * L.m + (uint64_t)L.q * L.B;
        uint64_t cnt = L.B;
        if (k + cnt > s.n - 1) cnt = s.n - 1 - k;   // h has n - 1 coefficients
        v.push_back({s.off_zh + k, cnt});
    }
    for (uint32_t k1 = 0; k1 < L.N; ++k1) v.push_back({s.off_xp + k1 * L.m + (uint64_t)L.q * L.B, L.B + 1});
    if (L.q == 0) {
        v.push_back({s.off_ya, 3});
        v.push_back({s.off_yg, 2});
    }
    return v;
}

// ------------------------------------------------------------------------------------- quotient segments
// Numerator index space [0, len), len = 8 sigma + 2n - 1 (prover.rs:211-225, multiplied through by X^(5 sigma)):
//   [0, 3s)            constants at 0,1 and 2s..2s+2, zeros elsewhere          "filler 0"
//   3s + [0, n)        x2 * witness_u                                          region R2
//   [3s + n, 5s)       zeros                                                   "filler 1"
//   5s + [0, n]        u, 2 x2 r_a u, constants                                region R3 (n + 1 entries)
//   [5s + n + 1, 8s)   zeros                                                   "filler 2"
//   8s + [0, n)        x2 * (u^2)_lo                                           region R4 lo
//   8s + n + [0, n-1)  x2 * (u^2)_hi                                           region R4 hi
// The quotient coefficient q_{k-1} = H_k = sum_{j >= k} N_j x1^(j-k) is dense over ALL of it, so every index has an owner:
// data regions follow the coefficient layout (block (k1, q) -> rank q), fillers are cut into N equal chunks.  Long
// stretches are cut further into sub-segments of at most `max_seg` indices: a sub-segment is the unit of the scan (one
// workgroup, one exchanged value).
enum SegKind { SEG_FILLER = 0, SEG_WITU = 1, SEG_U = 2, SEG_U2LO = 3, SEG_U2HI = 4 };

struct Segment {
    uint64_t a, b;       // numerator indices [a, b)
    uint64_t loc0;       // local array position of the first element (data regions)
    uint64_t qoff;       // offset of H_a (or of H_1 when a == 0) inside the rank's quotient-scalar vector
    uint32_t kind;
    uint32_t halo;       // SEG_U: 1 + index k1 of the block whose predecessor value u[start - 1] is needed at `a`, 0 = none
};

inline uint64_t numerator_len(uint64_t n) { return 8 * (n + 3) + 2 * n - 1; }

// all segments of rank `q` in increasing index order
inline std::vector<Segment> quotient_segments(uint64_t n, uint32_t N, uint32_t q, uint64_t max_seg) {
    const Layout L = make_layout(n, N, q);
    const uint64_t s = n + 3, len = numerator_len(n);
    std::vector<Segment> out;
    auto push_split = [&](uint64_t a, uint64_t b, uint32_t kind, uint64_t loc0, uint32_t halo) {
        for (uint64_t x = a; x < b; x += max_seg) {
            const uint64_t y = x + max_seg < b ? x + max_seg : b;
            out.push_back(Segment{x, y, loc0 + (x - a), 0, kind, x == a ? halo : 0u});
        }
    };
    auto filler = [&](uint64_t lo, uint64_t hi) {
        const uint64_t w = hi - lo, a = lo + w * q / N, b = lo + w * (q + 1) / N;
        if (a < b) push_split(a, b, SEG_FILLER, 0, 0);
    };
    auto blocks = [&](uint64_t base, uint32_t kind, uint64_t limit, bool extend_last) {
        for (uint32_t k1 = 0; k1 < N; ++k1) {
            const uint64_t k = k1 * L.m + (uint64_t)q * L.B;
            uint64_t cnt = L.B;
            if (extend_last && k + cnt == n) cnt += 1;        // region R3 has the extra index i = n (owned with u[n-1])
            if (k + cnt > limit) cnt = limit - k;
            uint32_t halo = 0;
            if (kind == SEG_U && k > 0) halo = 1 + k1;
            if (cnt) push_split(base + k, base + k + cnt, kind, (uint64_t)k1 * L.B, halo);
        }
    };
    filler(0, 3 * s);
    blocks(3 * s, SEG_WITU, n, false);
    filler(3 * s + n, 5 * s);
    blocks(5 * s, SEG_U, n + 1, true);
    filler(5 * s + n + 1, 8 * s);
    blocks(8 * s, SEG_U2LO, n, false);
    blocks(8 * s + n, SEG_U2HI, n - 1, false);
    (void)len;
    uint64_t off = 0;
    for (auto &g : out) {
        g.qoff = off;
        off += (g.b - g.a) - (g.a == 0 ? 1 : 0);   // index 0 is the remainder, not a quotient coefficient
    }
    return out;
}

// [d]_1 = sum_k H_k [x^(k-1) y^gamma z]  (prover.rs:229): base index k - 1 for every owned k >= 1
inline std::vector<Piece> pieces_d(const KeyShape &s, const std::vector<Segment> &segs) {
    std::vector<Piece> v;
    for (const auto &g : segs) {
        const uint64_t a = g.a == 0 ? 1 : g.a;
        if (a < g.b) {
            if (!v.empty() && v.back().cat_lo + v.back().count == s.off_ygz + a - 1) v.back().count += g.b - a;   // adjacent: merge
            else v.push_back({s.off_ygz + a - 1, g.b - a});
        }
    }
    return v;
}

// sub-segment size: at most ~384 segments per rank (one or two 512-lane workgroups per CU; the exchanged record is one Fr per
// segment), never below 2^13 indices, a power of two.  Measured at n = 2^21, N = 8 (profiles/r03_k_*): the scan's two kernels take
// 0.34 ms with 320 segments of 2^13 against 0.50 ms with 160 segments of 2^14 on 1024 lanes.
inline uint64_t pick_max_seg(uint64_t n, uint32_t N) {
    const uint64_t total = numerator_len(n) / N;
    uint64_t ms = (uint64_t)1 << 13;
    while (total / ms > 384) ms <<= 1;
    return ms;
}

}  // namespace pmlayout
