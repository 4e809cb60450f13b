// ORACLE -- test infrastructure only (see field64.hpp header; PARITY UNPINNED).
//
// CPU restatement of the Polymath prover hot path and of the setup that feeds it, on 64-bit
// Montgomery limbs.  It computes the SAME VALUES as /root/reference/src/{prover,generator,common}.rs
// by sparse routes (the reference's dense O(n*M) materialisation cannot run past n ~ 2^11,
// SURVEY.md finding 0.5); the sparse closed forms are checked against the literal dense
// transcription in oracle/pyref on small circuits (tests/test_oracle_cpp.py).
//
// Exported C symbols (po_*) mirror include/polymath_hip.h so the parity tests drive both with
// the same buffers.  Multi-threaded with std::thread so bench.py can time it as the
// "CPU restatement -- not arkworks" baseline.
#include "field64.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------ util
// PO_PROFILE=1: wall time of the prover's stages on stderr (where the CPU baseline of bench.py goes)
struct StageClock {
    const bool on = getenv("PO_PROFILE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[oracle] %-28s %9.3f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};
static void parallel_for(size_t n, int nthreads, const std::function<void(size_t, size_t, int)> &fn) {
    if (nthreads <= 1 || n < 2) { fn(0, n, 0); return; }
    std::vector<std::thread> th;
    size_t chunk = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        size_t lo = (size_t)t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back(fn, lo, hi, t);
    }
    for (auto &x : th) x.join();
}

static void hex_to_limbs(const char *hex, u64 *out, int n) {
    memset(out, 0, 8 * n);
    int len = (int)strlen(hex);
    for (int i = 0; i < len; ++i) {
        char ch = hex[len - 1 - i];
        u64 v = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : ch - 'A' + 10;
        out[i / 16] |= v << (4 * (i % 16));
    }
}

// --------------------------------------------------------------------------------- curves
struct BLS {
    typedef Fp<6, 1> Fq;
    typedef Fp<4, 0> Fr;
    static constexpr int ID = 0;
    static constexpr unsigned TWO_ADICITY = 32;
    static constexpr u64 FR_GENERATOR = 7;
    static constexpr u64 B = 4;
    static const char *p_hex() { return "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab"; }
    static const char *r_hex() { return "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001"; }
    static const char *gx_hex() { return "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"; }
    static const char *gy_hex() { return "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1"; }
};
struct BN {
    typedef Fp<4, 3> Fq;
    typedef Fp<4, 2> Fr;
    static constexpr int ID = 1;
    static constexpr unsigned TWO_ADICITY = 28;
    static constexpr u64 FR_GENERATOR = 5;
    static constexpr u64 B = 3;
    static const char *p_hex() { return "30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47"; }
    static const char *r_hex() { return "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001"; }
    static const char *gx_hex() { return "1"; }
    static const char *gy_hex() { return "2"; }
};

template <class C>
struct CurveCtx {
    typedef typename C::Fq Fq;
    typedef typename C::Fr Fr;
    Fq b, gx, gy;
    Fr two_adic_root;  // FR_GENERATOR^((r-1)/2^s)   [ark-ff MontConfig derive]
    static CurveCtx &get() {
        static CurveCtx c = make();
        return c;
    }
    static CurveCtx make() {
        u64 m[8];
        hex_to_limbs(C::p_hex(), m, Fq::LIMBS);
        Fq::init(m);
        hex_to_limbs(C::r_hex(), m, Fr::LIMBS);
        Fr::init(m);
        CurveCtx c;
        c.b = Fq::from_u64(C::B);
        u64 t[8];
        hex_to_limbs(C::gx_hex(), t, Fq::LIMBS);
        c.gx = Fq::from_raw(t).to_mont();
        hex_to_limbs(C::gy_hex(), t, Fq::LIMBS);
        c.gy = Fq::from_raw(t).to_mont();
        // trace = (r-1) >> s
        u64 e[4];
        memcpy(e, Fr::P.mod, 32);
        e[0] -= 1;
        unsigned s = C::TWO_ADICITY;
        for (int i = 0; i < 4; ++i) {
            u64 lo = (s == 64) ? 0 : e[i] >> s;
            u64 hi = (i + 1 < 4 && s) ? e[i + 1] << (64 - s) : 0;
            e[i] = (s >= 64) ? 0 : (lo | hi);
        }
        c.two_adic_root = Fr::from_u64(C::FR_GENERATOR).pow_limbs(e, 4);
        return c;
    }
    Fr root_of_unity(unsigned log_n) const {  // Radix2EvaluationDomain::new
        Fr w = two_adic_root;
        for (unsigned i = log_n; i < C::TWO_ADICITY; ++i) w = w.sqr();
        return w;
    }
};

// ------------------------------------------------------------------------- G1 (Jacobian)
template <class C>
struct Aff {
    typename C::Fq x, y;
    bool inf;
};
template <class C>
struct Jac {
    typedef typename C::Fq Fq;
    Fq X, Y, Z;
    static Jac identity() { return Jac{Fq::one(), Fq::one(), Fq::zero()}; }
    bool is_identity() const { return Z.is_zero(); }
    Jac dbl() const {  // dbl-2009-l (a = 0)
        if (is_identity() || Y.is_zero()) return identity();
        Fq A = X.sqr(), B = Y.sqr(), Cc = B.sqr();
        Fq D = ((X + B).sqr() - A - Cc).dbl();
        Fq E = A.dbl() + A, F = E.sqr();
        Jac r;
        r.X = F - D.dbl();
        r.Y = E * (D - r.X) - Cc.dbl().dbl().dbl();
        r.Z = (Y * Z).dbl();
        return r;
    }
    Jac add_affine(const Aff<C> &q) const {  // madd-2007-bl
        if (q.inf) return *this;
        if (is_identity()) return Jac{q.x, q.y, Fq::one()};
        Fq Z1Z1 = Z.sqr(), U2 = q.x * Z1Z1, S2 = q.y * Z * Z1Z1;
        Fq H = U2 - X, rr = S2 - Y;
        if (H.is_zero()) return rr.is_zero() ? dbl() : identity();
        Fq HH = H.sqr(), HHH = H * HH, V = X * HH;
        Jac r;
        r.X = rr.sqr() - HHH - V.dbl();
        r.Y = rr * (V - r.X) - Y * HHH;
        r.Z = Z * H;
        return r;
    }
    Jac add(const Jac &q) const {  // add-2007-bl
        if (q.is_identity()) return *this;
        if (is_identity()) return q;
        Fq Z1Z1 = Z.sqr(), Z2Z2 = q.Z.sqr();
        Fq U1 = X * Z2Z2, U2 = q.X * Z1Z1, S1 = Y * q.Z * Z2Z2, S2 = q.Y * Z * Z1Z1;
        Fq H = U2 - U1, rr = S2 - S1;
        if (H.is_zero()) return rr.is_zero() ? dbl() : identity();
        Fq HH = H.sqr(), HHH = H * HH, V = U1 * HH;
        Jac r;
        r.X = rr.sqr() - HHH - V.dbl();
        r.Y = rr * (V - r.X) - S1 * HHH;
        r.Z = Z * q.Z * H;
        return r;
    }
    Aff<C> to_affine() const {
        if (is_identity()) return Aff<C>{Fq::zero(), Fq::zero(), true};
        Fq zi = Z.inverse(), zi2 = zi.sqr();
        return Aff<C>{X * zi2, Y * zi2 * zi, false};
    }
};

template <class C>
static void batch_to_affine(const Jac<C> *in, Aff<C> *out, size_t n) {
    typedef typename C::Fq Fq;
    std::vector<Fq> z(n), scratch(n);
    for (size_t i = 0; i < n; ++i) z[i] = in[i].Z;
    batch_inverse(z.data(), n, scratch.data());
    for (size_t i = 0; i < n; ++i) {
        if (in[i].is_identity()) { out[i] = Aff<C>{Fq::zero(), Fq::zero(), true}; continue; }
        Fq zi2 = z[i].sqr();
        out[i] = Aff<C>{in[i].X * zi2, in[i].Y * zi2 * z[i], false};
    }
}

template <class C>
static Aff<C> load_affine(const uint8_t *p, size_t stride) {
    typedef typename C::Fq Fq;
    const int NQ = Fq::LIMBS;
    Aff<C> a;
    a.x = Fq::from_raw((const u64 *)p);
    a.y = Fq::from_raw((const u64 *)p + NQ);
    a.inf = (a.x.is_zero() && a.y.is_zero()) || (stride > (size_t)16 * NQ && p[16 * NQ] != 0);
    return a;
}
template <class C>
static void store_affine(const Aff<C> &a, u64 *out_xy, int *out_inf) {
    const int NQ = C::Fq::LIMBS;
    if (a.inf) { memset(out_xy, 0, 16 * NQ); if (out_inf) *out_inf = 1; return; }
    a.x.store(out_xy);
    a.y.store(out_xy + NQ);
    if (out_inf) *out_inf = 0;
}

template <class C>
static Jac<C> g1_mul(const Aff<C> &p, const typename C::Fr &k_mont) {
    typename C::Fr k = k_mont.from_mont();
    Jac<C> acc = Jac<C>::identity();
    for (int i = 3; i >= 0; --i)
        for (int b = 63; b >= 0; --b) {
            acc = acc.dbl();
            if ((k.l[i] >> b) & 1) acc = acc.add_affine(p);
        }
    return acc;
}

// ----------------------------------------------------------------------------- Pippenger
// Restates what E::G1::msm_unchecked computes (prover.rs:380-384): sum_i s_i * P_i over the
// zipped prefix.  The result is a canonical group element, so the bucket schedule is free.
static inline unsigned get_bits(const u64 *k, unsigned lo, unsigned c) {
    unsigned limb = lo / 64, off = lo % 64;
    if (limb >= 4) return 0;
    u64 v = k[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
    return (unsigned)(v & ((1ull << c) - 1));
}

// One Pippenger window over a contiguous range: sum_b (b+1) * bucket_b for the window's signed digits.
template <class C>
static Jac<C> msm_window(const uint8_t *bases, size_t stride, const typename C::Fr *canon, const unsigned char *carry_in,
                         size_t len, unsigned w, unsigned c) {
    std::vector<Jac<C>> buckets((size_t)1 << (c - 1), Jac<C>::identity());
    for (size_t i = 0; i < len; ++i) {
        int d = (int)get_bits(canon[i].l, w * c, c) + (int)carry_in[i];
        if (d > (1 << (c - 1))) d -= (1 << c);
        if (d == 0) continue;
        Aff<C> p = load_affine<C>(bases + i * stride, stride);
        if (p.inf) continue;
        if (d < 0) { p.y = p.y.neg(); d = -d; }
        buckets[d - 1] = buckets[d - 1].add_affine(p);
    }
    Jac<C> run = Jac<C>::identity(), acc = Jac<C>::identity();
    for (size_t b = buckets.size(); b-- > 0;) { run = run.add(buckets[b]); acc = acc.add(run); }
    return acc;
}

// Parallel Pippenger: the pairs are cut into chunks, every (chunk, window) is an independent task pulled
// from a shared counter -- arkworks itself only parallelises over the ~16 windows, which stops
// scaling at 16 cores.  Round 5: the number of chunks minimises the MAKESPAN -- rounds of `nthreads`
// tasks times the cost of one -- (round 4 made nchunks = threads / 16, i.e. ~1.1 tasks per thread at 256 threads: the
// second round of 33 tasks ran on an otherwise idle machine), and the window width comes from a cost model of the CHUNK,
// not from log2(chunk) - 3: per window a task pays `chunk` mixed additions (11 field products each) and 2 full additions
// (16 each) per bucket.
static double msm_task_cost(size_t chunk, unsigned c) {   // field products of one (chunk, window) task
    return (double)chunk * 11.0 + 2.0 * 16.0 * (double)((size_t)1 << (c - 1));
}
static unsigned msm_window_bits(size_t chunk) {
    double best = 1e300;
    unsigned bc = 3;
    for (unsigned c = 3; c <= 20; ++c) {
        const double cost = (double)((256 + c - 1) / c + 1) * msm_task_cost(chunk, c);
        if (cost < best) { best = cost; bc = c; }
    }
    return bc;
}
// chunks that minimise the MAKESPAN: rounds of `nthreads` tasks times the cost of a task
static size_t msm_chunks(size_t len, int nthreads) {
    if (nthreads <= 1 || len < 4096) return 1;
    double best = 1e300;
    size_t bn = 1;
    for (size_t nch = 1; nch <= (size_t)nthreads * 8 && len / nch >= 1024; ++nch) {
        const size_t chunk = (len + nch - 1) / nch;
        const unsigned c = msm_window_bits(chunk), nwin = (256 + c - 1) / c + 1;
        const size_t tasks = nch * nwin, rounds = (tasks + nthreads - 1) / nthreads;
        const double cost = (double)rounds * msm_task_cost(chunk, c);
        if (cost < best * 0.999) { best = cost; bn = nch; }
    }
    return bn;
}

template <class C>
static Jac<C> msm_parallel(const uint8_t *bases, size_t stride, const u64 *scalars, size_t len, int nthreads) {
    typedef typename C::Fr Fr;
    if (len == 0) return Jac<C>::identity();
    if (nthreads < 1) nthreads = 1;
    const size_t nchunks = msm_chunks(len, nthreads);
    size_t chunk = (len + nchunks - 1) / nchunks;
    const unsigned c = chunk >= 32 ? msm_window_bits(chunk) : 3;
    const unsigned nwin = (256 + c - 1) / c + 1;
    std::vector<Fr> canon(len);
    std::vector<unsigned char> carries((size_t)nwin * len, 0);   // carry INTO window w of scalar i
    parallel_for(len, nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) {
            canon[i] = Fr::from_raw(scalars + 4 * i).from_mont();
            unsigned carry = 0;
            for (unsigned w = 0; w < nwin; ++w) {
                carries[(size_t)w * len + i] = (unsigned char)carry;
                int d = (int)get_bits(canon[i].l, w * c, c) + (int)carry;
                carry = d > (1 << (c - 1)) ? 1 : 0;
            }
        }
    });
    std::vector<Jac<C>> wsum(nchunks * nwin, Jac<C>::identity());
    std::atomic<size_t> next(0);
    auto worker = [&]() {
        for (;;) {
            size_t t = next.fetch_add(1);
            if (t >= nchunks * nwin) break;
            size_t ch = t / nwin;
            unsigned w = (unsigned)(t % nwin);
            size_t lo = ch * chunk, hi = std::min(len, lo + chunk);
            if (lo >= hi) continue;
            wsum[t] = msm_window<C>(bases + lo * stride, stride, canon.data() + lo, carries.data() + (size_t)w * len + lo, hi - lo, w, c);
        }
    };
    if (nthreads == 1) worker();
    else {
        std::vector<std::thread> th;
        const size_t nt = std::min((size_t)nthreads, nchunks * nwin);
        for (size_t t = 0; t < nt; ++t) th.emplace_back(worker);
        for (auto &x : th) x.join();
    }
    // Horner per chunk (c doublings per window: ~256 per chunk), the chunks in parallel, then their sum
    std::vector<Jac<C>> csum(nchunks, Jac<C>::identity());
    parallel_for(nchunks, nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t ch = lo; ch < hi; ++ch) {
            Jac<C> acc = Jac<C>::identity();
            for (unsigned w = nwin; w-- > 0;) {
                for (unsigned k = 0; k < c; ++k) acc = acc.dbl();
                acc = acc.add(wsum[ch * nwin + w]);
            }
            csum[ch] = acc;
        }
    });
    Jac<C> total = Jac<C>::identity();
    for (size_t ch = 0; ch < nchunks; ++ch) total = total.add(csum[ch]);
    return total;
}

// ----------------------------------------------------------------------------------- NTT
// ark-poly Radix2EvaluationDomain::{fft_in_place, ifft_in_place}: natural order in/out,
// omega = two_adic_root^(2^(s - log n)), ifft multiplies by n^{-1}.
template <class C>
static int ntt_inplace(typename C::Fr *a, unsigned log_n, bool inverse, int nthreads) {
    typedef typename C::Fr Fr;
    if (log_n > C::TWO_ADICITY) return 3;
    size_t n = (size_t)1 << log_n;
    if (n == 1) return 0;
    Fr w = CurveCtx<C>::get().root_of_unity(log_n);
    if (inverse) w = w.inverse();
    parallel_for(n, nthreads, [&](size_t lo, size_t hi, int) {   // bit reversal: the pairs (i, j) with i < j are disjoint
        for (size_t i = lo; i < hi; ++i) {
            size_t j = 0;
            for (unsigned b = 0; b < log_n; ++b) j |= ((i >> b) & 1) << (log_n - 1 - b);
            if (i < j) std::swap(a[i], a[j]);
        }
    });
    std::vector<Fr> tw(n / 2);
    parallel_for(n / 2, nthreads, [&](size_t lo, size_t hi, int) {   // every range starts from its own power of w
        Fr cur = w.pow_u64((u64)lo);
        for (size_t i = lo; i < hi; ++i) { tw[i] = cur; cur = cur * w; }
    });
    for (unsigned s = 1; s <= log_n; ++s) {
        size_t m = (size_t)1 << s, half = m >> 1, step = n >> s;
        parallel_for(n / 2, nthreads, [&](size_t lo, size_t hi, int) {
            for (size_t idx = lo; idx < hi; ++idx) {
                size_t blk = idx / half, j = idx % half;
                size_t i0 = blk * m + j, i1 = i0 + half;
                Fr t = a[i1] * tw[j * step];
                Fr u = a[i0];
                a[i0] = u + t;
                a[i1] = u - t;
            }
        });
    }
    if (inverse) {
        Fr ninv = Fr::from_u64((u64)n).inverse();
        parallel_for(n, nthreads, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) a[i] = a[i] * ninv; });
    }
    return 0;
}

// ---------------------------------------------------------------------------- R1CS / SAP
struct Csr {
    u64 nrows;
    std::vector<u64> rowptr;
    std::vector<uint32_t> col;
    std::vector<u64> val;  // 4 limbs each
};
struct po_csr {
    u64 nrows;
    const u64 *rowptr;
    const uint32_t *col;
    const u64 *val;
};
// m_at (common.rs:100-105) returns the FIRST entry of a row with the requested column, so later
// duplicates of a column inside one row are invisible to the reference: drop them here.
static Csr load_csr(const po_csr *m) {
    Csr c;
    c.nrows = m->nrows;
    c.rowptr.assign(1, 0);
    for (u64 r = 0; r < m->nrows; ++r) {
        size_t start = c.col.size();
        for (u64 k = m->rowptr[r]; k < m->rowptr[r + 1]; ++k) {
            bool dup = false;
            for (size_t q = start; q < c.col.size(); ++q) if (c.col[q] == m->col[k]) { dup = true; break; }
            if (dup) continue;
            c.col.push_back(m->col[k]);
            for (int i = 0; i < 4; ++i) c.val.push_back(m->val[4 * k + i]);
        }
        c.rowptr.push_back(c.col.size());
    }
    return c;
}

template <class C>
struct Pk {
    typedef typename C::Fr Fr;
    u64 n, m0, mw, nr, sigma;
    unsigned log_n;
    Fr omega;
    Csr A, B, Cm;
    std::vector<Aff<C>> bases[6];
    // per-proof state (one proof in flight per pk handle in the oracle; fine for tests)
    std::vector<Fr> x, u_evals, w_evals, u, w, u2, h, wit_u, z_tail, quotient;
    Fr r_a[2];
    int phase = 0;
    int nthreads = 1;   // of the last phase-1 call: phase 2 has no thread argument in the po_* interface
};

template <class C>
static typename C::Fr row_dot(const Csr &m, u64 r, const std::vector<typename C::Fr> &z) {
    typedef typename C::Fr Fr;
    Fr acc = Fr::zero();
    for (u64 k = m.rowptr[r]; k < m.rowptr[r + 1]; ++k) acc = acc + Fr::from_raw(&m.val[4 * k]) * z[m.col[k]];
    return acc;
}

// Closed-form R1CS -> SAP witness map (equals U.z and W.z of common.rs:138-207
// and y of prover.rs:279-302).  Outputs u_evals[n], w_evals[n], z_tail = (x || w || y).
template <class C>
static void witness_map(const Pk<C> &pk, const std::vector<typename C::Fr> &x, const std::vector<typename C::Fr> &w,
                        std::vector<typename C::Fr> &ue, std::vector<typename C::Fr> &we,
                        std::vector<typename C::Fr> &z_tail, int nthreads) {
    typedef typename C::Fr Fr;
    u64 n = pk.n, m0 = pk.m0, mw = pk.mw, nr = pk.nr;
    std::vector<Fr> xw(x);
    xw.insert(xw.end(), w.begin(), w.end());
    ue.assign(n, Fr::zero());
    we.assign(n, Fr::zero());
    z_tail.assign(m0 + mw + m0 + nr, Fr::zero());
    Fr one = Fr::one(), two = one + one, four = two + two;
    for (u64 i = 0; i < m0 + mw; ++i) z_tail[i] = xw[i];
    // y = (0, (1-x_j)^2, ((A-B) xw)_i^2)    prover.rs:279-302
    Fr *y = &z_tail[m0 + mw];
    y[0] = Fr::zero();
    for (u64 j = 1; j < m0; ++j) y[j] = (one - x[j]).sqr();
    ue[0] = two;
    we[0] = four;
    for (u64 i = 1; i < m0; ++i) { ue[i] = one + x[i]; we[i] = four * x[i] + y[i]; }
    for (u64 i = 1; i < m0; ++i) { ue[m0 + i] = one - x[i]; we[m0 + i] = y[i]; }
    parallel_for(nr, nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t r = lo; r < hi; ++r) {
            Fr az = row_dot<C>(pk.A, r, xw), bz = row_dot<C>(pk.B, r, xw), cz = row_dot<C>(pk.Cm, r, xw);
            Fr d = az - bz, d2 = d.sqr();
            y[m0 + r] = d2;
            ue[2 * m0 + r] = az + bz;
            we[2 * m0 + r] = four * cz + d2;
            ue[2 * m0 + nr + r] = d;
            we[2 * m0 + nr + r] = d2;
        }
    });
}

// ---------------------------------------------------------------------------------- setup
template <class C>
struct FixedBase {  // 8-bit windows of the generator, affine
    std::vector<Aff<C>> table;  // [32][255]
    FixedBase() {
        CurveCtx<C> &cc = CurveCtx<C>::get();
        std::vector<Jac<C>> t(32 * 255);
        Aff<C> g{cc.gx, cc.gy, false};
        Jac<C> base{g.x, g.y, C::Fq::one()};
        for (int w = 0; w < 32; ++w) {
            Jac<C> acc = Jac<C>::identity();
            for (int d = 0; d < 255; ++d) { acc = acc.add(base); t[w * 255 + d] = acc; }
            for (int k = 0; k < 8; ++k) base = base.dbl();
        }
        table.resize(t.size());
        batch_to_affine<C>(t.data(), table.data(), t.size());
    }
    Jac<C> mul(const typename C::Fr &k_mont) const {
        typename C::Fr k = k_mont.from_mont();
        Jac<C> acc = Jac<C>::identity();
        for (int w = 0; w < 32; ++w) {
            unsigned d = (unsigned)(k.l[w / 8] >> (8 * (w % 8))) & 0xff;
            if (d) acc = acc.add_affine(table[w * 255 + d - 1]);
        }
        return acc;
    }
};

template <class C>
static void fixed_base_batch(const std::vector<typename C::Fr> &scalars, std::vector<Aff<C>> &out, int nthreads) {
    static FixedBase<C> fb;
    out.resize(scalars.size());
    parallel_for(scalars.size(), nthreads, [&](size_t lo, size_t hi, int) {
        const size_t CH = 1024;
        std::vector<Jac<C>> tmp(CH);
        for (size_t s = lo; s < hi; s += CH) {
            size_t e = std::min(hi, s + CH);
            for (size_t i = s; i < e; ++i) tmp[i - s] = fb.mul(scalars[i]);
            batch_to_affine<C>(tmp.data(), &out[s], e - s);
        }
    });
}

// generate_proving_key (generator.rs:24-167) with the trapdoors x, z supplied (the reference
// draws them at :72 and :77).  uj_wj_lcs (:112-136) is computed sparsely:
//   column j' of z_tail (j = j' + m0):
//     j' <  m0+mw  (R1CS column k=j'):  u = sum_r A[r][k](L1+L2) + B[r][k](L1-L2),
//                                       w = 4 sum_r C[r][k] L1  (+ 4 L[i] if k = i < m0)
//                                       with L1 = L[2m0+r], L2 = L[2m0+nr+r]
//     j' = m0+mw+t (y column t):        u = 0, w = L[t] + L[t+m0] (t < m0),
//                                       w = L[2m0+r] + L[2m0+nr+r]  (t = m0 + r)
template <class C>
static Pk<C> *setup(u64 m0, u64 mw, u64 nr, const po_csr *a, const po_csr *b, const po_csr *c,
                    const u64 *x_trap, const u64 *z_trap, int nthreads, int *status) {
    typedef typename C::Fr Fr;
    CurveCtx<C> &cc = CurveCtx<C>::get();
    auto pk = std::make_unique<Pk<C>>();
    pk->m0 = m0; pk->mw = mw; pk->nr = nr;
    pk->A = load_csr(a); pk->B = load_csr(b); pk->Cm = load_csr(c);
    u64 rows = 2 * (m0 + nr), cols = 3 * m0 + mw + nr;           // common.rs:131-135
    u64 n = 1; unsigned log_n = 0;
    while (n < rows) { n <<= 1; ++log_n; }                        // Radix2EvaluationDomain::new
    if (log_n > C::TWO_ADICITY) { *status = 3; return nullptr; }
    pk->n = n; pk->log_n = log_n; pk->sigma = n + 3;              // generator.rs:66-70
    pk->omega = cc.root_of_unity(log_n);
    if (!x_trap) return pk.release();                             // matrices-only handle (pk_load path)
    Fr x = Fr::from_raw(x_trap), z = Fr::from_raw(z_trap);
    u64 sigma = pk->sigma;
    Fr y = x.pow_u64(sigma), yinv = y.inverse();
    Fr y_alpha = yinv.pow_u64(3), y_to_minus_alpha = y.pow_u64(3), y_gamma = yinv.pow_u64(5);
    auto powers = [&](u64 count, Fr scale) {
        std::vector<Fr> s(count);
        Fr acc = scale;
        for (u64 j = 0; j < count; ++j) { s[j] = acc; acc = acc * x; }
        return s;
    };
    fixed_base_batch<C>(powers(n + 1, Fr::one()), pk->bases[0], nthreads);           // :82
    fixed_base_batch<C>(powers(3, y_alpha), pk->bases[1], nthreads);                  // :86
    fixed_base_batch<C>(powers(2, y_gamma), pk->bases[2], nthreads);                  // :90
    u64 dmax = 2 * (n - 1) + sigma * 8;                                               // :95-96
    fixed_base_batch<C>(powers(dmax + 1, y_gamma * z), pk->bases[3], nthreads);       // :97-99
    Fr zh = x.pow_u64(n) - Fr::one();                                                 // :106
    fixed_base_batch<C>(powers(n - 1, zh * y_to_minus_alpha), pk->bases[4], nthreads);// :107
    // Lagrange coefficients L_i(x) = zh/n * w^i / (x - w^i)    (:113)
    std::vector<Fr> L(n), den(n), scratch(n);
    {
        Fr wi = Fr::one();
        for (u64 i = 0; i < n; ++i) { den[i] = x - wi; L[i] = wi; wi = wi * pk->omega; }
        batch_inverse(den.data(), n, scratch.data());
        Fr k = zh * Fr::from_u64(n).inverse();
        for (u64 i = 0; i < n; ++i) L[i] = L[i] * den[i] * k;
    }
    u64 mcols = m0 + mw;
    std::vector<Fr> ucol(mcols, Fr::zero()), wcol(cols - m0, Fr::zero());
    Fr four = Fr::from_u64(4);
    for (u64 r = 0; r < nr; ++r) {
        Fr L1 = L[2 * m0 + r], L2 = L[2 * m0 + nr + r], sp = L1 + L2, sm = L1 - L2, L1x4 = four * L1;
        for (u64 k = pk->A.rowptr[r]; k < pk->A.rowptr[r + 1]; ++k)
            ucol[pk->A.col[k]] = ucol[pk->A.col[k]] + Fr::from_raw(&pk->A.val[4 * k]) * sp;
        for (u64 k = pk->B.rowptr[r]; k < pk->B.rowptr[r + 1]; ++k)
            ucol[pk->B.col[k]] = ucol[pk->B.col[k]] + Fr::from_raw(&pk->B.val[4 * k]) * sm;
        for (u64 k = pk->Cm.rowptr[r]; k < pk->Cm.rowptr[r + 1]; ++k)
            wcol[pk->Cm.col[k]] = wcol[pk->Cm.col[k]] + Fr::from_raw(&pk->Cm.val[4 * k]) * L1x4;
        wcol[mcols + m0 + r] = sp;
    }
    for (u64 i = 0; i < m0; ++i) {
        wcol[i] = wcol[i] + four * L[i];
        wcol[mcols + i] = L[i] + L[i + m0];
    }
    std::vector<Fr> lcs(cols - m0);
    for (u64 j = 0; j < cols - m0; ++j) {
        Fr u = j < mcols ? ucol[j] : Fr::zero();
        lcs[j] = (u * y_gamma + wcol[j]) * y_to_minus_alpha;                          // :134
    }
    fixed_base_batch<C>(lcs, pk->bases[5], nthreads);
    return pk.release();
}

// ---------------------------------------------------------------------------------- prove
template <class C>
static Aff<C> msm_vec(const std::vector<Aff<C>> &bases, const std::vector<typename C::Fr> &sc, int nthreads, int *status) {
    if (sc.size() > bases.size()) { *status = 2; return Aff<C>{C::Fq::zero(), C::Fq::zero(), true}; }  // prover.rs:381
    return msm_parallel<C>((const uint8_t *)bases.data(), sizeof(Aff<C>), (const u64 *)sc.data(), sc.size(), nthreads).to_affine();
}
template <class C>
static Aff<C> aff_add(const Aff<C> &a, const Aff<C> &b) {
    Jac<C> j = Jac<C>::identity();
    return j.add_affine(a).add_affine(b).to_affine();
}

template <class C>
static int prove_phase1(Pk<C> &pk, const u64 *x_in, const u64 *w_in, const u64 *r_a, int nthreads, Aff<C> &a_g1, Aff<C> &c_g1) {
    typedef typename C::Fr Fr;
    u64 n = pk.n, m0 = pk.m0;
    std::vector<Fr> x(m0), w(pk.mw);
    for (u64 i = 0; i < m0; ++i) x[i] = Fr::from_raw(x_in + 4 * i);
    for (u64 i = 0; i < pk.mw; ++i) w[i] = Fr::from_raw(w_in + 4 * i);
    pk.x = x;
    pk.r_a[0] = Fr::from_raw(r_a);
    pk.r_a[1] = Fr::from_raw(r_a + 4);
    pk.nthreads = nthreads;
    StageClock clk;
    witness_map<C>(pk, x, w, pk.u_evals, pk.w_evals, pk.z_tail, nthreads);          // prover.rs:75-96
    clk.mark("witness map");
    pk.u = pk.u_evals;
    pk.w = pk.w_evals;
    ntt_inplace<C>(pk.u.data(), pk.log_n, true, nthreads);                           // :94
    ntt_inplace<C>(pk.w.data(), pk.log_n, true, nthreads);                           // :96
    clk.mark("2 inverse transforms (n)");
    if (pk.log_n + 1 > C::TWO_ADICITY) return 3;                                     // :317
    pk.u2.assign(2 * n, Fr::zero());                                                 // square_polynomial :315-328
    std::copy(pk.u.begin(), pk.u.end(), pk.u2.begin());
    ntt_inplace<C>(pk.u2.data(), pk.log_n + 1, false, nthreads);
    for (auto &e : pk.u2) e = e.sqr();
    ntt_inplace<C>(pk.u2.data(), pk.log_n + 1, true, nthreads);
    clk.mark("square: 2 transforms (2n)");
    // h_num = u^2 - w ; divide by X^n - 1    :104-108
    pk.h.assign(n, Fr::zero());
    bool h_nonzero = false;
    for (u64 i = 0; i < n; ++i) {
        Fr lo = pk.u2[i] - pk.w[i], hi = pk.u2[n + i];
        if (!(lo + hi).is_zero()) return 4;                                          // :108
        pk.h[i] = hi;
        h_nonzero |= !hi.is_zero();
    }
    if (!h_nonzero || !pk.h[n - 1].is_zero()) return 5;                              // :107 (deg h <= n-2)
    // witness-only U part (prover.rs:156-162): columns >= m0 -> rows < 2 m0 vanish
    pk.wit_u = pk.u_evals;
    for (u64 i = 0; i < 2 * m0; ++i) pk.wit_u[i] = Fr::zero();
    ntt_inplace<C>(pk.wit_u.data(), pk.log_n, true, nthreads);
    clk.mark("h + witness-u transform");
    int st = 0;
    Fr two = Fr::from_u64(2);
    // compute_a_g1 :330-338
    std::vector<Fr> ra{pk.r_a[0], pk.r_a[1]};
    a_g1 = aff_add<C>(msm_vec<C>(pk.bases[0], pk.u, nthreads, &st), msm_vec<C>(pk.bases[1], ra, nthreads, &st));
    // compute_r_g1 :340-357
    std::vector<Fr> two_ra_u(n + 1, Fr::zero());
    for (u64 k = 0; k < n; ++k) {
        two_ra_u[k] = two_ra_u[k] + two * pk.r_a[0] * pk.u[k];
        two_ra_u[k + 1] = two_ra_u[k + 1] + two * pk.r_a[1] * pk.u[k];
    }
    clk.mark("[a]: msm(n)");
    std::vector<Fr> ra_sq{pk.r_a[0].sqr(), two * pk.r_a[0] * pk.r_a[1], pk.r_a[1].sqr()};
    Aff<C> r_g1 = aff_add<C>(aff_add<C>(msm_vec<C>(pk.bases[0], two_ra_u, nthreads, &st), msm_vec<C>(pk.bases[1], ra_sq, nthreads, &st)),
                             msm_vec<C>(pk.bases[2], ra, nthreads, &st));
    clk.mark("r: msm(n+1)");
    std::vector<Fr> hc(pk.h.begin(), pk.h.begin() + (n - 1));
    Aff<C> h_g1 = msm_vec<C>(pk.bases[4], hc, nthreads, &st);                        // :118
    clk.mark("h: msm(n-1)");
    Aff<C> lcs_g1 = msm_vec<C>(pk.bases[5], pk.z_tail, nthreads, &st);               // :120-121
    clk.mark("lcs: msm(M - m0)");
    c_g1 = aff_add<C>(aff_add<C>(lcs_g1, h_g1), r_g1);                               // :123
    if (st) return st;
    pk.phase = 1;
    return 0;
}

template <class C>
static int prove_phase2(Pk<C> &pk, const u64 *x1_in, u64 *out) {
    typedef typename C::Fr Fr;
    if (pk.phase < 1) return 8;
    Fr x1 = Fr::from_raw(x1_in), acc = Fr::zero();
    {   // u_poly.evaluate(&x1) :132 -- Horner per range, the ranges' values combined with x1^(range start)
        const int nt = pk.nthreads > 0 ? pk.nthreads : 1;
        std::vector<Fr> part((size_t)nt, Fr::zero());
        parallel_for(pk.n, nt, [&](size_t lo, size_t hi, int t) {
            Fr v = Fr::zero();
            for (size_t k = hi; k-- > lo;) v = v * x1 + pk.u[k];
            part[t] = v * x1.pow_u64((u64)lo);
        });
        for (const Fr &v : part) acc = acc + v;
    }
    acc.store(out);
    pk.phase = 2;
    return 0;
}

template <class C>
static int prove_phase3(Pk<C> &pk, const u64 *x1_in, const u64 *x2_in, const u64 *a_in, const u64 *c_in, int nthreads, Aff<C> &d_g1) {
    typedef typename C::Fr Fr;
    if (pk.phase < 1) return 8;
    u64 n = pk.n, sigma = pk.sigma;
    Fr x1 = Fr::from_raw(x1_in), x2 = Fr::from_raw(x2_in), a_at = Fr::from_raw(a_in), c_at = Fr::from_raw(c_in);
    Fr two = Fr::from_u64(2);
    u64 len = 8 * sigma + 2 * n - 1;
    StageClock clk;
    // Numerator N(X) = A(X) Y^-gamma + x2 C(X) Y^-gamma - (A(x1) + x2 C(x1)) Y^-gamma, coefficient by coefficient (round 5: one
    // parallel pass instead of the serial block-by-block assembly; the same sums in the same order per coefficient):
    //   A(X) Y^-gamma :145-152      u at 5 sigma, r_a at 2 sigma
    //   R(X) Y^-gamma :359-377      2 r_a u at 5 sigma (degree n), r_a^2 at 2 sigma, r_a at 0
    //   witness parts :160-175      wit_u at 3 sigma, w at 8 sigma  (the W witness part equals w itself: N6 == N2)
    //   h_numerator   :177-180      u^2 - w at 8 sigma (2n - 1 coefficients)
    std::vector<Fr> num(len);
    const Fr two_r0 = two * pk.r_a[0], two_r1 = two * pk.r_a[1];
    parallel_for(len, nthreads, [&](size_t lo, size_t hi, int) {
        for (u64 k = lo; k < hi; ++k) {
            Fr a = Fr::zero(), cc = Fr::zero();
            if (k >= 5 * sigma && k < 5 * sigma + n) a = pk.u[k - 5 * sigma];
            if (k == 2 * sigma) a = a + pk.r_a[0];
            if (k == 2 * sigma + 1) a = a + pk.r_a[1];
            if (k >= 5 * sigma && k <= 5 * sigma + n) {
                const u64 j = k - 5 * sigma;
                if (j < n) cc = cc + two_r0 * pk.u[j];
                if (j >= 1) cc = cc + two_r1 * pk.u[j - 1];
            }
            if (k == 2 * sigma) cc = cc + pk.r_a[0].sqr();
            if (k == 2 * sigma + 1) cc = cc + two * pk.r_a[0] * pk.r_a[1];
            if (k == 2 * sigma + 2) cc = cc + pk.r_a[1].sqr();
            if (k == 0) cc = cc + pk.r_a[0];
            if (k == 1) cc = cc + pk.r_a[1];
            if (k >= 3 * sigma && k < 3 * sigma + n) cc = cc + pk.wit_u[k - 3 * sigma];
            if (k >= 8 * sigma && k < 8 * sigma + n) cc = cc + pk.w[k - 8 * sigma];
            if (k >= 8 * sigma && k < 8 * sigma + 2 * n - 1) {
                const u64 j = k - 8 * sigma;
                cc = cc + (pk.u2[j] - (j < n ? pk.w[j] : Fr::zero()));
            }
            num[k] = a + x2 * cc;                                                    // :211-216
        }
    });
    num[5 * sigma] = num[5 * sigma] - a_at - x2 * c_at;
    clk.mark("numerator assembly");
    // synthetic division by (X - x1) :217-220: carry_k = num_k + x1 carry_{k+1}.  Ranges in parallel: each first folds its
    // coefficients alone (incoming carry 0), the range values are chained serially with x1^(range length), then every range
    // re-walks with its true incoming carry.  Field arithmetic is exact, so the values equal the serial recurrence's.
    pk.quotient.assign(len - 1, Fr::zero());
    Fr carry = Fr::zero();
    {
        const int nt = nthreads > 1 ? nthreads : 1;
        const u64 cnt = len - 1, per = (cnt + nt - 1) / nt;                         // indices 1 .. len - 1
        std::vector<Fr> fold((size_t)nt, Fr::zero()), incoming((size_t)nt, Fr::zero());
        parallel_for(cnt, nt, [&](size_t lo, size_t hi, int t) {
            Fr v = Fr::zero();
            for (u64 k = hi; k > lo; --k) v = num[k] + x1 * v;                      // coefficients lo + 1 .. hi
            fold[t] = v;
        });
        {   // chain from the top range down: carry into range t = value of everything above it
            Fr c_in = Fr::zero();
            for (int t = nt - 1; t >= 0; --t) {
                const u64 lo = (u64)t * per, hi = std::min(cnt, lo + per);
                if (lo >= hi) continue;
                incoming[t] = c_in;
                c_in = fold[t] + x1.pow_u64(hi - lo) * c_in;
            }
            carry = c_in;
        }
        parallel_for(cnt, nt, [&](size_t lo, size_t hi, int t) {
            Fr v = incoming[t];
            for (u64 k = hi; k > lo; --k) { v = num[k] + x1 * v; pk.quotient[k - 1] = v; }
        });
    }
    Fr rem = num[0] + x1 * carry;
    if (!rem.is_zero()) return 4;                                                    // :221
    int st = 0;
    clk.mark("division by (X - x1)");
    d_g1 = msm_vec<C>(pk.bases[3], pk.quotient, nthreads, &st);                      // :229
    clk.mark("[d]: msm(10n+22)");
    pk.phase = 3;
    return st;
}

// ------------------------------------------------------------------------------ C exports
#define DISPATCH(curve, EXPR_BLS, EXPR_BN) ((curve) == 0 ? (EXPR_BLS) : (curve) == 1 ? (EXPR_BN) : 1)

template <class F>
static int field_op(int op, const u64 *a, const u64 *b, u64 *out) {
    F x = F::from_raw(a), y = b ? F::from_raw(b) : F::zero(), r;
    switch (op) {
        case 0: r = x * y; break;
        case 1: r = x + y; break;
        case 2: r = x - y; break;
        case 3: r = x.inverse(); break;
        case 4: r = x.to_mont(); break;
        case 5: r = x.from_mont(); break;
        case 6: r = x.sqr(); break;
        default: return 1;
    }
    r.store(out);
    return 0;
}

struct po_pk { int curve; void *impl; };

#define PO_API extern "C" __attribute__((visibility("default")))

PO_API int po_init(void) { CurveCtx<BLS>::get(); CurveCtx<BN>::get(); return 0; }

PO_API int po_fr_op(int curve, int op, const u64 *a, const u64 *b, u64 *out) {
    po_init();
    return DISPATCH(curve, field_op<BLS::Fr>(op, a, b, out), field_op<BN::Fr>(op, a, b, out));
}
PO_API int po_fq_op(int curve, int op, const u64 *a, const u64 *b, u64 *out) {
    po_init();
    return DISPATCH(curve, field_op<BLS::Fq>(op, a, b, out), field_op<BN::Fq>(op, a, b, out));
}
PO_API int po_fr_root_of_unity(int curve, unsigned log_n, u64 *out) {
    po_init();
    if (curve == 0) CurveCtx<BLS>::get().root_of_unity(log_n).store(out);
    else CurveCtx<BN>::get().root_of_unity(log_n).store(out);
    return 0;
}
PO_API int po_g1_generator(int curve, u64 *out_xy) {
    po_init();
    if (curve == 0) { auto &c = CurveCtx<BLS>::get(); c.gx.store(out_xy); c.gy.store(out_xy + 6); }
    else { auto &c = CurveCtx<BN>::get(); c.gx.store(out_xy); c.gy.store(out_xy + 4); }
    return 0;
}
template <class C>
static int g1_mul_impl(const u64 *base_xy, const u64 *k, u64 *out_xy, int *out_inf) {
    Aff<C> p = load_affine<C>((const uint8_t *)base_xy, 16 * C::Fq::LIMBS);
    store_affine<C>(g1_mul<C>(p, C::Fr::from_raw(k)).to_affine(), out_xy, out_inf);
    return 0;
}
PO_API int po_g1_mul(int curve, const u64 *base_xy, const u64 *scalar, u64 *out_xy, int *out_inf) {
    po_init();
    return DISPATCH(curve, g1_mul_impl<BLS>(base_xy, scalar, out_xy, out_inf), g1_mul_impl<BN>(base_xy, scalar, out_xy, out_inf));
}
template <class C>
static int g1_sum_impl(const u64 *pts, const int *infs, size_t count, u64 *out_xy, int *out_inf) {
    Jac<C> acc = Jac<C>::identity();
    for (size_t i = 0; i < count; ++i) {
        Aff<C> p = load_affine<C>((const uint8_t *)(pts + i * 2 * C::Fq::LIMBS), 16 * C::Fq::LIMBS);
        if (infs && infs[i]) p.inf = true;
        acc = acc.add_affine(p);
    }
    store_affine<C>(acc.to_affine(), out_xy, out_inf);
    return 0;
}
PO_API int po_g1_sum(int curve, const u64 *pts, const int *infs, size_t count, u64 *out_xy, int *out_inf) {
    po_init();
    return DISPATCH(curve, g1_sum_impl<BLS>(pts, infs, count, out_xy, out_inf), g1_sum_impl<BN>(pts, infs, count, out_xy, out_inf));
}
template <class C>
static int g1_on_curve_impl(const u64 *xy) {
    Aff<C> p = load_affine<C>((const uint8_t *)xy, 16 * C::Fq::LIMBS);
    if (p.inf) return 1;
    return (p.y.sqr() == p.x.sqr() * p.x + CurveCtx<C>::get().b) ? 1 : 0;
}
PO_API int po_g1_is_on_curve(int curve, const u64 *xy) {
    po_init();
    return curve == 0 ? g1_on_curve_impl<BLS>(xy) : g1_on_curve_impl<BN>(xy);
}

PO_API int po_ntt(int curve, u64 *data, unsigned log_n, int inverse, int nthreads) {
    po_init();
    return DISPATCH(curve, ntt_inplace<BLS>((BLS::Fr *)data, log_n, inverse != 0, nthreads),
                    ntt_inplace<BN>((BN::Fr *)data, log_n, inverse != 0, nthreads));
}

template <class C>
static int msm_impl(const void *bases, size_t stride, const u64 *scalars, size_t len, int nthreads, u64 *out_xy, int *out_inf) {
    store_affine<C>(msm_parallel<C>((const uint8_t *)bases, stride, scalars, len, nthreads).to_affine(), out_xy, out_inf);
    return 0;
}
PO_API int po_msm_g1(int curve, const void *bases, size_t stride, const u64 *scalars, size_t len, int nthreads, u64 *out_xy, int *out_inf) {
    po_init();
    return DISPATCH(curve, msm_impl<BLS>(bases, stride, scalars, len, nthreads, out_xy, out_inf),
                    msm_impl<BN>(bases, stride, scalars, len, nthreads, out_xy, out_inf));
}

// P_i = (i+1) * G, affine, by running addition (SURVEY.md §8d MSM micro-inputs)
template <class C>
static int multiples_impl(size_t len, u64 *out_xy) {
    CurveCtx<C> &cc = CurveCtx<C>::get();
    Aff<C> g{cc.gx, cc.gy, false};
    const size_t CH = 4096;
    std::vector<Jac<C>> tmp(CH);
    std::vector<Aff<C>> aff(CH);
    Jac<C> acc = Jac<C>::identity();
    for (size_t s = 0; s < len; s += CH) {
        size_t e = std::min(len, s + CH);
        for (size_t i = s; i < e; ++i) { acc = acc.add_affine(g); tmp[i - s] = acc; }
        batch_to_affine<C>(tmp.data(), aff.data(), e - s);
        for (size_t i = s; i < e; ++i) store_affine<C>(aff[i - s], out_xy + i * 2 * C::Fq::LIMBS, nullptr);
    }
    return 0;
}
PO_API int po_g1_multiples(int curve, size_t len, u64 *out_xy) {
    po_init();
    return DISPATCH(curve, multiples_impl<BLS>(len, out_xy), multiples_impl<BN>(len, out_xy));
}

PO_API po_pk *po_pk_generate(int curve, u64 m0, u64 mw, u64 nr, const po_csr *a, const po_csr *b, const po_csr *c,
                      const u64 *x_trap, const u64 *z_trap, int nthreads, int *status) {
    po_init();
    int st = 0;
    void *impl = curve == 0 ? (void *)setup<BLS>(m0, mw, nr, a, b, c, x_trap, z_trap, nthreads, &st)
                            : (void *)setup<BN>(m0, mw, nr, a, b, c, x_trap, z_trap, nthreads, &st);
    if (status) *status = st;
    if (!impl) return nullptr;
    return new po_pk{curve, impl};
}
PO_API void po_pk_free(po_pk *pk) {
    if (!pk) return;
    if (pk->curve == 0) delete (Pk<BLS> *)pk->impl; else delete (Pk<BN> *)pk->impl;
    delete pk;
}
template <class C>
static int pk_info_impl(Pk<C> *p, u64 *n, u64 *m0, u64 *sigma, u64 *omega, u64 *lens) {
    *n = p->n; *m0 = p->m0; *sigma = p->sigma;
    p->omega.store(omega);
    for (int i = 0; i < 6; ++i) lens[i] = p->bases[i].size();
    return 0;
}
PO_API int po_pk_info(po_pk *pk, u64 *n, u64 *m0, u64 *sigma, u64 *omega, u64 *lens) {
    return pk->curve == 0 ? pk_info_impl((Pk<BLS> *)pk->impl, n, m0, sigma, omega, lens)
                          : pk_info_impl((Pk<BN> *)pk->impl, n, m0, sigma, omega, lens);
}
template <class C>
static int pk_export_impl(Pk<C> *p, int which, size_t off, size_t len, u64 *out) {
    if (which < 0 || which >= 6 || off + len > p->bases[which].size()) return 1;
    for (size_t i = 0; i < len; ++i) store_affine<C>(p->bases[which][off + i], out + i * 2 * C::Fq::LIMBS, nullptr);
    return 0;
}
PO_API int po_pk_export_bases(po_pk *pk, int which, size_t off, size_t len, u64 *out) {
    return pk->curve == 0 ? pk_export_impl((Pk<BLS> *)pk->impl, which, off, len, out)
                          : pk_export_impl((Pk<BN> *)pk->impl, which, off, len, out);
}
template <class C>
static int pk_import_impl(Pk<C> *p, int which, const void *pts, size_t stride, size_t len) {
    p->bases[which].resize(len);
    for (size_t i = 0; i < len; ++i) p->bases[which][i] = load_affine<C>((const uint8_t *)pts + i * stride, stride);
    return 0;
}
PO_API int po_pk_import_bases(po_pk *pk, int which, const void *pts, size_t stride, size_t len) {
    if (which < 0 || which >= 6) return 1;
    return pk->curve == 0 ? pk_import_impl((Pk<BLS> *)pk->impl, which, pts, stride, len)
                          : pk_import_impl((Pk<BN> *)pk->impl, which, pts, stride, len);
}

template <class C>
static int p1_impl(Pk<C> *p, const u64 *x, const u64 *w, const u64 *r_a, int nt, u64 *a_xy, int *a_inf, u64 *c_xy, int *c_inf) {
    Aff<C> a, c;
    int st = prove_phase1<C>(*p, x, w, r_a, nt, a, c);
    if (st) return st;
    store_affine<C>(a, a_xy, a_inf);
    store_affine<C>(c, c_xy, c_inf);
    return 0;
}
PO_API int po_prove_phase1(po_pk *pk, const u64 *x, const u64 *w, const u64 *r_a, int nthreads, u64 *a_xy, int *a_inf, u64 *c_xy, int *c_inf) {
    return pk->curve == 0 ? p1_impl((Pk<BLS> *)pk->impl, x, w, r_a, nthreads, a_xy, a_inf, c_xy, c_inf)
                          : p1_impl((Pk<BN> *)pk->impl, x, w, r_a, nthreads, a_xy, a_inf, c_xy, c_inf);
}
PO_API int po_prove_phase2(po_pk *pk, const u64 *x1, u64 *u_at_x1) {
    return pk->curve == 0 ? prove_phase2(*(Pk<BLS> *)pk->impl, x1, u_at_x1) : prove_phase2(*(Pk<BN> *)pk->impl, x1, u_at_x1);
}
template <class C>
static int p3_impl(Pk<C> *p, const u64 *x1, const u64 *x2, const u64 *a, const u64 *c, int nt, u64 *d_xy, int *d_inf) {
    Aff<C> d;
    int st = prove_phase3<C>(*p, x1, x2, a, c, nt, d);
    if (st) return st;
    store_affine<C>(d, d_xy, d_inf);
    return 0;
}
PO_API int po_prove_phase3(po_pk *pk, const u64 *x1, const u64 *x2, const u64 *a_at_x1, const u64 *c_at_x1, int nthreads, u64 *d_xy, int *d_inf) {
    return pk->curve == 0 ? p3_impl((Pk<BLS> *)pk->impl, x1, x2, a_at_x1, c_at_x1, nthreads, d_xy, d_inf)
                          : p3_impl((Pk<BN> *)pk->impl, x1, x2, a_at_x1, c_at_x1, nthreads, d_xy, d_inf);
}
template <class C>
static int tap_impl(Pk<C> *p, int which, u64 *out, size_t max_elems, size_t *n_elems) {
    const std::vector<typename C::Fr> *v = nullptr;
    switch (which) {
        case 0: v = &p->u_evals; break;
        case 1: v = &p->w_evals; break;
        case 2: v = &p->u; break;
        case 3: v = &p->w; break;
        case 4: v = &p->h; break;
        case 5: v = &p->wit_u; break;
        case 6: v = &p->z_tail; break;
        case 7: v = &p->quotient; break;
        default: return 1;
    }
    *n_elems = v->size();
    size_t k = std::min(max_elems, v->size());
    memcpy(out, v->data(), k * 32);
    return 0;
}
PO_API int po_prove_tap(po_pk *pk, int which, u64 *out, size_t max_elems, size_t *n_elems) {
    return pk->curve == 0 ? tap_impl((Pk<BLS> *)pk->impl, which, out, max_elems, n_elems)
                          : tap_impl((Pk<BN> *)pk->impl, which, out, max_elems, n_elems);
}

