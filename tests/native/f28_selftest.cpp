// Host build of the device field/EC templates (plain C++): checks the reduced-radix (28-bit limb)
// path of csrc/fq28.cuh against the dense 32-bit-limb path of csrc/field.cuh / ec.cuh, which the GPU
// parity tests pin against the oracle.  Built and run by tests/test_native_f28.py (CPU, no GPU).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../polymath_amd/csrc/fq28.cuh"

using namespace pm;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <class P>
static Fp<P> rand_fp() {
    Fp<P> r;
    for (int i = 0; i < P::N; ++i) r.l[i] = (uint32_t)next_u64();
    // squash below the modulus: clear top bits then one conditional subtraction via mul by one
    int top_bits = P::BITS - 32 * (P::N - 1);
    r.l[P::N - 1] &= (1u << (top_bits - 1)) - 1;
    return r;
}

template <class C>
static int run(const char *name, int iters) {
    typedef typename C::FqP Q;
    typedef typename C::FqRR RR;
    typedef Fp<Q> Fq;
    int fails = 0;
    // 1. products, including the extremes 0, 1, p-1
    std::vector<Fq> pool;
    pool.push_back(Fq::zero());
    pool.push_back(Fq::one());
    Fq pm1;
    for (int i = 0; i < Q::N; ++i) pm1.l[i] = Q::MOD[i];
    pm1.l[0] -= 1;
    pool.push_back(pm1);
    for (int i = 0; i < iters; ++i) pool.push_back(rand_fp<Q>());
    for (size_t i = 0; i + 1 < pool.size(); ++i) {
        Fq a = pool[i], b = pool[i + 1];
        Fq want = mul<Q>(a, b);
        F28<RR> fa = f28_unpack<RR>(fq_std_to_int<C>(a).l), fb = f28_unpack<RR>(fq_std_to_int<C>(b).l);
        Fq got = f28_to_std<RR>(f28_mul<RR>(fa, fb));
        if (!got.eq(want)) { if (fails++ < 5) printf("%s mul mismatch at %zu\n", name, i); }
        Fq gsq = f28_to_std<RR>(f28_sqr<RR>(f28_sub_k16<RR>(fa, fb)));      // loose input (e ~ 1.6), as in the mixed add
        Fq dsq = sqr<Q>(sub<Q>(a, b));
        if (!gsq.eq(dsq)) { if (fails++ < 5) printf("%s sqr mismatch at %zu\n", name, i); }
        // round trip of the radix conversions
        if (!fq_int_to_std<C>(fq_std_to_int<C>(a)).eq(a)) { if (fails++ < 5) printf("%s conv mismatch\n", name); }
        // lazy add/sub chain: (a + b) * (a - b) == a^2 - b^2
        F28<RR> s = f28_add<RR>(fa, fb), d = f28_sub_k4<RR>(fa, fb);
        Fq lhs = f28_to_std<RR>(f28_mul<RR>(s, d));
        Fq rhs = sub<Q>(sqr<Q>(a), sqr<Q>(b));
        if (!lhs.eq(rhs)) { if (fails++ < 5) printf("%s lazy mismatch at %zu\n", name, i); }
    }
    // 2. accumulate a chain of points both ways: P_k = (k+1) G built with the dense path
    Affine<C> g;
    for (int i = 0; i < Q::N; ++i) { g.x.l[i] = C::GX_MONT[i]; g.y.l[i] = C::GY_MONT[i]; }
    const int NP = 300;
    std::vector<Affine<C>> pts(NP);
    XYZZ<C> run_pt = XYZZ<C>::identity();
    for (int k = 0; k < NP; ++k) {
        xyzz_madd<C>(run_pt, g, false);
        pts[k] = xyzz_to_affine<C>(run_pt);
        if (!affine_on_curve<C>(pts[k])) { if (fails++ < 5) printf("%s off curve\n", name); }
    }
    XYZZ<C> dense = XYZZ<C>::identity();
    XYZZ28<C> rr;
    rr.X = rr.Y = rr.ZZ = rr.ZZZ = f28_zero<RR>();
    int exceptional = 0;
    for (int rep = 0; rep < 3; ++rep)
        for (int k = 0; k < NP; ++k) {
            bool negate = (next_u64() & 1) != 0;
            int idx = (rep == 2 && k == 5) ? 4 : k;       // force a repeated point (doubling) once
            const Affine<C> &p = pts[idx];
            xyzz_madd<C>(dense, p, negate);
            Affine<C> pi{fq_std_to_int<C>(p.x), fq_std_to_int<C>(p.y)};
            if (!xyzz28_madd<C>(rr, pi, negate)) {
                ++exceptional;
                XYZZ<C> tmp = xyzz28_to_std<C>(rr);
                xyzz_madd<C>(tmp, p, negate);
                rr = xyzz28_from_std<C>(tmp);
            }
        }
    Affine<C> a1 = xyzz_to_affine<C>(dense), a2 = xyzz_to_affine<C>(xyzz28_to_std<C>(rr));
    if (!(a1.x.eq(a2.x) && a1.y.eq(a2.y))) { fails++; printf("%s madd chain mismatch\n", name); }
    // 2b. full additions on F28 registers vs the dense add, through the internal-form memory records
    {
        XYZZ<C> dsum = XYZZ<C>::identity();
        XYZZ28<C> rsum;
        rsum.X = rsum.Y = rsum.ZZ = rsum.ZZZ = f28_zero<RR>();
        XYZZ28<C> msum = rsum;
        for (int k = 0; k + 1 < NP; k += 2) {
            XYZZ<C> pair = XYZZ<C>::from_affine(pts[k]);
            xyzz_madd<C>(pair, pts[k + 1], (k & 2) != 0);
            dsum = xyzz_add<C>(dsum, pair);
            XYZZ<C> rec = xyzz_std_to_internal<C>(pair);                    // what a task partial looks like in memory
            xyzz28_add_full<C>(rsum, xyzz28_load<C>(rec));
            xyzz28_add_into_full<C>(&msum, xyzz28_load<C>(rec));           // streamed-operand form (LDS accumulator)
            if (k == 10) {   // a + a: exceptional
                xyzz28_add_full<C>(rsum, xyzz28_load<C>(xyzz28_store<C>(rsum)));
                xyzz28_add_into_full<C>(&msum, xyzz28_load<C>(xyzz28_store<C>(msum)));
                dsum = xyzz_add<C>(dsum, dsum);
            }
        }
        Affine<C> b1 = xyzz_to_affine<C>(dsum), b2 = xyzz_to_affine<C>(xyzz_internal_to_std<C>(xyzz28_store<C>(rsum)));
        if (!(b1.x.eq(b2.x) && b1.y.eq(b2.y))) { fails++; printf("%s full-add chain mismatch\n", name); }
        Affine<C> b3 = xyzz_to_affine<C>(xyzz_internal_to_std<C>(xyzz28_store<C>(msum)));
        if (!(b1.x.eq(b3.x) && b1.y.eq(b3.y))) { fails++; printf("%s add-into chain mismatch\n", name); }
    }
    // 2c. doubling on F28 registers vs dense
    {
        XYZZ<C> d = XYZZ<C>::from_affine(pts[9]);
        xyzz_madd<C>(d, pts[3], false);
        XYZZ28<C> q = xyzz28_load<C>(xyzz_std_to_internal<C>(d));
        for (int k = 0; k < 20; ++k) { d = xyzz_dbl<C>(d); xyzz28_dbl<C>(q); }
        Affine<C> c1 = xyzz_to_affine<C>(d), c2 = xyzz_to_affine<C>(xyzz_internal_to_std<C>(xyzz28_store<C>(q)));
        if (!(c1.x.eq(c2.x) && c1.y.eq(c2.y))) { fails++; printf("%s doubling chain mismatch\n", name); }
    }
    // 3. P + (-P) and P + P through the exceptional path
    XYZZ28<C> e;
    e.X = e.Y = e.ZZ = e.ZZZ = f28_zero<RR>();
    Affine<C> pi{fq_std_to_int<C>(pts[7].x), fq_std_to_int<C>(pts[7].y)};
    bool ok1 = xyzz28_madd<C>(e, pi, false), ok2 = xyzz28_madd<C>(e, pi, true), ok3 = xyzz28_madd<C>(e, pi, false);
    if (!ok1 || ok2 || ok3) { fails++; printf("%s exceptional-case detection wrong %d %d %d\n", name, ok1, ok2, ok3); }
    printf("%s: %d failures (%d exceptional cases resolved)\n", name, fails, exceptional);
    return fails;
}

// The dense product the kernels use (field.cuh: mul_r28, 28-bit limbs, radix 2^(32 N)) against the textbook
// 32-bit CIOS (mul_cios) on random and extreme operands: identical canonical words; so is mul(), which host code takes on 64-bit limbs.
template <class P>
static int dense_mul_check(const char *name, int iters) {
    int bad = 0;
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    const int topbits = P::BITS - 32 * (P::N - 1);
    for (int it = 0; it < iters; ++it) {
        Fp<P> a, b;
        for (int i = 0; i < P::N; ++i) { a.l[i] = rnd(); b.l[i] = rnd(); }
        a.l[P::N - 1] &= (1u << (topbits - 1)) - 1;     // < p
        b.l[P::N - 1] &= (1u << (topbits - 1)) - 1;
        if (it == 0) { for (int i = 0; i < P::N; ++i) a.l[i] = P::MOD[i]; a.l[0] -= 1; b = a; }   // (p - 1)^2
        if (it == 1) a = Fp<P>::zero();
        if (it == 2) a = Fp<P>::one();
        if (it == 3) { for (int i = 0; i < P::N; ++i) a.l[i] = P::MOD[i]; a.l[0] -= 1; b = Fp<P>::one(); }
        if (!mul_r28<P>(a, b).eq(mul_cios<P>(a, b))) bad++;
        if (!mul<P>(a, b).eq(mul_cios<P>(a, b))) bad++;   // host code's product (64-bit limbs where the compiler has __int128)
    }
    printf("%s dense mul vs CIOS: %d mismatches of %d\n", name, bad, iters);
    return bad;
}


// f28_canonical_quot (one quotient step) against f28_canonical_lazy (the chain of conditional subtractions) on values k p + r,
// k < 2^(JMAX+1): random r, r = 0 / 1 / p - 1, every k -- the lazily grown tile values of the transforms, tight limbs, excess in the top limb.
template <class RR, int JMAX>
static int canonical_check(const char *name, int iters) {
    int fails = 0;
    constexpr int N = RR::N;
    for (int it = 0; it < iters; ++it) {
        // r < p in W-bit limbs
        F28<RR> r;
        if (it % 7 == 0) { for (int i = 0; i < N; ++i) r.l[i] = 0; }
        else if (it % 7 == 1) { for (int i = 0; i < N; ++i) r.l[i] = RR::MOD[i]; r.l[0] -= 1; }       // p - 1 (p is odd)
        else if (it % 7 == 2) { for (int i = 0; i < N; ++i) r.l[i] = 0; r.l[0] = 1; }
        else {
            for (int i = 0; i < N; ++i) r.l[i] = (uint32_t)next_u64() & RR::MASK;
            r.l[N - 1] %= RR::MOD[N - 1];                                                                // top limb below p's: r < p
        }
        const unsigned k = (unsigned)(next_u64() % (1u << (JMAX + 1)));
        // x = r + k p, carries propagated, excess in the top limb
        F28<RR> x;
        uint64_t c = 0;
        for (int i = 0; i < N; ++i) {
            const uint64_t t = (uint64_t)r.l[i] + (uint64_t)k * RR::MOD[i] + c;
            x.l[i] = i + 1 < N ? (uint32_t)(t & RR::MASK) : (uint32_t)t;
            c = t >> RR::W;
        }
        const F28<RR> a = f28_canonical_lazy<RR, JMAX>(x), b = f28_canonical_quot<RR, JMAX>(x);
        bool same = true, is_r = true;
        for (int i = 0; i < N; ++i) { same = same && a.l[i] == b.l[i]; is_r = is_r && b.l[i] == r.l[i]; }
        if (!same || !is_r) { if (fails++ < 5) printf("%s canonical_quot mismatch (k = %u, JMAX = %d)\n", name, k, JMAX); }
    }
    return fails;
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int f = run<BlsCurve>("bls12_381", iters) + run<BnCurve>("bn254", iters);
    f += dense_mul_check<BlsFrP>("BlsFr", 50 * iters) + dense_mul_check<BnFrP>("BnFr", 50 * iters) + dense_mul_check<BlsFqP>("BlsFq", 50 * iters) +
         dense_mul_check<BnFqP>("BnFq", 50 * iters);
    f += canonical_check<BlsFrRR29, 5>("BlsFr29", 200 * iters) + canonical_check<BlsFrRR29, 4>("BlsFr29", 200 * iters) +
         canonical_check<BnFrRR29, 5>("BnFr29", 200 * iters) + canonical_check<BnFrRR29, 4>("BnFr29", 200 * iters);
    printf("canonical_quot: %s\n", f ? "FAILURES" : "ok");
    return f ? 1 : 0;
}
