// Pairing check for the host mirror's Polymath::verify (/root/reference/src/verifier.rs:50-61:
// `E::multi_pairing(..).0.is_one()`), for both pairing engines of SURVEY.md §8 f-3: BLS12-381 (the one the reference
// instantiates, Cargo.toml:35) and BN254 (BASELINE.json configs[4]).  Verification is O(1) host work (SURVEY.md §2
// row 7) -- simplicity over speed: Fq12 = Fq[w]/(w^12 - A w^6 + B) with schoolbook products, G2 points untwisted into
// E(Fq12), generic affine line functions, final exponentiation by plain powering with (p^12 - 1)/r.
//   BLS12-381: w^12 = 2 w^6 - 2   (u = w^6 - 1), M-type twist (x / w^2, y / w^3), Miller loop over |x|;
//   BN254    : w^12 = 18 w^6 - 82 (i = w^6 - 9), D-type twist (x w^2, y w^3), optimal ate: loop over 6x + 2, then the
//              two Frobenius line steps with pi(Q) and -pi^2(Q).
// Any non-degenerate bilinear map gives the same accept/reject answer for a product-equals-one check;
// bilinearity is unit-tested (tests/native/host_selftest.cpp).  ~0.3 s per check.
#pragma once
#include <array>
#include <vector>

#include "../csrc/ec.cuh"

namespace pmhost {

// Per-curve parameters of PairingT
struct BlsPairingParams {
    typedef pm::BlsCurve Curve;
    typedef pm::BlsPairingConsts Consts;
    static constexpr unsigned MOD_A = 2, MOD_B = 2, SHIFT = 1;     // w^12 = A w^6 - B;  Fq2 unit = w^6 - SHIFT
    static constexpr bool TWIST_MUL = false, FROBENIUS_STEPS = false;
    static constexpr int LOOP_LIMBS = 2;
    static constexpr uint32_t LOOP[2] = {0x00010000u, 0xd2010000u};   // |x| = 0xd201000000010000
    static constexpr const char *G2X0 = "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8";
    static constexpr const char *G2X1 = "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e";
    static constexpr const char *G2Y0 = "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801";
    static constexpr const char *G2Y1 = "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be";
};
struct BnPairingParams {
    typedef pm::BnCurve Curve;
    typedef pm::BnPairingConsts Consts;
    static constexpr unsigned MOD_A = 18, MOD_B = 82, SHIFT = 9;
    static constexpr bool TWIST_MUL = true, FROBENIUS_STEPS = true;
    static constexpr int LOOP_LIMBS = 3;
    static constexpr uint32_t LOOP[3] = {0xbe763ba8u, 0x9d797039u, 0x00000001u};   // 6x + 2 = 29793968203157093288, x = 4965661367192848881
    // G2 generator of ark-bn254 / EIP-197
    static constexpr const char *G2X0 = "1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed";
    static constexpr const char *G2X1 = "198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2";
    static constexpr const char *G2Y0 = "12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa";
    static constexpr const char *G2Y1 = "090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b";
};

template <class PP>
struct PairingT {
    typedef typename PP::Curve Curve;
    typedef typename Curve::FqP Q;
    typedef pm::Fp<Q> Fq;
    static Fq fadd(const Fq &a, const Fq &b) { return pm::add<Q>(a, b); }
    static Fq fsub(const Fq &a, const Fq &b) { return pm::sub<Q>(a, b); }
    static Fq fmul(const Fq &a, const Fq &b) { return pm::mul<Q>(a, b); }
    static Fq fneg(const Fq &a) { return pm::neg<Q>(a); }
    static Fq finv(const Fq &a) { return pm::inverse<Q>(a); }
    static Fq small(unsigned v) { return pm::from_u64<Q>(v); }

    // ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2 + 1)
    struct Fq2 {
        Fq c0, c1;
        bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
        bool eq(const Fq2 &o) const { return c0.eq(o.c0) && c1.eq(o.c1); }
    };
    static Fq2 add2(const Fq2 &a, const Fq2 &b) { return {fadd(a.c0, b.c0), fadd(a.c1, b.c1)}; }
    static Fq2 sub2(const Fq2 &a, const Fq2 &b) { return {fsub(a.c0, b.c0), fsub(a.c1, b.c1)}; }
    static Fq2 neg2(const Fq2 &a) { return {fneg(a.c0), fneg(a.c1)}; }
    static Fq2 mul2(const Fq2 &a, const Fq2 &b) {
        return {fsub(fmul(a.c0, b.c0), fmul(a.c1, b.c1)), fadd(fmul(a.c0, b.c1), fmul(a.c1, b.c0))};
    }
    static Fq2 inv2(const Fq2 &a) {
        Fq d = finv(fadd(fmul(a.c0, a.c0), fmul(a.c1, a.c1)));
        return {fmul(a.c0, d), fneg(fmul(a.c1, d))};
    }
    struct G2 { Fq2 x, y; bool inf; };

    static G2 g2_add(const G2 &A, const G2 &B) {
        if (A.inf) return B;
        if (B.inf) return A;
        Fq2 m;
        if (A.x.eq(B.x)) {
            if (add2(A.y, B.y).is_zero()) return G2{{}, {}, true};
            Fq2 x2 = mul2(A.x, A.x), three = {small(3), Fq::zero()}, two = {small(2), Fq::zero()};
            m = mul2(mul2(three, x2), inv2(mul2(two, A.y)));
        } else {
            m = mul2(sub2(B.y, A.y), inv2(sub2(B.x, A.x)));
        }
        Fq2 x3 = sub2(sub2(mul2(m, m), A.x), B.x);
        return G2{x3, sub2(mul2(m, sub2(A.x, x3)), A.y), false};
    }
    static G2 g2_neg(const G2 &A) { return G2{A.x, neg2(A.y), A.inf}; }
    // k: canonical little-endian 32-bit limbs of a scalar
    static G2 g2_mul(const G2 &A, const uint32_t *k, int nlimbs) {
        G2 out{{}, {}, true};
        for (int i = nlimbs - 1; i >= 0; --i)
            for (int b = 31; b >= 0; --b) {
                out = g2_add(out, out);
                if ((k[i] >> b) & 1) out = g2_add(out, A);
            }
        return out;
    }
    static Fq2 twist_b() {   // b' of the twist y^2 = x^3 + b': b / xi (D-type, BN254) or b xi (M-type, BLS12-381)
        const Fq2 xi{small(PP::SHIFT), Fq::one()};          // SHIFT + u: 1 + u (BLS12-381), 9 + i (BN254)
        Fq2 cb{Fq::zero(), Fq::zero()};
        for (int i = 0; i < Q::N; ++i) cb.c0.l[i] = Curve::B_MONT[i];
        return PP::TWIST_MUL ? mul2(cb, inv2(xi)) : mul2(cb, xi);
    }
    static bool g2_on_curve(const G2 &A) {
        if (A.inf) return true;
        return mul2(A.y, A.y).eq(add2(mul2(mul2(A.x, A.x), A.x), twist_b()));
    }
    static G2 g2_generator() {
        auto H = [](const char *hex) {
            Fq v = Fq::zero();
            size_t len = strlen(hex);
            for (size_t i = 0; i < len; ++i) {
                char ch = hex[len - 1 - i];
                uint32_t d = (ch >= '0' && ch <= '9') ? ch - '0' : ch - 'a' + 10;
                v.l[i / 8] |= d << (4 * (i % 8));
            }
            return pm::to_mont<Q>(v);
        };
        G2 g;
        g.x = {H(PP::G2X0), H(PP::G2X1)};
        g.y = {H(PP::G2Y0), H(PP::G2Y1)};
        g.inf = false;
        return g;
    }

    // ------------------------------------------------------- Fq12 = Fq[w]/(w^12 - A w^6 + B)
    struct Fq12 {
        Fq c[12];
        static Fq12 zero() { Fq12 r; for (auto &v : r.c) v = Fq::zero(); return r; }
        static Fq12 one() { Fq12 r = zero(); r.c[0] = Fq::one(); return r; }
        static Fq12 scalar(const Fq &v) { Fq12 r = zero(); r.c[0] = v; return r; }
        bool eq(const Fq12 &o) const { for (int i = 0; i < 12; ++i) if (!c[i].eq(o.c[i])) return false; return true; }
        bool is_zero() const { for (int i = 0; i < 12; ++i) if (!c[i].is_zero()) return false; return true; }
    };
    static Fq12 add12(const Fq12 &a, const Fq12 &b) { Fq12 r; for (int i = 0; i < 12; ++i) r.c[i] = fadd(a.c[i], b.c[i]); return r; }
    static Fq12 sub12(const Fq12 &a, const Fq12 &b) { Fq12 r; for (int i = 0; i < 12; ++i) r.c[i] = fsub(a.c[i], b.c[i]); return r; }
    static Fq12 mul12(const Fq12 &a, const Fq12 &b) {
        Fq t[23];
        for (auto &v : t) v = Fq::zero();
        for (int i = 0; i < 12; ++i) {
            if (a.c[i].is_zero()) continue;
            for (int j = 0; j < 12; ++j) t[i + j] = fadd(t[i + j], fmul(a.c[i], b.c[j]));
        }
        const Fq ma = small(PP::MOD_A), mb = small(PP::MOD_B);
        for (int k = 22; k >= 12; --k) {   // w^k = A w^(k-6) - B w^(k-12)
            if (t[k].is_zero()) continue;
            t[k - 6] = fadd(t[k - 6], fmul(t[k], ma));
            t[k - 12] = fsub(t[k - 12], fmul(t[k], mb));
        }
        Fq12 r;
        for (int i = 0; i < 12; ++i) r.c[i] = t[i];
        return r;
    }
    static Fq12 muls12(const Fq12 &a, unsigned k) { Fq12 r; Fq s = small(k); for (int i = 0; i < 12; ++i) r.c[i] = fmul(a.c[i], s); return r; }

    // inverse by the extended Euclidean algorithm over Fq[w]
    static int deg(const std::vector<Fq> &p) { int d = (int)p.size() - 1; while (d > 0 && p[d].is_zero()) --d; return d; }
    static std::vector<Fq> poly_div(const std::vector<Fq> &a, const std::vector<Fq> &b) {   // quotient
        int da = deg(a), db = deg(b);
        std::vector<Fq> temp(a), o(a.size(), Fq::zero());
        Fq binv = finv(b[db]);
        for (int i = da - db; i >= 0; --i) {
            Fq q = fmul(temp[db + i], binv);
            o[i] = fadd(o[i], q);
            for (int c = 0; c <= db; ++c) temp[c + i] = fsub(temp[c + i], fmul(q, b[c]));
        }
        o.resize(deg(o) + 1);
        return o;
    }
    static Fq12 inv12(const Fq12 &a) {
        std::vector<Fq> lm(13, Fq::zero()), hm(13, Fq::zero()), low(13, Fq::zero()), high(13, Fq::zero());
        lm[0] = Fq::one();
        for (int i = 0; i < 12; ++i) low[i] = a.c[i];
        high[0] = small(PP::MOD_B); high[6] = fneg(small(PP::MOD_A)); high[12] = Fq::one();   // w^12 - A w^6 + B
        while (deg(low) > 0) {
            std::vector<Fq> r = poly_div(high, low);
            r.resize(13, Fq::zero());
            std::vector<Fq> nm(hm), nw(high);
            for (int i = 0; i < 13; ++i)
                for (int j = 0; j < 13 - i; ++j) {
                    nm[i + j] = fsub(nm[i + j], fmul(lm[i], r[j]));
                    nw[i + j] = fsub(nw[i + j], fmul(low[i], r[j]));
                }
            hm = lm; high = low; lm = nm; low = nw;
        }
        Fq li = finv(low[0]);
        Fq12 out;
        for (int i = 0; i < 12; ++i) out.c[i] = fmul(lm[i], li);
        return out;
    }
    static Fq12 div12(const Fq12 &a, const Fq12 &b) { return mul12(a, inv12(b)); }

    // ------------------------------------------------------------------------ E(Fq12), Miller loop
    struct P12 { Fq12 x, y; bool inf; };
    static P12 twist(const G2 &Qp) {   // E'(Fq2) -> E(Fq12);  u = w^6 - SHIFT
        P12 r;
        r.inf = Qp.inf;
        if (Qp.inf) return r;
        Fq12 nx = Fq12::zero(), ny = Fq12::zero(), w = Fq12::zero();
        const Fq sh = small(PP::SHIFT);
        nx.c[0] = fsub(Qp.x.c0, fmul(sh, Qp.x.c1)); nx.c[6] = Qp.x.c1;
        ny.c[0] = fsub(Qp.y.c0, fmul(sh, Qp.y.c1)); ny.c[6] = Qp.y.c1;
        w.c[1] = Fq::one();
        Fq12 w2 = mul12(w, w), w3 = mul12(w2, w);
        r.x = PP::TWIST_MUL ? mul12(nx, w2) : div12(nx, w2);
        r.y = PP::TWIST_MUL ? mul12(ny, w3) : div12(ny, w3);
        return r;
    }
    // a^e, e as little-endian 32-bit limbs (the Frobenius of the generic Fq12 representation is plain powering by p)
    static Fq12 pow12(const Fq12 &a, const uint32_t *e, int nlimbs) {
        Fq12 acc = Fq12::one();
        bool started = false;
        for (int i = nlimbs - 1; i >= 0; --i)
            for (int b = 31; b >= 0; --b) {
                if (started) acc = mul12(acc, acc);
                if ((e[i] >> b) & 1) { acc = started ? mul12(acc, a) : a; started = true; }
            }
        return acc;
    }
    static P12 dbl12(const P12 &p) {
        Fq12 m = div12(muls12(mul12(p.x, p.x), 3), muls12(p.y, 2));
        Fq12 nx = sub12(mul12(m, m), muls12(p.x, 2));
        return P12{nx, sub12(mul12(m, sub12(p.x, nx)), p.y), false};
    }
    static P12 add12p(const P12 &a, const P12 &b) {
        if (a.inf) return b;
        if (b.inf) return a;
        if (a.x.eq(b.x)) return a.y.eq(b.y) ? dbl12(a) : P12{{}, {}, true};
        Fq12 m = div12(sub12(b.y, a.y), sub12(b.x, a.x));
        Fq12 nx = sub12(sub12(mul12(m, m), a.x), b.x);
        return P12{nx, sub12(mul12(m, sub12(a.x, nx)), a.y), false};
    }
    static Fq12 line(const P12 &p1, const P12 &p2, const P12 &t) {
        if (!p1.x.eq(p2.x)) {
            Fq12 m = div12(sub12(p2.y, p1.y), sub12(p2.x, p1.x));
            return sub12(mul12(m, sub12(t.x, p1.x)), sub12(t.y, p1.y));
        }
        if (p1.y.eq(p2.y)) {
            Fq12 m = div12(muls12(mul12(p1.x, p1.x), 3), muls12(p1.y, 2));
            return sub12(mul12(m, sub12(t.x, p1.x)), sub12(t.y, p1.y));
        }
        return sub12(t.x, p1.x);
    }
    static Fq12 miller_loop(const G2 &Qp, const pm::Affine<Curve> &P, bool p_inf) {
        if (Qp.inf || p_inf) return Fq12::one();
        P12 Q12 = twist(Qp), P12p{Fq12::scalar(P.x), Fq12::scalar(P.y), false};
        P12 R = Q12;
        Fq12 f = Fq12::one();
        int top = 32 * PP::LOOP_LIMBS - 1;
        while (!((PP::LOOP[top >> 5] >> (top & 31)) & 1)) --top;
        for (int i = top - 1; i >= 0; --i) {
            f = mul12(mul12(f, f), line(R, R, P12p));
            R = dbl12(R);
            if ((PP::LOOP[i >> 5] >> (i & 31)) & 1) {
                f = mul12(f, line(R, Q12, P12p));
                R = add12p(R, Q12);
            }
        }
        if (PP::FROBENIUS_STEPS) {   // optimal ate on BN curves: Q1 = pi(Q), -Q2 = -pi^2(Q)
            P12 Q1{pow12(Q12.x, Q::MOD, Q::N), pow12(Q12.y, Q::MOD, Q::N), false};
            P12 nQ2{pow12(Q1.x, Q::MOD, Q::N), sub12(Fq12::zero(), pow12(Q1.y, Q::MOD, Q::N)), false};
            f = mul12(f, line(R, Q1, P12p));
            R = add12p(R, Q1);
            f = mul12(f, line(R, nQ2, P12p));
        }
        return f;
    }
    // f^((p^12 - 1) / r), exponent as 32-bit limbs (generated by tools/gen_constants.py)
    static Fq12 final_exponentiation(const Fq12 &f) {
        Fq12 acc = Fq12::one();
        bool started = false;
        for (int i = PP::Consts::FINAL_EXP_LIMBS - 1; i >= 0; --i)
            for (int b = 31; b >= 0; --b) {
                if (started) acc = mul12(acc, acc);
                if ((PP::Consts::FINAL_EXP[i] >> b) & 1) { acc = started ? mul12(acc, f) : f; started = true; }
            }
        return acc;
    }
    struct Pair { pm::Affine<Curve> p; bool p_inf; G2 q; };
    static bool product_is_one(const std::vector<Pair> &pairs) {
        Fq12 f = Fq12::one();
        for (const auto &pr : pairs) f = mul12(f, miller_loop(pr.q, pr.p, pr.p_inf));
        return final_exponentiation(f).eq(Fq12::one());
    }
};

typedef PairingT<BlsPairingParams> Bls12Pairing;
typedef PairingT<BnPairingParams> Bn254Pairing;
template <class C> struct PairingOf;
template <> struct PairingOf<pm::BlsCurve> { typedef Bls12Pairing type; };
template <> struct PairingOf<pm::BnCurve> { typedef Bn254Pairing type; };

}  // namespace pmhost
