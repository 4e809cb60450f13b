// G1 arithmetic for short-Weierstrass curves y^2 = x^3 + b (a = 0): BLS12-381 and BN254.
//
// Bucket accumulators use XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): the mixed add
// is 8M + 2S with no inversion, the cheapest complete-enough formula for Pippenger buckets
// (EFD madd-2008-s / add-2008-s / dbl-2008-s-1).  Bases stay affine (2 x Fq, Montgomery), the
// layout arkworks' ProvingKey holds them in (data_structures.rs:56-73).  The point at infinity is
// encoded as x = y = 0 for affine (not on the curve since b != 0) and ZZ = 0 for XYZZ.
#pragma once
#include "field.cuh"

namespace pm {

template <class C>
struct Affine {
    typedef Fp<typename C::FqP> Fq;
    Fq x, y;
    PM_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    PM_HD static Affine infinity() { return Affine{Fq::zero(), Fq::zero()}; }
};

template <class C>
struct XYZZ {
    typedef typename C::FqP P;
    typedef Fp<P> Fq;
    Fq X, Y, ZZ, ZZZ;

    PM_HD static XYZZ identity() { return XYZZ{Fq::zero(), Fq::zero(), Fq::zero(), Fq::zero()}; }
    PM_HD bool is_identity() const { return ZZ.is_zero(); }
    PM_HD static XYZZ from_affine(const Affine<C> &a) {
        if (a.is_inf()) return identity();
        return XYZZ{a.x, a.y, Fq::one(), Fq::one()};
    }
};

// dbl-2008-s-1 on an affine input (ZZ = ZZZ = 1): used when a bucket meets the same point twice.
template <class C>
PM_HD_COLD XYZZ<C> xyzz_dbl_affine(const Affine<C> &a) {
    typedef typename C::FqP P;
    typedef Fp<P> Fq;
    if (a.is_inf() || a.y.is_zero()) return XYZZ<C>::identity();
    Fq U = dbl<P>(a.y), V = sqr<P>(U), W = mul<P>(U, V), S = mul<P>(a.x, V);
    Fq x2 = sqr<P>(a.x), M = add<P>(dbl<P>(x2), x2);
    XYZZ<C> r;
    r.X = sub<P>(sqr<P>(M), dbl<P>(S));
    r.Y = sub<P>(mul<P>(M, sub<P>(S, r.X)), mul<P>(W, a.y));
    r.ZZ = V;
    r.ZZZ = W;
    return r;
}

template <class C>
PM_HD_COLD XYZZ<C> xyzz_dbl(const XYZZ<C> &p) {
    typedef typename C::FqP P;
    typedef Fp<P> Fq;
    if (p.is_identity() || p.Y.is_zero()) return XYZZ<C>::identity();
    Fq U = dbl<P>(p.Y), V = sqr<P>(U), W = mul<P>(U, V), S = mul<P>(p.X, V);
    Fq x2 = sqr<P>(p.X), M = add<P>(dbl<P>(x2), x2);
    XYZZ<C> r;
    r.X = sub<P>(sqr<P>(M), dbl<P>(S));
    r.Y = sub<P>(mul<P>(M, sub<P>(S, r.X)), mul<P>(W, p.Y));
    r.ZZ = mul<P>(V, p.ZZ);
    r.ZZZ = mul<P>(W, p.ZZZ);
    return r;
}

// acc += (x2, y2) with y2 negated when `negate`: madd-2008-s, 8M + 2S.
template <class C>
PM_HD void xyzz_madd(XYZZ<C> &acc, const Affine<C> &q, bool negate) {
    typedef typename C::FqP P;
    typedef Fp<P> Fq;
    if (q.is_inf()) return;
    Fq y2 = negate ? neg<P>(q.y) : q.y;
    if (acc.is_identity()) {
        acc.X = q.x;
        acc.Y = y2;
        acc.ZZ = Fq::one();
        acc.ZZZ = Fq::one();
        return;
    }
    Fq Pp = sub<P>(mul<P>(q.x, acc.ZZ), acc.X);
    Fq R = sub<P>(mul<P>(y2, acc.ZZZ), acc.Y);
    if (Pp.is_zero()) {
        if (R.is_zero()) {
            Affine<C> t{q.x, y2};
            acc = xyzz_dbl_affine<C>(t);
        } else {
            acc = XYZZ<C>::identity();
        }
        return;
    }
    Fq PP = sqr<P>(Pp), PPP = mul<P>(Pp, PP), Q = mul<P>(acc.X, PP);
    Fq X3 = sub<P>(sub<P>(sqr<P>(R), PPP), dbl<P>(Q));
    acc.Y = sub<P>(mul<P>(R, sub<P>(Q, X3)), mul<P>(acc.Y, PPP));
    acc.X = X3;
    acc.ZZ = mul<P>(acc.ZZ, PP);
    acc.ZZZ = mul<P>(acc.ZZZ, PPP);
}

// a + b, both XYZZ: add-2008-s, 12M + 2S.
template <class C>
PM_HD_COLD XYZZ<C> xyzz_add(const XYZZ<C> &a, const XYZZ<C> &b) {
    typedef typename C::FqP P;
    typedef Fp<P> Fq;
    if (a.is_identity()) return b;
    if (b.is_identity()) return a;
    Fq U1 = mul<P>(a.X, b.ZZ), U2 = mul<P>(b.X, a.ZZ);
    Fq S1 = mul<P>(a.Y, b.ZZZ), S2 = mul<P>(b.Y, a.ZZZ);
    Fq Pp = sub<P>(U2, U1), R = sub<P>(S2, S1);
    if (Pp.is_zero()) return R.is_zero() ? xyzz_dbl<C>(a) : XYZZ<C>::identity();
    Fq PP = sqr<P>(Pp), PPP = mul<P>(Pp, PP), Q = mul<P>(U1, PP);
    XYZZ<C> r;
    r.X = sub<P>(sub<P>(sqr<P>(R), PPP), dbl<P>(Q));
    r.Y = sub<P>(mul<P>(R, sub<P>(Q, r.X)), mul<P>(S1, PPP));
    r.ZZ = mul<P>(mul<P>(a.ZZ, b.ZZ), PP);
    r.ZZZ = mul<P>(mul<P>(a.ZZZ, b.ZZZ), PPP);
    return r;
}

// XYZZ -> affine (one Fermat inversion: x = X/ZZ, y = Y/ZZZ; 1/ZZ = (ZZ/ZZZ)^2 ... computed as
// i = 1/ZZZ, 1/ZZ = (i * ZZ)^2 since ZZ^3 = ZZZ^2  =>  (ZZ/ZZZ)^2 = ZZ^2/ZZ^3 = 1/ZZ).
template <class C>
PM_HD_COLD Affine<C> xyzz_to_affine(const XYZZ<C> &p) {
    typedef typename C::FqP P;
    typedef Fp<P> Fq;
    if (p.is_identity()) return Affine<C>::infinity();
    Fq i3 = inverse<P>(p.ZZZ);
    Fq i2 = sqr<P>(mul<P>(i3, p.ZZ));
    return Affine<C>{mul<P>(p.X, i2), mul<P>(p.Y, i3)};
}

template <class C>
PM_HD bool affine_on_curve(const Affine<C> &a) {
    typedef typename C::FqP P;
    typedef Fp<P> Fq;
    if (a.is_inf()) return true;
    Fq b;
    for (int i = 0; i < P::N; ++i) b.l[i] = C::B_MONT[i];
    return sqr<P>(a.y).eq(add<P>(mul<P>(sqr<P>(a.x), a.x), b));
}

}  // namespace pm
