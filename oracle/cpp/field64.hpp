// ORACLE -- test infrastructure only.  Never linked into or loaded by the product
// (polymath_amd/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY UNPINNED: the reference's arithmetic lives in un-vendored arkworks crates
// (/root/reference/Cargo.toml:13-31, 84-101) and the reference holds no golden vectors
// (tests/dummy.rs:69-72, tests/mimc.rs:214 only check verify(prove(..))).  This restatement is
// pinned against the independent big-integer restatement oracle/pyref (itself checked by a real
// pairing verifier) through tests/golden/*.json, and by algebraic identities.
//
// Prime fields in Montgomery form on 64-bit limbs, the representation ark-ff uses in memory
// (Fp<MontBackend<_, N>, N>: N x u64 little-endian, value * 2^(64N) mod p).  All Montgomery
// constants are derived at start-up from the modulus alone.
#pragma once
#include <cstdint>
#include <cstring>

typedef unsigned __int128 u128;
typedef uint64_t u64;

template <int N>
struct FieldParams {
    u64 mod[N];
    u64 inv;     // -mod^{-1} mod 2^64
    u64 R[N];    // 2^(64N) mod p  (Montgomery one)
    u64 R2[N];   // 2^(128N) mod p
    bool ready = false;
};

template <int N>
static inline bool big_geq(const u64 *a, const u64 *b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
template <int N>
static inline u64 big_sub(u64 *r, const u64 *a, const u64 *b) {
    u64 borrow = 0;
    for (int i = 0; i < N; ++i) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)t;
        borrow = (u64)(t >> 64) & 1;
    }
    return borrow;
}
template <int N>
static inline u64 big_add(u64 *r, const u64 *a, const u64 *b) {
    u64 carry = 0;
    for (int i = 0; i < N; ++i) {
        u128 t = (u128)a[i] + b[i] + carry;
        r[i] = (u64)t;
        carry = (u64)(t >> 64);
    }
    return carry;
}

template <int N, int ID>
struct Fp {
    u64 l[N];
    static FieldParams<N> P;
    static constexpr int LIMBS = N;

    static void init(const u64 *modulus) {
        memcpy(P.mod, modulus, sizeof(P.mod));
        if (modulus[N - 1] >> 63) __builtin_trap();      // operator* is the no-carry form: the top bit must be free
        // inv by Newton iteration on 2-adic inverse
        u64 x = 1;
        for (int i = 0; i < 6; ++i) x *= 2 - P.mod[0] * x;
        P.inv = (u64)0 - x;
        // R = 2^(64N) mod p by 64N modular doublings of 1; R2 by 64N more
        u64 t[N] = {1};
        for (int k = 0; k < 128 * N; ++k) {
            u64 c = big_add<N>(t, t, t);
            if (c || big_geq<N>(t, P.mod)) big_sub<N>(t, t, P.mod);
            if (k == 64 * N - 1) memcpy(P.R, t, sizeof(t));
        }
        memcpy(P.R2, t, sizeof(t));
        P.ready = true;
    }
    static Fp zero() { Fp r; memset(r.l, 0, sizeof(r.l)); return r; }
    static Fp one() { Fp r; memcpy(r.l, P.R, sizeof(r.l)); return r; }
    static Fp from_u64(u64 v) { Fp r = zero(); r.l[0] = v; return r.to_mont(); }
    static Fp from_raw(const u64 *p) { Fp r; memcpy(r.l, p, sizeof(r.l)); return r; }
    void store(u64 *p) const { memcpy(p, l, sizeof(l)); }
    bool is_zero() const { u64 a = 0; for (int i = 0; i < N; ++i) a |= l[i]; return a == 0; }
    bool operator==(const Fp &o) const { return memcmp(l, o.l, sizeof(l)) == 0; }
    bool operator!=(const Fp &o) const { return !(*this == o); }

    Fp operator+(const Fp &o) const {
        Fp r;
        u64 c = big_add<N>(r.l, l, o.l);
        if (c || big_geq<N>(r.l, P.mod)) big_sub<N>(r.l, r.l, P.mod);
        return r;
    }
    Fp operator-(const Fp &o) const {
        Fp r;
        if (big_sub<N>(r.l, l, o.l)) big_add<N>(r.l, r.l, P.mod);
        return r;
    }
    Fp neg() const { return is_zero() ? *this : (Fp::zero() - *this); }
    Fp dbl() const { return *this + *this; }

    // CIOS Montgomery multiplication, "no-carry" form: every modulus here leaves the top bit of its top limb free (381 of
    // 384, 255 / 254 of 256 bits -- checked in init), so the two carry words of a row add without overflowing and the
    // accumulator needs N limbs, not N + 2 (the observation arkworks' MontConfig derives CAN_USE_NO_CARRY_MUL_OPT from).
    // Round 5: 59 instead of 75 ns per Fq product on the build image's Xeon; the same canonical words.
    Fp operator*(const Fp &o) const {
        u64 T[N];
        const u64 inv = P.inv;
        u64 mod[N];
#pragma GCC unroll 8
        for (int j = 0; j < N; ++j) { T[j] = 0; mod[j] = P.mod[j]; }
#pragma GCC unroll 8
        for (int i = 0; i < N; ++i) {
            const u64 bi = o.l[i];
            u128 t = (u128)l[0] * bi + T[0];
            u64 c1 = (u64)(t >> 64);
            const u64 lo = (u64)t, m = lo * inv;
            u128 t2 = (u128)m * mod[0] + lo;
            u64 c2 = (u64)(t2 >> 64);
#pragma GCC unroll 8
            for (int j = 1; j < N; ++j) {
                t = (u128)l[j] * bi + T[j] + c1;
                c1 = (u64)(t >> 64);
                t2 = (u128)m * mod[j] + (u64)t + c2;
                c2 = (u64)(t2 >> 64);
                T[j - 1] = (u64)t2;
            }
            T[N - 1] = c1 + c2;
        }
        Fp r;
        if (big_geq<N>(T, P.mod)) big_sub<N>(r.l, T, P.mod);
        else memcpy(r.l, T, sizeof(r.l));
        return r;
    }
    Fp sqr() const { return *this * *this; }
    Fp to_mont() const { return *this * from_raw(P.R2); }
    Fp from_mont() const { Fp o = zero(); o.l[0] = 1; return *this * o; }  // canonical limbs

    Fp pow_limbs(const u64 *e, int nlimbs) const {
        Fp acc = one();
        bool started = false;
        for (int i = nlimbs - 1; i >= 0; --i)
            for (int b = 63; b >= 0; --b) {
                if (started) acc = acc.sqr();
                if ((e[i] >> b) & 1) { acc = acc * *this; started = true; }
            }
        return acc;
    }
    Fp pow_u64(u64 e) const { return pow_limbs(&e, 1); }
    Fp inverse() const {  // Fermat: a^(p-2)
        u64 e[N];
        u64 two[N] = {2};
        big_sub<N>(e, P.mod, two);
        return pow_limbs(e, N);
    }
};
template <int N, int ID>
FieldParams<N> Fp<N, ID>::P;

// Montgomery batch inversion; zeros stay zero.
template <class F>
static void batch_inverse(F *v, size_t n, F *scratch) {
    F acc = F::one();
    for (size_t i = 0; i < n; ++i) {
        scratch[i] = acc;
        if (!v[i].is_zero()) acc = acc * v[i];
    }
    F inv = acc.inverse();
    for (size_t i = n; i-- > 0;) {
        if (v[i].is_zero()) continue;
        F t = inv * scratch[i];
        inv = inv * v[i];
        v[i] = t;
    }
}
