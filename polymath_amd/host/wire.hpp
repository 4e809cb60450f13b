// ProvingKey / VerifyingKey wire format above the C ABI.
//
// Restates the `#[derive(CanonicalSerialize, CanonicalDeserialize)]` layout of
// /root/reference/src/data_structures.rs:25-73 and src/common.rs:112-127 in compressed mode:
//   VerifyingKey { e: PairingVK { one_g1, one_g2, x_g2, z_g2 }, n, m0, sigma, omega }
//   ProvingKey   { vk, sap_matrices { num_instance_variables, num_r1cs_witness_variables, num_r1cs_constraints,
//                  a, b, c : Vec<Vec<(F, usize)>> }, x_powers_g1, x_powers_y_alpha_g1, x_powers_zh_by_y_alpha_g1,
//                  x_powers_y_gamma_g1, x_powers_y_gamma_z_g1, uj_wj_lcs_by_y_alpha_g1 : Vec<G1Affine> }
// usize / u64 = 8 B little-endian, Vec = u64 length prefix, Fr = 32 B little-endian canonical, BLS12-381
// G1 / G2 in the zcash encoding (big-endian x, 3 flag bits; G2 = x.c1 || x.c0) [ark, from memory -- SURVEY App. C].
// A WireKey is the host-side image (points decompressed, matrices with row-duplicate columns KEPT, as the
// reference serialises what the synthesiser produced); pk_load / pk_export move it through pm_pk_load /
// pm_pk_export_bases.  Pinned by tests/golden/pk_wire.json and the published generator encodings.
#pragma once
#include "polymath.hpp"

namespace pmhost {

struct WireError : std::runtime_error {
    explicit WireError(const std::string &m) : std::runtime_error(m) {}
};

struct Reader {
    const uint8_t *p;
    size_t len, off = 0;
    Reader(const uint8_t *d, size_t n) : p(d), len(n) {}
    const uint8_t *take(size_t n) {
        if (off + n > len) throw WireError("truncated key");
        const uint8_t *r = p + off;
        off += n;
        return r;
    }
    uint64_t u64() {
        const uint8_t *b = take(8);
        uint64_t v = 0;
        for (int i = 7; i >= 0; --i) v = (v << 8) | b[i];
        return v;
    }
};

// ------------------------------------------------------------------ square roots (p = 3 mod 4)
template <class Q>
pm::Fp<Q> fq_pow_limbs(const pm::Fp<Q> &a, const uint32_t *e, int nlimbs) {
    pm::Fp<Q> acc = pm::Fp<Q>::one();
    for (int i = nlimbs - 1; i >= 0; --i)
        for (int b = 31; b >= 0; --b) {
            acc = pm::sqr<Q>(acc);
            if ((e[i] >> b) & 1) acc = pm::mul<Q>(acc, a);
        }
    return acc;
}

// sqrt(a) = a^((p+1)/4); false when a is a non-residue
template <class Q>
bool fq_sqrt(const pm::Fp<Q> &a, pm::Fp<Q> &out) {
    uint32_t e[Q::N];
    uint64_t carry = 1;   // (p + 1) >> 2
    for (int i = 0; i < Q::N; ++i) {
        uint64_t v = (uint64_t)Q::MOD[i] + carry;
        e[i] = (uint32_t)v;
        carry = v >> 32;
    }
    for (int i = 0; i < Q::N; ++i) e[i] = (e[i] >> 2) | (i + 1 < Q::N ? e[i + 1] << 30 : (uint32_t)(carry << 30));
    out = fq_pow_limbs<Q>(a, e, Q::N);
    return pm::sqr<Q>(out).eq(a);
}

// canonical-integer comparison of two Montgomery-form elements
template <class Q>
int fq_cmp(const pm::Fp<Q> &a, const pm::Fp<Q> &b) {
    pm::Fp<Q> x = pm::from_mont<Q>(a), y = pm::from_mont<Q>(b);
    for (int i = Q::N - 1; i >= 0; --i)
        if (x.l[i] != y.l[i]) return x.l[i] > y.l[i] ? 1 : -1;
    return 0;
}

template <class C>
struct Fq2OpsT {   // Fq2 = Fq[u]/(u^2+1) of either pairing engine
    typedef typename PairingOf<C>::type B;
    typedef typename B::Fq Fq;
    typedef typename B::Fq2 Fq2;
    typedef typename C::FqP Q;
    static bool sqrt(const Fq2 &a, Fq2 &out) {   // complex method
        Fq s;
        if (a.c1.is_zero()) {
            if (fq_sqrt<Q>(a.c0, s)) { out = Fq2{s, Fq::zero()}; return true; }
            if (fq_sqrt<Q>(B::fneg(a.c0), s)) { out = Fq2{Fq::zero(), s}; return true; }
            return false;
        }
        Fq alpha;
        if (!fq_sqrt<Q>(B::fadd(B::fmul(a.c0, a.c0), B::fmul(a.c1, a.c1)), alpha)) return false;
        const Fq inv2 = B::finv(B::small(2));
        Fq x0;
        if (!fq_sqrt<Q>(B::fmul(B::fadd(a.c0, alpha), inv2), x0) && !fq_sqrt<Q>(B::fmul(B::fsub(a.c0, alpha), inv2), x0)) return false;
        out = Fq2{x0, B::fmul(a.c1, B::finv(B::fadd(x0, x0)))};
        return true;
    }
    // a > b in arkworks' Fq2 order: c1 first, then c0
    static bool gt(const Fq2 &a, const Fq2 &b) {
        int c = fq_cmp<Q>(a.c1, b.c1);
        return c ? c > 0 : fq_cmp<Q>(a.c0, b.c0) > 0;
    }
};
typedef Fq2OpsT<pm::BlsCurve> Fq2Ops;

// ---------------------------------------------------------------------------- point codecs
// BLS12-381: the zcash encoding (big-endian, x.c1 || x.c0, flags 0x80 compressed / 0x40 infinity / 0x20 y > -y in the FIRST byte);
// BN254: ark-serialize's default short-Weierstrass form (little-endian, x.c0 || x.c1, flags 0x80 y > -y / 0x40 infinity in the
// LAST byte)                                                                                          [ark, from memory]
template <class Q>
inline void put_fq_be(const pm::Fp<Q> &mont, Bytes &out) {
    pm::Fp<Q> x = pm::from_mont<Q>(mont);
    uint8_t le[4 * Q::N];
    memcpy(le, x.l, sizeof(le));
    for (int i = 4 * Q::N - 1; i >= 0; --i) out.push_back(le[i]);
}
template <class Q>
inline void put_fq_le(const pm::Fp<Q> &mont, Bytes &out) {
    pm::Fp<Q> x = pm::from_mont<Q>(mont);
    const uint8_t *le = (const uint8_t *)x.l;
    out.insert(out.end(), le, le + 4 * Q::N);
}
template <class Q>
inline bool get_fq(const uint8_t *b, bool big_endian, uint8_t top_mask, pm::Fp<Q> &mont) {
    const int NB = 4 * Q::N;
    uint8_t le[4 * Q::N];
    for (int i = 0; i < NB; ++i) le[i] = big_endian ? b[NB - 1 - i] : b[i];
    le[NB - 1] &= top_mask;
    pm::Fp<Q> x;
    memcpy(x.l, le, NB);
    for (int i = Q::N - 1; i >= 0; --i) {      // x < p
        if (x.l[i] < Q::MOD[i]) break;
        if (x.l[i] > Q::MOD[i] || i == 0) return false;
    }
    mont = pm::to_mont<Q>(x);
    return true;
}

template <class C>
inline void ser_g2_c(const typename PairingOf<C>::type::G2 &g, Bytes &out) {
    typedef typename PairingOf<C>::type B;
    typedef typename C::FqP Q;
    const int NB = 4 * Q::N;
    const size_t at = out.size();
    if (C::ID == 0) {
        if (g.inf) { out.push_back(0xC0); out.insert(out.end(), 2 * NB - 1, 0); return; }
        put_fq_be<Q>(g.x.c1, out);
        put_fq_be<Q>(g.x.c0, out);
        out[at] |= 0x80 | (Fq2OpsT<C>::gt(g.y, B::neg2(g.y)) ? 0x20 : 0);
    } else {
        if (g.inf) { out.insert(out.end(), 2 * NB - 1, 0); out.push_back(0x40); return; }
        put_fq_le<Q>(g.x.c0, out);
        put_fq_le<Q>(g.x.c1, out);
        if (Fq2OpsT<C>::gt(g.y, B::neg2(g.y))) out[at + 2 * NB - 1] |= 0x80;
    }
}

template <class C>
inline typename PairingOf<C>::type::G2 deser_g2_c(Reader &rd) {
    typedef typename PairingOf<C>::type B;
    typedef typename C::FqP Q;
    const int NB = 4 * Q::N;
    const uint8_t *b = rd.take(2 * NB);
    typename B::G2 g;
    bool larger;
    if (C::ID == 0) {
        if (!(b[0] & 0x80)) throw WireError("G2: not a compressed point");
        g.inf = (b[0] & 0x40) != 0;
        larger = (b[0] & 0x20) != 0;
    } else {
        g.inf = (b[2 * NB - 1] & 0x40) != 0;
        larger = (b[2 * NB - 1] & 0x80) != 0;
    }
    if (C::ID != 0 && g.inf && larger) throw WireError("G2: both flag bits set");                 // SWFlags::from_u8 -> None
    if (g.inf) {
        // canonical infinity only: the flag with every other bit and byte zero (ark-bls12-381 read_g2_compressed checks x == 0;
        // a sign bit next to the infinity flag is EncodingFlags' InvalidData)
        for (int i = 0; i < 2 * NB; ++i) {
            const uint8_t v = C::ID == 0 ? (i == 0 ? (uint8_t)(b[0] & 0x3F) : b[i]) : (i == 2 * NB - 1 ? (uint8_t)(b[i] & 0x3F) : b[i]);
            if (v) throw WireError("G2: non-canonical encoding of the point at infinity");
        }
        g.x = g.y = typename B::Fq2{B::Fq::zero(), B::Fq::zero()};
        return g;
    }
    const bool ok = C::ID == 0 ? (get_fq<Q>(b, true, 0x1F, g.x.c1) && get_fq<Q>(b + NB, true, 0xFF, g.x.c0))
                               : (get_fq<Q>(b, false, 0xFF, g.x.c0) && get_fq<Q>(b + NB, false, 0x3F, g.x.c1));
    if (!ok) throw WireError("G2: coordinate >= p");
    typename B::Fq2 rhs = B::add2(B::mul2(B::mul2(g.x, g.x), g.x), B::twist_b());
    if (!Fq2OpsT<C>::sqrt(rhs, g.y)) throw WireError("G2: not on the twist");
    if (Fq2OpsT<C>::gt(g.y, B::neg2(g.y)) != larger) g.y = B::neg2(g.y);
    // Validate::Yes (what deserialize_compressed means): the twist has a cofactor on both curves, so a point of E'(Fq2) need not
    // lie in G2 -- and the pairing is only defined there.  [r] Q == O.
    if (!B::g2_mul(g, C::FrP::MOD, C::FrP::N).inf) throw WireError("G2: not in the prime-order subgroup");
    return g;
}
inline void ser_g2(const Bls12Pairing::G2 &g, Bytes &out) { ser_g2_c<pm::BlsCurve>(g, out); }
inline Bls12Pairing::G2 deser_g2(Reader &rd) { return deser_g2_c<pm::BlsCurve>(rd); }

// [r] P == O on E(Fq).  BN254's G1 has cofactor 1 (every curve point is in the group); BLS12-381's has cofactor
// 0x396c8c005555e1568c00aaab0000aaab, so a point on the curve need not be in G1.
template <class C>
inline bool g1_in_subgroup(const G1Point<C> &g) {
    if (C::ID != 0 || g.inf) return true;
    pm::XYZZ<C> acc = pm::XYZZ<C>::identity();
    for (int i = C::FrP::N - 1; i >= 0; --i)
        for (int b = 31; b >= 0; --b) {
            acc = pm::xyzz_dbl<C>(acc);
            if ((C::FrP::MOD[i] >> b) & 1) pm::xyzz_madd<C>(acc, g.p, false);
        }
    return acc.is_identity();
}

// validate = ark-serialize's Validate::Yes (deserialize_compressed: on the curve AND in the prime-order subgroup); false =
// deserialize_compressed_unchecked's subgroup part only -- the curve equation is always enforced by decompression.
template <class C>
G1Point<C> deser_g1(Reader &rd, bool validate = true) {
    typedef typename C::FqP Q;
    const int NB = Q::N * 4;
    const uint8_t *b = rd.take(NB);
    G1Point<C> g;
    memset(&g.p, 0, sizeof(g.p));
    uint8_t le[64];
    bool larger;
    if (C::ID == 0) {
        if (!(b[0] & 0x80)) throw WireError("G1: not a compressed point");
        g.inf = (b[0] & 0x40) != 0;
        larger = (b[0] & 0x20) != 0;
        for (int i = 0; i < NB; ++i) le[i] = b[NB - 1 - i];
        le[NB - 1] &= 0x1F;
    } else {
        g.inf = (b[NB - 1] & 0x40) != 0;
        larger = (b[NB - 1] & 0x80) != 0;
        memcpy(le, b, NB);
        le[NB - 1] &= 0x3F;
    }
    if (C::ID != 0 && g.inf && larger) throw WireError("G1: both flag bits set");                 // SWFlags::from_u8 -> None
    if (g.inf) {
        // canonical infinity only (ark-bls12-381 read_g1_compressed: x must be 0; EncodingFlags: no sign bit with infinity)
        if (larger) throw WireError("G1: sign bit on the point at infinity");
        for (int i = 0; i < NB; ++i)
            if (le[i]) throw WireError("G1: non-canonical encoding of the point at infinity");
        return g;
    }
    pm::Fp<Q> x;
    memcpy(x.l, le, NB);
    for (int i = Q::N - 1; i >= 0; --i) {
        if (x.l[i] < Q::MOD[i]) break;
        if (x.l[i] > Q::MOD[i] || i == 0) throw WireError("G1: coordinate >= p");
    }
    g.p.x = pm::to_mont<Q>(x);
    pm::Fp<Q> bb;
    for (int i = 0; i < Q::N; ++i) bb.l[i] = C::B_MONT[i];
    pm::Fp<Q> rhs = pm::add<Q>(pm::mul<Q>(pm::sqr<Q>(g.p.x), g.p.x), bb), y;
    if (!fq_sqrt<Q>(rhs, y)) throw WireError("G1: not on the curve");
    if ((fq_cmp<Q>(y, pm::neg<Q>(y)) > 0) != larger) y = pm::neg<Q>(y);
    g.p.y = y;
    if (validate && !g1_in_subgroup<C>(g)) throw WireError("G1: not in the prime-order subgroup");
    return g;
}

// ------------------------------------------------------------------------------------ keys
template <class C>
inline void ser_vk_c(const VerifyingKeyT<C> &vk, Bytes &out) {
    ser_g1<C>(vk.one_g1, out);
    ser_g2_c<C>(vk.one_g2, out);
    ser_g2_c<C>(vk.x_g2, out);
    ser_g2_c<C>(vk.z_g2, out);
    ser_u64(vk.n, out);
    ser_u64(vk.m0, out);
    ser_u64(vk.sigma, out);
    ser_fr<C>(vk.omega, out);
}
template <class C>
inline VerifyingKeyT<C> read_vk_c(Reader &rd) {
    VerifyingKeyT<C> vk;
    vk.one_g1 = deser_g1<C>(rd);
    vk.one_g2 = deser_g2_c<C>(rd);
    vk.x_g2 = deser_g2_c<C>(rd);
    vk.z_g2 = deser_g2_c<C>(rd);
    vk.n = rd.u64();
    vk.m0 = rd.u64();
    vk.sigma = rd.u64();
    vk.omega = FrOps<C>::from_le_bytes_canonical(rd.take(32));
    return vk;
}
inline void ser_vk(const VerifyingKey &vk, Bytes &out) { ser_vk_c<pm::BlsCurve>(vk, out); }
inline VerifyingKey read_vk(Reader &rd) { return read_vk_c<pm::BlsCurve>(rd); }

// Proof::deserialize_compressed (data_structures.rs:10-19): a_g1, c_g1, a_at_x1, d_g1
template <class C>
inline Proof<C> read_proof(const uint8_t *data, size_t len) {
    Reader rd(data, len);
    Proof<C> p;
    p.a_g1 = deser_g1<C>(rd);
    p.c_g1 = deser_g1<C>(rd);
    p.a_at_x1 = FrOps<C>::from_le_bytes_canonical(rd.take(32));
    p.d_g1 = deser_g1<C>(rd);
    if (rd.off != len) throw WireError("trailing bytes after the proof");
    return p;
}

// pm_base_vec ids in the struct's declaration order (data_structures.rs:60-72)
static const int PK_WIRE_VECTORS[6] = {PM_X_POWERS, PM_X_POWERS_Y_ALPHA, PM_X_POWERS_ZH_BY_Y_ALPHA, PM_X_POWERS_Y_GAMMA,
                                       PM_X_POWERS_Y_GAMMA_Z, PM_UJ_WJ_LCS_BY_Y_ALPHA};

template <class C>
struct WireKey {
    VerifyingKey vk;
    uint64_t m0 = 0, mw = 0, nr = 0;          // num_instance_variables, num_r1cs_witness_variables, num_r1cs_constraints
    CsrHost a, b, c;                           // rows of (value Montgomery, column), duplicates kept
    std::vector<G1Point<C>> vec[PM_NUM_BASE_VECS];   // indexed by pm_base_vec

    static void put_matrix(const CsrHost &m, Bytes &out) {
        const size_t rows = m.rowptr.size() - 1;
        ser_u64(rows, out);
        for (size_t r = 0; r < rows; ++r) {
            ser_u64(m.rowptr[r + 1] - m.rowptr[r], out);
            for (uint64_t k = m.rowptr[r]; k < m.rowptr[r + 1]; ++k) {
                typename FrOps<C>::Fr v;
                memcpy(v.l, &m.val[4 * k], 32);
                ser_fr<C>(v, out);
                ser_u64(m.col[k], out);
            }
        }
    }
    static CsrHost get_matrix(Reader &rd) {
        CsrHost m;
        const uint64_t rows = rd.u64();
        for (uint64_t r = 0; r < rows; ++r) {
            const uint64_t cnt = rd.u64();
            for (uint64_t k = 0; k < cnt; ++k) {
                typename FrOps<C>::Fr v = FrOps<C>::from_le_bytes_canonical(rd.take(32));
                const uint64_t col = rd.u64();
                if (col > 0xffffffffull) throw WireError("matrix column out of range");
                m.col.push_back((uint32_t)col);
                const uint64_t *limbs = (const uint64_t *)v.l;
                m.val.insert(m.val.end(), limbs, limbs + 4);
            }
            m.rowptr.push_back(m.col.size());
        }
        return m;
    }

    Bytes to_bytes() const {
        Bytes out;
        ser_vk(vk, out);
        ser_u64(m0, out);
        ser_u64(mw, out);
        ser_u64(nr, out);
        put_matrix(a, out);
        put_matrix(b, out);
        put_matrix(c, out);
        for (int which : PK_WIRE_VECTORS) {
            ser_u64(vec[which].size(), out);
            for (const auto &g : vec[which]) ser_g1<C>(g, out);
        }
        return out;
    }

    // validate: the subgroup check on every base point, as deserialize_compressed does (Validate::Yes); false =
    // deserialize_compressed_unchecked for the vectors (a 2^20-gate key holds 27 M points: 255 doublings each)
    static WireKey parse(const uint8_t *data, size_t len, bool validate = true) {
        static_assert(C::ID == 0, "the reference instantiates Bls12_381 only (Cargo.toml:35); G2 codecs are BLS12-381");
        Reader rd(data, len);
        WireKey k;
        k.vk = read_vk(rd);
        k.m0 = rd.u64();
        k.mw = rd.u64();
        k.nr = rd.u64();
        k.a = get_matrix(rd);
        k.b = get_matrix(rd);
        k.c = get_matrix(rd);
        for (int which : PK_WIRE_VECTORS) {
            const uint64_t cnt = rd.u64();
            if (cnt > (len - rd.off) / (C::FqP::N * 4)) throw WireError("truncated key");
            k.vec[which].reserve(cnt);
            for (uint64_t i = 0; i < cnt; ++i) k.vec[which].push_back(deser_g1<C>(rd, validate));
        }
        if (rd.off != len) throw WireError("trailing bytes after the key");
        return k;
    }
};

// WireKey -> device-resident key (pm_pk_load uploads, shards and builds the window tables)
template <class C>
ProvingKey<C> pk_load(Context &ctx, const WireKey<C> &k, int shard_rank = 0, int shard_count = 1) {
    std::vector<pm::Affine<C>> flat[PM_NUM_BASE_VECS];
    pm_base_array arr[PM_NUM_BASE_VECS];
    for (int v = 0; v < PM_NUM_BASE_VECS; ++v) {
        flat[v].resize(k.vec[v].size());
        for (size_t i = 0; i < k.vec[v].size(); ++i) {
            if (k.vec[v][i].inf) memset(&flat[v][i], 0, sizeof(pm::Affine<C>));   // the ABI's infinity: all-zero x, y
            else flat[v][i] = k.vec[v][i].p;
        }
        arr[v] = pm_base_array{flat[v].data(), flat[v].size(), sizeof(pm::Affine<C>)};
    }
    pm_csr va = k.a.view(), vb = k.b.view(), vc = k.c.view();
    ProvingKey<C> pk;
    int st = pm_pk_load(ctx.h, C::ID, k.vk.n, k.m0, k.mw, k.nr, k.vk.sigma, &va, &vb, &vc, arr, shard_rank, shard_count, &pk.h);
    if (st) throw PolymathError(0, st, std::string("pm_pk_load: ") + pm_last_error(ctx.h));
    // the host mirror hashes and evaluates pi(x1) with the vk's n / m0 / omega while the device derives its own from the
    // SAP header: a file whose two halves disagree would silently produce invalid proofs -- reject it here
    uint64_t dn = 0, dm0 = 0, dsigma = 0, lens[PM_NUM_BASE_VECS];
    pm::Fp<typename C::FrP> domega;
    pm_pk_info(pk.h, &dn, &dm0, &dsigma, (uint64_t *)domega.l, lens);
    if (k.vk.m0 != k.m0 || dm0 != k.m0) throw WireError("vk.m0 disagrees with the SAP matrices' m0");
    if (dn != k.vk.n || dsigma != k.vk.sigma) throw WireError("vk.n / vk.sigma disagree with the domain of the SAP matrices");
    if (!domega.eq(k.vk.omega)) throw WireError("vk.omega is not the generator of the size-n domain");
    for (int v = 0; v < PM_NUM_BASE_VECS; ++v)
        if (k.vec[v].size() != lens[v]) throw WireError("a base vector's length does not match the key's shape");
    pk.n = k.vk.n; pk.m0 = k.vk.m0; pk.sigma = k.vk.sigma; pk.omega = k.vk.omega;
    return pk;
}

// device-resident key + its vk + the matrices as synthesised -> WireKey (bases come back from HBM)
template <class C>
WireKey<C> pk_export(Context &ctx, const ProvingKey<C> &pk, const VerifyingKey &vk, uint64_t mw, uint64_t nr, const CsrHost &a,
                     const CsrHost &b, const CsrHost &c) {
    WireKey<C> k;
    k.vk = vk;
    k.m0 = pk.m0; k.mw = mw; k.nr = nr;
    k.a = a; k.b = b; k.c = c;
    uint64_t lens[PM_NUM_BASE_VECS];
    pm_pk_info(pk.h, nullptr, nullptr, nullptr, nullptr, lens);
    for (int v = 0; v < PM_NUM_BASE_VECS; ++v) {
        std::vector<pm::Affine<C>> flat(lens[v]);
        if (lens[v]) {
            int st = pm_pk_export_bases(ctx.h, pk.h, v, 0, lens[v], (uint64_t *)flat.data());
            if (st) throw PolymathError(0, st, std::string("pm_pk_export_bases: ") + pm_last_error(ctx.h));
        }
        k.vec[v].resize(lens[v]);
        for (size_t i = 0; i < flat.size(); ++i) {
            const uint32_t *w = (const uint32_t *)&flat[i];
            uint32_t any = 0;
            for (size_t t = 0; t < sizeof(pm::Affine<C>) / 4; ++t) any |= w[t];
            k.vec[v][i] = G1Point<C>{flat[i], any == 0};
        }
    }
    return k;
}

}  // namespace pmhost
