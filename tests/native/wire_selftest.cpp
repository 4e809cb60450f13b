// CPU-only test of the ProvingKey / VerifyingKey wire codec (polymath_amd/host/wire.hpp): no GPU, no
// library link.  argv = files holding the hex of a serialised ProvingKey (tests/golden/pk_wire.json).
#include <cstdio>
#include <fstream>
#include <string>
#include "../../polymath_amd/host/wire.hpp"
using namespace pmhost;
typedef pm::BlsCurve C;

static Bytes unhex(const std::string &s) {
    Bytes b;
    auto nib = [](char ch) { return ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10; };
    for (size_t i = 0; i + 1 < s.size(); i += 2) b.push_back((uint8_t)(nib(s[i]) << 4 | nib(s[i + 1])));
    return b;
}

int main(int argc, char **argv) {
    int fails = 0;
    // published compressed generators (zcash / IETF BLS12-381 serialisation)
    const std::string g1_hex = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb";
    const std::string g2_hex = "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                               "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8";
    {
        G1Point<C> g;
        for (int i = 0; i < 12; ++i) { g.p.x.l[i] = C::GX_MONT[i]; g.p.y.l[i] = C::GY_MONT[i]; }
        g.inf = false;
        Bytes o;
        ser_g1<C>(g, o);
        if (to_hex(o) != g1_hex) { fails++; printf("G1 generator encoding\n"); }
        Reader rd(o.data(), o.size());
        G1Point<C> back = deser_g1<C>(rd);
        if (back.inf || !back.p.x.eq(g.p.x) || !back.p.y.eq(g.p.y)) { fails++; printf("G1 generator decode\n"); }
        g.p.y = pm::neg<pm::BlsFqP>(g.p.y);
        Bytes on;
        ser_g1<C>(g, on);
        Reader rn(on.data(), on.size());
        G1Point<C> bn = deser_g1<C>(rn);
        if (!(on[0] & 0x20) || !bn.p.y.eq(g.p.y)) { fails++; printf("G1 sign flag\n"); }
        Bytes o2;
        Bls12Pairing::G2 q = Bls12Pairing::g2_generator();
        ser_g2(q, o2);
        if (to_hex(o2) != g2_hex) { fails++; printf("G2 generator encoding %s\n", to_hex(o2).c_str()); }
        Reader r2(o2.data(), o2.size());
        Bls12Pairing::G2 qb = deser_g2(r2);
        if (qb.inf || !qb.x.eq(q.x) || !qb.y.eq(q.y)) { fails++; printf("G2 generator decode\n"); }
        Bls12Pairing::G2 qn = Bls12Pairing::g2_neg(q);
        Bytes o3;
        ser_g2(qn, o3);
        Reader r3(o3.data(), o3.size());
        Bls12Pairing::G2 qnb = deser_g2(r3);
        if (!qnb.y.eq(qn.y) || o3 == o2) { fails++; printf("G2 sign flag\n"); }
        Bytes bad(48, 0);
        bad[0] = 0x80; bad[47] = 1;    // x = 1: x^3 + 4 is a non-residue
        bool threw = false;
        try { Reader rb(bad.data(), bad.size()); deser_g1<C>(rb); } catch (const WireError &) { threw = true; }
        if (!threw) { fails++; printf("off-curve x accepted\n"); }
    }
    for (int a = 1; a < argc; ++a) {
        std::ifstream f(argv[a]);
        std::string hex;
        f >> hex;
        Bytes data = unhex(hex);
        try {
            WireKey<C> k = WireKey<C>::parse(data.data(), data.size());
            Bytes again = k.to_bytes();
            size_t pts = 0;
            for (auto &v : k.vec) pts += v.size();
            if (again != data) { fails++; printf("round trip differs: %s\n", argv[a]); }
            else printf("wire ok n=%llu m0=%llu nr=%llu points=%zu bytes=%zu\n", (unsigned long long)k.vk.n, (unsigned long long)k.vk.m0,
                        (unsigned long long)k.nr, pts, data.size());
            if (!Bls12Pairing::g2_on_curve(k.vk.x_g2) || !Bls12Pairing::g2_on_curve(k.vk.z_g2)) { fails++; printf("vk G2 off curve\n"); }
        } catch (const std::exception &e) { fails++; printf("parse failed: %s (%s)\n", e.what(), argv[a]); }
        bool t1 = false, t2 = false;
        try { WireKey<C>::parse(data.data(), data.size() - 1); } catch (const WireError &) { t1 = true; }
        Bytes more = data;
        more.push_back(0);
        try { WireKey<C>::parse(more.data(), more.size()); } catch (const WireError &) { t2 = true; }
        if (!t1 || !t2) { fails++; printf("truncated / trailing input accepted\n"); }
    }
    printf("wire selftest: %d failures\n", fails);
    return fails ? 1 : 0;
}
