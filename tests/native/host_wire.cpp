// GPU: a serialised ProvingKey (polymath_amd/host/wire.hpp) -> pm_pk_load -> proof; and back out of HBM.
//   host_wire <pk.hex> <assignment.txt>     assignment: "inst v..", "wit v..", "ra a b" (canonical hex)
// Prints "proof <hex>" (Merlin transcript) and "export identical=<0|1>".
#include <cstdio>
#include <fstream>
#include <sstream>
#include "../../polymath_amd/host/wire.hpp"
using namespace pmhost;
typedef pm::BlsCurve C;
typedef FrOps<C> F;
typedef F::Fr Fr;

static Bytes unhex(const std::string &s) {
    Bytes b;
    auto nib = [](char ch) { return ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10; };
    for (size_t i = 0; i + 1 < s.size(); i += 2) b.push_back((uint8_t)(nib(s[i]) << 4 | nib(s[i + 1])));
    return b;
}
static Fr fr_from_hex(std::string h) {   // big-endian hex, canonical
    if (h.rfind("0x", 0) == 0) h = h.substr(2);
    while (h.size() < 64) h = "0" + h;
    Bytes be = unhex(h);
    uint8_t le[32];
    for (int i = 0; i < 32; ++i) le[i] = be[31 - i];
    return F::from_le_bytes_canonical(le);
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1]);
    std::string hex;
    f >> hex;
    Bytes data = unhex(hex);
    std::vector<Fr> inst, wit, ra;
    std::ifstream g(argv[2]);
    std::string line;
    while (std::getline(g, line)) {
        std::istringstream ss(line);
        std::string tag, v;
        ss >> tag;
        std::vector<Fr> &dst = tag == "inst" ? inst : tag == "wit" ? wit : ra;
        while (ss >> v) dst.push_back(fr_from_hex(v));
    }
    try {
        Context ctx(0);
        WireKey<C> key = WireKey<C>::parse(data.data(), data.size());
        ProvingKey<C> pk = pk_load<C>(ctx, key);
        Polymath<C, MerlinFieldTranscript<C>> pm(ctx);
        Fr r_a[2] = {ra.at(0), ra.at(1)};
        Proof<C> proof = pm.prove_with_assignment(pk, inst, wit, r_a);
        printf("proof %s\n", to_hex(proof.to_bytes()).c_str());
        WireKey<C> back = pk_export<C>(ctx, pk, key.vk, key.mw, key.nr, key.a, key.b, key.c);
        printf("export identical=%d\n", (int)(back.to_bytes() == data));
        // a key file whose vk disagrees with its SAP header must be rejected, not silently produce invalid proofs
        int rejected = 0;
        for (int which = 0; which < 3; ++which) {
            WireKey<C> bad = key;
            if (which == 0) bad.vk.m0 += 1;
            if (which == 1) bad.vk.omega = FrOps<C>::mul(bad.vk.omega, bad.vk.omega);
            if (which == 2) bad.vec[PM_X_POWERS].push_back(bad.vec[PM_X_POWERS].back());
            try { (void)pk_load<C>(ctx, bad); } catch (const WireError &) { ++rejected; }
        }
        printf("inconsistent keys rejected=%d\n", rejected);
    } catch (const std::exception &e) {
        printf("error %s\n", e.what());
        return 1;
    }
    return 0;
}
