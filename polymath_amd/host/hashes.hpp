// Host-side hashes behind the reference's three Fiat-Shamir transcripts
// (/root/reference/src/transcript/{merlin,keccak256,blake3}.rs): Keccak-f[1600] (Keccak-256 and
// STROBE-128 / Merlin v1.0) and BLAKE3, restated from their public specifications -- the crates
// merlin 3.0.0 / sha3 / blake3 are third-party dependencies of the reference (Cargo.toml:29-31).
// Known-answer vectors: tests/native/host_selftest.cpp.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace pmhost {

typedef std::vector<uint8_t> Bytes;

inline uint64_t rol64(uint64_t v, unsigned n) { return n ? (v << n) | (v >> (64 - n)) : v; }

inline void keccak_f1600(uint64_t a[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
        0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
        0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
        0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const unsigned RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int rnd = 0; rnd < 24; ++rnd) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) {
            int x = i % 5, y = i / 5;
            b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[i] ^ d[x], RHO[i]);
        }
        for (int y = 0; y < 25; y += 5)
            for (int x = 0; x < 5; ++x) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
        a[0] ^= RC[rnd];
    }
}

inline Bytes keccak_sponge256(const Bytes &data, uint8_t pad) {
    const size_t rate = 136;
    Bytes msg(data);
    msg.push_back(pad);
    while (msg.size() % rate) msg.push_back(0);
    msg.back() |= 0x80;
    uint64_t st[25] = {0};
    for (size_t off = 0; off < msg.size(); off += rate) {
        for (size_t i = 0; i < rate / 8; ++i) {
            uint64_t w;
            memcpy(&w, &msg[off + 8 * i], 8);
            st[i] ^= w;
        }
        keccak_f1600(st);
    }
    Bytes out(32);
    memcpy(out.data(), st, 32);
    return out;
}
inline Bytes keccak256(const Bytes &d) { return keccak_sponge256(d, 0x01); }  // legacy Keccak (sha3::Keccak256)

// ---------------------------------------------------------------------------------- BLAKE3
namespace b3 {
static const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const int PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };
inline uint32_t ror(uint32_t v, int n) { return (v >> n) | (v << (32 - n)); }
inline void g(uint32_t *s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx; s[d] = ror(s[d] ^ s[a], 16); s[c] = s[c] + s[d]; s[b] = ror(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my; s[d] = ror(s[d] ^ s[a], 8);  s[c] = s[c] + s[d]; s[b] = ror(s[b] ^ s[c], 7);
}
inline void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t blen, uint32_t flags, uint32_t out[8]) {
    uint32_t s[16], m[16], t[16];
    memcpy(s, cv, 32);
    memcpy(s + 8, IV, 16);
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = blen; s[15] = flags;
    memcpy(m, block, 64);
    for (int r = 0; r < 7; ++r) {
        g(s, 0, 4, 8, 12, m[0], m[1]); g(s, 1, 5, 9, 13, m[2], m[3]); g(s, 2, 6, 10, 14, m[4], m[5]); g(s, 3, 7, 11, 15, m[6], m[7]);
        g(s, 0, 5, 10, 15, m[8], m[9]); g(s, 1, 6, 11, 12, m[10], m[11]); g(s, 2, 7, 8, 13, m[12], m[13]); g(s, 3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; ++i) t[i] = m[PERM[i]];
        memcpy(m, t, 64);
    }
    for (int i = 0; i < 8; ++i) out[i] = s[i] ^ s[i + 8];
}
inline void words(const uint8_t *p, size_t len, uint32_t w[16]) {
    uint8_t buf[64] = {0};
    if (len) memcpy(buf, p, len);   // p may be null for the empty message
    memcpy(w, buf, 64);
}
struct CV { uint32_t v[8]; };
inline CV chunk_cv(const uint8_t *p, size_t len, uint64_t counter, bool root) {
    CV cv;
    memcpy(cv.v, IV, 32);
    size_t nblocks = len ? (len + 63) / 64 : 1;
    for (size_t i = 0; i < nblocks; ++i) {
        size_t bl = (i == nblocks - 1) ? len - 64 * i : 64;
        uint32_t w[16], fl = (i == 0 ? CHUNK_START : 0) | (i == nblocks - 1 ? CHUNK_END : 0);
        if (root && i == nblocks - 1) fl |= ROOT;
        words(p + 64 * i, bl, w);
        uint32_t out[8];
        compress(cv.v, w, counter, (uint32_t)bl, fl, out);
        memcpy(cv.v, out, 32);
    }
    return cv;
}
inline CV merge(const std::vector<CV> &nodes, size_t lo, size_t hi, bool root) {
    if (hi - lo == 1) return nodes[lo];
    size_t split = 1;
    while (split * 2 < hi - lo) split *= 2;
    CV l = merge(nodes, lo, lo + split, false), r = merge(nodes, lo + split, hi, false);
    uint32_t blk[16];
    memcpy(blk, l.v, 32);
    memcpy(blk + 8, r.v, 32);
    CV out;
    compress(IV, blk, 0, 64, PARENT | (root ? ROOT : 0), out.v);
    return out;
}
}  // namespace b3

inline Bytes blake3(const Bytes &data) {
    size_t nchunks = data.size() ? (data.size() + 1023) / 1024 : 1;
    b3::CV res;
    if (nchunks == 1) {
        res = b3::chunk_cv(data.data(), data.size(), 0, true);
    } else {
        std::vector<b3::CV> cvs(nchunks);
        for (size_t i = 0; i < nchunks; ++i)
            cvs[i] = b3::chunk_cv(data.data() + 1024 * i, i == nchunks - 1 ? data.size() - 1024 * i : 1024, i, false);
        res = b3::merge(cvs, 0, nchunks, true);
    }
    Bytes out(32);
    memcpy(out.data(), res.v, 32);
    return out;
}

// ---------------------------------------------------------------------- STROBE-128 / Merlin
class Strobe128 {
    static const int R = 166;
    uint8_t st[200];
    int pos = 0, pos_begin = 0, cur_flags = 0;
    void f() {
        uint64_t lanes[25];
        memcpy(lanes, st, 200);
        keccak_f1600(lanes);
        memcpy(st, lanes, 200);
    }
    void run_f() {
        st[pos] ^= (uint8_t)pos_begin;
        st[pos + 1] ^= 0x04;
        st[R + 1] ^= 0x80;
        f();
        pos = pos_begin = 0;
    }
    void absorb(const uint8_t *d, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            st[pos++] ^= d[i];
            if (pos == R) run_f();
        }
    }
    void begin_op(int flags, bool more) {
        if (more) return;
        uint8_t hdr[2] = {(uint8_t)pos_begin, (uint8_t)flags};
        pos_begin = pos + 1;
        cur_flags = flags;
        absorb(hdr, 2);
        if ((flags & (4 | 32)) && pos != 0) run_f();
    }

public:
    explicit Strobe128(const char *label) {
        memset(st, 0, sizeof(st));
        const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
        memcpy(st, hdr, 6);
        memcpy(st + 6, "STROBEv1.0.2", 12);
        f();
        meta_ad((const uint8_t *)label, strlen(label), false);
    }
    void meta_ad(const uint8_t *d, size_t n, bool more) { begin_op(16 | 2, more); absorb(d, n); }
    void ad(const uint8_t *d, size_t n, bool more) { begin_op(2, more); absorb(d, n); }
    void prf(uint8_t *out, size_t n) {
        begin_op(1 | 2 | 4, false);
        for (size_t i = 0; i < n; ++i) {
            out[i] = st[pos];
            st[pos++] = 0;
            if (pos == R) run_f();
        }
    }
};

class MerlinTranscript {  // merlin::Transcript
    Strobe128 s;

public:
    explicit MerlinTranscript(const std::string &label) : s("Merlin v1.0") { append_message("dom-sep", (const uint8_t *)label.data(), label.size()); }
    void append_message(const char *label, const uint8_t *msg, size_t n) {
        uint32_t len = (uint32_t)n;
        s.meta_ad((const uint8_t *)label, strlen(label), false);
        s.meta_ad((const uint8_t *)&len, 4, true);
        s.ad(msg, n, false);
    }
    void challenge_bytes(const char *label, uint8_t *out, size_t n) {
        uint32_t len = (uint32_t)n;
        s.meta_ad((const uint8_t *)label, strlen(label), false);
        s.meta_ad((const uint8_t *)&len, 4, true);
        s.prf(out, n);
    }
};

}  // namespace pmhost
