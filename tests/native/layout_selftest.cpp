// CPU-only checks of the PM_SHARD_VECTOR layout (polymath_amd/host/layout.hpp): index maps are bijections, every rank's
// quotient segments tile the numerator index space exactly once, the MSM piece lists cover every pair once (plus the
// documented N^2 extra (2 r1 u_{e-1}, X_e) pairs of [c]), and the counts the prover assumes hold.
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
#include "../../polymath_amd/host/layout.hpp"
using namespace pmlayout;

int main() {
    int fails = 0, cases = 0;
    const uint64_t ns[] = {4, 16, 64, 256, 4096, 1u << 16};
    for (uint64_t n : ns)
        for (uint32_t N : {1u, 2u, 4u, 8u, 16u}) {
            if (!layout_ok(n, N)) continue;
            ++cases;
            const uint64_t m0 = 2, mw = n / 3 + 1, nr = n / 2 - m0 - (n > 16 ? 3 : 0);   // rows 2(m0 + nr) <= n
            const KeyShape ks = key_shape(n, m0, mw, nr);
            const uint64_t len = numerator_len(n);
            std::vector<int> ce(n, 0), ee(n, 0);
            std::vector<int> cover(len, 0);
            std::map<uint64_t, int> base_hits_a, base_hits_c, base_hits_d;
            uint64_t tot_a = 0, tot_c = 0, tot_d = 0;
            for (uint64_t max_seg : {(uint64_t)8, (uint64_t)1 << 13}) {
                std::fill(cover.begin(), cover.end(), 0);
                uint64_t dsum = 0;
                for (uint32_t q = 0; q < N; ++q) {
                    const std::vector<Segment> segs = quotient_segments(n, N, q, max_seg);
                    uint64_t qoff = 0;
                    for (const auto &g : segs) {
                        if (g.b <= g.a || g.b - g.a > max_seg || g.qoff != qoff) { ++fails; printf("bad segment n=%llu N=%u\n", (unsigned long long)n, N); }
                        for (uint64_t k = g.a; k < g.b; ++k) cover[k]++;
                        qoff += (g.b - g.a) - (g.a == 0 ? 1 : 0);
                    }
                    for (const auto &pc : pieces_d(ks, segs)) dsum += pc.count;
                }
                for (uint64_t k = 0; k < len; ++k) if (cover[k] != 1) { ++fails; printf("index %llu covered %d times (n=%llu N=%u)\n", (unsigned long long)k, cover[k], (unsigned long long)n, N); break; }
                if (dsum != len - 1) { ++fails; printf("d pairs %llu != %llu\n", (unsigned long long)dsum, (unsigned long long)(len - 1)); }
            }
            for (uint32_t q = 0; q < N; ++q) {
                const Layout L = make_layout(n, N, q);
                for (uint64_t p = 0; p < L.m; ++p) {
                    const uint64_t k = coeff_global(L, p), i = eval_global(L, p);
                    if (k >= n || i >= n || coeff_owner(L, k) != q || coeff_local(L, k) != p || eval_owner(L, i) != q || eval_local(L, i) != p) { ++fails; printf("map mismatch\n"); }
                    else { ce[k]++; ee[i]++; }
                }
                for (const auto &pc : pieces_a(ks, L)) { tot_a += pc.count; for (uint64_t t = 0; t < pc.count; ++t) base_hits_a[pc.cat_lo + t]++; }
                for (const auto &pc : pieces_c(ks, L)) { tot_c += pc.count; for (uint64_t t = 0; t < pc.count; ++t) base_hits_c[pc.cat_lo + t]++; }
                // the counts prove_sharded.hip assumes
                uint64_t ca = 0, cc = 0;
                for (const auto &pc : pieces_a(ks, L)) ca += pc.count;
                for (const auto &pc : pieces_c(ks, L)) cc += pc.count;
                const uint64_t zl = ztail_lo(ks.Lz, N, q), zh = ztail_lo(ks.Lz, N, q + 1);
                if (ca != L.m + (q == 0 ? 2 : 0)) { ++fails; printf("a count\n"); }
                if (cc != (zh - zl) + (L.m - (q == N - 1 ? 1 : 0)) + (uint64_t)N * (L.B + 1) + (q == 0 ? 5 : 0)) { ++fails; printf("c count\n"); }
            }
            for (uint64_t k = 0; k < n; ++k) if (ce[k] != 1 || ee[k] != 1) { ++fails; printf("not a bijection\n"); break; }
            if (tot_a != n + 2) { ++fails; printf("a total\n"); }
            if (tot_c != ks.Lz + (n - 1) + n + (uint64_t)N * N + 5) { ++fails; printf("c total %llu\n", (unsigned long long)tot_c); }
            // [a]: x_powers[0..n) and y_alpha[0..2) once; [c]: lcs, zh once, x_powers[k] once or twice (block ends), ya[0..3), yg[0..2)
            for (auto &kv : base_hits_a) if (kv.second != 1) { ++fails; printf("a base twice\n"); break; }
            uint64_t twice = 0;
            for (auto &kv : base_hits_c) {
                const bool xp = kv.first >= ks.off_xp && kv.first < ks.off_ya;
                if (kv.second == 2 && xp) ++twice;
                else if (kv.second != 1) { ++fails; printf("c base hit %d times\n", kv.second); break; }
            }
            if (twice != (uint64_t)N * N - 1) { ++fails; printf("c shared block-end bases %llu (n=%llu N=%u)\n", (unsigned long long)twice, (unsigned long long)n, N); }
        }
    printf("layout selftest: %d cases, %d failures\n", cases, fails);
    return fails ? 1 : 0;
}
