// Internal declarations shared by the translation units of libpolymath_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/polymath_hip.h"
#include "../host/layout.hpp"
#include "ec.cuh"

namespace pm {

// ---- error plumbing: every HIP failure becomes PM_ERR_HIP with a message, never an abort ----
struct pm_error_sink {
    std::string msg;
};

#define PM_HIP(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + \
                         std::to_string(__LINE__);                                               \
            return PM_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)

#define PM_TRY(expr)                  \
    do {                              \
        int _s = (expr);              \
        if (_s != PM_OK) return _s;   \
    } while (0)

// Growable device buffer owned by a context (never shrinks: the prover reuses its workspaces).
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            if (e != hipSuccess) return e;
            p = nullptr;
            bytes = 0;
        }
        size_t want = need + need / 8;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, need);
            want = need;
        }
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T *as() const { return (T *)p; }
};

enum TimingSlot {
    T_WITNESS_MAP = 0,
    T_NTT = 1,
    T_POLY = 2,
    T_MSM_SORT = 3,
    T_MSM_ACCUMULATE = 4,
    T_MSM_REDUCE = 5,
    T_MSM_TOTAL = 6,
    T_PHASE = 7,
    T_NUM_SLOTS = 8
};

// What one bucket pipeline hands from its sort to its accumulation and from there to the reduction.
struct MsmSet {
    DevBuf sorted, counts, bucket_off, task_off, order, partials, task_cnt;
};
struct MsmWorkspace {
    MsmSet set;
    DevBuf digits, cursor, wsum, region, sub, digits2, len_bins, block_cnt, hot;
};

// Window tables of a resident base vector (setup.hip: tables_build): point (w, i) = 2^(c w) P_i lives at
// index w * stride + i of the table array.  c == 0: no tables (per-window Pippenger).
// Windows are BALANCED: the 256 scalar bits (255 + signed-digit carry) are split into nwin windows of
// width c or c - 1 (wider ones first), so no window is narrow -- with one shared bucket set a short top
// window would pile all of its digits into a few hot buckets.
struct MsmTables {
    unsigned c = 0, nwin = 0;        // c = widest window; buckets = 2^(c-1)
    unsigned off[33] = {0};          // bit offset of window w (off[nwin] = 256)
    unsigned char width[32] = {0};   // bits of window w
    size_t stride = 0;               // points per window
    size_t base_index = 0;           // first point of this MSM inside window 0
    const unsigned char *inf = nullptr;   // device: one flag byte per point of window 0 (1 = point at infinity)
    const void *table = nullptr;          // device: TablePoint<C>[nwin][stride]
    // WIDE mode (table == nullptr, wide == true): no tables in HBM -- every window keeps its OWN set of 2^(c-1) buckets (key =
    // w 2^(c-1) + bucket), the points are gathered from the plain base array, the nwin window sums are combined by a Horner
    // chain of doublings on the host.  Same sort / accumulate / reduce kernels as the table mode, so c can be 19-20 (13-14
    // additions per pair) where the LDS-histogram pipeline of one-shot MSMs is limited to c = 16.
    bool wide = false;
};

struct TwiddleCache {
    int curve = -1;
    unsigned log_n = 0;
    unsigned long long stamp = 0;   // last use (LRU eviction)
    DevBuf fwd, inv;          // n/2 twiddles each, standard Montgomery form
    DevBuf fwd_int, inv_int;  // the same powers in the reduced-radix internal form (ntt.hip butterflies): dense 8 x u32 for the
                              // 32-bit-limb tile kernels, 10 x u32 28-bit-limb records for the 28-bit ones (ntt_l28_domain)
};

}  // namespace pm

struct pm_bases {
    int curve;
    int device;
    size_t len;
    void *d_points;  // Affine<C>[len], internal Montgomery form
    void *d_inf;     // infinity flags of the table set (tables.inf)
    void *d_table;   // TablePoint<C>[nwin][len] after pm_bases_precompute (tables.table)
    pm::MsmTables tables;
};

struct pm_pk {
    int curve, device;
    uint64_t n, m0, mw, nr, sigma;
    unsigned log_n;
    uint64_t omega[4];
    int shard_rank, shard_count;
    // R1CS matrices on the device (CSR; duplicates of a column inside a row removed, see api)
    uint64_t *d_rowptr[3];
    uint32_t *d_col[3];
    uint64_t *d_val[3];
    uint64_t nnz[3];
    // all base vectors in one device allocation, ordered
    //   [uj_wj_lcs | x_powers_zh | x_powers | y_alpha | y_gamma | y_gamma_z]
    // so that the pair ranges of the merged MSMs are contiguous (DESIGN.md "MSM merging").
    void *d_bases;
    uint64_t base_len[PM_NUM_BASE_VECS];  // logical lengths (whole key)
    uint64_t seg_off[PM_NUM_BASE_VECS];   // element offset of each vector inside the LOGICAL concatenation
    uint64_t total_points;                // logical
    // shard: this device holds logical points [res_lo[k], res_hi[k]) of MSM k (0 = a, 1 = c, 2 = d)
    // stored at d_bases + res_dev_off[k]
    uint64_t res_lo[3], res_hi[3], res_dev_off[3];
    uint64_t msm_lo[3], msm_len[3];  // logical pair range of each merged MSM inside the concatenation
    // Generalisation of [res_lo, res_hi): MSM k's resident pairs are `pieces[k]` (ranges of the logical concatenation),
    // stored back to back at d_bases + res_dev_off[k]; res_cnt[k] = their total.  layout 0 (PM_SHARD_PAIRS): one
    // contiguous piece per MSM, the prover's scalar vectors are whole and this rank reads them at res_lo[k].
    // layout 1 (PM_SHARD_VECTOR, host/layout.hpp): the pieces follow the blocked coefficient layout, the prover builds
    // only this rank's scalars (in piece order) and `segs` is this rank's part of the quotient index space.
    int layout;
    std::vector<pmlayout::Piece> pieces[3];
    uint64_t res_cnt[3];
    std::vector<pmlayout::Segment> segs;
    uint64_t max_seg;
    uint64_t max_piece = (uint64_t)1 << 27;   // msm_max_piece of the creating context (pm_pk_msm_plan reports plans without a context)
    struct SegRef { uint64_t a, b; uint32_t rank, idx; };
    std::vector<SegRef> all_segs;    // the segments of ALL ranks in increasing index order (the carry chain of the division scan)
    size_t seg_slots;                // max over ranks of the segment count
    void *d_segs;                    // device copy of `segs`
    void *d_all_segs = nullptr;      // device copy of `all_segs`
    // Window tables, ONE SET PER MERGED MSM (its own window width: the optimum depends on the pair count):
    // d_tab[k] holds [nwin_k][res_hi[k] - res_lo[k]] points, window 0 being a copy of the MSM's resident slice.
    pm::MsmTables tables[3];
    void *d_tab[3], *d_tab_inf[3];   // tables and their infinity flags (tables[k].inf)
};

// One persistent host thread per context that runs a helper job (the overlapped [a]_1 MSM of prove.hip): created on first
// use, parked on a condition variable between proofs -- a std::thread per proof costs 30-50 us, which matters once a
// proof's share of an 8-GPU job is ~15 ms.
struct pm_worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, stop = false;
    bool submit(std::function<void()> fn) {      // false: the thread could not be created (caller runs the job inline)
        std::unique_lock<std::mutex> lk(mu);
        if (!th.joinable()) {
            try {
                th = std::thread([this] {
                    std::unique_lock<std::mutex> l(mu);
                    for (;;) {
                        cv.wait(l, [this] { return has_job || stop; });
                        if (stop) return;
                        std::function<void()> f = std::move(job);
                        has_job = false;
                        l.unlock();
                        f();
                        l.lock();
                        done = true;
                        cv.notify_all();
                    }
                });
            } catch (const std::system_error &) {
                return false;
            }
        }
        job = std::move(fn);
        has_job = true;
        done = false;
        cv.notify_all();
        return true;
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
    }
    ~pm_worker() {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this] { return done; });
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

// pm_option values of a context (include/polymath_hip.h); defaults and their environment names in api.hip: options_defaults
struct pm_options {
    long long v[PM_NUM_OPTIONS];
};

struct PendingTimer {
    int slot;
    hipEvent_t a, b;
};

struct pm_ctx {
    int device;
    pm_options opt;           // pm_ctx_set_option; the helper context `aux` carries a copy
    std::vector<PendingTimer> pending_timers;
    std::vector<hipEvent_t> event_pool;   // recycled stage-timer events: ~50 create / destroy pairs per proof are 0.2 ms of host time
    hipStream_t stream;
    std::string err;
    double timing_ms[pm::T_NUM_SLOTS];
    hipEvent_t ev_sc_a;       // recorded when the [a]_1 scalars are ready (prove.hip: the helper stream waits on it)
    pm::MsmWorkspace msm;
    pm::TwiddleCache tw[8];   // the sharded prover works with log m, log n and log 2n tables of both directions
    unsigned long long tw_clock;
    pm::DevBuf scratch, flags, ntt_tmp;   // ntt_tmp: the out-of-place first / last passes of ntt_run
    void *h_pinned;           // 4 KiB of pinned host memory: the asynchronous MSM's result slot (msm.hip: msm_begin / msm_end)
    int msm_async;            // 0 none pending, 1 enqueued (msm_end synchronises), 2 ran synchronously (result parked in the slot)
    bool ntt_lds_attr[2];     // ntt.hip: the tile kernels' dynamic-LDS limit has been raised on this context's device (per curve id)
    pm_comm *comm;            // this rank's communicator (pm_ctx_set_comm); null on single-GPU contexts
    pm_worker worker;         // runs the helper context's MSM concurrently with this context's own work
    pm_ctx *aux;              // helper context (own stream + MSM workspace) for the second of two concurrent MSMs
    pm::DevBuf fb_table[2];   // setup.hip: 8-bit-window multiples of the G1 generator, per curve id (built on first use)
    // proof in flight
    const pm_pk *pk;
    int phase;
    std::vector<uint8_t> seg_all;   // sharded prover: phase 2's exchanged records ([u(x1) partial | P_s | Q_s] per rank) for phase 3
    uint64_t x1_host[4];      // ... and the x1 they were taken at
    uint64_t ra_host[8];      // r_a of the proof in flight (phase 3's numerator constants need it on the host)
    bool keep_timings;   // pm_host_prove: the stage slots accumulate over the three phases of one proof
    bool lazy_timings;   // ... and are read by pm_last_timings instead of at the end of every phase (timing_flush)
    pm::DevBuf xw, ue, we, u, w, wit_u, u2, sc_a, sc_c, quotient, ztail, lvl[6], ra;
    // PM_SHARD_VECTOR prover (prove_sharded.hip): transform temporaries, halo coefficients, roots of the cross-rank butterfly
    pm::DevBuf sh_a, sh_b, sh_c, halo, shard_roots;
    uint64_t shard_roots_n;
    uint32_t shard_roots_N;
    int shard_roots_curve;
};

namespace pm {

// Developer aid: PM_PROFILE_HOST=1 prints, at the end of every phase of the sharded prover, how the HOST thread's wall time of that
// phase splits over labelled sections (stderr; one line per phase and rank).  What a kernel trace cannot show: waits, collectives,
// host arithmetic.  Costs two clock reads per section when off.
struct HostProfile {
    bool on;
    const char *phase;
    int rank;
    std::chrono::steady_clock::time_point t0, last;
    std::vector<std::pair<const char *, double>> parts;
    HostProfile(const char *ph, int r) : phase(ph), rank(r) {
        static const bool enabled = [] { const char *e = getenv("PM_PROFILE_HOST"); return e && e[0] == '1'; }();
        on = enabled;
        if (on) t0 = last = std::chrono::steady_clock::now();
    }
    void mark(const char *label) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        parts.emplace_back(label, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
    ~HostProfile() {
        if (!on) return;
        std::string line = std::string("[pm host rank ") + std::to_string(rank) + "] " + phase + ":";
        char buf[96];
        for (auto &p : parts) { snprintf(buf, sizeof buf, " %s=%.3f", p.first, p.second); line += buf; }
        snprintf(buf, sizeof buf, " | total=%.3f ms\n", std::chrono::duration<double, std::milli>(last - t0).count());
        line += buf;
        fputs(line.c_str(), stderr);
    }
};

// ---- per-curve entry points implemented in the .hip translation units -----------------------
template <class C>
int ntt_run(pm_ctx *ctx, Fp<typename C::FrP> *d_data, unsigned log_n, bool inverse);
// device table of omega_{2^log_n}^j (or its inverse), j < 2^(log_n - 1); cached per context
template <class C>
int twiddles_get(pm_ctx *ctx, unsigned log_n, bool inv_dir, const Fp<typename C::FrP> **out);

// tables == nullptr (or c == 0): d_bases points at the MSM's first base.  Otherwise d_bases is unused, the
// points come from tables->table and tables->base_index locates the MSM's first base inside window 0.
template <class C>
int msm_run(pm_ctx *ctx, const Affine<C> *d_bases, const Fp<typename C::FrP> *d_scalars, size_t len,
            Affine<C> *h_out, int *h_inf, const MsmTables *tables = nullptr);

// the same in two halves (msm.hip): enqueue on ctx->stream without waiting / wait and convert
template <class C>
int msm_begin(pm_ctx *ctx, const Affine<C> *d_bases, const Fp<typename C::FrP> *d_scalars, size_t len, const MsmTables *tables);
template <class C>
int msm_end(pm_ctx *ctx, Affine<C> *h_out, int *h_inf);
template <class C>
int msm_resident_begin(pm_ctx *ctx, const pm_pk *pk, int which, const Fp<typename C::FrP> *d_scalars, uint64_t lo = 0, uint64_t count = 0);
template <class C>
int msm_resident_end(pm_ctx *ctx, uint64_t *out_xy, int *out_inf);

// msm_reduce.hip: the bucket reduction of the table-mode MSM (one set of NB >= 4096 buckets whose task partials sit in ctx->msm),
// three launches on ctx->stream; *out = the sum sum_b (b + 1) B_b, internal form, inside the workspace.
template <class C>
int reduce_two_level(pm_ctx *ctx, size_t NB, XYZZ<C> **out, unsigned nsets = 1, size_t bucket0 = 0);   // bucket0: first bucket of the first set

// Wide mode: buckets of the set of a window that is one bit narrower than the plan's widest (c bits; the 256 % nwin wider windows
// come first): half of 2^(c-1) where that is still whole 2^15-bucket sort regions.  Planner (setup.hip: wide_plan), driver and
// kernels (msm.hip: win_base) agree through this.
inline size_t wide_narrow_buckets(unsigned nwin, unsigned c) {
    const size_t nb1 = (size_t)1 << (c - 1);
    return (256 % nwin != 0 && (nb1 >> 1) >= ((size_t)1 << 15)) ? nb1 >> 1 : nb1;
}
inline size_t wide_total_buckets(unsigned nwin, unsigned c) {
    const size_t nb1 = (size_t)1 << (c - 1), nbn = wide_narrow_buckets(nwin, c);
    return nbn == nb1 ? nb1 * nwin : nb1 * (256 % nwin) + nbn * (nwin - 256 % nwin);
}

// One bucket pipeline covers at most this many pairs (sorted-entry positions are u32: windows x pairs < 2^32); longer
// MSMs run in pieces summed on the host.  2^27 in production; PM_OPT_MSM_MAX_PIECE_LOG (developer / test knob) lowers it
// so that the piece-split path can be exercised at small sizes.
inline size_t msm_max_piece(const pm_ctx *ctx) {
    const long long lg = ctx->opt.v[PM_OPT_MSM_MAX_PIECE_LOG];
    return (size_t)1 << (lg < 4 ? 4 : lg > 27 ? 27 : lg);
}

// choose c and the number of windows for a key whose longest MSM has `max_len` pairs
void msm_plan_query(size_t len, unsigned scalar_bits, unsigned *nwin, unsigned *c);
MsmTables tables_plan(size_t total_pairs, unsigned n_msm, size_t resident_points, unsigned scalar_bits, unsigned force_c = 0);
// the plan of an MSM that gets no tables (MsmTables::wide); c == 0 if none applies (short MSMs: the per-window pipeline)
MsmTables wide_plan(size_t piece, unsigned force_c = 0);
// window 0 of d_table <- the `count` internal-form affine points at d_points; then windows 1..nwin-1
template <class C>
struct TablePoint;   // fq28.cuh: 128-byte, 28-bit-limb record
template <class C>
int tables_build(pm_ctx *ctx, const Affine<C> *d_points, TablePoint<C> *d_table, size_t count, const MsmTables &t);
// flags[i] = 1 iff points[i] is the point at infinity (all-zero x, y)
template <class C>
int infinity_flags(pm_ctx *ctx, const Affine<C> *d_points, size_t count, unsigned char *d_flags);

template <class C>
int bases_generate_multiples(pm_ctx *ctx, size_t len, Affine<C> *d_out);

template <class C>
int fixed_base_batch(pm_ctx *ctx, const Fp<typename C::FrP> *d_scalars, size_t len, Affine<C> *d_out);

// Device base vectors are kept in the INTERNAL Montgomery radix of the reduced-radix accumulate kernel
// (fq28.cuh); this converts a device array in place at the API boundary (upload / generate / export).
template <class C>
int bases_convert(pm_ctx *ctx, Affine<C> *d_points, size_t len, bool to_internal);

// setup.hip: the uj_wj_lcs scalars of generator.rs:112-136 on the device (Lagrange coefficients at x + the sparse pass over the
// key's CSR matrices); `lagrange` and `work` are scratch the caller releases
template <class C>
int lcs_scalars(pm_ctx *ctx, const pm_pk *pk, const Fp<typename C::FrP> &x, const Fp<typename C::FrP> &omega, const Fp<typename C::FrP> &kscale,
                const Fp<typename C::FrP> &y_gamma, const Fp<typename C::FrP> &y_to_minus_alpha, DevBuf &lagrange, DevBuf &work,
                Fp<typename C::FrP> *d_lcs);

template <class C>
int powers_fill(pm_ctx *ctx, Fp<typename C::FrP> *d_out, size_t count, const Fp<typename C::FrP> &scale,
                const Fp<typename C::FrP> &x);

// MSM `which` (0 = a, 1 = c, 2 = d) over this rank's resident pairs; d_scalars in the order of pk->pieces[which]
template <class C>
int msm_resident(pm_ctx *ctx, const pm_pk *pk, int which, const Fp<typename C::FrP> *d_scalars, uint64_t *out_xy, int *out_inf);

template <class C>
int prove_phase1_impl(pm_ctx *ctx, const pm_pk *pk, const uint64_t *x, const uint64_t *w, const uint64_t *r_a,
                      uint64_t *a_xy, int *a_inf, uint64_t *c_xy, int *c_inf, bool assignment_on_device);
template <class C>
int prove_phase2_impl(pm_ctx *ctx, const uint64_t *x1, uint64_t *u_at_x1);
template <class C>
int prove_phase3_impl(pm_ctx *ctx, const uint64_t *x1, const uint64_t *x2, const uint64_t *a_at_x1,
                      const uint64_t *c_at_x1, uint64_t *d_xy, int *d_inf);

template <class C>
int prove_phase1_sharded(pm_ctx *ctx, const pm_pk *pk, const uint64_t *x, const uint64_t *w, const uint64_t *r_a, uint64_t *a_xy, int *a_inf,
                         uint64_t *c_xy, int *c_inf, bool assignment_on_device);
template <class C>
int prove_phase2_sharded(pm_ctx *ctx, const uint64_t *x1, uint64_t *u_at_x1);
template <class C>
int prove_phase3_sharded(pm_ctx *ctx, const uint64_t *x1, const uint64_t *x2, const uint64_t *a_at_x1, const uint64_t *c_at_x1, uint64_t *d_xy,
                         int *d_inf);

// The helper context of `ctx` (own stream, own MSM workspace), created on first use with a copy of ctx's options; nullptr if it
// cannot be created (the callers then run their two jobs back to back).
inline pm_ctx *ctx_aux(pm_ctx *ctx) {
    if (!ctx->aux && pm_ctx_create(ctx->device, &ctx->aux) != PM_OK) ctx->aux = nullptr;
    if (ctx->aux) ctx->aux->opt = ctx->opt;
    return ctx->aux;
}

// The context's pinned host staging: [0, 4096) the asynchronous MSM's result slots (msm.hip) and phase 3's remainder, then
// PINNED_STAGE_BYTES for small device-to-host records that must land without stalling the host (prove_sharded.hip: phase 1's
// flags and halo coefficients -- a pageable destination makes hipMemcpyAsync wait for the stream).  nullptr on failure.
constexpr size_t PINNED_SLOTS_BYTES = 4096, PINNED_STAGE_BYTES = 32768;
inline void *ctx_pinned(pm_ctx *ctx) {
    if (!ctx->h_pinned && hipHostMalloc(&ctx->h_pinned, PINNED_SLOTS_BYTES + PINNED_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) ctx->h_pinned = nullptr;
    return ctx->h_pinned;
}

inline void timing_reset(pm_ctx *ctx) {
    for (int i = 0; i < T_NUM_SLOTS; ++i) ctx->timing_ms[i] = 0;
    // timers nobody asked for (lazy mode, below): their events go back to the pool unread -- the work they bracket finished with the
    // call that recorded them
    for (auto &t : ctx->pending_timers) { ctx->event_pool.push_back(t.a); ctx->event_pool.push_back(t.b); }
    ctx->pending_timers.clear();
}

// hipEvent stage timers (replace start_timer!/end_timer!, prover.rs:32-61).  Events are only
// recorded while a call runs; elapsed times are read in timing_flush() after the call's final
// stream synchronisation, so timing adds no host round trips.
struct StageTimer {
    pm_ctx *ctx;
    int slot;
    hipEvent_t a, b;
    bool ok;
    static bool take(pm_ctx *c, hipEvent_t *e) {
        if (!c->event_pool.empty()) { *e = c->event_pool.back(); c->event_pool.pop_back(); return true; }
        return hipEventCreate(e) == hipSuccess;
    }
    hipStream_t stream;
    StageTimer(pm_ctx *c, int s, hipStream_t on = nullptr) : ctx(c), slot(s), ok(false), stream(on ? on : c->stream) {
        if (!take(c, &a)) return;
        if (!take(c, &b)) { c->event_pool.push_back(a); return; }
        ok = hipEventRecord(a, stream) == hipSuccess;
        if (!ok) { c->event_pool.push_back(a); c->event_pool.push_back(b); }
    }
    void stop() {
        if (!ok) return;
        ok = false;
        (void)hipEventRecord(b, stream);
        ctx->pending_timers.push_back(PendingTimer{slot, a, b});
    }
    ~StageTimer() { stop(); }
};

// Reading ~18 event pairs costs ~40 us of host time per phase, with the GPU idle between the phases of a proof: pm_host_prove sets
// lazy_timings, the phases then leave their timers pending and pm_last_timings reads them when (if) somebody asks.
inline void timing_flush_now(pm_ctx *ctx) {
    for (auto &t : ctx->pending_timers) {
        float ms = 0;
        if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess)
            ctx->timing_ms[t.slot] += ms;
        ctx->event_pool.push_back(t.a);
        ctx->event_pool.push_back(t.b);
    }
    ctx->pending_timers.clear();
}
inline void timing_flush(pm_ctx *ctx) {
    if (!ctx->lazy_timings) timing_flush_now(ctx);
}
// the helper context of the second MSM pipeline: reset / hand its stage times (or, lazily, its pending timers) to the owner
inline void timing_reset_aux(pm_ctx *ctx, pm_ctx *aux) {
    for (int i = 0; i < T_NUM_SLOTS; ++i) aux->timing_ms[i] = 0;   // its pending timers stay: the sharded prover's w transform ran on it
    aux->lazy_timings = ctx->lazy_timings;
}
inline void timing_absorb_aux(pm_ctx *ctx, pm_ctx *aux) {
    for (int s = 0; s < T_NUM_SLOTS; ++s) { ctx->timing_ms[s] += aux->timing_ms[s]; aux->timing_ms[s] = 0; }
    // lazy mode: the helper's unread timers (and their events) become the owner's; the helper's pool gets as many events back from the
    // owner's, or it would create new ones every proof while the owner's pool grows.  Called by the owner's thread, the helper idle.
    for (auto &t : aux->pending_timers) {
        ctx->pending_timers.push_back(t);
        for (int k = 0; k < 2 && !ctx->event_pool.empty(); ++k) { aux->event_pool.push_back(ctx->event_pool.back()); ctx->event_pool.pop_back(); }
    }
    aux->pending_timers.clear();
}

// Declared BEFORE a call's StageTimers: whatever path the call returns by (error statuses included), the stage timers
// that were started are read and their events destroyed -- stale entries would otherwise be added to the next
// proof's slots.
struct TimingGuard {
    pm_ctx *ctx;
    ~TimingGuard() { timing_flush(ctx); if (ctx->aux) timing_flush(ctx->aux); }
};

}  // namespace pm
