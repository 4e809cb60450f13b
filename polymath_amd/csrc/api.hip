// C ABI of libpolymath_hip.so (include/polymath_hip.h): context / key management and dispatch
// on the curve.  All compute is in the .hip kernels; the host code here only moves buffers,
// and removes row-duplicate R1CS entries the reference cannot see (m_at, common.rs:100-105) when a matrix has any.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <thread>

#include "internal.h"
#include "comm.h"
#include "fq28.cuh"
#include "../host/hashes.hpp"

using namespace pm;

extern "C" void pm_host_keccak_f1600(uint64_t state[25]) { pmhost::keccak_f1600(state); }

// ------------------------------------------------------------------------------ helpers
// body(lo, hi, thread) over [0, count) in contiguous chunks on up to 32 host threads (one below 2^16 items).  An exception in a
// worker (bad_alloc) is caught there and re-thrown on the calling thread after the join -- never a std::terminate.
template <class F>
static void parallel_chunks(uint64_t count, F body) {
    unsigned T = std::thread::hardware_concurrency();
    if (T > 32) T = 32;
    if (T < 1 || count < ((uint64_t)1 << 16)) T = 1;
    static const int env_threads = [] { const char *e = getenv("PM_HOST_THREADS"); return e ? std::max(1, atoi(e)) : 0; }();   // read once
    if (env_threads) T = (unsigned)env_threads;
    if (T == 1) { body((uint64_t)0, count, 0u); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> errs(T);
    auto guarded_body = [&](uint64_t lo, uint64_t hi, unsigned t) {
        try { body(lo, hi, t); } catch (...) { errs[t] = std::current_exception(); }
    };
    for (unsigned t = 1; t < T; ++t) {
        try {
            th.emplace_back([=, &guarded_body] { guarded_body(count * t / T, count * (t + 1) / T, t); });
        } catch (const std::system_error &) {            // no more threads: this chunk runs here
            guarded_body(count * t / T, count * (t + 1) / T, t);
        }
    }
    guarded_body((uint64_t)0, count / T, 0u);
    for (auto &x : th) x.join();
    for (auto &e : errs)
        if (e) std::rethrow_exception(e);
}

template <class C>
static void repack_bases(const void *src, size_t stride, size_t len, Affine<C> *dst) {
    const size_t PT = sizeof(Affine<C>);
    const uint8_t *s = (const uint8_t *)src;
    for (size_t i = 0; i < len; ++i) {
        memcpy(&dst[i], s + i * stride, PT);
        if (stride > PT && s[i * stride + PT] != 0) dst[i] = Affine<C>::infinity();  // arkworks `infinity: bool`
    }
}

#define PM_DISPATCH(curve, CALL_BLS, CALL_BN) \
    ((curve) == PM_BLS12_381 ? (CALL_BLS) : (curve) == PM_BN254 ? (CALL_BN) : (int)PM_ERR_INVALID_ARG)

static int set_device(pm_ctx *ctx) {
    PM_HIP(ctx, hipSetDevice(ctx->device));
    return PM_OK;
}

// ------------------------------------------------------------------------------ context
extern "C" int pm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- options (include/polymath_hip.h: pm_option).  The environment gives a NEW context its defaults; it is read here, once per
// context, and nowhere on a proving path.
struct OptionSpec { const char *env; long long def, lo, hi; };
static const OptionSpec OPTION_SPECS[PM_NUM_OPTIONS] = {
    /* PM_OPT_MSM_OVERLAP       */ {"PM_MSM_OVERLAP", 1, 0, 1},
    /* PM_OPT_NTT_OVERLAP       */ {"PM_NTT_OVERLAP", 1, 0, 1},
    /* PM_OPT_TABLES            */ {"PM_TABLES", PM_TABLES_AUTO, PM_TABLES_OFF, PM_TABLES_NO_WIDE},
    /* PM_OPT_MSM_MAX_PIECE_LOG */ {"PM_MSM_MAX_PIECE_LOG", 27, 4, 27},
    /* PM_OPT_MAX_SEG_LOG       */ {"PM_MAX_SEG_LOG", 0, 0, 40},
    /* PM_OPT_INFLIGHT_CONTEXTS */ {"PM_INFLIGHT_CONTEXTS", 1, 1, 64},
    /* PM_OPT_MSM_TASK_LEN      */ {"PM_MSM_SEG", 0, 0, 1 << 20},
    /* PM_OPT_TABLE_WINDOW_BITS */ {"PM_TABLE_C", 0, 0, 24},
};
static void options_defaults(pm_options *o) {
    for (int k = 0; k < PM_NUM_OPTIONS; ++k) {
        long long v = OPTION_SPECS[k].def;
        if (const char *e = getenv(OPTION_SPECS[k].env)) {
            if (k == PM_OPT_TABLES) v = e[0] == 'w' ? PM_TABLES_WIDE : e[0] == '0' ? PM_TABLES_OFF : PM_TABLES_AUTO;   // 0 | 1 | wide
            else v = atoll(e);
            if (v < OPTION_SPECS[k].lo || v > OPTION_SPECS[k].hi) v = OPTION_SPECS[k].def;
        }
        o->v[k] = v;
    }
    if (o->v[PM_OPT_TABLES] == PM_TABLES_AUTO) {
        const char *e = getenv("PM_WIDE");
        if (e && e[0] == '0') o->v[PM_OPT_TABLES] = PM_TABLES_NO_WIDE;
    }
}

extern "C" int pm_ctx_set_option(pm_ctx *ctx, int option, long long value) {
    if (!ctx || option < 0 || option >= PM_NUM_OPTIONS) return PM_ERR_INVALID_ARG;
    if (value < OPTION_SPECS[option].lo || value > OPTION_SPECS[option].hi) return PM_ERR_INVALID_ARG;
    ctx->opt.v[option] = value;
    if (ctx->aux) ctx->aux->opt.v[option] = value;
    return PM_OK;
}
extern "C" int pm_ctx_get_option(const pm_ctx *ctx, int option, long long *value) {
    if (!ctx || !value || option < 0 || option >= PM_NUM_OPTIONS) return PM_ERR_INVALID_ARG;
    *value = ctx->opt.v[option];
    return PM_OK;
}

extern "C" int pm_ctx_create(int device, pm_ctx **out) {
    if (!out) return PM_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PM_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return PM_ERR_INVALID_ARG;
    pm_ctx *ctx = new (std::nothrow) pm_ctx();
    if (!ctx) return PM_ERR_INVALID_ARG;
    ctx->device = device;
    options_defaults(&ctx->opt);
    ctx->pk = nullptr;
    ctx->phase = 0;
    ctx->aux = nullptr;
    ctx->comm = nullptr;
    ctx->tw_clock = 0;
    ctx->shard_roots_n = 0; ctx->shard_roots_N = 0; ctx->shard_roots_curve = -1;
    ctx->ntt_lds_attr[0] = ctx->ntt_lds_attr[1] = false;
    ctx->h_pinned = nullptr;
    ctx->msm_async = 0;
    ctx->keep_timings = false;
    ctx->lazy_timings = false;
    timing_reset(ctx);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return PM_ERR_HIP;
    }
    if (hipEventCreateWithFlags(&ctx->ev_sc_a, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return PM_ERR_HIP;
    }
    *out = ctx;
    return PM_OK;
}

extern "C" void pm_ctx_destroy(pm_ctx *ctx) {
    if (!ctx) return;
    if (ctx->aux) pm_ctx_destroy(ctx->aux);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    timing_flush(ctx);
    MsmWorkspace &m = ctx->msm;
    for (DevBuf *b : {&m.set.sorted, &m.set.counts, &m.set.bucket_off, &m.set.task_off, &m.set.order, &m.set.partials, &m.set.task_cnt}) b->release();
    for (DevBuf *b : {&m.digits, &m.cursor, &m.wsum,
                      &m.region, &m.sub, &m.digits2, &m.len_bins, &m.block_cnt, &m.hot, &ctx->scratch, &ctx->flags, &ctx->xw, &ctx->ue, &ctx->we, &ctx->u, &ctx->w,
                      &ctx->wit_u, &ctx->u2, &ctx->sc_a, &ctx->sc_c, &ctx->quotient, &ctx->ztail, &ctx->ra, &ctx->sh_a, &ctx->sh_b, &ctx->sh_c, &ctx->halo,
                      &ctx->shard_roots, &ctx->ntt_tmp})
        b->release();
    for (auto &b : ctx->lvl) b.release();
    for (auto &b : ctx->fb_table) b.release();
    for (auto &t : ctx->tw) { t.fwd.release(); t.inv.release(); t.fwd_int.release(); t.inv_int.release(); }
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    for (auto &t : ctx->pending_timers) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }   // pm_host_prove leaves them unread (lazy_timings)
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    (void)hipEventDestroy(ctx->ev_sc_a);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char *pm_last_error(const pm_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int pm_last_timings(pm_ctx *ctx, double *ms_out, int n_slots) {
    if (!ctx || !ms_out) return PM_ERR_INVALID_ARG;
    if (!ctx->pending_timers.empty()) {   // pm_host_prove leaves its stage timers unread (internal.h: lazy_timings)
        if (hipSetDevice(ctx->device) != hipSuccess) return PM_ERR_HIP;
        timing_flush_now(ctx);
    }
    for (int i = 0; i < n_slots; ++i) ms_out[i] = i < T_NUM_SLOTS ? ctx->timing_ms[i] : 0.0;
    return PM_OK;
}

// ---------------------------------------------------------------------------------- NTT
template <class C>
static int ntt_host(pm_ctx *ctx, uint64_t *data, unsigned log_n, int inverse) {
    typedef Fp<typename C::FrP> Fr;
    if (log_n > (unsigned)C::TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
    size_t bytes = sizeof(Fr) << log_n;
    PM_HIP(ctx, ctx->scratch.reserve(bytes));
    PM_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    timing_reset(ctx);
    PM_TRY(ntt_run<C>(ctx, ctx->scratch.as<Fr>(), log_n, inverse != 0));
    PM_HIP(ctx, hipMemcpyAsync(data, ctx->scratch.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    timing_flush(ctx);
    return PM_OK;
}

extern "C" int pm_ntt(pm_ctx *ctx, int curve, uint64_t *data, unsigned log_n, int inverse) {
    if (!ctx || !data) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    return PM_DISPATCH(curve, ntt_host<BlsCurve>(ctx, data, log_n, inverse), ntt_host<BnCurve>(ctx, data, log_n, inverse));
}

template <class C>
static int ntt_dev(pm_ctx *ctx, uint64_t *d, unsigned log_n, int inverse) {
    timing_reset(ctx);
    PM_TRY(ntt_run<C>(ctx, (Fp<typename C::FrP> *)d, log_n, inverse != 0));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    timing_flush(ctx);
    return PM_OK;
}

extern "C" int pm_ntt_device(pm_ctx *ctx, int curve, uint64_t *d_data, unsigned log_n, int inverse) {
    if (!ctx || !d_data) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    return PM_DISPATCH(curve, ntt_dev<BlsCurve>(ctx, d_data, log_n, inverse), ntt_dev<BnCurve>(ctx, d_data, log_n, inverse));
}

// ---------------------------------------------------------------------------------- MSM
struct BasesDeleter {   // frees the device allocations with the handle: error paths cannot leak HBM
    void operator()(pm_bases *b) const {
        if (!b) return;
        (void)hipSetDevice(b->device);
        if (b->d_points) (void)hipFree(b->d_points);
        if (b->d_inf) (void)hipFree(b->d_inf);
        if (b->d_table) (void)hipFree(b->d_table);
        delete b;
    }
};
typedef std::unique_ptr<pm_bases, BasesDeleter> BasesPtr;

template <class C>
static int bases_upload_impl(pm_ctx *ctx, const void *bases, size_t stride, size_t len, pm_bases **out) {
    if (stride < sizeof(Affine<C>)) return PM_ERR_INVALID_ARG;
    BasesPtr b(new pm_bases{C::ID, ctx->device, len, nullptr});
    if (len) {
        std::vector<Affine<C>> packed(len);
        repack_bases<C>(bases, stride, len, packed.data());
        PM_HIP(ctx, hipMalloc(&b->d_points, len * sizeof(Affine<C>)));
        PM_HIP(ctx, hipMemcpy(b->d_points, packed.data(), len * sizeof(Affine<C>), hipMemcpyHostToDevice));
        PM_TRY(bases_convert<C>(ctx, (Affine<C> *)b->d_points, len, true));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out = b.release();
    return PM_OK;
}

extern "C" int pm_bases_upload(pm_ctx *ctx, int curve, const void *bases, size_t base_stride, size_t len, pm_bases **out) {
    if (!ctx || !out || (len && !bases)) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    return PM_DISPATCH(curve, bases_upload_impl<BlsCurve>(ctx, bases, base_stride, len, out),
                       bases_upload_impl<BnCurve>(ctx, bases, base_stride, len, out));
}

template <class C>
static int bases_multiples_impl(pm_ctx *ctx, size_t len, pm_bases **out) {
    BasesPtr b(new pm_bases{C::ID, ctx->device, len, nullptr});
    if (len) {
        PM_HIP(ctx, hipMalloc(&b->d_points, len * sizeof(Affine<C>)));
        PM_TRY(bases_generate_multiples<C>(ctx, len, (Affine<C> *)b->d_points));
    }
    *out = b.release();
    return PM_OK;
}

extern "C" int pm_bases_generate_multiples(pm_ctx *ctx, int curve, size_t len, pm_bases **out) {
    if (!ctx || !out) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    return PM_DISPATCH(curve, bases_multiples_impl<BlsCurve>(ctx, len, out), bases_multiples_impl<BnCurve>(ctx, len, out));
}

// device -> host copy of resident (internal-form) points, converted back to standard Montgomery form
template <class C>
static int download_points(pm_ctx *ctx, const void *d_src, size_t len, uint64_t *out_xy) {
    if (!len) return PM_OK;
    PM_HIP(ctx, ctx->scratch.reserve(len * sizeof(Affine<C>)));
    PM_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, d_src, len * sizeof(Affine<C>), hipMemcpyDeviceToDevice, ctx->stream));
    PM_TRY(bases_convert<C>(ctx, ctx->scratch.as<Affine<C>>(), len, false));
    PM_HIP(ctx, hipMemcpyAsync(out_xy, ctx->scratch.p, len * sizeof(Affine<C>), hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

extern "C" int pm_bases_download(pm_ctx *ctx, const pm_bases *b, size_t offset, size_t len, uint64_t *out_xy) {
    if (!ctx || !b || !out_xy || offset + len > b->len) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    size_t pt = b->curve == PM_BLS12_381 ? sizeof(Affine<BlsCurve>) : sizeof(Affine<BnCurve>);
    const void *src = (const uint8_t *)b->d_points + offset * pt;
    return PM_DISPATCH(b->curve, download_points<BlsCurve>(ctx, src, len, out_xy), download_points<BnCurve>(ctx, src, len, out_xy));
}

template <class C>
static int bases_precompute_impl(pm_ctx *ctx, pm_bases *b) {
    if (b->tables.c || !b->len) return PM_OK;
    MsmTables tb = tables_plan(b->len, 1, b->len, (unsigned)C::FrP::BITS, (unsigned)ctx->opt.v[PM_OPT_TABLE_WINDOW_BITS]);
    if (!tb.c) return PM_OK;
    void *d_table = nullptr, *d_flags = nullptr;
    int st = PM_OK;
    if (hipMalloc(&d_table, b->len * tb.nwin * sizeof(TablePoint<C>)) != hipSuccess || hipMalloc(&d_flags, b->len) != hipSuccess) {
        ctx->err = "pm_bases_precompute: out of device memory for the window tables";
        st = PM_ERR_HIP;
    }
    if (st == PM_OK) st = infinity_flags<C>(ctx, (const Affine<C> *)b->d_points, b->len, (unsigned char *)d_flags);
    if (st == PM_OK) st = tables_build<C>(ctx, (const Affine<C> *)b->d_points, (TablePoint<C> *)d_table, b->len, tb);
    if (st != PM_OK) {
        if (d_table) (void)hipFree(d_table);
        if (d_flags) (void)hipFree(d_flags);
        return st;
    }
    b->d_table = d_table;
    b->d_inf = d_flags;
    tb.inf = (const unsigned char *)d_flags;
    tb.table = d_table;
    b->tables = tb;
    return PM_OK;
}

extern "C" int pm_bases_precompute(pm_ctx *ctx, pm_bases *b) {
    if (!ctx || !b) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    return PM_DISPATCH(b->curve, bases_precompute_impl<BlsCurve>(ctx, b), bases_precompute_impl<BnCurve>(ctx, b));
}

extern "C" size_t pm_bases_len(const pm_bases *b) { return b ? b->len : 0; }

extern "C" void pm_bases_free(pm_bases *b) { BasesDeleter()(b); }

template <class C>
static int msm_resident_impl(pm_ctx *ctx, const pm_bases *bases, size_t off, const uint64_t *scalars, int on_device,
                             size_t len, uint64_t *out_xy, int *out_inf) {
    typedef Fp<typename C::FrP> Fr;
    const Fr *d_sc = (const Fr *)scalars;
    if (!on_device && len) {
        PM_HIP(ctx, ctx->scratch.reserve(len * sizeof(Fr)));
        PM_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, scalars, len * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        d_sc = ctx->scratch.as<Fr>();
    }
    Affine<C> r;
    int inf = 1;
    timing_reset(ctx);
    if (bases->tables.c) {
        MsmTables tb = bases->tables;
        tb.base_index = off;
        PM_TRY(msm_run<C>(ctx, (const Affine<C> *)nullptr, d_sc, len, &r, &inf, &tb));
    } else {
        PM_TRY(msm_run<C>(ctx, (const Affine<C> *)bases->d_points + off, d_sc, len, &r, &inf));
    }
    timing_flush(ctx);
    if (inf) memset(out_xy, 0, sizeof(Affine<C>));
    else memcpy(out_xy, &r, sizeof(Affine<C>));
    *out_inf = inf;
    return PM_OK;
}

extern "C" int pm_msm_g1_resident(pm_ctx *ctx, const pm_bases *bases, size_t base_offset, const uint64_t *scalars,
                                  int scalars_on_device, size_t len, uint64_t *out_xy, int *out_inf) {
    if (!ctx || !bases || !out_xy || !out_inf || (len && !scalars)) return PM_ERR_INVALID_ARG;
    if (base_offset + len > bases->len) return PM_ERR_LEN_MISMATCH;  // prover.rs:381
    PM_TRY(set_device(ctx));
    return PM_DISPATCH(bases->curve,
                       msm_resident_impl<BlsCurve>(ctx, bases, base_offset, scalars, scalars_on_device, len, out_xy, out_inf),
                       msm_resident_impl<BnCurve>(ctx, bases, base_offset, scalars, scalars_on_device, len, out_xy, out_inf));
}

extern "C" int pm_msm_g1(pm_ctx *ctx, int curve, const void *bases, size_t base_stride, const uint64_t *scalars,
                         size_t len, uint64_t *out_xy, int *out_inf) {
    if (!ctx || !out_xy || !out_inf || (len && (!bases || !scalars))) return PM_ERR_INVALID_ARG;
    pm_bases *b = nullptr;
    PM_TRY(pm_bases_upload(ctx, curve, bases, base_stride, len, &b));
    int st = pm_msm_g1_resident(ctx, b, 0, scalars, 0, len, out_xy, out_inf);
    pm_bases_free(b);
    return st;
}

template <class C>
static int g1_sum_impl(const uint64_t *pts, const int *infs, size_t count, uint64_t *out_xy, int *out_inf) {
    XYZZ<C> acc = XYZZ<C>::identity();
    for (size_t i = 0; i < count; ++i) {
        Affine<C> a;
        memcpy(&a, (const uint8_t *)pts + i * sizeof(Affine<C>), sizeof(Affine<C>));
        if (infs && infs[i]) continue;
        xyzz_madd<C>(acc, a, false);
    }
    Affine<C> r = xyzz_to_affine<C>(acc);
    int inf = acc.is_identity() ? 1 : 0;
    if (inf) memset(out_xy, 0, sizeof(Affine<C>));
    else memcpy(out_xy, &r, sizeof(Affine<C>));
    *out_inf = inf;
    return PM_OK;
}

extern "C" int pm_g1_sum(int curve, const uint64_t *points_xy, const int *infs, size_t count, uint64_t *out_xy, int *out_inf) {
    if (!out_xy || !out_inf || (count && !points_xy)) return PM_ERR_INVALID_ARG;
    return PM_DISPATCH(curve, g1_sum_impl<BlsCurve>(points_xy, infs, count, out_xy, out_inf),
                       g1_sum_impl<BnCurve>(points_xy, infs, count, out_xy, out_inf));
}

// ------------------------------------------------------------------------- proving key
struct HostCsr {
    std::vector<uint64_t> rowptr;
    std::vector<uint32_t> col;
    std::vector<uint64_t> val;
};

// m_at (common.rs:100-105) only ever sees the FIRST entry of a row with a given column.
static int dedupe_csr(const pm_csr *m, uint64_t nr, uint64_t ncols, HostCsr &out) {
    if (!m || m->nrows != nr || !m->rowptr) return PM_ERR_INVALID_ARG;
    out.rowptr.assign(1, 0);
    for (uint64_t r = 0; r < nr; ++r) {
        size_t start = out.col.size();
        for (uint64_t k = m->rowptr[r]; k < m->rowptr[r + 1]; ++k) {
            if (m->col[k] >= ncols) return PM_ERR_INVALID_ARG;
            bool dup = false;
            for (size_t q = start; q < out.col.size(); ++q)
                if (out.col[q] == m->col[k]) { dup = true; break; }
            if (dup) continue;
            out.col.push_back(m->col[k]);
            for (int i = 0; i < 4; ++i) out.val.push_back(m->val[4 * k + i]);
        }
        out.rowptr.push_back(out.col.size());
    }
    return PM_OK;
}

static void pk_release(pm_pk *pk) {
    if (!pk) return;
    (void)hipSetDevice(pk->device);
    for (int i = 0; i < 3; ++i) {
        if (pk->d_rowptr[i]) (void)hipFree(pk->d_rowptr[i]);
        if (pk->d_col[i]) (void)hipFree(pk->d_col[i]);
        if (pk->d_val[i]) (void)hipFree(pk->d_val[i]);
    }
    if (pk->d_bases) (void)hipFree(pk->d_bases);
    if (pk->d_segs) (void)hipFree(pk->d_segs);
    if (pk->d_all_segs) (void)hipFree(pk->d_all_segs);
    for (int k = 0; k < 3; ++k) {
        if (pk->d_tab[k]) (void)hipFree(pk->d_tab[k]);
        if (pk->d_tab_inf[k]) (void)hipFree(pk->d_tab_inf[k]);
    }
    delete pk;
}

extern "C" void pm_pk_free(pm_pk *pk) { pk_release(pk); }

// Shapes, domain, the logical base concatenation and this shard's resident ranges.
template <class C>
static int pk_init_layout(pm_ctx *ctx, pm_pk *pk, uint64_t m0, uint64_t mw, uint64_t nr, int shard_rank, int shard_count, int layout = PM_SHARD_PAIRS) {
    typedef typename C::FrP P;
    if (m0 < 1 || shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return PM_ERR_INVALID_ARG;
    if (layout != PM_SHARD_PAIRS && layout != PM_SHARD_VECTOR) return PM_ERR_INVALID_ARG;
    pk->layout = layout;
    pk->curve = C::ID;
    pk->device = ctx->device;
    pk->m0 = m0; pk->mw = mw; pk->nr = nr;
    uint64_t rows = 2 * (m0 + nr);                     // common.rs:131-135
    uint64_t n = 1;
    unsigned log_n = 0;
    while (n < rows) { n <<= 1; ++log_n; }             // Radix2EvaluationDomain::new, generator.rs:60
    if (log_n > (unsigned)C::TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
    pk->n = n; pk->log_n = log_n; pk->sigma = n + 3;   // generator.rs:66-70
    Fp<P> w;
    for (int i = 0; i < P::N; ++i) w.l[i] = C::ROOT_MONT[i];
    for (unsigned i = log_n; i < (unsigned)C::TWO_ADICITY; ++i) w = sqr<P>(w);
    memcpy(pk->omega, w.l, 32);
    pk->shard_rank = shard_rank; pk->shard_count = shard_count;
    const uint64_t Lz = 2 * m0 + mw + nr;
    pk->base_len[PM_X_POWERS] = n + 1;
    pk->base_len[PM_X_POWERS_Y_ALPHA] = 3;
    pk->base_len[PM_X_POWERS_Y_GAMMA] = 2;
    pk->base_len[PM_X_POWERS_Y_GAMMA_Z] = 2 * (n - 1) + 8 * pk->sigma + 1;
    pk->base_len[PM_X_POWERS_ZH_BY_Y_ALPHA] = n - 1;
    pk->base_len[PM_UJ_WJ_LCS_BY_Y_ALPHA] = Lz;
    const int order[6] = {PM_UJ_WJ_LCS_BY_Y_ALPHA, PM_X_POWERS_ZH_BY_Y_ALPHA, PM_X_POWERS, PM_X_POWERS_Y_ALPHA,
                          PM_X_POWERS_Y_GAMMA, PM_X_POWERS_Y_GAMMA_Z};
    uint64_t off = 0;
    for (int k = 0; k < 6; ++k) { pk->seg_off[order[k]] = off; off += pk->base_len[order[k]]; }
    pk->total_points = off;
    pk->msm_lo[0] = pk->seg_off[PM_X_POWERS];            pk->msm_len[0] = n + 3;
    pk->msm_lo[1] = 0;                                   pk->msm_len[1] = pk->seg_off[PM_X_POWERS_Y_GAMMA_Z];
    pk->msm_lo[2] = pk->seg_off[PM_X_POWERS_Y_GAMMA_Z];  pk->msm_len[2] = pk->base_len[PM_X_POWERS_Y_GAMMA_Z] - 1;
    for (int k = 0; k < 3; ++k) {
        pk->res_lo[k] = pk->msm_len[k] * (uint64_t)shard_rank / (uint64_t)shard_count;
        pk->res_hi[k] = pk->msm_len[k] * (uint64_t)(shard_rank + 1) / (uint64_t)shard_count;
    }
    if (layout == PM_SHARD_VECTOR) {
        if (!pmlayout::layout_ok(n, (uint32_t)shard_count)) return PM_ERR_INVALID_ARG;   // N a power of two, N^2 | n
        const pmlayout::Layout L = pmlayout::make_layout(n, (uint32_t)shard_count, (uint32_t)shard_rank);
        const pmlayout::KeyShape ks = pmlayout::key_shape(n, m0, mw, nr);
        pk->max_seg = pmlayout::pick_max_seg(n, (uint32_t)shard_count);
        if (ctx->opt.v[PM_OPT_MAX_SEG_LOG] > 0) pk->max_seg = (uint64_t)1 << ctx->opt.v[PM_OPT_MAX_SEG_LOG];   // test knob: many tiny sub-segments
        pk->segs = pmlayout::quotient_segments(n, (uint32_t)shard_count, (uint32_t)shard_rank, pk->max_seg);
        pk->seg_slots = 0;
        for (uint32_t r = 0; r < (uint32_t)shard_count; ++r) {
            const std::vector<pmlayout::Segment> sr = (int)r == shard_rank ? pk->segs : pmlayout::quotient_segments(n, (uint32_t)shard_count, r, pk->max_seg);
            pk->seg_slots = std::max(pk->seg_slots, sr.size());
            for (size_t i = 0; i < sr.size(); ++i) pk->all_segs.push_back(pm_pk::SegRef{sr[i].a, sr[i].b, r, (uint32_t)i});
        }
        std::sort(pk->all_segs.begin(), pk->all_segs.end(), [](const pm_pk::SegRef &x, const pm_pk::SegRef &y) { return x.a < y.a; });
        uint64_t at = 0;                                    // the ranks' segments must tile [0, len) exactly
        for (const auto &e : pk->all_segs) {
            if (e.a != at || e.b <= e.a) return PM_ERR_STATE;
            at = e.b;
        }
        if (at != pmlayout::numerator_len(n)) return PM_ERR_STATE;
        PM_HIP(ctx, hipMalloc(&pk->d_segs, pk->segs.size() * sizeof(pmlayout::Segment)));
        PM_HIP(ctx, hipMemcpy(pk->d_segs, pk->segs.data(), pk->segs.size() * sizeof(pmlayout::Segment), hipMemcpyHostToDevice));
        // the carry chain of the division scan runs on the device too (prove_sharded.hip: k_seg_chain): all ranks' segments, in index order
        PM_HIP(ctx, hipMalloc(&pk->d_all_segs, pk->all_segs.size() * sizeof(pm_pk::SegRef)));
        PM_HIP(ctx, hipMemcpy(pk->d_all_segs, pk->all_segs.data(), pk->all_segs.size() * sizeof(pm_pk::SegRef), hipMemcpyHostToDevice));
        pk->pieces[0] = pmlayout::pieces_a(ks, L);
        pk->pieces[1] = pmlayout::pieces_c(ks, L);
        pk->pieces[2] = pmlayout::pieces_d(ks, pk->segs);
        for (int k = 0; k < 3; ++k) { pk->res_lo[k] = 0; pk->res_hi[k] = 0; }   // not contiguous: unused in this layout
    } else {
        for (int k = 0; k < 3; ++k) pk->pieces[k] = {pmlayout::Piece{pk->msm_lo[k] + pk->res_lo[k], pk->res_hi[k] - pk->res_lo[k]}};
    }
    for (int k = 0; k < 3; ++k) {
        pk->res_cnt[k] = 0;
        for (const auto &pc : pk->pieces[k]) pk->res_cnt[k] += pc.count;
    }
    if (shard_count == 1 && layout == PM_SHARD_PAIRS) {   // the whole logical concatenation is resident
        for (int k = 0; k < 3; ++k) pk->res_dev_off[k] = pk->msm_lo[k];
    } else {  // device layout [c pairs | a pairs | d pairs]
        pk->res_dev_off[1] = 0;
        pk->res_dev_off[0] = pk->res_cnt[1];
        pk->res_dev_off[2] = pk->res_dev_off[0] + pk->res_cnt[0];
    }
    return PM_OK;
}

static uint64_t pk_resident_points(const pm_pk *pk) {
    if (pk->shard_count == 1 && pk->layout == PM_SHARD_PAIRS) return pk->total_points;
    return pk->res_cnt[0] + pk->res_cnt[1] + pk->res_cnt[2];
}

template <class C>
static int pk_upload_matrices(pm_ctx *ctx, pm_pk *pk, const pm_csr *a, const pm_csr *b, const pm_csr *c) {
    const pm_csr *m[3] = {a, b, c};
    const uint64_t nr = pk->nr, ncols = pk->m0 + pk->mw;
    for (int i = 0; i < 3; ++i) {
        if (!m[i] || m[i]->nrows != nr || !m[i]->rowptr) return PM_ERR_INVALID_ARG;
        // One parallel pass over the rows: column range, and whether ANY row repeats a column (m_at, common.rs:100-105, only ever
        // sees the first entry).  Without repeats -- every circuit ark-relations' `to_matrices` produces after its own
        // linear-combination merging, and the synthetic ones -- the caller's arrays go to the device as they are: no host copy
        // (round 4: the copying first-entry pass was 1.05 s of host time at 2^24 gates).
        std::atomic<int> bad{0}, repeats{0};
        const pm_csr *mi = m[i];
        parallel_chunks(nr, [&](uint64_t lo, uint64_t hi, unsigned) {
            for (uint64_t r = lo; r < hi; ++r) {
                const uint64_t k0 = mi->rowptr[r], k1 = mi->rowptr[r + 1];
                if (k1 < k0) { bad = 1; return; }
                for (uint64_t k = k0; k < k1; ++k) {
                    if (mi->col[k] >= ncols) { bad = 1; return; }
                    for (uint64_t q = k0; q < k; ++q)
                        if (mi->col[q] == mi->col[k]) { repeats = 1; break; }
                }
            }
        });
        if (bad) return PM_ERR_INVALID_ARG;
        HostCsr h;
        const uint64_t *rowptr = mi->rowptr, *val = mi->val;
        const uint32_t *col = mi->col;
        uint64_t nnz = nr ? mi->rowptr[nr] - mi->rowptr[0] : 0;
        if (repeats || (nr && mi->rowptr[0] != 0)) {
            PM_TRY(dedupe_csr(mi, nr, ncols, h));
            rowptr = h.rowptr.data(); col = h.col.data(); val = h.val.data();
            nnz = h.col.size();
        }
        const uint64_t zero_row = 0;
        pk->nnz[i] = nnz;
        PM_HIP(ctx, hipMalloc((void **)&pk->d_rowptr[i], (nr + 1) * 8));
        PM_HIP(ctx, hipMemcpy(pk->d_rowptr[i], nr ? rowptr : &zero_row, (nr + 1) * 8, hipMemcpyHostToDevice));
        const size_t nz = nnz ? nnz : 1;
        PM_HIP(ctx, hipMalloc((void **)&pk->d_col[i], nz * 4));
        PM_HIP(ctx, hipMalloc((void **)&pk->d_val[i], nz * 32));
        if (nnz) {
            PM_HIP(ctx, hipMemcpy(pk->d_col[i], col, nnz * 4, hipMemcpyHostToDevice));
            PM_HIP(ctx, hipMemcpy(pk->d_val[i], val, nnz * 32, hipMemcpyHostToDevice));
        }
    }
    return PM_OK;
}

// Apply `fill(vec, start, count, dst)` to every piece of the logical concatenation range [lo, hi).
template <class C, class F>
static int for_cat_range(const pm_pk *pk, uint64_t lo, uint64_t hi, Affine<C> *dst, F fill) {
    const int order[6] = {PM_UJ_WJ_LCS_BY_Y_ALPHA, PM_X_POWERS_ZH_BY_Y_ALPHA, PM_X_POWERS, PM_X_POWERS_Y_ALPHA,
                          PM_X_POWERS_Y_GAMMA, PM_X_POWERS_Y_GAMMA_Z};
    for (int k = 0; k < 6; ++k) {
        int v = order[k];
        uint64_t s = pk->seg_off[v], e = s + pk->base_len[v];
        uint64_t a = std::max(lo, s), b = std::min(hi, e);
        if (a >= b) continue;
        PM_TRY(fill(v, a - s, b - a, dst + (a - lo)));
    }
    return PM_OK;
}

// Allocates the resident base array and fills it through `fill`; then, when they fit (sized for 288 GB of
// HBM), builds one set of window tables per merged MSM on the device.
template <class C, class F>
static int pk_fill_bases(pm_ctx *ctx, pm_pk *pk, F fill) {
    const uint64_t resident = pk_resident_points(pk);
    PM_HIP(ctx, hipMalloc(&pk->d_bases, (resident ? resident : 1) * sizeof(Affine<C>)));
    Affine<C> *d = (Affine<C> *)pk->d_bases;
    if (pk->shard_count == 1 && pk->layout == PM_SHARD_PAIRS) {
        PM_TRY(for_cat_range<C>(pk, 0, pk->total_points, d, fill));
    } else {
        for (int k = 0; k < 3; ++k) {
            uint64_t at = pk->res_dev_off[k];
            for (const auto &pc : pk->pieces[k]) {
                if (pc.count) PM_TRY(for_cat_range<C>(pk, pc.cat_lo, pc.cat_lo + pc.count, d + at, fill));
                at += pc.count;
            }
        }
    }
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const long long tmode = ctx->opt.v[PM_OPT_TABLES];
    pk->max_piece = (uint64_t)msm_max_piece(ctx);
    if (tmode == PM_TABLES_OFF) return PM_OK;
    // Budget: free HBM minus the per-proof vectors (~40 Fr per domain point; 26 are in use) and the MSM workspaces: 16 B
    // per (pair, window) entry of the sort's two ping-pong arrays and the sorted indices, 13-16 windows, plus the
    // bucket-side arrays -> 256 B per pair of one <= 2^27-pair piece, for the main context's largest MSM and for the
    // helper context's [a]_1 MSM.  Tables are granted per MSM, smallest first, while they fit; an MSM without tables
    // runs the per-window pipeline on the plain array.
    size_t free_b = 0, total_b = 0;
    PM_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    // PM_INFLIGHT_CONTEXTS: how many contexts will prove on this resident key at once (each owns the per-proof vectors
    // and an MSM workspace; default 1).  The helper context of the overlapped [a]_1 MSM (prove.hip) has a workspace of
    // its own as well.
    const int inflight = (int)ctx->opt.v[PM_OPT_INFLIGHT_CONTEXTS];
    const uint64_t vec_n = pk->layout == PM_SHARD_VECTOR ? pk->n / (uint64_t)pk->shard_count : pk->n;   // length of a rank's vectors
    double budget = 0.9 * (double)free_b - (double)inflight * 64.0 * 40.0 * (double)vec_n;
    {
        const uint64_t len_d = pk->res_cnt[2], len_a = pk->res_cnt[0];
        budget -= (double)inflight * 256.0 * (double)(len_d < pk->max_piece ? len_d : pk->max_piece);
        budget -= (double)inflight * 256.0 * (double)len_a;
    }
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int x, int y) { return pk->res_cnt[x] < pk->res_cnt[y]; });
    for (int q = 0; q < 3; ++q) {
        const int k = order[q];
        const uint64_t len = pk->res_cnt[k];
        if (!len) continue;
        MsmTables tb = tables_plan((size_t)len, 1, (size_t)len, (unsigned)C::FrP::BITS, (unsigned)ctx->opt.v[PM_OPT_TABLE_WINDOW_BITS]);
        // no table plan at all (nwin x points would overflow the u32 table indices: the 335 M-pair [d]_1 of a 2^24-gate key) counts
        // as "does not fit"
        const double need = tb.c ? (double)len * tb.nwin * sizeof(TablePoint<C>) + (double)len : 1e300;
        const bool force_wide = tmode == PM_TABLES_WIDE;      // test / tuning knob -- no tables for any MSM, wide mode at any size
        if (need > budget || force_wide) {
            // No room for this MSM's tables (the 10n-pair [d]_1 of a 2^24-gate key on one GPU: 515 GB): WIDE mode -- the same
            // sort / accumulate / reduce kernels on the plain base array, one bucket set per window, 13 additions per pair
            // instead of the 16 of the one-shot pipeline.  Costs one infinity-flag byte per point.  PM_TABLES_NO_WIDE disables.
            const size_t piece = (size_t)(len < pk->max_piece ? len : pk->max_piece);
            MsmTables wt = tmode == PM_TABLES_NO_WIDE ? MsmTables() : wide_plan(piece, (unsigned)ctx->opt.v[PM_OPT_TABLE_WINDOW_BITS]);
            if (wt.c && (piece >= ((size_t)1 << 18) || force_wide) && (double)len < budget) {
                budget -= (double)len;
                PM_HIP(ctx, hipMalloc(&pk->d_tab_inf[k], len));
                PM_TRY(infinity_flags<C>(ctx, d + pk->res_dev_off[k], (size_t)len, (unsigned char *)pk->d_tab_inf[k]));
                wt.inf = (const unsigned char *)pk->d_tab_inf[k];
                pk->tables[k] = wt;
            }
            continue;
        }
        budget -= need;
        PM_HIP(ctx, hipMalloc(&pk->d_tab[k], len * tb.nwin * sizeof(TablePoint<C>)));
        PM_HIP(ctx, hipMalloc(&pk->d_tab_inf[k], len));
        PM_TRY(infinity_flags<C>(ctx, d + pk->res_dev_off[k], (size_t)len, (unsigned char *)pk->d_tab_inf[k]));
        PM_TRY(tables_build<C>(ctx, d + pk->res_dev_off[k], (TablePoint<C> *)pk->d_tab[k], (size_t)len, tb));
        tb.inf = (const unsigned char *)pk->d_tab_inf[k];
        tb.table = pk->d_tab[k];
        pk->tables[k] = tb;
    }
    return PM_OK;
}

template <class C>
static int pk_load_impl(pm_ctx *ctx, uint64_t n, uint64_t m0, uint64_t mw, uint64_t nr, uint64_t sigma, const pm_csr *a,
                        const pm_csr *b, const pm_csr *c, const pm_base_array *bases, int shard_rank, int shard_count, int layout,
                        pm_pk **out) {
    pm_pk *pk = new pm_pk();
    auto guard = [&](int st) { if (st != PM_OK) pk_release(pk); return st; };
    int st = pk_init_layout<C>(ctx, pk, m0, mw, nr, shard_rank, shard_count, layout);
    if (st) return guard(st);
    if (pk->n != n || pk->sigma != sigma) return guard(PM_ERR_INVALID_ARG);
    for (int v = 0; v < PM_NUM_BASE_VECS; ++v)
        if (bases[v].len < pk->base_len[v] || bases[v].stride < sizeof(Affine<C>) || !bases[v].points)
            return guard(PM_ERR_LEN_MISMATCH);
    st = pk_upload_matrices<C>(ctx, pk, a, b, c);
    if (st) return guard(st);
    std::vector<Affine<C>> tmp;
    st = pk_fill_bases<C>(ctx, pk, [&](int v, uint64_t start, uint64_t count, Affine<C> *dst) -> int {
        tmp.resize(count);
        repack_bases<C>((const uint8_t *)bases[v].points + start * bases[v].stride, bases[v].stride, count, tmp.data());
        PM_HIP(ctx, hipMemcpy(dst, tmp.data(), count * sizeof(Affine<C>), hipMemcpyHostToDevice));
        PM_TRY(bases_convert<C>(ctx, dst, count, true));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PM_OK;
    });
    if (st) return guard(st);
    *out = pk;
    return PM_OK;
}

extern "C" int pm_pk_load_sharded(pm_ctx *ctx, int curve, uint64_t n, uint64_t m0, uint64_t mw, uint64_t nr, uint64_t sigma,
                                  const pm_csr *a, const pm_csr *b, const pm_csr *c, const pm_base_array bases[PM_NUM_BASE_VECS],
                                  int shard_rank, int shard_count, int layout, pm_pk **out) {
    if (!ctx || !a || !b || !c || !bases || !out) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    return PM_DISPATCH(curve, pk_load_impl<BlsCurve>(ctx, n, m0, mw, nr, sigma, a, b, c, bases, shard_rank, shard_count, layout, out),
                       pk_load_impl<BnCurve>(ctx, n, m0, mw, nr, sigma, a, b, c, bases, shard_rank, shard_count, layout, out));
}

extern "C" int pm_pk_load(pm_ctx *ctx, int curve, uint64_t n, uint64_t m0, uint64_t mw, uint64_t nr, uint64_t sigma,
                          const pm_csr *a, const pm_csr *b, const pm_csr *c, const pm_base_array bases[PM_NUM_BASE_VECS],
                          int shard_rank, int shard_count, pm_pk **out) {
    return pm_pk_load_sharded(ctx, curve, n, m0, mw, nr, sigma, a, b, c, bases, shard_rank, shard_count, PM_SHARD_PAIRS, out);
}

// generate_proving_key (generator.rs:24-167) with the trapdoors supplied.  The dense uj_wj_lcs loop
// (:112-136) becomes one sparse pass over A, B, C:
//   column j' of z_tail (j = j' + m0), L1 = L[2m0+r], L2 = L[2m0+nr+r]:
//     R1CS column k = j' < m0+mw:  u = sum_r A[r][k](L1+L2) + B[r][k](L1-L2)
//                                  w = 4 sum_r C[r][k] L1  (+ 4 L[k] if k < m0)
//     y column t = j' - (m0+mw):   u = 0, w = L[t] + L[t+m0] (t < m0) | L1 + L2 (t = m0 + r)
template <class C>
static int pk_generate_impl(pm_ctx *ctx, uint64_t m0, uint64_t mw, uint64_t nr, const pm_csr *a, const pm_csr *b,
                            const pm_csr *c, const uint64_t *x_trap, const uint64_t *z_trap, int shard_rank,
                            int shard_count, int layout, pm_pk **out) {
    typedef typename C::FrP P;
    typedef Fp<P> Fr;
    pm_pk *pk = new pm_pk();
    auto guard = [&](int st) { if (st != PM_OK) pk_release(pk); return st; };
    HostProfile hp("setup", shard_rank);            // PM_PROFILE_HOST=1: where the host thread's time goes (stderr)
    int st = pk_init_layout<C>(ctx, pk, m0, mw, nr, shard_rank, shard_count, layout);
    if (st) return guard(st);
    hp.mark("layout");
    st = pk_upload_matrices<C>(ctx, pk, a, b, c);
    if (st) return guard(st);
    hp.mark("matrices: first-entry pass + upload");
    const uint64_t n = pk->n, sigma = pk->sigma, Lz = 2 * m0 + mw + nr;
    Fr x, z, omega;
    memcpy(x.l, x_trap, 32);
    memcpy(z.l, z_trap, 32);
    memcpy(omega.l, pk->omega, 32);
    Fr xn = pow_u64<P>(x, n), one = Fr::one();
    if (xn.eq(one) || pow_u64<P>(z, n).eq(one)) return guard(PM_ERR_INVALID_ARG);  // sample_element_outside_domain
    Fr y = pow_u64<P>(x, sigma), yinv = inverse<P>(y);                            // generator.rs:73
    Fr y_alpha = pow_u64<P>(yinv, 3), y_to_minus_alpha = pow_u64<P>(y, 3), y_gamma = pow_u64<P>(yinv, 5);
    Fr zh = sub<P>(xn, one);                                                       // :106
    // uj_wj_lcs scalars (generator.rs:112-136) on the device: Lagrange coefficients at x by chunked batch inversion, the sparse pass
    // over the CSR matrices already uploaded above (setup.hip: lcs_scalars).  Rounds 1-3 ran both on <= 32 host threads.
    DevBuf d_lagrange, d_work, d_lcs_buf;
    struct Release { DevBuf &a, &b, &c; ~Release() { a.release(); b.release(); c.release(); } } release_lcs{d_lagrange, d_work, d_lcs_buf};
    {
        if (hipMalloc(&d_lcs_buf.p, (Lz ? Lz : 1) * sizeof(Fr)) != hipSuccess) { ctx->err = "out of device memory for the lcs scalars"; return guard(PM_ERR_HIP); }
        d_lcs_buf.bytes = (Lz ? Lz : 1) * sizeof(Fr);
        const Fr kscale = mul<P>(zh, inverse<P>(from_u64<P>(n)));
        st = lcs_scalars<C>(ctx, pk, x, omega, kscale, y_gamma, y_to_minus_alpha, d_lagrange, d_work, d_lcs_buf.as<Fr>());
        if (st) return guard(st);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "lcs scalars: stream failed"; return guard(PM_ERR_HIP); }
        d_lagrange.release();
        d_work.release();
    }
    hp.mark("lcs scalars (device, synchronised)");
    const Fr *d_lcs = d_lcs_buf.as<Fr>();
    // per-vector scale of the x-power vectors (generator.rs:82-109)
    Fr scale[PM_NUM_BASE_VECS];
    scale[PM_X_POWERS] = one;
    scale[PM_X_POWERS_Y_ALPHA] = y_alpha;
    scale[PM_X_POWERS_Y_GAMMA] = y_gamma;
    scale[PM_X_POWERS_Y_GAMMA_Z] = mul<P>(y_gamma, z);
    scale[PM_X_POWERS_ZH_BY_Y_ALPHA] = mul<P>(zh, y_to_minus_alpha);
    const uint64_t CH = (uint64_t)1 << 22;
    st = pk_fill_bases<C>(ctx, pk, [&](int v, uint64_t start, uint64_t count, Affine<C> *dst) -> int {
        PM_HIP(ctx, ctx->scratch.reserve(std::min(count, CH) * sizeof(Fr)));
        Fr *d_sc = ctx->scratch.as<Fr>();
        for (uint64_t s = 0; s < count; s += CH) {
            uint64_t cnt = std::min(CH, count - s);
            const Fr *src = d_sc;
            if (v == PM_UJ_WJ_LCS_BY_Y_ALPHA) {
                src = d_lcs + start + s;
            } else {
                Fr first = mul<P>(scale[v], pow_u64<P>(x, start + s));
                PM_TRY(powers_fill<C>(ctx, d_sc, cnt, first, x));
            }
            PM_TRY(fixed_base_batch<C>(ctx, src, cnt, dst + s));
            PM_TRY(bases_convert<C>(ctx, dst + s, cnt, true));
            PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        return PM_OK;
    });
    if (st) return guard(st);
    hp.mark("bases + tables (device, synchronised)");
    *out = pk;
    return PM_OK;
}

extern "C" int pm_pk_generate_sharded(pm_ctx *ctx, int curve, uint64_t m0, uint64_t mw, uint64_t nr, const pm_csr *a,
                                      const pm_csr *b, const pm_csr *c, const uint64_t *x_trapdoor, const uint64_t *z_trapdoor,
                                      int shard_rank, int shard_count, int layout, pm_pk **out) {
    if (!ctx || !a || !b || !c || !x_trapdoor || !z_trapdoor || !out) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    try {
        return PM_DISPATCH(curve, pk_generate_impl<BlsCurve>(ctx, m0, mw, nr, a, b, c, x_trapdoor, z_trapdoor, shard_rank, shard_count, layout, out),
                           pk_generate_impl<BnCurve>(ctx, m0, mw, nr, a, b, c, x_trapdoor, z_trapdoor, shard_rank, shard_count, layout, out));
    } catch (const std::exception &e) {       // host allocation failure in the O(n) setup vectors: a status, never an abort
        ctx->err = e.what();
        return PM_ERR_STATE;
    }
}

extern "C" int pm_pk_generate(pm_ctx *ctx, int curve, uint64_t m0, uint64_t mw, uint64_t nr, const pm_csr *a,
                              const pm_csr *b, const pm_csr *c, const uint64_t *x_trapdoor, const uint64_t *z_trapdoor,
                              int shard_rank, int shard_count, pm_pk **out) {
    return pm_pk_generate_sharded(ctx, curve, m0, mw, nr, a, b, c, x_trapdoor, z_trapdoor, shard_rank, shard_count, PM_SHARD_PAIRS, out);
}

// ---- layout helpers for hosts and tests (polymath_amd/host/layout.hpp): pure index arithmetic, no GPU
extern "C" int pm_layout_indices(uint64_t n, int shard_count, int shard_rank, int coefficients, uint64_t *out /* n / shard_count */) {
    if (!out || shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count || !pmlayout::layout_ok(n, (uint32_t)shard_count)) return PM_ERR_INVALID_ARG;
    const pmlayout::Layout L = pmlayout::make_layout(n, (uint32_t)shard_count, (uint32_t)shard_rank);
    for (uint64_t p = 0; p < L.m; ++p) out[p] = coefficients ? pmlayout::coeff_global(L, p) : pmlayout::eval_global(L, p);
    return PM_OK;
}

extern "C" int pm_pk_msm_pieces(const pm_pk *pk, int which, uint64_t *cat_lo, uint64_t *count, size_t capacity, size_t *n_pieces) {
    if (!pk || which < 0 || which > 2 || !n_pieces) return PM_ERR_INVALID_ARG;
    *n_pieces = pk->pieces[which].size();
    for (size_t i = 0; i < pk->pieces[which].size() && i < capacity; ++i) {
        if (cat_lo) cat_lo[i] = pk->pieces[which][i].cat_lo;
        if (count) count[i] = pk->pieces[which][i].count;
    }
    return PM_OK;
}

extern "C" int pm_pk_info(const pm_pk *pk, uint64_t *n, uint64_t *m0, uint64_t *sigma, uint64_t *omega,
                          uint64_t base_lens[PM_NUM_BASE_VECS]) {
    if (!pk) return PM_ERR_INVALID_ARG;
    if (n) *n = pk->n;
    if (m0) *m0 = pk->m0;
    if (sigma) *sigma = pk->sigma;
    if (omega) memcpy(omega, pk->omega, 32);
    if (base_lens) for (int i = 0; i < PM_NUM_BASE_VECS; ++i) base_lens[i] = pk->base_len[i];
    return PM_OK;
}

extern "C" int pm_pk_msm_plan(const pm_pk *pk, int which, uint64_t *pairs, unsigned *windows, unsigned *window_bits, int *tables) {
    if (!pk || which < 0 || which > 2) return PM_ERR_INVALID_ARG;
    const uint64_t len = pk->res_cnt[which];
    unsigned nwin = 0, c = 0;
    if (pk->tables[which].c) {
        nwin = pk->tables[which].nwin;
        c = pk->tables[which].c;
    } else {
        const uint64_t piece = len < pk->max_piece ? len : pk->max_piece;
        pm::msm_plan_query((size_t)piece, pk->curve == PM_BLS12_381 ? (unsigned)BlsFrP::BITS : (unsigned)BnFrP::BITS, &nwin, &c);
    }
    if (pairs) *pairs = len;
    if (windows) *windows = nwin;
    if (window_bits) *window_bits = c;
    if (tables) *tables = pk->tables[which].c && !pk->tables[which].wide ? 1 : 0;
    return PM_OK;
}

extern "C" int pm_pk_export_bases(pm_ctx *ctx, const pm_pk *pk, int which, size_t offset, size_t len, uint64_t *out_xy) {
    if (!ctx || !pk || !out_xy || which < 0 || which >= PM_NUM_BASE_VECS) return PM_ERR_INVALID_ARG;
    if (offset + len > pk->base_len[which]) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    size_t pt = pk->curve == PM_BLS12_381 ? sizeof(Affine<BlsCurve>) : sizeof(Affine<BnCurve>);
    uint64_t lo = pk->seg_off[which] + offset, hi = lo + len;
    uint64_t dev_off = 0;
    bool found = false;
    if (pk->shard_count == 1 && pk->layout == PM_SHARD_PAIRS) {
        dev_off = lo;
        found = true;
    } else {
        for (int k = 0; k < 3 && !found; ++k) {
            uint64_t at = pk->res_dev_off[k];
            for (const auto &pc : pk->pieces[k]) {
                if (lo >= pc.cat_lo && hi <= pc.cat_lo + pc.count) { dev_off = at + (lo - pc.cat_lo); found = true; break; }
                at += pc.count;
            }
        }
    }
    if (!found) return PM_ERR_INVALID_ARG;  // not resident on this shard
    const void *src = (const uint8_t *)pk->d_bases + dev_off * pt;
    return PM_DISPATCH(pk->curve, download_points<BlsCurve>(ctx, src, len, out_xy), download_points<BnCurve>(ctx, src, len, out_xy));
}

// -------------------------------------------------------------------------------- prove
// extern "C" entry points never let a C++ exception (bad_alloc in a host vector, system_error from a thread) unwind
// into the embedding host (a Rust shim or ctypes): it becomes a status with the message in pm_last_error.
template <class F>
static int guarded(pm_ctx *ctx, F body) {
    struct Turn {   // local serialised emulation (comm.hip): a phase runs holding the turn; no-op otherwise
        pm_comm *c;
        explicit Turn(pm_comm *cc) : c(cc) { if (c) c->phase_begin(); }
        ~Turn() { if (c) c->phase_end(); }
    } turn(ctx->comm);
    int st;
    try {
        st = body();
    } catch (const std::bad_alloc &) {
        ctx->err = "host allocation failed";
        st = PM_ERR_STATE;
    } catch (const std::exception &e) {
        ctx->err = e.what();
        st = PM_ERR_STATE;
    }
    // Fail-fast (include/polymath_hip.h, pm_comm contract): the prover's own verdicts are the same on every rank; any other
    // failure is this rank's alone, and its peers are about to wait for it in the next collective -- abort the communicator.
    const bool shared_verdict = st == PM_OK || st == PM_ERR_LEN_MISMATCH || st == PM_ERR_DOMAIN_TOO_LARGE || st == PM_ERR_REMAINDER_NONZERO ||
                                st == PM_ERR_DEGREE_BOUND || st == PM_ERR_INVALID_ARG;
    if (!shared_verdict && ctx->comm && !ctx->comm->failed) ctx->comm->abort(("a prover phase failed on this rank: " + ctx->err).c_str());
    return st;
}

extern "C" int pm_prove_phase1(pm_ctx *ctx, const pm_pk *pk, const uint64_t *x, const uint64_t *w, const uint64_t *r_a,
                               uint64_t *a_g1_xy, int *a_inf, uint64_t *c_g1_xy, int *c_inf) {
    if (!ctx || !pk || !x || !r_a || !a_g1_xy || !a_inf || !c_g1_xy || !c_inf || (pk->mw && !w)) return PM_ERR_INVALID_ARG;
    if (pk->device != ctx->device) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    return guarded(ctx, [&] {
        if (pk->layout == PM_SHARD_VECTOR)
            return PM_DISPATCH(pk->curve, prove_phase1_sharded<BlsCurve>(ctx, pk, x, w, r_a, a_g1_xy, a_inf, c_g1_xy, c_inf, false),
                               prove_phase1_sharded<BnCurve>(ctx, pk, x, w, r_a, a_g1_xy, a_inf, c_g1_xy, c_inf, false));
        return PM_DISPATCH(pk->curve, prove_phase1_impl<BlsCurve>(ctx, pk, x, w, r_a, a_g1_xy, a_inf, c_g1_xy, c_inf, false),
                           prove_phase1_impl<BnCurve>(ctx, pk, x, w, r_a, a_g1_xy, a_inf, c_g1_xy, c_inf, false));
    });
}

extern "C" int pm_prove_phase1_device(pm_ctx *ctx, const pm_pk *pk, const uint64_t *d_x, const uint64_t *d_w, const uint64_t *r_a,
                                      uint64_t *a_g1_xy, int *a_inf, uint64_t *c_g1_xy, int *c_inf) {
    if (!ctx || !pk || !d_x || !r_a || !a_g1_xy || !a_inf || !c_g1_xy || !c_inf || (pk->mw && !d_w)) return PM_ERR_INVALID_ARG;
    if (pk->device != ctx->device) return PM_ERR_INVALID_ARG;
    PM_TRY(set_device(ctx));
    return guarded(ctx, [&] {
        if (pk->layout == PM_SHARD_VECTOR)
            return PM_DISPATCH(pk->curve, prove_phase1_sharded<BlsCurve>(ctx, pk, d_x, d_w, r_a, a_g1_xy, a_inf, c_g1_xy, c_inf, true),
                               prove_phase1_sharded<BnCurve>(ctx, pk, d_x, d_w, r_a, a_g1_xy, a_inf, c_g1_xy, c_inf, true));
        return PM_DISPATCH(pk->curve, prove_phase1_impl<BlsCurve>(ctx, pk, d_x, d_w, r_a, a_g1_xy, a_inf, c_g1_xy, c_inf, true),
                           prove_phase1_impl<BnCurve>(ctx, pk, d_x, d_w, r_a, a_g1_xy, a_inf, c_g1_xy, c_inf, true));
    });
}

extern "C" int pm_prove_phase2(pm_ctx *ctx, const uint64_t *x1, uint64_t *u_at_x1) {
    if (!ctx || !x1 || !u_at_x1) return PM_ERR_INVALID_ARG;
    if (!ctx->pk) return PM_ERR_STATE;
    PM_TRY(set_device(ctx));
    return guarded(ctx, [&] {
        if (ctx->pk->layout == PM_SHARD_VECTOR)
            return PM_DISPATCH(ctx->pk->curve, prove_phase2_sharded<BlsCurve>(ctx, x1, u_at_x1), prove_phase2_sharded<BnCurve>(ctx, x1, u_at_x1));
        return PM_DISPATCH(ctx->pk->curve, prove_phase2_impl<BlsCurve>(ctx, x1, u_at_x1), prove_phase2_impl<BnCurve>(ctx, x1, u_at_x1));
    });
}

extern "C" int pm_prove_phase3(pm_ctx *ctx, const uint64_t *x1, const uint64_t *x2, const uint64_t *a_at_x1,
                               const uint64_t *c_at_x1, uint64_t *d_g1_xy, int *d_inf) {
    if (!ctx || !x1 || !x2 || !a_at_x1 || !c_at_x1 || !d_g1_xy || !d_inf) return PM_ERR_INVALID_ARG;
    if (!ctx->pk) return PM_ERR_STATE;
    PM_TRY(set_device(ctx));
    return guarded(ctx, [&] {
        if (ctx->pk->layout == PM_SHARD_VECTOR)
            return PM_DISPATCH(ctx->pk->curve, prove_phase3_sharded<BlsCurve>(ctx, x1, x2, a_at_x1, c_at_x1, d_g1_xy, d_inf),
                               prove_phase3_sharded<BnCurve>(ctx, x1, x2, a_at_x1, c_at_x1, d_g1_xy, d_inf));
        return PM_DISPATCH(ctx->pk->curve, prove_phase3_impl<BlsCurve>(ctx, x1, x2, a_at_x1, c_at_x1, d_g1_xy, d_inf),
                           prove_phase3_impl<BnCurve>(ctx, x1, x2, a_at_x1, c_at_x1, d_g1_xy, d_inf));
    });
}

extern "C" int pm_prove_tap(pm_ctx *ctx, int which, uint64_t *out, size_t max_elems, size_t *n_elems) {
    if (!ctx || !out || !n_elems) return PM_ERR_INVALID_ARG;
    if (!ctx->pk || ctx->phase < 1) return PM_ERR_STATE;
    PM_TRY(set_device(ctx));
    const pm_pk *pk = ctx->pk;
    const uint64_t n = pk->n, Lz = 2 * pk->m0 + pk->mw + pk->nr;
    const void *src = nullptr;
    size_t cnt = 0;
    if (pk->layout == PM_SHARD_VECTOR) {
        // this rank's LOCAL vectors (coefficients in the blocked layout: pm_layout_indices); the evaluations are consumed
        // by the in-place transforms and are not available
        const uint64_t N = (uint64_t)pk->shard_count, q = (uint64_t)pk->shard_rank, m = n / N;
        const uint64_t zcnt = pmlayout::ztail_lo(Lz, (uint32_t)N, (uint32_t)q + 1) - pmlayout::ztail_lo(Lz, (uint32_t)N, (uint32_t)q);
        switch (which) {
            case 2: src = ctx->u.p; cnt = m; break;
            case 3: src = ctx->w.p; cnt = m; break;
            case 4: src = (const uint8_t *)ctx->sc_c.p + zcnt * 32; cnt = m - (q == N - 1 ? 1 : 0); break;
            case 5: src = ctx->wit_u.p; cnt = m; break;
            case 6: src = ctx->sc_c.p; cnt = zcnt; break;
            case 7:
                if (ctx->phase < 3) return PM_ERR_STATE;
                src = ctx->quotient.p; cnt = pk->res_cnt[2]; break;
            default: return PM_ERR_INVALID_ARG;
        }
        *n_elems = cnt;
        const size_t kk = std::min(cnt, max_elems);
        if (kk) PM_HIP(ctx, hipMemcpy(out, src, kk * 32, hipMemcpyDeviceToHost));
        return PM_OK;
    }
    switch (which) {
        case 0: src = ctx->ue.p; cnt = n; break;
        case 1: src = ctx->we.p; cnt = n; break;
        case 2: src = ctx->u.p; cnt = n; break;
        case 3: src = ctx->w.p; cnt = n; break;
        case 4: src = (const uint8_t *)ctx->sc_c.p + Lz * 32; cnt = n - 1; break;
        case 5: src = ctx->wit_u.p; cnt = n; break;
        case 6: src = ctx->sc_c.p; cnt = Lz; break;
        case 7:
            if (ctx->phase < 3) return PM_ERR_STATE;
            src = ctx->quotient.p; cnt = 8 * pk->sigma + 2 * n - 2; break;
        default: return PM_ERR_INVALID_ARG;
    }
    *n_elems = cnt;
    size_t k = std::min(cnt, max_elems);
    PM_HIP(ctx, hipMemcpy(out, src, k * 32, hipMemcpyDeviceToHost));
    return PM_OK;
}
