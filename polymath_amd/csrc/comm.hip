// pm_comm: the exchange layer of a multi-GPU proof.  Two collectives are all the sharded prover needs:
//   all_to_all  equal blocks of DEVICE memory, stream-ordered (the transpose of the four-step NTT: one per transform);
//   all_gather  a few hundred bytes of HOST memory (partial G1 points, status flags, Horner partial sums and halo
//               coefficients, the per-segment values of the division scan).
// Implementations:
//   RCCL   (pm_comm_rccl_create)  -- one process per GPU over xGMI; librccl is dlopen'ed at first use, so the library
//                                    has no link-time dependency on it and single-GPU hosts never load it.  The
//                                    all-to-all is one ncclAllToAll on the context's stream: no host synchronisation.
//   local  (pm_comm_local_create) -- N ranks as N threads of ONE process (any mix of devices): rendezvous through a
//                                    barrier, device-to-device copies.  What the single-GPU lockstep tests and the
//                                    per-rank emulation of bench.py run on; also a valid single-process multi-GPU mode.
//   callbacks (pm_comm_from_callbacks) -- the host brings its own transport (torch.distributed in tests).
// Fail-fast: no collective waits for a dead peer for ever.  Every communicator has a deadline (pm_comm_set_timeout_ms,
// PM_COMM_TIMEOUT_MS; 120 s by default) and a sticky `failed` flag; once it is set every collective returns PM_ERR_COMM.
//   local : the rendezvous is a timed wait; a rank that aborts (pm_comm_abort, or a phase of the sharded prover that
//           returns an error its peers cannot know) wakes everybody at once.
//   RCCL  : a watchdog thread follows an event recorded behind every collective; past the deadline, or on an
//           asynchronous RCCL error, it calls ncclCommAbort, which ends the device-side wait, and marks the
//           communicator failed -- the blocked stream synchronisation returns and the phase reports PM_ERR_COMM.
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "internal.h"
#include "comm.h"

using namespace pm;

namespace {

// ---------------------------------------------------------------------------------------------------- local
struct LocalGroup {
    int world;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void *> send;
    std::vector<void *> recv;
    std::atomic<bool> failed{false};   // sticky: once a rank failed a collective, every later one reports it
    std::string why;                   // first failure (guarded by mu)
    // pm_comm_local_set_serialize: between collectives only ONE rank runs at a time (a turnstile), so that N ranks emulated on
    // one GPU do not time-slice it: each rank's kernels then take what they would take alone (tools/shard_emulation.py).
    bool serialize = false;
    std::mutex turn;
    explicit LocalGroup(int w) : world(w), send(w), recv(w) {}
    // false: the group is dead (a peer aborted, or did not arrive within timeout_ms)
    bool barrier(long timeout_ms, int rank) {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        const bool woke = cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return generation != gen || failed.load(); });
        if (!woke) {
            if (why.empty()) why = "rank " + std::to_string(rank) + ": a peer did not reach the collective within " + std::to_string(timeout_ms) + " ms";
            failed = true;
            cv.notify_all();
            return false;
        }
        return generation != gen;   // a completed rendezvous counts even if the group failed right after it
    }
    void fail(const std::string &w) {
        std::unique_lock<std::mutex> lk(mu);
        if (why.empty()) why = w;
        failed = true;
        cv.notify_all();
    }
    std::string reason() {
        std::unique_lock<std::mutex> lk(mu);
        return why;
    }
};

struct LocalComm : pm_comm {
    std::shared_ptr<LocalGroup> g;
    bool holds_turn = false;
    // serialised emulation: the time this rank spent RUNNING (holding the turn), i.e. without the waits for its peers --
    // what the rank would take on a GPU of its own, exchanges aside (pm_comm_busy_ms; tools/shard_emulation.py)
    std::chrono::steady_clock::time_point mark;
    void release_turn() {
        if (g->serialize && holds_turn) {
            busy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - mark).count();
            g->turn.unlock();
            holds_turn = false;
        }
    }
    void take_turn() {
        if (g->serialize && !holds_turn) {
            g->turn.lock();
            holds_turn = true;
            mark = std::chrono::steady_clock::now();
        }
    }
    void phase_begin() override { take_turn(); }
    ~LocalComm() override { release_turn(); }
    void phase_end() override { release_turn(); }
    const char *kind() const override { return "local"; }
    void abort(const char *why) override {
        g->fail(std::string("rank ") + std::to_string(rank) + " aborted: " + (why ? why : ""));
        dead();
    }
    int dead() {                                        // the group failed: this rank's view of it
        (void)fail_once(g->reason());
        return PM_ERR_COMM;
    }
    int all_to_all(const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) override {
        if (g->failed) return dead();
        if (hipStreamSynchronize(stream) != hipSuccess) g->fail("local all_to_all: stream sync failed on rank " + std::to_string(rank));
        release_turn();
        g->send[rank] = d_send;
        g->recv[rank] = d_recv;
        bool ok = g->barrier(timeout_ms, rank);         // every rank's send buffer is complete and published
        take_turn();
        for (int p = 0; p < world && ok && !g->failed; ++p) {
            const uint8_t *src = (const uint8_t *)g->send[p] + (size_t)rank * bytes;
            if (hipMemcpyAsync((uint8_t *)d_recv + (size_t)p * bytes, src, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                g->fail("local all_to_all: copy failed on rank " + std::to_string(rank));
        }
        if (hipStreamSynchronize(stream) != hipSuccess) g->fail("local all_to_all: stream sync failed on rank " + std::to_string(rank));
        release_turn();
        ok = g->barrier(timeout_ms, rank) && ok;        // nobody reuses a send buffer before every peer has read it
        take_turn();
        return ok && !g->failed ? (int)PM_OK : dead();
    }
    int all_gather_device(const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) override {
        if (g->failed) return dead();
        if (hipStreamSynchronize(stream) != hipSuccess) g->fail("local all_gather_device: stream sync failed on rank " + std::to_string(rank));
        release_turn();
        g->send[rank] = d_send;
        bool ok = g->barrier(timeout_ms, rank);
        take_turn();
        for (int p = 0; p < world && ok && !g->failed; ++p)
            if (hipMemcpyAsync((uint8_t *)d_recv + (size_t)p * bytes, g->send[p], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                g->fail("local all_gather_device: copy failed on rank " + std::to_string(rank));
        if (hipStreamSynchronize(stream) != hipSuccess) g->fail("local all_gather_device: stream sync failed on rank " + std::to_string(rank));
        release_turn();
        ok = g->barrier(timeout_ms, rank) && ok;
        take_turn();
        return ok && !g->failed ? (int)PM_OK : dead();
    }
    int all_gather(const void *send_h, void *recv_h, size_t bytes, hipStream_t) override {
        if (g->failed) return dead();
        release_turn();
        g->send[rank] = send_h;
        bool ok = g->barrier(timeout_ms, rank);
        if (ok)
            for (int p = 0; p < world; ++p) memcpy((uint8_t *)recv_h + (size_t)p * bytes, g->send[p], bytes);
        ok = g->barrier(timeout_ms, rank) && ok;
        take_turn();
        return ok && !g->failed ? (int)PM_OK : dead();
    }
};

// ----------------------------------------------------------------------------------------------------- RCCL
// The handful of RCCL entry points, resolved at run time.  Under torch the process already holds a librccl (same
// soname): dlopen then returns that instance instead of loading a second copy.
struct Id128 { char b[128]; };   // ncclUniqueId, passed BY VALUE to ncclCommInitRank
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;
    int (*CommGetAsyncError)(void *, int *) = nullptr;
    int (*AllToAll)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};

static RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) { api.err = std::string("librccl not found: ") + dlerror(); return; }
        auto sym = [&](const char *n, bool required = true) {
            void *p = dlsym(api.lib, n);
            if (!p && required && api.err.empty()) api.err = std::string("librccl lacks ") + n;
            return p;
        };
        api.GetUniqueId = (int (*)(void *))sym("ncclGetUniqueId");
        api.CommInitRank = (int (*)(void **, int, Id128, int))sym("ncclCommInitRank");
        api.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
        api.CommAbort = (int (*)(void *))sym("ncclCommAbort", false);
        api.CommGetAsyncError = (int (*)(void *, int *))sym("ncclCommGetAsyncError", false);
        api.AllToAll = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))sym("ncclAllToAll");
        api.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))sym("ncclAllGather");
        api.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    });
    return &api;
}
constexpr int NCCL_UINT8 = 1;   // ncclUint8 (rccl.h)

struct RcclComm : pm_comm {
    void *comm = nullptr;
    int device = 0;
    hipStream_t side = nullptr;     // stream of the small host all-gathers
    // staging of the host all-gather, [send | world x recv] on both sides; the host side is PINNED so that the two copies
    // are real asynchronous DMA transfers and the call costs one stream synchronisation
    void *d_stage = nullptr, *h_stage = nullptr;
    size_t stage_bytes = 0;
    // watchdog: an event behind every collective; the thread polls the oldest one
    static constexpr int RING = 32;
    hipEvent_t ring[RING] = {};
    int ring_next = 0;
    // Two events per collective: `start` in front of it, `ev` behind it.  The deadline runs from the moment `start` completes --
    // the kernels a phase queued ahead of the collective (first-call twiddle tables, a large circuit's transforms) do not eat
    // into it -- with a ceiling of UNARMED_FACTOR deadlines from the enqueue for a stream that never gets there.
    struct Watch { hipEvent_t start, ev; unsigned long long seq; bool armed; std::chrono::steady_clock::time_point deadline; };
    static constexpr int UNARMED_FACTOR = 8;
    hipEvent_t ring_start[RING] = {};
    unsigned long long seq_next = 1;
    hipEvent_t begun = nullptr;        // start event of the collective being enqueued (watch_begin .. watch)
    std::deque<Watch> pending;
    std::mutex wmu;
    std::condition_variable wcv;
    std::thread wd;
    bool stop = false, aborted = false;
    const char *kind() const override { return "rccl"; }
    int fail(const char *what, int rc) {
        RcclApi *a = rccl_api();
        (void)fail_once(std::string(what) + ": " + (a->GetErrorString ? a->GetErrorString(rc) : "rccl error"));
        return PM_ERR_COMM;
    }
    int dead() { return PM_ERR_COMM; }
    void abort_comm(const std::string &why) {          // under wmu
        if (aborted) return;
        aborted = true;
        (void)fail_once(why);
        fprintf(stderr, "[pm_comm rank %d] %s -- aborting the RCCL communicator\n", rank, why.c_str());
        if (comm && rccl_api()->CommAbort) rccl_api()->CommAbort(comm);
    }
    void abort(const char *why) override {
        std::unique_lock<std::mutex> lk(wmu);
        abort_comm(std::string("rank ") + std::to_string(rank) + " aborted: " + (why ? why : ""));
    }
    void watchdog() {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(wmu);
        for (;;) {
            wcv.wait(lk, [&] { return stop || !pending.empty(); });
            if (stop) return;
            const Watch w = pending.front();
            lk.unlock();
            const hipError_t q = hipEventQuery(w.ev);
            const hipError_t qs = w.armed ? hipSuccess : hipEventQuery(w.start);
            int async = 0;
            if (rccl_api()->CommGetAsyncError && comm && !aborted) (void)rccl_api()->CommGetAsyncError(comm, &async);
            lk.lock();
            if (stop) return;
            // the entry may have been re-armed (ring wrap) or dropped while the lock was released: decide on the CURRENT front
            if (pending.empty() || pending.front().seq != w.seq) continue;
            Watch &cur = pending.front();
            if (q == hipSuccess) {
                pending.pop_front();
                continue;
            }
            const auto now = std::chrono::steady_clock::now();
            if (!cur.armed && qs == hipSuccess) {          // the collective is at the head of its stream: the clock starts now
                cur.armed = true;
                cur.deadline = now + std::chrono::milliseconds(timeout_ms);
            }
            if (async != 0) {
                abort_comm(std::string("asynchronous RCCL error: ") + (rccl_api()->GetErrorString ? rccl_api()->GetErrorString(async) : "?"));
                pending.clear();
            } else if (q != hipErrorNotReady) {
                abort_comm("a collective's completion event failed (device error)");
                pending.clear();
            } else if (now > cur.deadline) {
                abort_comm(cur.armed ? "a collective did not complete within " + std::to_string(timeout_ms) + " ms (peer dead or stalled)"
                                     : "a collective was not reached by its stream within " + std::to_string((long long)timeout_ms * UNARMED_FACTOR) + " ms");
                pending.clear();
            } else {
                wcv.wait_for(lk, std::chrono::milliseconds(5));
            }
        }
    }
    void watch_begin(hipStream_t stream) {               // in front of a collective about to be enqueued on `stream`
        std::unique_lock<std::mutex> lk(wmu);
        begun = ring_start[ring_next];
        if (hipEventRecord(begun, stream) != hipSuccess) begun = nullptr;
    }
    void watch(hipStream_t stream) {                     // behind a collective just enqueued on `stream`
        std::unique_lock<std::mutex> lk(wmu);
        hipEvent_t ev = ring[ring_next], start = begun;
        begun = nullptr;
        ring_next = (ring_next + 1) % RING;
        for (auto it = pending.begin(); it != pending.end();)   // the ring wrapped: that entry is re-armed below
            it = it->ev == ev ? pending.erase(it) : it + 1;
        if (hipEventRecord(ev, stream) != hipSuccess) return;
        const auto now = std::chrono::steady_clock::now();
        if (start) pending.push_back(Watch{start, ev, seq_next++, false, now + std::chrono::milliseconds((long long)timeout_ms * UNARMED_FACTOR)});
        else pending.push_back(Watch{ev, ev, seq_next++, true, now + std::chrono::milliseconds(timeout_ms)});
        wcv.notify_all();
    }
    int start() {
        for (int i = 0; i < RING; ++i)
            if (hipEventCreateWithFlags(&ring[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ring_start[i], hipEventDisableTiming) != hipSuccess) return PM_ERR_HIP;
        try {
            wd = std::thread([this] { watchdog(); });
        } catch (const std::system_error &) {
            return PM_ERR_STATE;
        }
        return PM_OK;
    }
    ~RcclComm() override {
        {
            std::unique_lock<std::mutex> lk(wmu);
            stop = true;
            wcv.notify_all();
        }
        if (wd.joinable()) wd.join();
        (void)hipSetDevice(device);
        if (comm && !aborted) rccl_api()->CommDestroy(comm);
        for (int i = 0; i < RING; ++i) {
            if (ring[i]) (void)hipEventDestroy(ring[i]);
            if (ring_start[i]) (void)hipEventDestroy(ring_start[i]);
        }
        if (d_stage) (void)hipFree(d_stage);
        if (h_stage) (void)hipHostFree(h_stage);
        if (side) (void)hipStreamDestroy(side);
    }
    int all_to_all(const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) override {
        if (failed) return dead();
        watch_begin(stream);
        const int rc = rccl_api()->AllToAll(d_send, d_recv, bytes, NCCL_UINT8, comm, stream);
        if (rc) return fail("ncclAllToAll", rc);
        watch(stream);
        return PM_OK;
    }
    int all_gather_device(const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) override {
        if (failed) return dead();
        watch_begin(stream);
        const int rc = rccl_api()->AllGather(d_send, d_recv, bytes, NCCL_UINT8, comm, stream);
        if (rc) return fail("ncclAllGather (device)", rc);
        watch(stream);
        return PM_OK;
    }
    int all_gather(const void *send_h, void *recv_h, size_t bytes, hipStream_t stream) override {
        if (failed) return dead();
        hipStream_t side = stream ? stream : this->side;
        const size_t need = bytes * (size_t)(world + 1);
        if (need > stage_bytes) {
            if (d_stage) (void)hipFree(d_stage);
            if (h_stage) (void)hipHostFree(h_stage);
            d_stage = h_stage = nullptr;
            stage_bytes = need < 65536 ? 65536 : need;
            if (hipMalloc(&d_stage, stage_bytes) != hipSuccess || hipHostMalloc(&h_stage, stage_bytes, hipHostMallocDefault) != hipSuccess) {
                set_error("all_gather staging allocation failed");
                stage_bytes = 0;
                return PM_ERR_HIP;
            }
        }
        uint8_t *ds = (uint8_t *)d_stage, *dr = ds + bytes, *hs = (uint8_t *)h_stage, *hr = hs + bytes;
        memcpy(hs, send_h, bytes);
        if (hipMemcpyAsync(ds, hs, bytes, hipMemcpyHostToDevice, side) != hipSuccess) { set_error("all_gather H2D failed"); return PM_ERR_HIP; }
        watch_begin(side);
        const int rc = rccl_api()->AllGather(ds, dr, bytes, NCCL_UINT8, comm, side);
        if (rc) return fail("ncclAllGather", rc);
        watch(side);
        if (hipMemcpyAsync(hr, dr, bytes * (size_t)world, hipMemcpyDeviceToHost, side) != hipSuccess || hipStreamSynchronize(side) != hipSuccess) {
            if (failed) return dead();
            set_error("all_gather D2H failed");
            return PM_ERR_HIP;
        }
        if (failed) return dead();                       // the watchdog ended the wait: the bytes are not the peers'
        memcpy(recv_h, hr, bytes * (size_t)world);
        return PM_OK;
    }
};

// ------------------------------------------------------------------------------------------------ callbacks
// Deadlines are the transport's own here (torch.distributed: the process group's timeout); a failing callback makes the
// communicator fail for good, like the others.
struct CallbackComm : pm_comm {
    pm_comm_ops ops;
    const char *kind() const override { return "callbacks"; }
    int done(int st, const char *what) {
        if (st != PM_OK) (void)fail_once(std::string(what) + " callback failed with status " + std::to_string(st));
        return st == PM_OK ? (int)PM_OK : (int)PM_ERR_COMM;
    }
    int all_to_all(const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) override {
        if (failed) return PM_ERR_COMM;
        return ops.all_to_all ? done(ops.all_to_all(ops.user, d_send, d_recv, bytes, (void *)stream), "all_to_all") : (int)PM_ERR_INVALID_ARG;
    }
    int all_gather(const void *send_h, void *recv_h, size_t bytes, hipStream_t) override {
        if (failed) return PM_ERR_COMM;
        return ops.all_gather ? done(ops.all_gather(ops.user, send_h, recv_h, bytes), "all_gather") : (int)PM_ERR_INVALID_ARG;
    }
    // the ops table has no device all-gather: an all-to-all whose `world` send blocks are all this rank's block is one
    void *d_rep = nullptr;
    size_t rep_bytes = 0;
    ~CallbackComm() override { if (d_rep) (void)hipFree(d_rep); }
    int all_gather_device(const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) override {
        if (failed) return PM_ERR_COMM;
        const size_t need = bytes * (size_t)world;
        if (need > rep_bytes) {
            if (d_rep) (void)hipFree(d_rep);
            d_rep = nullptr;
            rep_bytes = 0;
            if (hipMalloc(&d_rep, need) != hipSuccess) { set_error("all_gather_device staging allocation failed"); return PM_ERR_HIP; }
            rep_bytes = need;
        }
        for (int p = 0; p < world; ++p)
            if (hipMemcpyAsync((uint8_t *)d_rep + (size_t)p * bytes, d_send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
                set_error("all_gather_device replicate failed");
                return PM_ERR_HIP;
            }
        return all_to_all(d_rep, d_recv, bytes, stream);
    }
};

static long default_timeout_ms() {
    const char *e = getenv("PM_COMM_TIMEOUT_MS");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : 120000;
}

}  // namespace

// ----------------------------------------------------------------------------------------------------- C ABI
extern "C" int pm_comm_local_create(int world, pm_comm **out /* world handles */) {
    if (world < 1 || !out) return PM_ERR_INVALID_ARG;
    auto g = std::make_shared<LocalGroup>(world);
    for (int r = 0; r < world; ++r) {
        LocalComm *c = new LocalComm();
        c->rank = r;
        c->world = world;
        c->g = g;
        c->timeout_ms = default_timeout_ms();
        out[r] = c;
    }
    return PM_OK;
}

extern "C" int pm_comm_rccl_unique_id(void *out_128_bytes) {
    if (!out_128_bytes) return PM_ERR_INVALID_ARG;
    RcclApi *a = rccl_api();
    if (!a->GetUniqueId) return PM_ERR_STATE;
    return a->GetUniqueId(out_128_bytes) ? (int)PM_ERR_HIP : (int)PM_OK;
}

extern "C" int pm_comm_rccl_create(const void *unique_id_128_bytes, int rank, int world, int device, pm_comm **out) {
    if (!unique_id_128_bytes || !out || world < 1 || rank < 0 || rank >= world) return PM_ERR_INVALID_ARG;
    *out = nullptr;
    RcclApi *a = rccl_api();
    if (!a->CommInitRank || !a->AllToAll || !a->AllGather || !a->CommDestroy) return PM_ERR_STATE;
    if (hipSetDevice(device) != hipSuccess) return PM_ERR_HIP;
    std::unique_ptr<RcclComm> c(new RcclComm());
    c->rank = rank;
    c->world = world;
    c->device = device;
    Id128 id;
    memcpy(id.b, unique_id_128_bytes, 128);
    c->timeout_ms = default_timeout_ms();
    if (a->CommInitRank(&c->comm, world, id, rank)) return PM_ERR_COMM;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) return PM_ERR_HIP;
    PM_TRY(c->start());
    *out = c.release();
    return PM_OK;
}

extern "C" int pm_comm_from_callbacks(const pm_comm_ops *ops, int rank, int world, pm_comm **out) {
    if (!ops || !out || world < 1 || rank < 0 || rank >= world) return PM_ERR_INVALID_ARG;
    CallbackComm *c = new CallbackComm();
    c->rank = rank;
    c->world = world;
    c->ops = *ops;
    c->timeout_ms = default_timeout_ms();
    *out = c;
    return PM_OK;
}

extern "C" void pm_comm_destroy(pm_comm *c) { delete c; }
extern "C" int pm_comm_rank(const pm_comm *c) { return c ? c->rank : -1; }
extern "C" int pm_comm_world(const pm_comm *c) { return c ? c->world : 0; }
extern "C" double pm_comm_busy_ms(pm_comm *c, int reset) {
    if (!c) return 0.0;
    const double v = c->busy_ms;
    if (reset) c->busy_ms = 0.0;
    return v;
}
extern "C" const char *pm_comm_last_error(const pm_comm *c) {
    if (!c) return "null comm";
    static thread_local std::string copy;    // taken under the communicator's lock: the watchdog may be writing the original
    copy = c->error();
    return copy.c_str();
}
extern "C" int pm_comm_local_set_serialize(pm_comm *c, int on) {
    LocalComm *l = dynamic_cast<LocalComm *>(c);
    if (!l) return PM_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lk(l->g->mu);
    if (l->g->generation != 0 || l->holds_turn) return PM_ERR_STATE;   // before the group's first collective only
    l->g->serialize = on != 0;
    return PM_OK;
}
extern "C" const char *pm_comm_kind(const pm_comm *c) { return c ? c->kind() : "none"; }
extern "C" int pm_comm_set_timeout_ms(pm_comm *c, long timeout_ms) {
    if (!c || timeout_ms <= 0) return PM_ERR_INVALID_ARG;
    c->timeout_ms = timeout_ms;
    return PM_OK;
}
extern "C" int pm_comm_abort(pm_comm *c, const char *why) {
    if (!c) return PM_ERR_INVALID_ARG;
    c->abort(why ? why : "pm_comm_abort");
    return PM_OK;
}
extern "C" int pm_comm_failed(const pm_comm *c) { return c && c->failed ? 1 : 0; }

extern "C" int pm_comm_all_gather(pm_comm *c, const void *send, void *recv, size_t bytes) {
    if (!c || !send || !recv) return PM_ERR_INVALID_ARG;
    const int st = c->all_gather(send, recv, bytes, nullptr);
    c->phase_end();   // a host-level collective: no prover phase is running, nothing to hold a turn for
    return st;
}

extern "C" int pm_comm_all_gather_device(pm_comm *c, const void *d_send, void *d_recv, size_t bytes, void *hip_stream) {
    if (!c || !d_send || !d_recv) return PM_ERR_INVALID_ARG;
    return c->all_gather_device(d_send, d_recv, bytes, (hipStream_t)hip_stream);
}

extern "C" int pm_comm_all_to_all(pm_comm *c, const void *d_send, void *d_recv, size_t bytes_per_peer, void *hip_stream) {
    if (!c || !d_send || !d_recv) return PM_ERR_INVALID_ARG;
    return c->all_to_all(d_send, d_recv, bytes_per_peer, (hipStream_t)hip_stream);
}

extern "C" int pm_ctx_set_comm(pm_ctx *ctx, pm_comm *comm) {
    if (!ctx) return PM_ERR_INVALID_ARG;
    ctx->comm = comm;
    return PM_OK;
}

// sum over ranks of `count` partial G1 points, in place: all-gather (x||y, inf) + pm_g1_sum (RCCL has no
// elliptic-curve reduction op; ncclSum over limbs would be wrong).  The native form of pm_combine_fn.
extern "C" int pm_comm_combine_points(pm_comm *c, int curve, int count, uint64_t *xy, int *inf) {
    if (!c || !xy || !inf || count < 1 || count > 8) return PM_ERR_INVALID_ARG;
    const size_t words = curve == PM_BLS12_381 ? 12 : 8;
    const size_t rec = words + 1;                                    // u64 words per point record
    std::vector<uint64_t> mine(rec * count), all(rec * count * (size_t)c->world);
    for (int j = 0; j < count; ++j) {
        memcpy(&mine[j * rec], xy + j * words, words * 8);
        mine[j * rec + words] = (uint64_t)(inf[j] != 0);
    }
    const int st = c->all_gather(mine.data(), all.data(), mine.size() * 8, nullptr);
    c->phase_end();   // called by the host glue BETWEEN the prover's phases
    PM_TRY(st);
    std::vector<uint64_t> pts(words * (size_t)c->world);
    std::vector<int> infs(c->world);
    for (int j = 0; j < count; ++j) {
        for (int r = 0; r < c->world; ++r) {
            const uint64_t *src = &all[((size_t)r * count + j) * rec];
            memcpy(&pts[r * words], src, words * 8);
            infs[r] = (int)src[words];
        }
        PM_TRY(pm_g1_sum(curve, pts.data(), infs.data(), (size_t)c->world, xy + j * words, &inf[j]));
    }
    return PM_OK;
}
