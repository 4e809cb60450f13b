// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so.1 so that the RCCL branch of polymath_amd/csrc/comm.hip (RcclComm:
// unique-id hand-off, ncclCommInitRank ordering, per-peer byte counts, two user streams on one communicator, the watchdog's
// ncclCommAbort, ncclCommGetAsyncError) runs with REAL PEER PROCESSES on a box that has ONE GPU.  Real RCCL refuses two ranks
// on one device, and this pool has no multi-GPU node for builder runs; nothing here says anything about xGMI, RCCL's kernels or
// their bandwidth.  It is never loaded by the product: tests put its directory in LD_LIBRARY_PATH of rank processes that do
// not import torch, where comm.hip's dlopen("librccl.so.1") then finds this file instead of /opt/rocm/lib/librccl.so.1.
//
// Exports exactly the eight symbols comm.hip:198-205 resolves, with rccl.h's signatures:
//   ncclGetUniqueId  ncclCommInitRank  ncclCommDestroy  ncclCommAbort  ncclCommGetAsyncError  ncclAllToAll  ncclAllGather
//   ncclGetErrorString
// Semantics kept from NCCL, because they are what comm.hip's code depends on:
//   * collectives are ASYNCHRONOUS and stream-ordered: the call enqueues and returns; the user stream is held by a device-side
//     wait (a one-lane kernel polling a pinned host flag) until the data has arrived -- so a dead or stalled peer leaves the
//     stream blocked on the DEVICE, exactly the state comm.hip's watchdog exists for, and ncclCommAbort is what ends it;
//   * one communicator executes its collectives in issue order whatever streams they were issued on (NCCL's implicit launch
//     order), and every rank must issue the same sequence: each collective's (kind, byte count, ordinal) is compared across
//     the ranks when they meet, and a mismatch FAILS the communicator (ncclInvalidUsage as the asynchronous error, a line on
//     stderr) instead of exchanging the wrong buffers;
//   * ncclCommInitRank blocks until all ranks of the id have joined;
//   * a peer process that has died surfaces as the asynchronous error ncclRemoteError (PM_FAKE_RCCL_NO_LIVENESS=1 turns this
//     off, leaving only the caller's own deadline -- what a stalled-but-alive peer looks like).
// Transport: each rank stages its send buffer in a POSIX shared-memory segment (device -> host), the ranks meet on counters
// in a shared control block, each rank copies its blocks out of its peers' segments (host -> device).  Bytes move through the
// host; that is the point of a stand-in.
//
// Diagnostics: with PM_FAKE_RCCL_LOG=<path prefix> every communicator writes <prefix>.rank<r>.json at destroy / abort: the
// number of collectives of each kind, the number of DISTINCT user streams they were issued on, the deepest queue.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <set>
#include <string>
#include <thread>

namespace {

enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
       ncclRemoteError = 6, ncclInProgress = 7 };

constexpr int MAX_RANKS = 16;
constexpr uint32_t MAGIC = 0x70666b31;   // "pfk1"

struct Slot {
    std::atomic<uint32_t> joined;
    std::atomic<int64_t> pid;
    std::atomic<uint64_t> posted;     // collectives whose send data is staged and whose descriptor is published
    std::atomic<uint64_t> drained;    // collectives this rank has finished reading its peers' segments for
    std::atomic<uint32_t> gone;       // the rank aborted or destroyed its communicator
    // descriptor of collective number `posted - 1`
    std::atomic<uint32_t> kind;
    std::atomic<uint64_t> bytes;
};

struct Control {
    std::atomic<uint32_t> magic;
    std::atomic<uint32_t> world;      // 0 until the first rank arrives
    uint64_t stage_bytes;
    Slot slot[MAX_RANKS];
};

struct IdBlob {          // what travels in the 128 bytes of ncclUniqueId
    uint32_t magic;
    char name[100];      // shm name of the control block
};

__global__ void k_wait_released(const uint64_t *released, uint64_t want, const uint32_t *abort_flag, uint32_t *timed_out, long long max_ticks) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(released, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
        if (wall_clock64() - t0 > max_ticks) {      // the stand-in's own backstop: a test bug must not hang the GPU box
            __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __builtin_amdgcn_s_sleep(64);
    }
}

long env_long(const char *name, long dflt) {
    const char *e = getenv(name);
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : dflt;
}

bool process_alive(int64_t pid) {
    char path[64], buf[512];
    snprintf(path, sizeof path, "/proc/%lld/stat", (long long)pid);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')');          // "pid (comm) S ..."
    if (!p || !p[1] || !p[2]) return false;
    const char state = p[2];
    return state != 'Z' && state != 'X' && state != 'x';
}

void *map_shm(const char *name, size_t bytes, bool create) {
    int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return nullptr; }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    return p == MAP_FAILED ? nullptr : p;
}

struct Op {
    int kind;                 // 0 = all-to-all, 1 = all-gather
    const void *send;
    void *recv;
    size_t bytes;             // per peer (all-to-all) / per rank (all-gather)
    hipEvent_t ready;         // recorded on the user stream at enqueue: the send buffer is complete
    uint64_t ordinal;         // 1-based
};

struct FakeComm {
    int rank = 0, world = 1, device = 0;
    std::string ctl_name;
    Control *ctl = nullptr;
    uint8_t *stage[MAX_RANKS] = {};   // every rank's staging segment, mapped
    size_t stage_bytes = 0;
    hipStream_t copy_stream = nullptr;
    // device-side wait
    uint64_t *released = nullptr;     // pinned, coherent: ordinal of the last collective whose data has arrived
    uint32_t *abort_flag = nullptr;   // pinned
    uint32_t *timed_out = nullptr;    // pinned
    long long max_ticks = 0;
    // worker
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Op> queue;
    std::thread worker;
    bool stop = false;
    std::atomic<bool> aborted{false};
    std::atomic<int> async_error{0};
    uint64_t next_ordinal = 1;
    bool liveness = true;
    // diagnostics
    uint64_t n_a2a = 0, n_ag = 0;
    size_t deepest = 0;
    std::set<void *> streams;
    std::string log_prefix;

    void fail_async(int code, const char *what, uint64_t ordinal) {
        int expected = 0;
        if (async_error.compare_exchange_strong(expected, code))
            fprintf(stderr, "[fake_rccl rank %d] collective #%llu: %s\n", rank, (unsigned long long)ordinal, what);
    }

    // wait until every rank's counter `which` has reached `want`; false: aborted here, a peer gone / dead, or another failure
    bool meet(std::atomic<uint64_t> Slot::*which, uint64_t want, uint64_t ordinal) {
        auto last_check = std::chrono::steady_clock::now();
        for (;;) {
            bool all = true;
            for (int p = 0; p < world; ++p)
                if ((ctl->slot[p].*which).load(std::memory_order_acquire) < want) { all = false; break; }
            if (all) return true;
            if (aborted.load() || async_error.load()) return false;
            const auto now = std::chrono::steady_clock::now();
            if (now - last_check > std::chrono::milliseconds(50)) {
                last_check = now;
                for (int p = 0; p < world; ++p) {
                    if (p == rank || (ctl->slot[p].*which).load(std::memory_order_acquire) >= want) continue;
                    if (ctl->slot[p].gone.load() || (liveness && !process_alive(ctl->slot[p].pid.load()))) {
                        fail_async(ncclRemoteError, "a peer process exited or tore its communicator down", ordinal);
                        return false;
                    }
                }
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }

    void release_up_to(uint64_t ordinal) { __atomic_store_n(released, ordinal, __ATOMIC_RELEASE); }

    bool run(const Op &op) {
        Slot &me = ctl->slot[rank];
        const size_t send_bytes = op.kind == 0 ? op.bytes * (size_t)world : op.bytes;
        if (send_bytes > stage_bytes) {
            fail_async(ncclInvalidArgument, "send buffer larger than the stand-in's staging segment (PM_FAKE_RCCL_STAGE_MB)", op.ordinal);
            return false;
        }
        if (hipEventSynchronize(op.ready) != hipSuccess) { fail_async(ncclUnhandledCudaError, "waiting for the user stream failed", op.ordinal); return false; }
        // never the null stream: it would wait for the device-side waits parked on the caller's (blocking) streams
        if (send_bytes && (hipMemcpyAsync(stage[rank], op.send, send_bytes, hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
                           hipStreamSynchronize(copy_stream) != hipSuccess)) {
            fail_async(ncclUnhandledCudaError, "device -> staging copy failed", op.ordinal);
            return false;
        }
        me.kind.store((uint32_t)op.kind, std::memory_order_relaxed);
        me.bytes.store(op.bytes, std::memory_order_relaxed);
        me.posted.store(op.ordinal, std::memory_order_release);
        if (!meet(&Slot::posted, op.ordinal, op.ordinal)) return false;
        // NCCL's rule: every rank issues the same collectives in the same order on a communicator
        for (int p = 0; p < world; ++p) {
            const bool same_ordinal = ctl->slot[p].posted.load(std::memory_order_acquire) == op.ordinal;
            if (!same_ordinal || ctl->slot[p].kind.load() != (uint32_t)op.kind || ctl->slot[p].bytes.load() != op.bytes) {
                char msg[200];
                snprintf(msg, sizeof msg, "ORDER MISMATCH with rank %d (here kind %d, %zu bytes; there kind %u, %llu bytes, ordinal %llu)", p, op.kind, op.bytes,
                         ctl->slot[p].kind.load(), (unsigned long long)ctl->slot[p].bytes.load(), (unsigned long long)ctl->slot[p].posted.load());
                fail_async(ncclInvalidUsage, msg, op.ordinal);
                return false;
            }
        }
        for (int p = 0; p < world; ++p) {
            const uint8_t *src = op.kind == 0 ? stage[p] + (size_t)rank * op.bytes : stage[p];
            if (op.bytes && hipMemcpyAsync((uint8_t *)op.recv + (size_t)p * op.bytes, src, op.bytes, hipMemcpyHostToDevice, copy_stream) != hipSuccess) {
                fail_async(ncclUnhandledCudaError, "staging -> device copy failed", op.ordinal);
                return false;
            }
        }
        if (hipStreamSynchronize(copy_stream) != hipSuccess) { fail_async(ncclUnhandledCudaError, "staging -> device copy failed", op.ordinal); return false; }
        me.drained.store(op.ordinal, std::memory_order_release);
        // the data is here: let the user stream go on.  The second meeting only protects the staging segments from reuse.
        release_up_to(op.ordinal);
        return meet(&Slot::drained, op.ordinal, op.ordinal);
    }

    void work() {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        bool broken = false;
        for (;;) {
            cv.wait(lk, [&] { return stop || !queue.empty(); });
            if (queue.empty()) { if (stop) return; continue; }
            const Op op = queue.front();
            lk.unlock();
            if (!broken && !aborted.load()) broken = !run(op);
            (void)hipEventDestroy(op.ready);
            lk.lock();
            queue.pop_front();
            cv.notify_all();
            // a broken communicator never releases its device-side waits by itself: like NCCL's kernels they sit on the stream
            // until ncclCommAbort (or the stand-in's backstop) ends them
        }
    }

    void write_log(const char *how) {
        if (log_prefix.empty()) return;
        const std::string path = log_prefix + ".rank" + std::to_string(rank) + ".json";
        FILE *f = fopen(path.c_str(), "w");
        if (!f) return;
        fprintf(f, "{\"stand_in\": \"tests/native/fake_rccl.hip\", \"rank\": %d, \"world\": %d, \"pid\": %lld, \"all_to_all\": %llu, \"all_gather\": %llu, "
                   "\"distinct_user_streams\": %zu, \"deepest_queue\": %zu, \"async_error\": %d, \"device_wait_timed_out\": %u, \"end\": \"%s\"}\n",
                rank, world, (long long)getpid(), (unsigned long long)n_a2a, (unsigned long long)n_ag, streams.size(), deepest, async_error.load(),
                timed_out ? *timed_out : 0u, how);
        fclose(f);
    }

    void shutdown(bool abort_now, const char *how) {
        if (abort_now) {
            aborted.store(true);
            if (abort_flag) __atomic_store_n(abort_flag, 1u, __ATOMIC_RELEASE);     // ends every device-side wait of this communicator
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            if (!abort_now) cv.wait(lk, [&] { return queue.empty(); });           // destroy: the queued collectives complete first
            stop = true;
            cv.notify_all();
        }
        if (worker.joinable()) worker.join();
        if (ctl) ctl->slot[rank].gone.store(1);
        write_log(how);
    }

    ~FakeComm() {
        (void)hipSetDevice(device);
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
        // the pinned flags stay allocated: a device-side wait of an aborted communicator may still be draining on a user stream
        for (int p = 0; p < world; ++p)
            if (stage[p]) munmap(stage[p], stage_bytes);
        if (ctl) {
            if (rank == 0) {       // names go away with rank 0; mappings live on in the peers until they unmap
                for (int p = 0; p < world; ++p) shm_unlink((ctl_name + "_s" + std::to_string(p)).c_str());
                shm_unlink(ctl_name.c_str());
            }
            munmap(ctl, sizeof(Control));
        }
    }
};

int enqueue(FakeComm *c, int kind, const void *send, void *recv, size_t count, int datatype, hipStream_t stream) {
    static const size_t elem[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};
    if (!c || datatype < 0 || datatype > 9) return ncclInvalidArgument;
    if (c->aborted.load()) return ncclInvalidUsage;
    if (const int e = c->async_error.load()) return e;
    Op op;
    op.kind = kind;
    op.send = send;
    op.recv = recv;
    op.bytes = count * elem[datatype];
    if (hipEventCreateWithFlags(&op.ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(op.ready, stream) != hipSuccess) { (void)hipEventDestroy(op.ready); return ncclUnhandledCudaError; }
    {
        std::unique_lock<std::mutex> lk(c->mu);
        op.ordinal = c->next_ordinal++;
        c->queue.push_back(op);
        if (c->queue.size() > c->deepest) c->deepest = c->queue.size();
        (kind == 0 ? c->n_a2a : c->n_ag)++;
        c->streams.insert((void *)stream);
        c->cv.notify_all();
    }
    // the device-side wait: everything enqueued on `stream` after this call runs once the collective's data has arrived
    hipLaunchKernelGGL(k_wait_released, dim3(1), dim3(1), 0, stream, c->released, op.ordinal, c->abort_flag, c->timed_out, c->max_ticks);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int ncclGetUniqueId(void *out) {
    if (!out) return ncclInvalidArgument;
    static std::atomic<unsigned> counter{0};
    IdBlob id;
    memset(&id, 0, sizeof id);
    id.magic = MAGIC;
    snprintf(id.name, sizeof id.name, "/pm_fake_rccl_%lld_%u_%llx", (long long)getpid(), counter++,
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    Control *ctl = (Control *)map_shm(id.name, sizeof(Control), true);
    if (!ctl) return ncclSystemError;
    ctl->stage_bytes = (uint64_t)env_long("PM_FAKE_RCCL_STAGE_MB", 64) << 20;
    ctl->world.store(0);
    ctl->magic.store(MAGIC, std::memory_order_release);
    munmap(ctl, sizeof(Control));
    static_assert(sizeof(IdBlob) <= 128, "ncclUniqueId is 128 bytes");
    memset(out, 0, 128);
    memcpy(out, &id, sizeof id);
    return ncclSuccess;
}

struct UniqueId128 { char b[128]; };

__attribute__((visibility("default"))) int ncclCommInitRank(void **comm_out, int nranks, UniqueId128 id_by_value, int rank) {
    if (!comm_out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    IdBlob id;
    memcpy(&id, id_by_value.b, sizeof id);
    if (id.magic != MAGIC || !memchr(id.name, 0, sizeof id.name)) return ncclInvalidArgument;   // an id this library did not make
    FakeComm *c = new FakeComm();
    c->rank = rank;
    c->world = nranks;
    c->ctl_name = id.name;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    c->ctl = (Control *)map_shm(id.name, sizeof(Control), false);
    if (!c->ctl || c->ctl->magic.load(std::memory_order_acquire) != MAGIC) { delete c; return ncclSystemError; }
    uint32_t expected = 0;
    if (!c->ctl->world.compare_exchange_strong(expected, (uint32_t)nranks) && expected != (uint32_t)nranks) {
        fprintf(stderr, "[fake_rccl rank %d] ncclCommInitRank: nranks %d, but a peer said %u\n", rank, nranks, expected);
        delete c;
        return ncclInvalidArgument;
    }
    c->stage_bytes = c->ctl->stage_bytes;
    Slot &me = c->ctl->slot[rank];
    if (me.joined.load()) {
        fprintf(stderr, "[fake_rccl] ncclCommInitRank: rank %d joined twice\n", rank);
        c->ctl = nullptr;
        delete c;
        return ncclInvalidUsage;
    }
    // this rank's staging segment, created before the rank counts as joined
    const std::string mine = c->ctl_name + "_s" + std::to_string(rank);
    c->stage[rank] = (uint8_t *)map_shm(mine.c_str(), c->stage_bytes, true);
    if (!c->stage[rank]) { delete c; return ncclSystemError; }
    me.pid.store((int64_t)getpid());
    me.posted.store(0);
    me.drained.store(0);
    me.gone.store(0);
    me.joined.store(1, std::memory_order_release);
    // ncclCommInitRank is a collective: it returns when every rank of the id has joined
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(env_long("PM_FAKE_RCCL_INIT_TIMEOUT_S", 120));
    for (int p = 0; p < nranks; ++p) {
        while (!c->ctl->slot[p].joined.load(std::memory_order_acquire)) {
            if (std::chrono::steady_clock::now() > deadline) {
                fprintf(stderr, "[fake_rccl rank %d] ncclCommInitRank: rank %d never joined\n", rank, p);
                delete c;
                return ncclSystemError;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (p != rank) {
            c->stage[p] = (uint8_t *)map_shm((c->ctl_name + "_s" + std::to_string(p)).c_str(), c->stage_bytes, false);
            if (!c->stage[p]) { delete c; return ncclSystemError; }
        }
    }
    void *flags = nullptr;
    if (hipHostMalloc(&flags, 4096, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return ncclUnhandledCudaError;
    }
    memset(flags, 0, 4096);
    c->released = (uint64_t *)flags;
    c->abort_flag = (uint32_t *)((uint8_t *)flags + 256);
    c->timed_out = (uint32_t *)((uint8_t *)flags + 512);
    c->max_ticks = (long long)env_long("PM_FAKE_RCCL_DEVICE_WAIT_CAP_S", 180) * 100000000ll;      // wall_clock64: 100 MHz
    c->liveness = getenv("PM_FAKE_RCCL_NO_LIVENESS") == nullptr;
    if (const char *lp = getenv("PM_FAKE_RCCL_LOG")) c->log_prefix = lp;
    c->worker = std::thread([c] { c->work(); });
    *comm_out = c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) int ncclCommDestroy(void *comm) {
    FakeComm *c = (FakeComm *)comm;
    if (!c) return ncclInvalidArgument;
    c->shutdown(c->async_error.load() != 0, "destroy");
    delete c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) int ncclCommAbort(void *comm) {
    FakeComm *c = (FakeComm *)comm;
    if (!c) return ncclInvalidArgument;
    c->shutdown(true, "abort");
    delete c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) int ncclCommGetAsyncError(void *comm, int *async_error) {
    FakeComm *c = (FakeComm *)comm;
    if (!c || !async_error) return ncclInvalidArgument;
    *async_error = c->async_error.load();
    return ncclSuccess;
}

__attribute__((visibility("default"))) int ncclAllToAll(const void *send, void *recv, size_t count, int datatype, void *comm, hipStream_t stream) {
    return enqueue((FakeComm *)comm, 0, send, recv, count, datatype, stream);
}

__attribute__((visibility("default"))) int ncclAllGather(const void *send, void *recv, size_t sendcount, int datatype, void *comm, hipStream_t stream) {
    return enqueue((FakeComm *)comm, 1, send, recv, sendcount, datatype, stream);
}

__attribute__((visibility("default"))) const char *ncclGetErrorString(int code) {
    switch (code) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (stand-in)";
        case ncclSystemError: return "unhandled system error (stand-in)";
        case ncclInternalError: return "internal error (stand-in)";
        case ncclInvalidArgument: return "invalid argument (stand-in)";
        case ncclInvalidUsage: return "invalid usage (stand-in: collectives issued in different orders on different ranks?)";
        case ncclRemoteError: return "remote process exited or there was a network error (stand-in)";
        case ncclInProgress: return "operation in progress (stand-in)";
        default: return "unknown result code (stand-in)";
    }
}

}  // extern "C"
