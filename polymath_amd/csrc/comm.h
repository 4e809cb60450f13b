// Internal view of pm_comm (comm.hip implements it; prove_sharded.hip and host_prove.hip call it).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <string>

struct pm_comm {
    int rank = 0, world = 1;
    // the first failure's message: written under err_mu BEFORE `failed` is published, read through error() -- a watchdog
    // thread or a peer's abort may fail the communicator while the proving thread looks at it
    mutable std::mutex err_mu;
    std::string err;
    std::string error() const {
        std::lock_guard<std::mutex> lk(err_mu);
        return err;
    }
    void set_error(const std::string &m) {          // a message without failing the communicator (staging allocation, copies)
        std::lock_guard<std::mutex> lk(err_mu);
        err = m;
    }
    bool fail_once(const std::string &m) {          // true: this call failed the communicator
        std::lock_guard<std::mutex> lk(err_mu);
        if (failed.load(std::memory_order_acquire)) return false;
        err = m;
        failed.store(true, std::memory_order_release);
        return true;
    }
    // Fail-fast contract (include/polymath_hip.h): a collective never waits longer than `timeout_ms` for its peers; a rank
    // whose phase fails for a reason its peers cannot know calls abort().  Either way `failed` becomes sticky and every
    // later collective of the communicator returns PM_ERR_COMM at once.
    std::atomic<bool> failed{false};
    long timeout_ms = 120000;
    virtual ~pm_comm() {}
    virtual const char *kind() const = 0;
    // send/recv: world blocks of `bytes` each; block p of `send` goes to rank p, block p of `recv` comes from rank p.
    // Ordered after everything already enqueued on `stream`; the received data is visible to work enqueued after the call.
    virtual int all_to_all(const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) = 0;
    // host buffers: recv holds world x bytes, rank r's contribution at r * bytes.  Blocking.  `stream`: the caller's
    // stream (RCCL runs every collective of a communicator on ONE stream, in program order); null = the comm's own.
    virtual int all_gather(const void *send, void *recv, size_t bytes, hipStream_t stream) = 0;
    // DEVICE buffers: d_recv holds world x bytes, rank r's block at r * bytes; stream-ordered like all_to_all.  The broadcast
    // of the assignment: every rank uploads 1/world of the witness over its own PCIe link, the
    // fabric delivers the rest.
    virtual int all_gather_device(const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) = 0;
    // this rank gives up: wake / unblock whoever can be reached (local group: all peers at once; RCCL: ncclCommAbort here,
    // the peers run into their own deadline)
    virtual void abort(const char *why) { (void)fail_once(why ? why : "aborted"); }
    // end of a prover phase: nothing to exchange until the host calls again (local serialised emulation: pass the turn on)
    virtual void phase_end() {}
    virtual void phase_begin() {}
    double busy_ms = 0.0;   // local serialised emulation only
};
