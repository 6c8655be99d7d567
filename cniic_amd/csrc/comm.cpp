// comm.cpp -- RCCL collectives enqueued on the context's own stream (the all-reduce of the K
// partial centroid sums sits between the assign and the update kernels with no host round trip).
//
// librccl is bound at run time (dlopen): the library loads and every single-GPU entry point works on a box
// without RCCL; cniic_comm_* then returns CNIIC_ERR_UNSUPPORTED.  In a Python process torch has already
// mapped its own librccl.so.1, and dlopen by soname returns that copy, so the process holds one RCCL.
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "common.hpp"

namespace cniic {

namespace {
struct Rccl {
    bool ok = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                          // optional
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;  // optional
};

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
        r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
    });
    return r;
}
}  // namespace

struct Comm {
    Ctx *c = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, nranks = 1;
    // a host's own transport instead of RCCL (MPI, a socket, gloo ...): called with the stream drained, must leave the
    // in-place unsigned sum over all ranks in the host buffer when it returns (cniic_comm_create_host)
    int32_t (*host_sum)(void *user, void *buf_host, uint64_t count, int32_t elem_bytes) = nullptr;
    void *user = nullptr;
    std::vector<uint8_t> bounce;
    Mailbox *mb = nullptr;  // the one-shot exchange instead of RCCL (cniic_comm_create_mailbox, k_mailbox.hip)
    bool dead = false;  // aborted after a failure on this rank or a peer: every later collective fails at once
    uint64_t timeout_ms = default_timeout_ms();  // how long a loop waits for a batch that contains collectives (0: for ever)
    static uint64_t default_timeout_ms() {
        const char *e = getenv("CNIIC_COLLECTIVE_TIMEOUT_MS");
        return e ? strtoull(e, nullptr, 10) : 120000ull;
    }
};

int comm_unique_id(uint8_t *id128) {
    const Rccl &r = rccl();
    if (!r.ok) return CNIIC_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return CNIIC_ERR_HIP;
    memcpy(id128, &id, 128);
    return CNIIC_OK;
}

int comm_create(Ctx *c, const uint8_t *id128, uint32_t rank, uint32_t nranks, Comm **out) {
    const Rccl &r = rccl();
    if (!r.ok) return c->fail(CNIIC_ERR_UNSUPPORTED, "comm_create: librccl not found");
    if (nranks == 0 || rank >= nranks) return c->fail(CNIIC_ERR_BAD_ARG, "comm_create: rank %u of %u", rank, nranks);
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t cm = nullptr;
    const ncclResult_t e = r.CommInitRank(&cm, (int)nranks, id, (int)rank);
    if (e != ncclSuccess) return c->fail(CNIIC_ERR_HIP, "ncclCommInitRank: %s", r.GetErrorString(e));
    *out = new Comm{c, cm, rank, nranks};
    return CNIIC_OK;
}

int comm_create_host(Ctx *c, uint32_t rank, uint32_t nranks, int32_t (*fn)(void *, void *, uint64_t, int32_t), void *user, Comm **out) {
    if (!fn || nranks == 0 || rank >= nranks) return c->fail(CNIIC_ERR_BAD_ARG, "comm_create_host: rank %u of %u, callback %p", rank, nranks, (void *)fn);
    Comm *m = new Comm{c, nullptr, rank, nranks};
    m->host_sum = fn;
    m->user = user;
    *out = m;
    return CNIIC_OK;
}

int comm_create_mailbox(Ctx *c, uint32_t rank, uint32_t nranks, uint64_t max_bytes, uint8_t *handle64, Comm **out) {
    Mailbox *mb = nullptr;
    CNIIC_TRY(mailbox_create(c, rank, nranks, max_bytes, handle64, &mb));
    Comm *m = new Comm{c, nullptr, rank, nranks};
    m->mb = mb;
    *out = m;
    return CNIIC_OK;
}

int comm_connect_mailbox(Comm *cm, const uint8_t *handles) {
    if (!cm->mb) return cm->c->fail(CNIIC_ERR_BAD_ARG, "comm_connect_mailbox: not a mailbox communicator");
    return mailbox_connect(cm->mb, handles);
}

void comm_destroy(Comm *cm) {
    if (!cm) return;
    if (cm->mb) mailbox_destroy(cm->mb);
    if (cm->comm) (void)rccl().CommDestroy(cm->comm);
    delete cm;
}

// (mailboxes: a word in every peer's mailbox ends their waits; a peer that is simply gone ends them by the kernel's own deadline)
// A rank that fails inside a loop of collectives must not simply return: its peers sit in (or are about to enter) an
// all-reduce that will never complete.  RCCL: abort the communicator -- the peers' collectives then end with an error
// (they poll comm_async_error while they wait, below) instead of hanging.  Host transport: the caller's callback is told
// once with (buf = NULL, count = 0, elem_bytes = -1) so that it can tear its own transport down.
void comm_abort(Comm *cm) {
    if (!cm || cm->dead) return;
    cm->dead = true;
    if (cm->host_sum) { (void)cm->host_sum(cm->user, nullptr, 0, -1); return; }
    if (cm->mb) { mailbox_abort(cm->mb); return; }  // a word in every peer's mailbox: their waits end at once
    if (cm->comm && rccl().CommAbort) { (void)rccl().CommAbort(cm->comm); cm->comm = nullptr; }
}

// has the communicator seen a failure (of this rank or, reported by the transport, of a peer)?  CNIIC_OK = healthy
int comm_async_error(Comm *cm) {
    if (!cm) return CNIIC_OK;
    if (cm->dead) return CNIIC_ERR_RCCL;
    if (cm->mb) {
        const int st = mailbox_status(cm->mb);
        if (st) return cm->c->fail(CNIIC_ERR_RCCL, st == 2 ? "mailbox exchange: a peer aborted" : "mailbox exchange: a peer's data did not arrive within the communicator's timeout");
    }
    if (cm->comm && rccl().CommGetAsyncError) {
        ncclResult_t st = ncclSuccess;
        if (rccl().CommGetAsyncError(cm->comm, &st) == ncclSuccess && st != ncclSuccess && st != ncclInProgress)
            return cm->c->fail(CNIIC_ERR_RCCL, "RCCL reports an asynchronous error: %s", rccl().GetErrorString(st));
    }
    return CNIIC_OK;
}

uint64_t comm_timeout_ms(const Comm *cm) { return cm ? cm->timeout_ms : 0; }
void comm_set_timeout_ms(Comm *cm, uint64_t ms) { if (cm) cm->timeout_ms = ms; }

Ctx *comm_ctx(Comm *cm) { return cm->c; }
uint32_t comm_size(const Comm *cm) { return cm->nranks; }

// in-place sum over the ranks, on the context's stream; kind: 0 = u8, 1 = u32, 2 = u64
int comm_all_reduce(Comm *cm, void *buf_d, uint64_t count, int kind) {
    Ctx *c = cm->c;
    const ncclDataType_t dt = kind == 0 ? ncclUint8 : kind == 1 ? ncclUint32 : ncclUint64;
    if (kind < 0 || kind > 2) return c->fail(CNIIC_ERR_BAD_ARG, "all_reduce: unknown element kind %d", kind);
    if (cm->dead) return c->fail(CNIIC_ERR_RCCL, "all_reduce: the communicator was aborted after a failure");
    if (cm->mb) {
        CNIIC_TRY(comm_async_error(cm));
        return mailbox_all_reduce(cm->mb, buf_d, count, kind, cm->timeout_ms);
    }
    if (cm->host_sum) {  // through the host: drain the stream, bounce, let the caller's transport sum, put it back
        const int eb = kind == 0 ? 1 : kind == 1 ? 4 : 8;
        cm->bounce.resize((size_t)count * eb);
        CNIIC_HIP_TRY(c, hipMemcpyAsync(cm->bounce.data(), buf_d, cm->bounce.size(), hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (cm->host_sum(cm->user, cm->bounce.data(), count, eb) != 0) { cm->dead = true; return c->fail(CNIIC_ERR_RCCL, "all_reduce: the host transport failed"); }
        CNIIC_HIP_TRY(c, hipMemcpyAsync(buf_d, cm->bounce.data(), cm->bounce.size(), hipMemcpyHostToDevice, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        return CNIIC_OK;
    }
    const ncclResult_t e = rccl().AllReduce(buf_d, buf_d, (size_t)count, dt, ncclSum, cm->comm, c->stream);
    if (e != ncclSuccess) return c->fail(CNIIC_ERR_RCCL, "ncclAllReduce: %s", rccl().GetErrorString(e));
    return CNIIC_OK;
}

}  // namespace cniic
