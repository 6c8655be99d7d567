// common.hpp -- shared host-side plumbing of libcniic_hip.so (context, scratch HBM, error
// handling).  gfx950 only; no CPU fallback anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cniic_hip.h"

namespace cniic {

constexpr uint64_t kDefaultSeed = 0x636E696963ULL;

struct KernelTime {
    double   ms = 0.0;
    uint64_t launches = 0;
};

struct Ctx;

// Caching HBM allocator: hipMalloc / hipFree cost tens of microseconds and hipFree synchronises
// the device, so scratch buffers are recycled per context (a codec call allocates ~20 of them).
struct DevPool {
    struct Block { void *p; uint64_t cap; };
    std::vector<Block> free_blocks;
    uint64_t cached_bytes = 0;
    hipError_t get(uint64_t bytes, void **p, uint64_t *cap) {
        // best fit among cached blocks that are not more than 2x too large
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < free_blocks.size(); i++)
            if (free_blocks[i].cap >= bytes && free_blocks[i].cap <= 2 * bytes + 4096 &&
                (best == SIZE_MAX || free_blocks[i].cap < free_blocks[best].cap))
                best = i;
        if (best != SIZE_MAX) {
            *p = free_blocks[best].p;
            *cap = free_blocks[best].cap;
            cached_bytes -= *cap;
            free_blocks.erase(free_blocks.begin() + (long)best);
            return hipSuccess;
        }
        const uint64_t rounded = (bytes + 255) & ~255ull;
        hipError_t e = hipMalloc(p, rounded);
        if (e != hipSuccess && !free_blocks.empty()) {  // out of memory: drop the cache and retry
            trim();
            e = hipMalloc(p, rounded);
        }
        *cap = rounded;
        return e;
    }
    void put(void *p, uint64_t cap) {
        free_blocks.push_back({p, cap});
        cached_bytes += cap;
    }
    void trim() {
        for (auto &b : free_blocks) (void)hipFree(b.p);
        free_blocks.clear();
        cached_bytes = 0;
    }
};

// the pool of the context whose call is running on this thread (set by the ABI entry points)
inline DevPool *&current_pool() {
    static thread_local DevPool *p = nullptr;
    return p;
}

// RAII device allocation on the context's device (recycled through the context's pool).
struct DevBuf {
    void    *p = nullptr;
    uint64_t bytes = 0;
    uint64_t cap = 0;
    DevPool *pool = nullptr;
    bool     owned = true;   // false: a view into another DevBuf (never released)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes), cap(o.cap), pool(o.pool), owned(o.owned) { o.p = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; cap = o.cap; pool = o.pool; owned = o.owned; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    hipError_t alloc(uint64_t n) {
        release();
        if (n == 0) n = 16;
        pool = current_pool();
        hipError_t e;
        if (pool) e = pool->get(n, &p, &cap);
        else { e = hipMalloc(&p, n); cap = n; }
        if (e == hipSuccess) bytes = n; else p = nullptr;
        return e;
    }
    void view(void *ptr, uint64_t n) { release(); p = ptr; bytes = n; cap = 0; pool = nullptr; owned = false; }
    void release() {
        if (!p) return;
        if (owned) { if (pool) pool->put(p, cap); else (void)hipFree(p); }
        p = nullptr; bytes = 0; owned = true;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Knobs of the test-suite, the probes under tools/ and measuring builds (CNIIC_TEST_*, CNIIC_DBG_*, route forcing, tuning constants) are
// read from the environment ONLY by the testing build (-DCNIIC_TESTING: libcniic_hip_testing.so, which tests/conftest.py and tools/ ask
// for with CNIIC_USE_TESTING_LIB=1).  In the release library test_env() is a constant nullptr: no getenv, and the names are not even in
// the binary (tests/test_abi.py checks `strings`).  What a host may set on a release build are the context options of the ABI, whose
// documented environment fallbacks (Ctx::opt below) stay.
#ifdef CNIIC_TESTING
inline const char *test_env(const char *name) { return getenv(name); }
#else
inline const char *test_env(const char *) { return nullptr; }
#endif

struct Ctx {
    int         device = 0;
    hipStream_t stream = nullptr;
    bool        own_stream = false;
    std::mutex  mu;
    std::string err;
    std::map<std::string, KernelTime> ktimes;  // per-call dominant-kernel timings (HIP events)
    hipEvent_t  ev0 = nullptr, ev1 = nullptr;
    bool    timers = getenv("CNIIC_KERNEL_TIMERS") != nullptr;  // per-stage HIP-event timers (they synchronise)
    // cniic_ctx_set_opt: values a host set for this context (bit i of opt_set); unset options read their environment variable per call
    uint64_t opt_val[CNIIC_OPT_COUNT] = {};
    uint32_t opt_set = 0;
    uint64_t opt(int id, const char *env, uint64_t dflt) const {
        if (id > 0 && id < CNIIC_OPT_COUNT && ((opt_set >> id) & 1u)) return opt_val[id];
        const char *e = env ? getenv(env) : nullptr;
        return e ? strtoull(e, nullptr, 10) : dflt;
    }
    DevPool pool;           // recycled scratch HBM (all DevBufs created inside an ABI call)
    // persistent scratch: dense symbol tables (zeroed on demand), grown lazily
    DevBuf dense;           // u32[2^24] or u32[2^27]
    DevBuf dense27, dense27_pages;  // `delta`: u32[2^27] SignedColor counts + a flag per 4096-entry page, all zero between calls (k_delta.hip)
    bool   dense27_clean = false;
    DevBuf hilbert_lut;     // state-machine tables of the 2^n Hilbert scan (k_hilbert.hip)
    void  *pinned = nullptr; // 4 KiB of pinned host memory: two KmDevState slots for lagged convergence polling
    void  *pinned_ps = nullptr;  // pinned: how the persistent K-means launch ended (PsExit, k_kmeans_persist.hip)
    uint32_t ps_div = 1;         // the persistent K-means launch takes 1 / ps_div of the CUs (worker contexts of a batch encode)
    uint32_t ps_backoff = 0, ps_backoff_left = 0;   // after a persistent launch whose grid was not resident together in time (somebody else's kernels on the CUs: a 2 s
                                                    // wait before the launches take over) the next 4, 8, ... 256 K-means runs of this context do not try one
    hipEvent_t poll_ev[2] = {nullptr, nullptr};
    void  *pinned_res = nullptr;  // pinned landing area of K-means result blocks (grown on demand)
    uint64_t pinned_res_bytes = 0;
    hipEvent_t res_ev = nullptr;
    uint64_t *pinned_u = nullptr;  // 64 KiB of pinned host memory: [0] sp_build's distinct-colour count, [1] the point list's length,
                                   // [8 ..] this image's pixels per cluster (shared palette); u_ev: behind the copy of [0]
    hipEvent_t u_ev = nullptr;
    std::shared_ptr<void> trie_scratch; // the decoder's parsed leaf table (LeafTable, codec.cpp), kept between calls: 90 MB of fresh pages cost 25 ms
    std::shared_ptr<void> huf_scratch;  // host arrays of the Huffman tree build, kept between calls (HuffScratch, codec.cpp)
    void  *pinned_huf = nullptr;  // pinned host memory of a `delta` encode: distinct symbols, counts, codes, the serialised decoder
    uint64_t pinned_huf_bytes = 0;
    hipEvent_t huf_ev = nullptr;   // behind the D2H copies of the compacted histogram (huf_encode_all_dev)
    std::shared_ptr<void> scan_leaves;  // the built-in scan of large rectangles: per image size, the recursion's leaves and class tables (k_hilbert.hip)
    DevBuf scan_xy;         // cniic_ctx_set_scan: an injected scan of scan_w x scan_h images, (x, y) per position (uint2[w h])
    uint32_t scan_w = 0, scan_h = 0;
    std::vector<void *> batch_workers;  // cniic_codec_encode_batch: worker contexts (cniic_ctx *), created on first use
    const void *poll_owner = nullptr;  // the K-means state whose lagged polls own `pinned` / poll_ev (one loop at a time per context)

    int fail(int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define CNIIC_HIP_TRY(ctx, expr)                                                          \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return (ctx)->fail(CNIIC_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                               hipGetErrorString(_e));                                    \
    } while (0)

// Waits for the context's stream without giving the core up: a blocking wait of a few hundred microseconds sends the core to
// sleep, and the host work that follows (the Huffman tree of a `delta` encode) then runs at half speed.
inline hipError_t ctx_spin_sync(Ctx *c) {
    hipError_t e;
    while ((e = hipStreamQuery(c->stream)) == hipErrorNotReady) {}
    return e;
}

inline hipError_t ctx_pinned_huf(Ctx *c, uint64_t bytes) {  // at least `bytes` of pinned memory at c->pinned_huf (contents lost when it grows)
    if (c->pinned_huf_bytes >= bytes) return hipSuccess;
    if (c->pinned_huf) (void)hipHostFree(c->pinned_huf);
    c->pinned_huf = nullptr; c->pinned_huf_bytes = 0;
    const uint64_t want = bytes + bytes / 4 + 65536;
    const hipError_t e = hipHostMalloc(&c->pinned_huf, want, hipHostMallocDefault);
    if (e == hipSuccess) c->pinned_huf_bytes = want;
    return e;
}

inline hipError_t ctx_pinned_u(Ctx *c) {
    if (c->pinned_u) return hipSuccess;
    return hipHostMalloc(reinterpret_cast<void **>(&c->pinned_u), 64 * 1024, hipHostMallocDefault);
}

#define CNIIC_TRY(expr)              \
    do {                             \
        int _rc = (expr);            \
        if (_rc != CNIIC_OK) return _rc; \
    } while (0)

// Is p device-accessible memory (hipMalloc / torch allocation)?  Host pointers unknown to HIP
// report hipErrorInvalidValue, which is cleared.
inline bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// Input view: device pointer used as is, host pointer staged into an owned HBM buffer.
template <class T> struct In {
    const T *d = nullptr;
    DevBuf   own;
    int bind(Ctx *c, const T *p, uint64_t n) {
        if (n == 0 || !p) { d = nullptr; return CNIIC_OK; }
        if (is_device_ptr(p)) { d = p; return CNIIC_OK; }
        CNIIC_HIP_TRY(c, own.alloc(n * sizeof(T)));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(own.p, p, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
        d = own.as<T>();
        return CNIIC_OK;
    }
};

// Output view: device pointer written in place, host pointer filled by a D2H copy in finish().
template <class T> struct Out {
    T       *d = nullptr;
    T       *host = nullptr;
    uint64_t n = 0;
    DevBuf   own;
    int bind(Ctx *c, T *p, uint64_t count) {
        n = count;
        if (!p || count == 0) { d = nullptr; host = nullptr; return CNIIC_OK; }
        if (is_device_ptr(p)) { d = p; host = nullptr; return CNIIC_OK; }
        CNIIC_HIP_TRY(c, own.alloc(count * sizeof(T)));
        d = own.as<T>();
        host = p;
        return CNIIC_OK;
    }
    // copy the first `count` elements back (async; caller syncs the stream)
    int finish(Ctx *c, uint64_t count) {
        if (host && count) CNIIC_HIP_TRY(c, hipMemcpyAsync(host, d, count * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        return CNIIC_OK;
    }
    int finish(Ctx *c) { return finish(c, n); }
};

inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// Device-resident state of a K-means loop (kmeans.rs:21-39): lets iterations be enqueued back to
// back; kernels of iterations past convergence exit on `done`.
constexpr uint32_t kHistRing = 64;
struct KmDevState {
    uint64_t iter;         // iterations completed
    uint32_t done;         // 1 once an iteration moved nothing (kmeans.rs:26) or max_iters was hit
    uint32_t pad;
    uint64_t moved_last;
    uint64_t reseeds;
    uint64_t active;
    uint64_t pair_evals;
    uint64_t changed_ring[kHistRing];
    uint32_t nmoved_ring[kHistRing];  // centroids whose value changed in the update that closed iteration i (colour K-means: picks the schedule of the next launch)
};

// One launch's view of the scalar state, written by that launch into slot (launch number % kPollRing) of a pinned ring:
// what the host reads for launch L is then the same on every rank, whatever the timing (multi-GPU loops must all leave
// after the same batch), without a copy kernel in the stream.
struct PollRec { uint64_t iter; uint32_t done; uint32_t seq; uint64_t moved_last, reseeds, active, pair_evals; };
constexpr uint32_t kPollRing = 16;

struct Comm;
void comm_abort(Comm *cm);
int comm_async_error(Comm *cm);
uint64_t comm_timeout_ms(const Comm *cm);

// Lagged convergence polling.  The K-means loops enqueue batches of (assign, update) launches; every
// kernel exits at once when the device-side `done` flag is set.  After each batch the state is copied to
// a pinned slot and an event recorded, but the host only waits for the copy of the PREVIOUS batch, so the
// GPU never idles for a host round trip; the price is at most one extra batch of no-op launches.
struct LaggedPoll {
    Ctx *c;
    const void *dstate;
    int slot = 0, pending = 0;
    bool mapped = false;  // the kernels write the state into the third pinned slot themselves: no copy to enqueue
    bool ring = false;    // ... or every launch into its own slot of the ring behind the three slots (deterministic per launch)
    uint32_t last_prev = 0;
    Comm *watch = nullptr;  // set by loops with collectives
    bool owner = false;     // this poll holds the context's slots (Ctx::poll_owner)
    LaggedPoll(Ctx *ctx, const void *dstate_d) : c(ctx), dstate(dstate_d) {}
    LaggedPoll(const LaggedPoll &) = delete;
    LaggedPoll &operator=(const LaggedPoll &) = delete;
    ~LaggedPoll() { if (owner && c->poll_owner == dstate) c->poll_owner = nullptr; }
    static constexpr size_t ring_offset() { return (3 * sizeof(KmDevState) + 63) & ~size_t(63); }
    int ring_slot(PollRec **dev_ptr) {
        static_assert(ring_offset() + kPollRing * sizeof(PollRec) <= 4096, "the poll ring must fit the pinned page");
        PollRec *r = reinterpret_cast<PollRec *>(static_cast<uint8_t *>(c->pinned) + ring_offset());
        memset(r, 0xff, kPollRing * sizeof(PollRec));  // no slot carries a valid launch number yet
        void *d = nullptr;
        CNIIC_HIP_TRY(c, hipHostGetDevicePointer(&d, r, 0));
        *dev_ptr = static_cast<PollRec *>(d);
        ring = true;
        return CNIIC_OK;
    }
    // device address of the slot the kernels write in mapped mode (zeroed here)
    int mapped_slot(KmDevState **dev_ptr) {
        KmDevState *slots = static_cast<KmDevState *>(c->pinned);
        memset(&slots[2], 0, sizeof(KmDevState));
        void *d = nullptr;
        CNIIC_HIP_TRY(c, hipHostGetDevicePointer(&d, &slots[2], 0));
        *dev_ptr = static_cast<KmDevState *>(d);
        mapped = true;
        return CNIIC_OK;
    }
    int prepare() {
        static_assert(3 * sizeof(KmDevState) <= 4096, "the poll slots must fit the pinned page");
        // The slots, the ring and the two events are the CONTEXT's: one K-means loop at a time polls through them.  A second
        // state that starts polling while another still holds them would read the first one's records -- refused instead.
        if (c->poll_owner && c->poll_owner != dstate)
            return c->fail(CNIIC_ERR_BAD_ARG, "kmeans: another K-means state of this context is polling (one lagged-poll loop per context at a time; "
                                              "destroy it, or give the second session its own context)");
        c->poll_owner = dstate;
        owner = true;
        if (!c->pinned) CNIIC_HIP_TRY(c, hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault));
        for (int i = 0; i < 2; i++)
            if (!c->poll_ev[i]) CNIIC_HIP_TRY(c, hipEventCreateWithFlags(&c->poll_ev[i], hipEventDisableTiming));
        return CNIIC_OK;
    }
    // call after enqueuing a batch; returns the state as of the batch BEFORE it in *h (valid when *have).
    // last_launch: number of the batch's last launch (ring mode)
    int after_batch(KmDevState *h, bool *have, uint32_t last_launch = 0) {
        KmDevState *slots = static_cast<KmDevState *>(c->pinned);
        if (!mapped && !ring) CNIIC_HIP_TRY(c, hipMemcpyAsync(&slots[slot], dstate, sizeof(KmDevState), hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipEventRecord(c->poll_ev[slot], c->stream));
        *have = pending > 0;
        if (pending) {
            if (watch) {  // a loop with collectives: a peer that failed leaves this rank's stream stuck in an all-reduce -- look at the communicator while waiting
                // A dead peer is not always reported through ncclCommGetAsyncError (intra-node P2P / SHM transports), so the wait has
                // a deadline of its own: when a batch has not finished after watch_timeout_ms (cniic_comm_set_timeout /
                // CNIIC_COLLECTIVE_TIMEOUT_MS, default 120 s; 0 = wait for ever) the communicator is aborted -- by km_rgbw_run, on any
                // error return -- and the call ends with CNIIC_ERR_RCCL instead of hanging.
                const uint64_t limit_ms = comm_timeout_ms(watch);
                const auto t_wait = std::chrono::steady_clock::now();
                for (;;) {
                    const hipError_t q = hipEventQuery(c->poll_ev[slot ^ 1]);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) return c->fail(CNIIC_ERR_HIP, "kmeans: waiting for a batch: %s", hipGetErrorString(q));
                    CNIIC_TRY(comm_async_error(watch));
                    if (limit_ms && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_wait).count() >= (long long)limit_ms)
                        return c->fail(CNIIC_ERR_RCCL, "kmeans: a batch with collectives did not finish within %llu ms (a peer is gone?): communicator aborted",
                                       (unsigned long long)limit_ms);
                    std::this_thread::sleep_for(std::chrono::microseconds(20));
                }
            }
            CNIIC_HIP_TRY(c, hipEventSynchronize(c->poll_ev[slot ^ 1]));
            if (ring) {  // the record the previous batch's last launch wrote: complete before its event, untouched for kPollRing launches
                const volatile PollRec *r = reinterpret_cast<const volatile PollRec *>(static_cast<uint8_t *>(c->pinned) + ring_offset()) + last_prev % kPollRing;
                if (r->seq != last_prev) return c->fail(CNIIC_ERR_HIP, "kmeans: launch %u left no state record (found %u)", last_prev, r->seq);
                KmDevState t{};
                t.done = r->done; t.iter = r->iter; t.moved_last = r->moved_last; t.reseeds = r->reseeds; t.active = r->active; t.pair_evals = r->pair_evals;
                *h = t;
            } else if (mapped) {  // at least as new as that batch; the flag is read first, the scalars it guards after it
                volatile KmDevState *m = &slots[2];
                KmDevState t{};
                t.done = m->done;
                std::atomic_thread_fence(std::memory_order_acquire);
                t.iter = m->iter; t.moved_last = m->moved_last; t.reseeds = m->reseeds; t.active = m->active; t.pair_evals = m->pair_evals;
                *h = t;
            } else {
                *h = slots[slot ^ 1];
            }
        }
        slot ^= 1;
        pending = 1;
        last_prev = last_launch;
        return CNIIC_OK;
    }
    // state after everything enqueued so far
    int drain(KmDevState *h) {
        KmDevState *slots = static_cast<KmDevState *>(c->pinned);
        CNIIC_HIP_TRY(c, hipEventSynchronize(c->poll_ev[slot ^ 1]));
        *h = slots[slot ^ 1];
        return CNIIC_OK;
    }
};

// ---- host-side stage marks (CNIIC_TRACE_HOST=1): where an ABI call spends its wall time ----
struct HostTrace {
    bool on;
    std::vector<std::pair<const char *, double>> marks;
    HostTrace() : on(test_env("CNIIC_TRACE_HOST") != nullptr) {}
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void mark(const char *what) { if (on) marks.emplace_back(what, now()); }
    void dump() {
        if (!on || marks.empty()) return;
        for (size_t i = 1; i < marks.size(); i++) fprintf(stderr, "[host] %-28s %8.1f us\n", marks[i].first, marks[i].second - marks[i - 1].second);
        fprintf(stderr, "[host] %-28s %8.1f us\n", "total", marks.back().second - marks.front().second);
        marks.clear();
    }
};
HostTrace &host_trace();

// ---- kernel timing of the dominant kernels (HIP events on the ctx stream) ----
struct ScopedKernelTimer {
    Ctx        *c;
    const char *name;
    bool        on;
    // stop() waits for the GPU, so stage timers are off unless asked for (CNIIC_KM_PROFILE in the call's
    // options, or CNIIC_KERNEL_TIMERS=1 in the environment)
    ScopedKernelTimer(Ctx *ctx, const char *nm) : ScopedKernelTimer(ctx, nm, ctx->timers) {}
    ScopedKernelTimer(Ctx *ctx, const char *nm, bool enable) : c(ctx), name(nm), on(enable) {
        if (on) (void)hipEventRecord(c->ev0, c->stream);
    }
    // must be called after the launch; synchronises on the stop event
    void stop(uint64_t launches = 1) {
        if (!on) return;
        (void)hipEventRecord(c->ev1, c->stream);
        (void)hipEventSynchronize(c->ev1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
        KernelTime &kt = c->ktimes[name];
        kt.ms += ms;
        kt.launches += launches;
        on = false;
    }
};

// =========================================================================== device entry points
// (host launchers implemented in the .hip files)

// ---- k_hist.hip ----
// Dense-table histogram of packed keys.  table: u32[1<<bits], must be zero on entry.
int hist_rgb_dense(Ctx *c, const uint8_t *rgb_d, uint64_t npx, uint32_t *table_d);
int hist_syms_dense(Ctx *c, const uint32_t *syms_d, uint64_t n, uint32_t *table_d, uint32_t bits);
// Compaction of a dense count table into ascending (key,count) pairs (3-phase scan).
struct CompactPlan {
    DevBuf   blockoff, blockmax;
    uint64_t n_unique = 0;
    uint64_t max_count = 0;          // the largest count in the table (how many radix passes the leaves' sort by count needs)
    uint32_t bits = 0;
    const uint8_t *pages = nullptr;  // (optional) a flag per 4096-entry page: pages without one are skipped
};
// occupancy across ranks: nibble per key (k_hist.hip), and the index of all occupied keys built from the summed nibbles
int occupancy_pack(Ctx *c, const uint32_t *table_d, uint32_t *occ_d);  // u32[2^24] counts -> u32[2^21] nibble words
int gidx_build(Ctx *c, const uint32_t *occ_d, DevBuf &bits, DevBuf &wprefix, uint64_t *U_h, DevBuf *total_keep = nullptr);
int gidx_finish(Ctx *c, uint32_t *wprefix_d, const uint32_t *blocktot_d, uint64_t *total_d);  // 256 blocks of 1024 words

// ---- k_points.hip: pixels partitioned by colour super-cell (large cluster-colors encodes) ----
struct SpPlan {
    uint64_t npx = 0, U = 0;     // pixels, distinct colours
    uint32_t nchunks = 0;
    DevBuf cnt, pre, bstart;     // pixels per (chunk, bucket), those in earlier chunks, first entry of every bucket
    DevBuf part, prank;          // u16 per pixel: colour inside its bucket (bucket order), place in its chunk's bucket-sorted order (pixel order)
    DevBuf cell_count;           // distinct colours per K-means cell
    DevBuf sstart, sbin, scnt;   // per bucket: its distinct colours (colour inside the bucket, pixel count) staged in cell-major order
    DevBuf bits, wprefix;        // occupancy bitmap of the 2^24 colours + popcount prefix (GIdx)
    DevBuf total;                // u64 on the device: distinct colours (the kernels of the set-up read it there)
};
int sp_build(Ctx *c, const uint8_t *rgb_d, uint64_t npx, SpPlan *plan);  // asynchronous: the count arrives with sp_wait_count
int sp_wait_count(Ctx *c, SpPlan *plan);                                  // plan->U on the host (waits for its copy only)
// distinct colours, counts, initial labels -> the K-means state's cell-major arrays; (gbits, gprefix, Ug): index of the point list
int sp_emit(Ctx *c, const SpPlan *plan, const uint32_t *cell_start_d, uint32_t *ckeys_d, uint32_t *cweight_d, void *labels_d, bool wide,
            uint32_t K, const void *gbits_d, const uint32_t *gprefix_d, uint64_t Ug, const uint64_t *Ug_dev /* overrides Ug when set */);
int sp_occupancy(Ctx *c, const SpPlan *plan, uint32_t *occ_d);  // this image's colours as summable nibbles, u32[2^21]
// final cell-major labels -> label of every pixel, image order
int sp_pixel_labels(Ctx *c, const SpPlan *plan, const uint8_t *rgb_d, const uint32_t *cell_start_d, const uint32_t *ckeys_d,
                    const void *labels_d, bool wide, void *pixlab_d);
// cell_count_d (optional, 24-bit tables): zeroed u32[32768] receiving the occupied bins per K-means colour cell
int hist_compact_count(Ctx *c, const uint32_t *table_d, uint32_t bits, CompactPlan *plan, uint32_t *cell_count_d = nullptr,
                       const uint8_t *pages_d = nullptr);
// After this call the table holds, for every occupied bin, its RANK (index into the compacted
// list) + 1; empty bins stay 0.  Outputs are optional device arrays of plan->n_unique entries.
int hist_compact_write(Ctx *c, uint32_t *table_d, const CompactPlan *plan, uint32_t *keys_d, uint64_t *counts_d,
                       uint32_t *weights_d);
int dense_table(Ctx *c, uint32_t bits, uint32_t **table_d);  // zeroed scratch table of the ctx

// ---- k_kmeans_rgbw.hip ----
struct KmRgbwState;  // opaque device state of one rgbw K-means problem
int km_rgbw_create(Ctx *c, const uint32_t *keys_d, const uint32_t *weight_d, uint64_t U, uint32_t shard,
                   uint32_t nshards, uint32_t K, const cniic_kmeans_opts *opts, void *partials_dev,
                   const uint32_t *rank_table_d /* dense key -> rank+1 table, or null */, KmRgbwState **out,
                   const uint32_t *cell_count_d = nullptr /* with rank_table_d: points per cell, if already counted */,
                   const void *gbits_d = nullptr, const uint32_t *gprefix_d = nullptr, uint64_t Ug = 0
                   /* the points are this rank's share of Ug colours: positions in the list of all occupied keys */,
                   bool points_follow = false /* keys_d / weight_d null: the caller writes the cell-major arrays (sp_emit) */,
                   const uint64_t *points_dev = nullptr /* with points_follow: U and Ug are upper bounds, the count is here
                                                           on the device; km_rgbw_set_points before anything else */);
int km_rgbw_set_points(KmRgbwState *s, uint64_t U, uint64_t Ulist = 0 /* 0: the points are the whole list */);
void km_rgbw_cell_arrays(KmRgbwState *s, uint32_t **cell_start_d, uint32_t **ckeys_d, uint32_t **cweight_d);
void km_rgbw_destroy(KmRgbwState *s);
int km_rgbw_set_state(KmRgbwState *s, const uint8_t *centroids_h, const uint32_t *labels_d_u32);
int km_rgbw_assign(KmRgbwState *s);                       // async: assign + partial sums -> partials
int km_rgbw_update(KmRgbwState *s);                       // async: centroids from partials
// ---- comm.cpp: RCCL on the context's stream (bound at run time) ----
struct Comm;
int comm_unique_id(uint8_t *id128);
int comm_create(Ctx *c, const uint8_t *id128, uint32_t rank, uint32_t nranks, Comm **out);
int comm_create_host(Ctx *c, uint32_t rank, uint32_t nranks, int32_t (*fn)(void *, void *, uint64_t, int32_t), void *user, Comm **out);
void comm_destroy(Comm *cm);
Ctx *comm_ctx(Comm *cm);
uint32_t comm_size(const Comm *cm);
int comm_all_reduce(Comm *cm, void *buf_d, uint64_t count, int kind);  // in place, sum; kind 0 = u8, 1 = u32, 2 = u64
void comm_abort(Comm *cm);        // after a failure on this rank: the peers' collectives end with an error instead of hanging
int comm_async_error(Comm *cm);   // CNIIC_OK while healthy; an error once this rank aborted or the transport reports a peer's failure
uint64_t comm_timeout_ms(const Comm *cm);           // deadline of a loop's wait for a batch with collectives (0: none)
void comm_set_timeout_ms(Comm *cm, uint64_t ms);
int comm_create_mailbox(Ctx *c, uint32_t rank, uint32_t nranks, uint64_t max_bytes, uint8_t *handle64, Comm **out);
int comm_connect_mailbox(Comm *cm, const uint8_t *handles);

// ---- k_mailbox.hip: the one-shot exchange (every rank writes its buffer into a slot of every peer's mailbox)
struct Mailbox;
int mailbox_create(Ctx *c, uint32_t rank, uint32_t nranks, uint64_t cap_bytes, uint8_t *handle64, Mailbox **out);
int mailbox_connect(Mailbox *m, const uint8_t *handles);  // nranks x 64 bytes, in rank order
int mailbox_all_reduce(Mailbox *m, void *buf_d, uint64_t count, int kind, uint64_t timeout_ms);
int mailbox_status(const Mailbox *m);  // 0 healthy, 1 a wait ran out, 2 a peer aborted
void mailbox_abort(Mailbox *m);
void mailbox_destroy(Mailbox *m);

constexpr int kKmRetry = 1000;    // km_rgbw_result_end after a deferred run: the persistent launch had given up, the launch-per-iteration loop has run since -- what the
                                  // caller enqueued on the labels must be enqueued again (never returned through the C ABI)
int km_rgbw_run(KmRgbwState *s, Comm *cm = nullptr, bool may_defer = false);   // full loop to convergence; with cm the partial sums are all-reduced in-stream each iteration;
                                                           // may_defer: the caller goes on to km_rgbw_result_begin / _end (which may answer kKmRetry, kmeans_rgbw.hpp) and wants no wait in between
int km_rgbw_poll_changed(KmRgbwState *s, uint64_t *changed);  // syncs
int km_rgbw_poll(KmRgbwState *s, cniic_kmeans_stats *st, uint32_t *done);  // syncs
bool km_rgbw_run_stats(KmRgbwState *s, cniic_kmeans_stats *st);  // of the last km_rgbw_run; no stream work
int km_rgbw_poll_lagged(KmRgbwState *s, cniic_kmeans_stats *st, uint32_t *done, uint32_t *have);  // waits for the previous call's copy only
int km_rgbw_result(KmRgbwState *s, uint8_t *centroids_h, uint32_t *labels_d_u32, uint64_t *members_h,
                   uint64_t *wsum_h, cniic_kmeans_stats *stats);
// the same in two halves: _begin enqueues the copy of the result block, _end waits for it (work enqueued
// in between overlaps the wait and whatever the host does with the result)
int km_rgbw_result_begin(KmRgbwState *s);
int km_rgbw_result_end(KmRgbwState *s, uint8_t *centroids_h, uint64_t *members_h, uint64_t *wsum_h, cniic_kmeans_stats *stats);
int km_rgbw_time_assign(KmRgbwState *s, int reps, double *ms_per_launch);
int km_rgbw_partials(KmRgbwState *s, uint64_t *sums_h, uint64_t *wsum_h, uint64_t *members_h, uint64_t *changed_h);
void *km_rgbw_partials_dev(KmRgbwState *s);
const uint32_t *km_rgbw_centroids_dev(KmRgbwState *s);
bool km_rgbw_is_wide(KmRgbwState *s);                     // u16 labels (K > 256) instead of u8
int km_rgbw_labels_canonical(KmRgbwState *s, void *dst_d); // u8/u16 labels of all points, canonical order
void *km_rgbw_labels_internal(KmRgbwState *s, uint64_t *elem_bytes);
int km_rgbw_fold_initial(KmRgbwState *s);
// sharded runs: labels of this shard's cells (zeros elsewhere) out of / merged labels into the state
int km_rgbw_export_labels(KmRgbwState *s, void *dst_d);
int km_rgbw_import_labels(KmRgbwState *s, const void *src_d);
uint64_t km_rgbw_points(KmRgbwState *s);

// ---- k_kmeans_xyrgb.hip ----
int km_xyrgb_run(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K,
                 const cniic_kmeans_opts *opts, cniic_colorpos *centroids_h, uint32_t *labels_d_u32,
                 uint64_t *members_h, cniic_kmeans_stats *stats);
int km_xyrgb_run_wide(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K, const cniic_kmeans_opts *opts, cniic_colorpos *centroids_h,
                      uint32_t *labels_d_u32, uint64_t *members_h, cniic_kmeans_stats *stats);   // k_kmeans_wide.hip: any K, any sides
int km_xyrgb_step(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K,
                  const cniic_colorpos *centroids_h, uint32_t *labels_d_u32, uint64_t *sums_h,
                  uint64_t *wsum_h, uint64_t *members_h, uint64_t *changed_h, const cniic_kmeans_opts *opts);

// ---- k_misc.hip ----
// per-pixel colour -> centroid colour through the rank table left by hist_compact
int remap_rgb(Ctx *c, const uint8_t *rgb_d, uint64_t npx, const uint32_t *rank_table_d,
              const uint32_t *lut_rgb_d /* packed centroid colour per unique colour rank */, uint8_t *out_d);
int expand_codes_by_label(Ctx *c, const uint8_t *labels8_d, const uint16_t *labels16_d, uint64_t U, const uint8_t *clen_d,
                          const uint64_t *ccode_d, uint8_t *len_d, uint64_t *code_d);
// (local_counts_d == nullptr: the weights come from weights_d[i] instead of local_counts_d[keys_d[i]])
int local_cluster_weights(Ctx *c, const uint32_t *keys_d, const void *labels_d, bool wide, uint64_t U,
                          const uint32_t *local_counts_d, uint32_t K, uint64_t *out_d, const uint32_t *weights_d = nullptr);
int label_lut(Ctx *c, const uint32_t *labels_d, uint64_t U, const uint32_t *cent_d, uint32_t *lut_d);
int rank_from_keys(Ctx *c, const uint32_t *keys_d, uint64_t U, uint32_t *table_d);
int voronoi_paint(Ctx *c, const cniic_colorpos *cent_d, uint32_t K, uint32_t w, uint32_t h, uint8_t *out_d, bool small_coords = false);
int mse_rgb(Ctx *c, const uint8_t *a_d, const uint8_t *b_d, uint64_t npx, double *mse_h);
int synth_image(Ctx *c, int kind, uint64_t seed, uint32_t w, uint32_t h, uint8_t *out_d);
int rgb_to_keys(Ctx *c, const uint8_t *rgb_d, uint64_t npx, uint32_t *keys_d);

// ---- colour-space cells of the cluster-colors K-means (numbering: cell_of in device_utils.hpp) ----
#ifndef CNIIC_CELL_SHIFT
#define CNIIC_CELL_SHIFT 3
#endif
constexpr int kCellShift = CNIIC_CELL_SHIFT;                    // 2^shift colours per cell side
constexpr uint32_t kCellsPerDim = 256 >> kCellShift;           // 32
constexpr uint32_t kNumCells = kCellsPerDim * kCellsPerDim * kCellsPerDim;  // 32768

// ---- k_hdecode.hip: parallel Huffman decode (self-synchronising subsequences), keys -> pixels, FromDiff as a scan ----
struct LeafTable;
struct UdSums { DevBuf buf; bool filled = false; };   // FromDiff's channel sums per 4096 symbols, added up by the decoder while it writes them (k_hdecode.hip)
// payload in host memory, or (payload_dev) anywhere in HBM; mode 0: out_d = nsyms packed keys (u32, 16-byte aligned), mode 1: nsyms RGB
// triples (4-byte aligned).  status: 0 ok, 1 stream ends early, 2 did not settle (decode on the host)
int huff_decode_dev(Ctx *c, const LeafTable &lt, const uint8_t *payload, bool payload_dev, uint64_t payload_bytes, uint64_t nsyms,
                    int mode, void *out_d, int *status, UdSums *sums = nullptr);
// ... with the table of leaves already in HBM (code u64[n] | key u32[n] at off_key | len u8[n] at off_len)
int huff_decode_tables_dev(Ctx *c, const uint8_t *tab_d, uint64_t n, uint64_t off_key, uint64_t off_len, uint32_t max_len, uint32_t first_key,
                           const uint8_t *payload, bool payload_dev, uint64_t payload_bytes, uint64_t nsyms, int mode, void *out_d, int *status, UdSums *sums = nullptr);
// k_trieparse.hip: Dec::deserialize on the GPU (decoders of millions of leaves).  status: 0 ok, 1 malformed / truncated, 2 too deep
int huff_parse_leaves_dev(Ctx *c, int sym_kind, const uint8_t *stream_d, uint64_t nbytes, uint64_t pos0, DevBuf *tab, uint64_t *n_leaves,
                          uint64_t *off_key, uint64_t *off_len, uint32_t *max_len, uint64_t *payload_pos, int *status);
int keys_to_rgb(Ctx *c, const uint32_t *keys_d, uint64_t n, uint8_t *rgb_d);
int delta_undiff_dev(Ctx *c, const uint32_t *keys_d, uint64_t n, uint8_t *lin_d, uint32_t *bad_h);
int delta_undiff_scatter_dev(Ctx *c, const uint32_t *keys_d, uint32_t w, uint32_t h, uint8_t *rgb_out_d, uint32_t *bad_h, UdSums *sums = nullptr);  // + the walk along the scan

// ---- k_rle.hip: exact run-length coding of a linearised image (hilbertc.rs:100-196) ----
struct RlePlan { uint64_t n = 0, nruns = 0; uint32_t nchunks = 0; DevBuf flags, run_off; };
int rle_plan(Ctx *c, const uint8_t *lin_d, uint64_t n, RlePlan *plan);                       // counts the runs (syncs)
int rle_emit(Ctx *c, const uint8_t *lin_d, const RlePlan *plan, uint32_t *out_words_d);      // 12-byte records
int rle_expand_dev(Ctx *c, const uint8_t *rec_d, uint64_t R, uint64_t tail_bytes, uint64_t n, uint8_t *lin_d, int *status);  // RleDecoder

// ---- k_hilbert.hip ----
int hilbert_xy(Ctx *c, uint32_t w, uint32_t h, uint32_t *xy_d);
int hilbert_linearize(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint8_t *out_d);
// gather + delta; syms_d (packed SIGNED keys, may be null) and/or histogram into table_d (u32[2^27], may be null)
int hilbert_delta(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t *syms_d, uint32_t *table_d);
int hilbert_scatter(Ctx *c, const uint8_t *lin_d, uint32_t w, uint32_t h, uint8_t *rgb_out_d);
int scan_inject(Ctx *c, uint32_t w, uint32_t h, const uint32_t *xy, bool xy_dev);   // cniic_ctx_set_scan (xy == null: the built-in scan again)
int hilbert_undiff_scatter(Ctx *c, const uint32_t *keys_d, const int32_t *chunk_off_d, uint32_t w, uint32_t h, uint8_t *rgb_out_d, uint32_t *bad_d, bool *fused);

// ---- k_delta.hip: the `delta` encoder's passes over a 16-bit symbol stream ----
constexpr uint32_t kPageShift = 12;  // a page of a dense table = the 4096 entries one block of the compaction reads
int delta_table(Ctx *c, uint32_t **table_d, uint8_t **pages_d);  // the context's clean 2^27-bin table + page flags
int delta_table_clean(Ctx *c);                                   // touched pages back to zero (enqueued)
uint64_t delta_stream_len(uint64_t n);                           // u16 entries of the symbol stream of n pixels
int delta_gather_hist(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint16_t *hot16_d, uint32_t *table_d, uint8_t *pages_d,
                      uint32_t *coldkeys_d, uint8_t *chunk_cold_d, uint32_t *overflow_d);
struct DeltaPackScratch { DevBuf cb, co, edge, hot, hotlen; };
int delta_pack16_count(Ctx *c, const uint16_t *hot16_d, uint64_t n, uint32_t *coldkeys_d, const uint8_t *chunk_cold_d, uint32_t *dense_d,
                       const uint32_t *keys_d, const uint8_t *len_d, const uint64_t *code_d, uint64_t U, uint64_t *total_d, DeltaPackScratch *keep);
int delta_pack16_write(Ctx *c, const uint16_t *hot16_d, uint64_t n, const uint32_t *coldkeys_d, const uint8_t *len_d, const uint64_t *code_d, uint8_t *out_d,
                       uint64_t bit_base, DeltaPackScratch *keep);

// ---- k_huff.hip ----
// the leaves count << 32 | rank sorted by (count, rank): stable radix sort (see k_huff.hip); the result is in *out_d = buf_a or buf_b
int huff_sort_leaves_dev(Ctx *c, const uint64_t *counts_d, uint32_t n, uint64_t max_count, uint64_t *buf_a, uint64_t *buf_b, uint64_t **out_d);
int huff_sort_u64(Ctx *c, uint64_t *buf_a, uint64_t *buf_b, uint32_t n, uint32_t lo_bit, uint64_t **out_d);
// the tree, the codes and every leaf's place in the serialised decoder from the sorted leaves, the host merging RUNS of equal count only
int huff_tree_from_runs(Ctx *c, const uint64_t *sorted_d, const uint64_t *counts_d, uint32_t n, int sym_kind, uint8_t *len_d, uint64_t *code_d,
                        uint64_t *off_d, uint64_t *nbits_h, bool *built);
// codes and the serialised decoder of a tree the host built, for large alphabets (see k_huff.hip)
int huff_tree_codes(Ctx *c, const uint32_t *left_d, const uint32_t *right_d, const uint32_t *nleaves_d, const uint64_t *counts_d, uint32_t n,
                    uint32_t root, int sym_kind, uint8_t *len_d, uint64_t *code_d, uint64_t *off_d, uint64_t *totals_d);
int huff_tree_serialize_dev(Ctx *c, const uint32_t *keys_d, const uint64_t *off_d, uint32_t n, int sym_kind, uint8_t *trie_d, uint64_t trie_bytes);
// chunk_off[i] = exclusive prefix (u64) of chunk_bits[0 .. nchunks); *total_d = the sum
int pack_scan(Ctx *c, const uint32_t *chunk_bits_d, uint32_t nchunks, uint64_t *chunk_off_d, uint64_t *total_d);
uint32_t pack_img_cap();  // tests: CNIIC_TEST_PACK_IMG_WORDS caps the packs' LDS bit image
// MSB-first bit-pack of n symbols at bit offset bit_base of out_d (4-byte aligned, pre-zeroed,
// large enough; bytes before bit_base may already hold the stream header).
// Generic path: symbol -> rank (dense table left by the compaction) -> len_d / code_d per rank.
int huff_pack_keys(Ctx *c, const uint32_t *keys_or_null_d, const uint8_t *rgb_or_null_d, uint64_t n,
                   const uint32_t *rank_table_d, const uint8_t *len_d, const uint64_t *code_d,
                   uint8_t *out_d, uint64_t bit_base, uint64_t *nbits_h);
// the same in two steps: the symbols' ranks as a stream (needs no code: runs while the host builds the tree), then the pack
int huff_rank_stream(Ctx *c, const uint32_t *keys_or_null_d, const uint8_t *rgb_or_null_d, uint64_t n, const uint32_t *rank_table_d,
                     uint32_t *ranks_d, bool one_based);
int huff_pack_ranks(Ctx *c, const uint32_t *ranks_d, uint64_t n, const uint8_t *len_d, const uint64_t *code_d, uint8_t *out_d,
                    uint64_t bit_base, uint64_t *nbits_h);
// same result with one random read per symbol (needs every code length <= 26): the dense table is
// overwritten with len<<26|code per key; packed_d = n u32 of scratch (may alias syms)
int huff_pack_code32(Ctx *c, const uint32_t *syms_or_null_d, const uint8_t *rgb_or_null_d, uint64_t n, uint32_t *table_d,
                     const uint32_t *keys_d, const uint8_t *len_d, const uint64_t *code_d, uint64_t U, uint32_t *packed_d,
                     uint8_t *out_d, uint64_t bit_base, uint64_t *nbits_h);
// the same for SignedColor (`delta`) symbols: the (length, code) words of the cube of small differences sit in LDS
int huff_pack_code32_hot(Ctx *c, const uint32_t *syms_d, uint64_t n, uint32_t *dense_d, const uint32_t *keys_d, const uint8_t *len_d,
                         const uint64_t *code_d, uint64_t U, uint32_t *packed_d, uint8_t *out_d, uint64_t bit_base, uint64_t *nbits_h);
// cluster-colors path: pixel -> cluster label through a dense colour->label table, codes per cluster
int pixel_labels(Ctx *c, const uint8_t *rgb_d, uint64_t n, const void *key2label_d, bool wide, void *pixlab_d);
int huff_pack_labels(Ctx *c, const void *pixlab_d, uint64_t n, bool wide, uint32_t K, const uint8_t *clen_d,
                     const uint64_t *ccode_d, uint8_t *out_d, uint64_t bit_base, uint64_t *nbits_h);
int scatter_labels_by_key(Ctx *c, const uint32_t *keys_d, const void *labels_d, bool wide, uint64_t U, void *key2label_d);
// a batch of equally sized frames sharing one palette: labels per (frame, cluster), and the label pack of every frame in one go
int frame_label_hist(Ctx *c, const void *pixlab_d, uint64_t npf, uint64_t lab_stride, uint32_t frames, bool wide, uint32_t K, uint32_t *out_d /* u32[frames][K] */);
int huff_pack_labels_frames(Ctx *c, const void *pixlab_d, uint64_t npf, uint64_t lab_stride, uint32_t frames, bool wide, uint32_t K, const uint8_t *clen_d,
                            const uint64_t *ccode_d, uint8_t *out_d, uint64_t stride, const uint64_t *bit_base_h, uint64_t *totals_h, const uint64_t *bit_base_d = nullptr);
// the Huffman codes and stream headers of a batch's frames on the GPU (K <= 256; k_huff.hip k_frame_trees)
int frame_trees(Ctx *c, const uint32_t *cnt_d, const uint32_t *cent_d, uint32_t frames, uint32_t K, uint32_t w, uint32_t h, uint8_t *out_d, uint64_t stride,
                uint8_t *clen_d, uint64_t *ccode_d, uint64_t *bit_base_d, uint64_t *nbits_d, uint64_t *lens_d, uint32_t *err_d);

}  // namespace cniic
