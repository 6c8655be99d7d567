// k_mailbox.hip -- one-shot all-reduce over mailboxes (the exchange step of the shared-palette K-means: the
// K partial centroid sums, 5 K + 2 u64 words = 10 KiB at K = 256, once per iteration).
//
// RCCL moves such a buffer around a ring: N - 1 reduce-scatter steps and N - 1 all-gather steps, each a hop over one xGMI
// link with its own synchronisation.  The GPUs of an MI355X node are fully connected (seven direct links each), and 10 KiB
// is latency, not bandwidth: here every rank WRITES its buffer into a slot of every peer's mailbox (one hop, all links at
// once), raises a flag behind the data, waits for the N flags of its own mailbox and adds the N slots IN RANK ORDER.
// Unsigned integer sums: the result is bit-identical on every rank and to RCCL's, whatever the order of arrival.
//
//   mailbox of rank r (fine-grained HBM of r's GPU, mapped by every peer through HIP IPC):
//       header   u32 abort                                  (a failing rank sets it on every peer: their waits end)
//                u32 poisoned                               (set by the OWNER's kernel on its first timeout / abort: every later
//                                                            exchange of this rank returns at once, buffer untouched -- ADVICE r03)
//       flags    u32 [2 parities][N sources][slices]        (= the sequence number of the all-reduce the slice belongs to)
//       slots    u8  [2 parities][N sources][cap bytes]
//   one all-reduce = ONE kernel per rank, one block per 4 KiB slice: load the slice, store it to the N mailboxes, fence,
//   N flags; N lanes wait for the own mailbox's flags; add the slots.  Two parities make reuse safe: a rank can only start
//   all-reduce s + 2 after it finished s + 1, which needed every peer's flag of s + 1, which a peer raises in a kernel that
//   its stream runs after the one that read the slots of s.
//
// A wait is bounded inside the kernel (the communicator's timeout; the device's constant-rate clock), so a dead peer ends
// as an error word in mapped host memory that the K-means loop's polling sees (comm_async_error), never as a hung GPU.
// RCCL stays the default transport (cniic_comm_create); this one is asked for (cniic_comm_create_mailbox).
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>

#include "common.hpp"

namespace cniic {

namespace {
constexpr uint32_t kMbMaxRanks = 16;
constexpr uint32_t kMbSlice = 4096;      // bytes per block
constexpr uint32_t kMbThreads = 256;
constexpr uint32_t kMbHeader = 256;      // bytes in front of the flags
constexpr uint64_t kMbDefaultCap = 1ull << 20;
constexpr uint64_t kMbLongestWaitMs = 600000;  // "wait for ever" is not offered to a spinning kernel

struct MbArgs {
    uint8_t *peer[kMbMaxRanks];  // every rank's mailbox as mapped in this process (own: the allocation itself)
    uint32_t rank, nranks, seq, parity;
    uint32_t nslices_cap;
    uint64_t cap, flags_off, slots_off;
    uint64_t wait_ticks;
    uint32_t *status;            // mapped host word: 0 healthy, 1 a wait ran out, 2 a peer aborted
};

template <typename T> __device__ __forceinline__ T mb_load_sys(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// BYTES: the words are four u8 lanes each (occupancy nibbles summed over <= 15 ranks: a lane never carries)
template <bool BYTES> __device__ __forceinline__ uint32_t mb_add(uint32_t a, uint32_t b) {
    if (!BYTES) return a + b;
    const uint32_t lo = ((a & 0x00ff00ffu) + (b & 0x00ff00ffu)) & 0x00ff00ffu;
    const uint32_t hi = (((a >> 8) & 0x00ff00ffu) + ((b >> 8) & 0x00ff00ffu)) & 0x00ff00ffu;
    return lo | (hi << 8);
}
template <bool BYTES> __device__ __forceinline__ unsigned long long mb_add(unsigned long long a, unsigned long long b) { return a + b; }

// `count` elements of T; with BYTES `count` is in BYTES and T = u32 (a last partial word is read and written byte by byte)
template <typename T, bool BYTES>
__global__ __launch_bounds__(kMbThreads) void k_mb_all_reduce(MbArgs a, T *buf, uint64_t count) {
    constexpr uint32_t E = kMbSlice / sizeof(T), PER = E / kMbThreads;
    const uint32_t tid = threadIdx.x, sl = blockIdx.x;
    __shared__ uint32_t s_bad;
    // a mailbox that has seen a timeout or an abort is dead for good: the exchanges already enqueued behind the failed one (the loop
    // keeps two iterations in flight, several slices each) must not each spin a full timeout for the same dead peer, nor add stale
    // slots into the caller's sums
    if (tid == 0) s_bad = __hip_atomic_load(reinterpret_cast<const uint32_t *>(a.peer[a.rank]) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (s_bad) return;
    const uint64_t nwords = BYTES ? (count + 3) / 4 : count;
    const uint64_t base = (uint64_t)sl * E;
    T v[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint64_t i = base + k * kMbThreads + tid;
        T x = 0;
        if (BYTES && i < nwords && i * 4 + 4 > count) {
            const uint8_t *b = reinterpret_cast<const uint8_t *>(buf);
            for (uint64_t j = i * 4; j < count; j++) x |= (T)b[j] << (8 * (j - i * 4));
        } else if (i < nwords) {
            x = buf[i];
        }
        v[k] = x;
    }
    const uint64_t slot = a.slots_off + ((uint64_t)a.parity * a.nranks + a.rank) * a.cap + (uint64_t)sl * kMbSlice;
    for (uint32_t p = 0; p < a.nranks; p++) {
        T *dst = reinterpret_cast<T *>(a.peer[p] + slot);
#pragma unroll
        for (uint32_t k = 0; k < PER; k++)
            if (base + k * kMbThreads + tid < nwords) __hip_atomic_store(dst + k * kMbThreads + tid, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (tid < a.nranks) {  // the data is out: this rank's flag of the slice, on every peer
        uint32_t *f = reinterpret_cast<uint32_t *>(a.peer[tid] + a.flags_off) + ((uint64_t)a.parity * a.nranks + a.rank) * a.nslices_cap + sl;
        __hip_atomic_store(f, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid < a.nranks) {  // ... and the flags of the own mailbox, one lane per source
        const uint32_t *f = reinterpret_cast<const uint32_t *>(a.peer[a.rank] + a.flags_off) + ((uint64_t)a.parity * a.nranks + tid) * a.nslices_cap + sl;
        const uint32_t *ab = reinterpret_cast<const uint32_t *>(a.peer[a.rank]);
        const unsigned long long t0 = wall_clock64();
        uint32_t spins = 0, why = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
            if ((++spins & 31u) == 0) {
                if (__hip_atomic_load(ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { why = 2; break; }
                if (wall_clock64() - t0 > a.wait_ticks) { why = 1; break; }
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (why) {
            __hip_atomic_store(a.status, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<uint32_t *>(a.peer[a.rank]) + 1, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // poisoned, sticky
            s_bad = why;
        }
    }
    __syncthreads();
    if (s_bad) return;   // a slice some source never delivered: the sums stay what they were, the status word tells the host
    __threadfence_system();
    const uint8_t *src = a.peer[a.rank] + a.slots_off + (uint64_t)a.parity * a.nranks * a.cap + (uint64_t)sl * kMbSlice;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint64_t i = base + k * kMbThreads + tid;
        if (i >= nwords) continue;
        T acc = 0;
        for (uint32_t r = 0; r < a.nranks; r++)  // rank order: the same additions on every rank
            acc = mb_add<BYTES>(acc, mb_load_sys(reinterpret_cast<const T *>(src + (uint64_t)r * a.cap) + k * kMbThreads + tid));
        if (BYTES && i * 4 + 4 > count) {
            uint8_t *b = reinterpret_cast<uint8_t *>(buf);
            for (uint64_t j = i * 4; j < count; j++) b[j] = (uint8_t)(acc >> (8 * (j - i * 4)));
        } else {
            buf[i] = acc;
        }
    }
}

__global__ void k_mb_abort(MbArgs a) {
    if (threadIdx.x < a.nranks) __hip_atomic_store(reinterpret_cast<uint32_t *>(a.peer[threadIdx.x]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ranks that live in ONE process (tests; a host that drives several contexts from one process) cannot open each other's IPC
// handles -- the runtime refuses a handle in the process that made it -- so the allocations are also looked up here.
// Such ranks must sit on streams with hardware queues of their own (a process has GPU_MAX_HW_QUEUES = 4; streams beyond
// that share): a rank whose kernel is queued BEHIND a peer's waiting kernel never runs, and the peer's wait runs out.
std::mutex g_reg_mu;
std::map<std::string, uint8_t *> g_reg;
}  // namespace

struct Mailbox {
    Ctx *c = nullptr;
    uint32_t rank = 0, nranks = 1;
    uint64_t cap = 0, total = 0;
    uint32_t nslices_cap = 0;
    uint8_t *base = nullptr;
    uint8_t *peer[kMbMaxRanks] = {};
    bool opened[kMbMaxRanks] = {};
    bool connected = false;
    uint32_t seq = 0;
    uint32_t *status = nullptr;
    int wall_khz = 100000;
    hipStream_t side = nullptr;  // for the abort note (the context's stream may be sitting in a wait); made when first needed:
                                 // a process has few hardware queues, and streams that share one run their kernels in turn
    std::string key;
    MbArgs args(uint64_t timeout_ms) const {
        MbArgs a{};
        for (uint32_t p = 0; p < nranks; p++) a.peer[p] = peer[p];
        a.rank = rank; a.nranks = nranks; a.nslices_cap = nslices_cap; a.cap = cap;
        a.flags_off = kMbHeader;
        a.slots_off = kMbHeader + (((uint64_t)2 * nranks * nslices_cap * 4 + 255) & ~255ull);
        const uint64_t ms = timeout_ms == 0 || timeout_ms > kMbLongestWaitMs ? kMbLongestWaitMs : timeout_ms;
        a.wait_ticks = ms * (uint64_t)wall_khz;
        a.status = status;
        return a;
    }
};

int mailbox_create(Ctx *c, uint32_t rank, uint32_t nranks, uint64_t cap_bytes, uint8_t *handle64, Mailbox **out) {
    if (nranks == 0 || nranks > kMbMaxRanks || rank >= nranks)
        return c->fail(CNIIC_ERR_BAD_ARG, "comm_create_mailbox: rank %u of %u (at most %u ranks)", rank, nranks, kMbMaxRanks);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "an IPC handle is 64 bytes");
    std::unique_ptr<Mailbox> m(new Mailbox);
    m->c = c; m->rank = rank; m->nranks = nranks;
    m->cap = ((cap_bytes ? cap_bytes : kMbDefaultCap) + kMbSlice - 1) / kMbSlice * kMbSlice;
    m->nslices_cap = (uint32_t)(m->cap / kMbSlice);
    const MbArgs a = m->args(1);
    m->total = a.slots_off + (uint64_t)2 * nranks * m->cap;
    CNIIC_HIP_TRY(c, hipExtMallocWithFlags(reinterpret_cast<void **>(&m->base), m->total, hipDeviceMallocFinegrained));
    hipError_t e = hipMemset(m->base, 0, a.slots_off);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, m->base);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&m->status), 64, hipHostMallocMapped);
    if (e != hipSuccess) {
        (void)hipFree(m->base);
        if (m->status) (void)hipHostFree(m->status);
        return c->fail(CNIIC_ERR_HIP, "comm_create_mailbox: %s", hipGetErrorString(e));
    }
    *m->status = 0;
    (void)hipDeviceGetAttribute(&m->wall_khz, hipDeviceAttributeWallClockRate, c->device);
    if (m->wall_khz <= 0) m->wall_khz = 100000;
    memcpy(handle64, &h, 64);
    m->key.assign(reinterpret_cast<const char *>(handle64), 64);
    { std::lock_guard<std::mutex> g(g_reg_mu); g_reg[m->key] = m->base; }
    *out = m.release();
    return CNIIC_OK;
}

int mailbox_connect(Mailbox *m, const uint8_t *handles) {
    Ctx *c = m->c;
    if (m->connected) return c->fail(CNIIC_ERR_BAD_ARG, "comm_connect_mailbox: already connected");
    if (memcmp(handles + (size_t)m->rank * 64, m->key.data(), 64) != 0)
        return c->fail(CNIIC_ERR_BAD_ARG, "comm_connect_mailbox: entry %u of the handles is not this rank's own", m->rank);
    for (uint32_t p = 0; p < m->nranks; p++) {
        if (p == m->rank) { m->peer[p] = m->base; continue; }
        const std::string k(reinterpret_cast<const char *>(handles + (size_t)p * 64), 64);
        {
            std::lock_guard<std::mutex> g(g_reg_mu);
            auto it = g_reg.find(k);
            if (it != g_reg.end()) { m->peer[p] = it->second; continue; }
        }
        hipIpcMemHandle_t h;
        memcpy(&h, k.data(), 64);
        void *q = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return c->fail(CNIIC_ERR_HIP, "comm_connect_mailbox: the mailbox of rank %u cannot be mapped: %s", p, hipGetErrorString(e));
        m->peer[p] = static_cast<uint8_t *>(q);
        m->opened[p] = true;
    }
    m->connected = true;
    return CNIIC_OK;
}

// 0 while healthy; 1: a wait for a peer's slice ran out; 2: a peer aborted
int mailbox_status(const Mailbox *m) { return m && m->status ? (int)*reinterpret_cast<volatile uint32_t *>(m->status) : 0; }

int mailbox_all_reduce(Mailbox *m, void *buf_d, uint64_t count, int kind, uint64_t timeout_ms) {
    Ctx *c = m->c;
    if (!m->connected) return c->fail(CNIIC_ERR_BAD_ARG, "all_reduce: the mailboxes are not connected yet (cniic_comm_connect_mailbox)");
    if (const int st = mailbox_status(m))   // sticky: once a wait ran out or a peer aborted nothing more is enqueued
        return c->fail(CNIIC_ERR_RCCL, "all_reduce over mailboxes: the exchange is dead (%s)", st == 1 ? "a wait for a peer ran out" : "a peer aborted");
    const uint32_t eb = kind == 0 ? 1 : kind == 1 ? 4 : 8;
    if ((uintptr_t)buf_d % (kind == 2 ? 8 : 4)) return c->fail(CNIIC_ERR_BAD_ARG, "all_reduce over mailboxes: the buffer must be %d-byte aligned", kind == 2 ? 8 : 4);
    const uint64_t per = m->cap / eb;  // elements per launch
    for (uint64_t at = 0; at < count; at += per) {
        const uint64_t n = std::min(per, count - at);
        MbArgs a = m->args(timeout_ms);
        a.seq = ++m->seq;
        a.parity = a.seq & 1u;
        const uint32_t blocks = (uint32_t)((n * eb + kMbSlice - 1) / kMbSlice);
        uint8_t *p = static_cast<uint8_t *>(buf_d) + at * eb;
        if (kind == 2) hipLaunchKernelGGL((k_mb_all_reduce<unsigned long long, false>), dim3(blocks), dim3(kMbThreads), 0, c->stream, a, reinterpret_cast<unsigned long long *>(p), n);
        else if (kind == 1) hipLaunchKernelGGL((k_mb_all_reduce<uint32_t, false>), dim3(blocks), dim3(kMbThreads), 0, c->stream, a, reinterpret_cast<uint32_t *>(p), n);
        else hipLaunchKernelGGL((k_mb_all_reduce<uint32_t, true>), dim3(blocks), dim3(kMbThreads), 0, c->stream, a, reinterpret_cast<uint32_t *>(p), n);
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

void mailbox_abort(Mailbox *m) {
    if (!m || !m->connected) return;
    if (!m->side && hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess) { m->side = nullptr; return; }
    hipLaunchKernelGGL(k_mb_abort, dim3(1), dim3(64), 0, m->side, m->args(1));
    (void)hipStreamSynchronize(m->side);
}

void mailbox_destroy(Mailbox *m) {
    if (!m) return;
    // nothing of this rank may still be running on the mailbox when it is unmapped and freed: the exchanges on the context's stream
    // (a poisoned mailbox makes them return at once) and the abort note on the side stream.  Peers are not waited for -- a peer that
    // still stores into this mailbox holds its own mapping of the allocation, which outlives this free.
    if (m->c) (void)hipStreamSynchronize(m->c->stream);   // (a null stream is the default stream: waited for as well)
    if (m->side) (void)hipStreamSynchronize(m->side);
    { std::lock_guard<std::mutex> g(g_reg_mu); g_reg.erase(m->key); }
    for (uint32_t p = 0; p < m->nranks; p++)
        if (m->opened[p]) (void)hipIpcCloseMemHandle(m->peer[p]);
    if (m->side) (void)hipStreamDestroy(m->side);
    if (m->base) (void)hipFree(m->base);
    if (m->status) (void)hipHostFree(m->status);
    delete m;
}

}  // namespace cniic
