// k_kmeans_persist.hip -- kmeans::cluster::<ColorCount> as ONE launch (reference: src/kmeans.rs:21-39, the `while changed` loop,
// with assign_points kmeans.rs:330-416 and update_centroids kmeans.rs:110-143; Point = ColorCount, src/codec/clusterc.rs:68-114).
//
// Why (round 4's numbers, NOTES.md D): with one launch per iteration a full-schedule launch of the 4096^2 encode was 11.9 us of fixed
// cost + 6.8 us of candidate builds + 16.7 us of sweeps, of which 13.4 us were waits for points that never change; 61 dependent launches.
// Here the grid is one block of 16 waves per CU, every block owns a fixed range of the cell-major point list and keeps it IN LDS for the
// whole run, one packed word per colour (its 9-bit position inside its 8^3 cell, an 8-bit pixel count with an escape, its label: pk_make),
// together with its cells' descriptors and skip records.  An iteration is: sweep from LDS -> signed deltas of the movers into the
// block's LDS accumulators -> flush with agent-scope atomics -> grid barrier -> every block turns (its running sums + the iteration's
// deltas) into the K centroids it needs anyway.  Cells that do not fit a block's LDS (images of more than ~8 M colours) keep their
// packed words in memory (the XCD's L2 serves them) and are otherwise handled alike.
//
// What crosses the barrier is written with memory-side atomics (the deltas) or write-through stores (the buffer block 0 clears) and read
// with returning atomics: no cache holds a copy that could be stale, so the barrier itself needs no fence -- relaxed agent-scope
// atomics on counters that each sit on a line of their own, one counter per XCD and one on top (tools/persist_probe.hip checks every
// word of every round under uneven load: profiles/r05_persist_probe.txt; 4-byte sc1 loads of the same words DID read stale halves there).
// Every spin is bounded by the wall clock and watches an abort word: a grid that is not resident at once (somebody else's kernel on the
// CUs) ends with status `aborted`, the arrays the classic loop starts from are untouched, and km_rgbw_run falls back to it.
#include <atomic>

#include <hip/hip_ext.h>

#include "kmeans_rgbw.hpp"

namespace cniic {

constexpr int kPsThreads = 1024, kPsWaves = kPsThreads / 64;
constexpr uint32_t kPsScap = 128;                      // entries of a shared super-cell list (a longer list: the cell builds from the table)
constexpr uint32_t kPsAccWords = 5 * 256;              // u64 accumulators (K <= 256)
// dynamic LDS: [acc u64 5 x 256 | tab uint2 256 | S uint2 kPsSlots x 128 | wmask u64 16 x 4 | per block: cstart u32[C + 1], ccell u16[C], rec u32[C][10], points u32[...]]
constexpr uint32_t kPsOffTab = kPsAccWords * 8, kPsOffS = kPsOffTab + 256 * 8, kPsOffMask = kPsOffS + kPsSlots * kPsScap * 8;
static_assert(kPsOffCell == kPsOffMask + kPsWaves * 4 * 8, "ps_cell_bytes / kPsOffCell (kmeans_rgbw.hpp) describe this layout");

__device__ __forceinline__ uint32_t ps_xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ unsigned long long ps_aread(unsigned long long *p) {   // what memory holds, whatever any cache holds
    return __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ps_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ps_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one thread per block.  false: the wait ran out or somebody gave up (the abort word is set: every block leaves)
__device__ __forceinline__ bool ps_spin(uint32_t *word, uint32_t old, PsBar *b, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    uint32_t spins = 0;
    while (ps_ld(word) == old) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0) {
            if (ps_ld(&b->abort_.v)) return false;
            if (wall_clock64() - t0 > ticks) { ps_st(&b->abort_.v, 1u); return false; }
        }
    }
    return true;
}
__device__ __forceinline__ bool ps_barrier_flat(PsBar *b, uint32_t nblocks, unsigned long long ticks) {
    const uint32_t g = ps_ld(&b->gen.v);
    if (__hip_atomic_fetch_add(&b->count.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
        ps_st(&b->count.v, 0u);
        ps_st(&b->gen.v, g + 1);
        return true;
    }
    return ps_spin(&b->gen.v, g, b, ticks);
}
// two levels: the last block of an XCD to arrive reports to the top counter, the last XCD releases the others, each releases its own blocks
__device__ __forceinline__ bool ps_barrier_xcd(PsBar *b, uint32_t x, uint32_t nx_blocks, uint32_t nxcd, unsigned long long ticks) {
    bool ok = true;
    const uint32_t g = ps_ld(&b->xgen[x].v);
    if (__hip_atomic_fetch_add(&b->xcount[x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nx_blocks - 1) {
        ps_st(&b->xcount[x].v, 0u);
        const uint32_t tg = ps_ld(&b->topgen.v);
        if (__hip_atomic_fetch_add(&b->top.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nxcd - 1) {
            ps_st(&b->top.v, 0u);
            ps_st(&b->topgen.v, tg + 1);
        } else ok = ps_spin(&b->topgen.v, tg, b, ticks);
        ps_st(&b->xgen[x].v, g + 1);   // (also after a failed wait: the abort word is set and the block's own XCD must not wait out its clock)
    } else ok = ps_spin(&b->xgen[x].v, g, b, ticks);
    return ok;
}

// ---------------------------------------------------------------- the block ranges
// Block g owns the cells [mb0, mb1) of the compacted list (equal estimated cost, as k_wave_ranges deals them to waves), its points are
// consecutive in the cell-major arrays; the first [mb0, msplit) have their packed words in the block's LDS -- as many whole cells as
// fit behind the block's descriptors and records.  fail: a block would own more cells than its LDS can describe (then no block starts).
__global__ __launch_bounds__(1024) void k_ps_ranges(const uint32_t *__restrict__ ne_cost, const uint32_t *__restrict__ ne_start, const uint32_t *__restrict__ ne_count,
                                                    uint32_t G, uint32_t dyn_bytes, PsRange *__restrict__ ranges, uint32_t *__restrict__ fail) {
    __shared__ uint32_t s_b[1025];
    const uint32_t M = *ne_count;
    const uint64_t total = ne_cost[M];
    for (uint32_t g = threadIdx.x; g <= G; g += blockDim.x) {
        const uint64_t c_lo = total * g / G;
        uint32_t a = 0, b = M;   // first cell whose cost prefix is >= c_lo
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (ne_cost[mid] < c_lo) a = mid + 1; else b = mid; }
        s_b[g] = g == G ? M : a;
    }
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) {
        const uint32_t mb0 = s_b[g], mb1 = s_b[g + 1], C = mb1 - mb0;
        uint32_t msplit = mb0;
        if (C > kPsMaxCells || kPsOffCell + ps_cell_bytes(C) > dyn_bytes) atomicAdd(fail, 1u);
        else {
            const uint32_t cap = (dyn_bytes - kPsOffCell - ps_cell_bytes(C)) / 4u, q0 = ne_start[mb0];
            uint32_t a = mb0, b = mb1;   // the last m in [mb0, mb1] whose points before it fit
            while (a < b) { const uint32_t mid = (a + b + 1) >> 1; if (ne_start[mid] - q0 <= cap) a = mid; else b = mid - 1; }
            msplit = a;
        }
        ranges[g] = PsRange{mb0, mb1, msplit, 0u};
    }
}

// ---------------------------------------------------------------- a cell's candidates as a bitmask of cluster ids
// The members of `list` (ascending id; n entries) that can be nearest somewhere in cell c's cube: pivot = the member nearest the cube's
// centre, kept = whoever the pivot does not dominate over the whole cube (Dominance, kmeans_rgbw.hpp).  The mask, the pivot's colour and
// id go into the cell's record (skip schedule).  TABLE: `list` is the block's centroid table itself (position = id: the ballots ARE the
// mask words, and the mask is complete -- it holds every centroid of the table the pivot does not dominate).
template <bool TABLE>
__device__ __forceinline__ uint32_t ps_build(const uint2 *list, uint32_t n, uint32_t c, int lane, unsigned long long *wm, uint32_t *rec, unsigned long long (&nm)[4]) {
    constexpr int32_t ext = (1 << kCellShift) - 1;
    const CellBox bx = cell_box(c);
    const uint32_t pe = nearest_to_centre(list, n, bx, ext, lane);
    const uint2 pvc = list[pe];
    const uint32_t pv = (uint32_t)__builtin_amdgcn_readfirstlane((int)pvc.x), pid = 255u - ((uint32_t)__builtin_amdgcn_readfirstlane((int)pvc.y) & 255u);
    Dominance dm;
    dm.set(bx, ext, pv);
    if constexpr (TABLE) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t k = 64 * w + lane;
            nm[w] = __ballot(k < n && dm.worst(list[k < n ? k : 0].x) >= 0);
        }
    } else {
        if (lane < 4) wm[lane] = 0ull;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t e0 = 0; e0 < n; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < n) {
                const uint2 cc = list[e];
                if (dm.worst(cc.x) >= 0) { const uint32_t k = 255u - (cc.y & 255u); atomicOr(&wm[k >> 6], 1ull << (k & 63)); }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t *w32 = reinterpret_cast<const uint32_t *>(wm);
#pragma unroll
        for (int w = 0; w < 4; w++)
            nm[w] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)w32[2 * w + 1]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)w32[2 * w]);
    }
    uint32_t wv = 0;
#pragma unroll
    for (int t = 0; t < 8; t++)
        if (lane == t) wv = (uint32_t)(nm[t >> 1] >> (32 * (t & 1)));
    if (lane < 8) rec[2 + lane] = wv;
    if (lane == 8) rec[0] = pv;
    if (lane == 9) rec[1] = c | (pid << 16) | (TABLE ? kRecComplete : 0u);
    return (uint32_t)(__popcll(nm[0]) + __popcll(nm[1]) + __popcll(nm[2]) + __popcll(nm[3]));
}

// ---------------------------------------------------------------- sweeps
// best packed key (distance | 255 - id) of every slot's colour over the candidates of the mask: the set bits are walked on the scalar
// unit, each candidate one broadcast read of the block's table + 3 vector instructions per slot
__device__ __forceinline__ void ps_best(const uint32_t (&key)[kSweep], const unsigned long long (&nm)[4], const uint2 *tab, uint32_t (&best)[kSweep]) {
#pragma unroll
    for (int u = 0; u < kSweep; u++) best[u] = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned long long mm = nm[w];
        while (mm) {
            const uint32_t k = 64 * w + (uint32_t)__builtin_ctzll(mm);
            mm &= mm - 1;
            const uint2 cc = tab[k];
#pragma unroll
            for (int u = 0; u < kSweep; u++) best[u] = max(best[u], (dot4u8(key[u], cc.x, 0) << 9) + cc.y);
        }
    }
}

// the signed deltas of one mover into the block's accumulators (clusterc.rs:92-98: sums of channel x count, of counts, of members)
__device__ __forceinline__ void ps_book_move(unsigned long long *acc, uint32_t K, uint32_t pp, uint64_t w, uint32_t ol, uint32_t nl) {
    const unsigned long long rw = ((pp >> 16) & 255) * w, gw = ((pp >> 8) & 255) * w, bw = (pp & 255) * w;
    atomicAdd(&acc[3 * nl + 0], rw); atomicAdd(&acc[3 * nl + 1], gw); atomicAdd(&acc[3 * nl + 2], bw);
    atomicAdd(&acc[3 * K + nl], (unsigned long long)w); atomicAdd(&acc[4 * K + nl], 1ull);
    atomicAdd(&acc[3 * ol + 0], 0ull - rw); atomicAdd(&acc[3 * ol + 1], 0ull - gw); atomicAdd(&acc[3 * ol + 2], 0ull - bw);
    atomicAdd(&acc[3 * K + ol], 0ull - (unsigned long long)w); atomicAdd(&acc[4 * K + ol], 0ull - 1ull);
}

// One sweep of an iteration after the first: the 64 x kSweep packed words w (positions base + 64 u + lane of the block's point range,
// those below e) against the candidates of the mask.  Stay unless another centroid is STRICTLY closer (kmeans.rs:375), lowest id among
// equals (the key's low byte).  A mover's label byte is rewritten in place (st: the words' home, LDS or memory), its deltas booked.
// agg (the first iterations after iteration 0, where centroids still travel and whole cells change hands): the movers that share the
// first mover's (old, new) pair are summed in the wave and booked by one lane, round by round (ten LDS atomics per point on the same
// ten words run one lane at a time).
template <typename StoreLabel>
__device__ __forceinline__ void ps_sweep(const uint32_t (&wd)[kSweep], uint32_t base, uint32_t e, int lane, const unsigned long long (&nm)[4], uint32_t ncand,
                                         const uint2 *tab, uint32_t K, uint32_t cbk, const uint32_t *__restrict__ cwq, unsigned long long *acc, uint32_t &moved,
                                         bool agg, StoreLabel store_label) {
    uint32_t key[kSweep], cur[kSweep], wt[kSweep];
    bool heavy = false;
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        key[u] = pk_key(wd[u], cbk);
        cur[u] = wd[u] >> 24;
        wt[u] = (wd[u] >> 16) & 255u;
        heavy = heavy | (wt[u] == 255u);
    }
    if (ncand == 1) {
        // More than half of the cells lie inside one cluster's region: ONE candidate, and as a rule every point already carries its
        // label; the lone candidate beats every other centroid for every colour of the cube, so nothing can move.
        const uint32_t only = nm[0] ? (uint32_t)__builtin_ctzll(nm[0]) : nm[1] ? 64u + (uint32_t)__builtin_ctzll(nm[1]) : nm[2] ? 128u + (uint32_t)__builtin_ctzll(nm[2]) : 192u + (uint32_t)__builtin_ctzll(nm[3]);
        bool same = true;
#pragma unroll
        for (int u = 0; u < kSweep; u++) same = same & ((base + u * 64 + lane >= e) | (cur[u] == only));
        if (__ballot(!same) == 0ull) return;
    }
    uint32_t best[kSweep];
    ps_best(key, nm, tab, best);
    bool mv[kSweep], any = false;
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        const uint2 cc = tab[cur[u]];
        const uint32_t kc = (dot4u8(key[u], cc.x, 0) << 9) + cc.y;
        mv[u] = (base + u * 64 + lane < e) & ((best[u] >> 8) > (kc >> 8));  // strictly closer (kmeans.rs:375)
        any = any | mv[u];
    }
    if (!__ballot(any)) return;
    if (__ballot(heavy & any)) {   // a pixel count of 255 and more is looked up (rare in a photograph)
#pragma unroll
        for (int u = 0; u < kSweep; u++)
            if (mv[u] && wt[u] == 255u) wt[u] = cwq[base + u * 64 + lane];
    }
    uint32_t nl[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        nl[u] = mv[u] ? 255u - (best[u] & 255u) : cur[u];
        if (mv[u]) { store_label(base + u * 64 + lane, nl[u]); moved++; }
    }
    if (agg) {
        static_assert(kSweep == 4, "four slots per lane");
        uint32_t nmv = 0;
#pragma unroll
        for (int u = 0; u < kSweep; u++) nmv += (uint32_t)__popcll(__ballot(mv[u]));
        if (nmv >= kAggMin) {
#pragma unroll 1
            for (int round = 0; round < 6; round++) {
                const unsigned long long b0 = __ballot(mv[0]), b1 = __ballot(mv[1]), b2 = __ballot(mv[2]), b3 = __ballot(mv[3]);
                if (!(b0 | b1 | b2 | b3)) return;
                uint32_t pn, po;
                if (b0) { const int l = __builtin_ctzll(b0); pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[0], l); po = (uint32_t)__builtin_amdgcn_readlane((int)cur[0], l); }
                else if (b1) { const int l = __builtin_ctzll(b1); pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[1], l); po = (uint32_t)__builtin_amdgcn_readlane((int)cur[1], l); }
                else if (b2) { const int l = __builtin_ctzll(b2); pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[2], l); po = (uint32_t)__builtin_amdgcn_readlane((int)cur[2], l); }
                else { const int l = __builtin_ctzll(b3); pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[3], l); po = (uint32_t)__builtin_amdgcn_readlane((int)cur[3], l); }
                uint32_t cnt = 0;
                bool mt[kSweep];
#pragma unroll
                for (int u = 0; u < kSweep; u++) {
                    mt[u] = mv[u] && nl[u] == pn && cur[u] == po;
                    cnt += (uint32_t)__popcll(__ballot(mt[u]));
                    mv[u] = mv[u] && !mt[u];
                }
                if (cnt < kAggMin) {   // a handful books itself, and so does everybody who is left
#pragma unroll
                    for (int u = 0; u < kSweep; u++) mv[u] = mv[u] || mt[u];
                    break;
                }
                auto book = [&](int shift, uint32_t mask, size_t at_new, size_t at_old) {
                    unsigned long long v = 0;
#pragma unroll
                    for (int u = 0; u < kSweep; u++)
                        if (mt[u]) v += (unsigned long long)(mask ? (key[u] >> shift) & mask : 1u) * wt[u];
                    v = wave_reduce_sum64(v);
                    if (lane == 0) { atomicAdd(&acc[at_new], v); atomicAdd(&acc[at_old], 0ull - v); }
                };
                book(16, 255u, 3 * (size_t)pn + 0, 3 * (size_t)po + 0);
                book(8, 255u, 3 * (size_t)pn + 1, 3 * (size_t)po + 1);
                book(0, 255u, 3 * (size_t)pn + 2, 3 * (size_t)po + 2);
                book(0, 0u, 3 * (size_t)K + pn, 3 * (size_t)K + po);
                if (lane == 0) { atomicAdd(&acc[4 * K + pn], (unsigned long long)cnt); atomicAdd(&acc[4 * K + po], 0ull - (unsigned long long)cnt); }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kSweep; u++)
        if (mv[u]) ps_book_move(acc, K, key[u], wt[u], cur[u], nl[u]);
}

// The sweep of iteration 0: colours, pixel counts and the initial labels (init_assignment, kmeans.rs:61-78) come from the cell-major
// arrays, EVERY point adds to the sums of the cluster it ends in (the running sums start at zero), and the packed word of every point
// is written to its home.  A sweep lies inside one 8^3 cell and its points join one, two, three clusters: round by round, the cluster of
// the first point still to be booked, every point that joins it summed in the wave, one lane adds the totals.
template <typename StoreWord>
__device__ __forceinline__ void ps_sweep_first(const uint32_t (&p)[kSweep], const uint32_t (&cur)[kSweep], const uint32_t (&wt)[kSweep], uint32_t base, uint32_t e, int lane,
                                               const unsigned long long (&nm)[4], const uint2 *tab, uint32_t K, unsigned long long *acc, uint32_t &moved, StoreWord store_word) {
    uint32_t best[kSweep];
    ps_best(p, nm, tab, best);
    uint32_t nl[kSweep];
    bool rem[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        const uint32_t idx = base + u * 64 + lane;
        nl[u] = cur[u];
        rem[u] = idx < e;
        if (rem[u]) {
            const uint2 cc = tab[cur[u]];
            const uint32_t kc = (dot4u8(p[u], cc.x, 0) << 9) + cc.y;
            if ((best[u] >> 8) > (kc >> 8)) { nl[u] = 255u - (best[u] & 255u); moved++; }  // strictly closer (kmeans.rs:375)
            store_word(idx, pk_make(p[u], wt[u], nl[u]));
        }
    }
    static_assert(kSweep == 4, "four slots per lane");
#pragma unroll 1
    for (int round = 0; round < 6; round++) {
        const unsigned long long b0 = __ballot(rem[0]), b1 = __ballot(rem[1]), b2 = __ballot(rem[2]), b3 = __ballot(rem[3]);
        if (!(b0 | b1 | b2 | b3)) return;
        uint32_t pn;
        if (b0) pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[0], __builtin_ctzll(b0));
        else if (b1) pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[1], __builtin_ctzll(b1));
        else if (b2) pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[2], __builtin_ctzll(b2));
        else pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[3], __builtin_ctzll(b3));
        uint32_t cnt = 0, mbits = 0;
#pragma unroll
        for (int u = 0; u < kSweep; u++) {
            const bool match = rem[u] && nl[u] == pn;
            mbits |= match ? 1u << u : 0u;
            cnt += (uint32_t)__popcll(__ballot(match));
            rem[u] = rem[u] && !match;
        }
        auto book = [&](int shift, uint32_t mask, size_t at) {
            unsigned long long v = 0;
#pragma unroll
            for (int u = 0; u < kSweep; u++)
                if ((mbits >> u) & 1u) v += (unsigned long long)(mask ? (p[u] >> shift) & mask : 1u) * wt[u];
            v = wave_reduce_sum64(v);
            if (lane == 0) atomicAdd(&acc[at], v);
        };
        book(16, 255u, 3 * (size_t)pn + 0);
        book(8, 255u, 3 * (size_t)pn + 1);
        book(0, 255u, 3 * (size_t)pn + 2);
        book(0, 0u, 3 * (size_t)K + pn);
        if (lane == 0) atomicAdd(&acc[4 * K + pn], (unsigned long long)cnt);
    }
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        if (rem[u]) {
            const uint32_t pp = p[u], n_ = nl[u];
            const uint64_t w = wt[u];
            atomicAdd(&acc[3 * n_ + 0], ((pp >> 16) & 255) * w); atomicAdd(&acc[3 * n_ + 1], ((pp >> 8) & 255) * w); atomicAdd(&acc[3 * n_ + 2], (pp & 255) * w);
            atomicAdd(&acc[3 * K + n_], (unsigned long long)w); atomicAdd(&acc[4 * K + n_], 1ull);
        }
    }
}

// the 64 x kSweep entries from position `from` (global, cell-major) of the classic arrays as buffer loads; entries at `end` and beyond read 0
__device__ __forceinline__ void ps_load_first(const uint32_t *__restrict__ ckeys, const uint8_t *__restrict__ labels, const uint32_t *__restrict__ cweight,
                                              uint32_t from, uint32_t end, int lane, uint32_t (&p)[kSweep], uint32_t (&cur)[kSweep], uint32_t (&wt)[kSweep]) {
    const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)from), e = (uint32_t)__builtin_amdgcn_readfirstlane((int)end);
    const uint32_t lim = e > f ? e : 0u;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(ckeys), 0, (int)(lim * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(labels), 0, (int)lim, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(cweight), 0, (int)(lim * 4u), 0x00020000);
    const uint32_t q = f + (uint32_t)lane;
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        p[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rk, (int)(q * 4u + u * 256u), 0, 0);
        cur[u] = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rl, (int)(q + u * 64u), 0, 0);
        wt[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rw, (int)(q * 4u + u * 256u), 0, 0);
    }
}

// ---------------------------------------------------------------- the kernel
struct PsArgs {
    const uint32_t *ckeys, *cweight;      // cell-major colours and pixel counts
    uint8_t *labels;                      // cell-major labels: the initial assignment on entry, the result on a regular exit
    const uint32_t *ne_cell, *ne_start;   // compacted non-empty cells: id, first position
    const PsRange *ranges;
    const uint32_t *ranges_fail;
    uint32_t *pk;                         // packed words of the points that do not fit their block's LDS, by cell-major position
    const uint2 *cconst0;                 // the initial centroids (k_rgbw_init_cent)
    unsigned long long *partials;         // 3 x kPsPartWords, zero on entry
    PsBar *bar;                           // zero on entry
    uint2 *cconst_g;                      // results, written by block 0 on a regular exit
    uint32_t *cent_g;
    uint64_t *members_out, *wsum_out;
    KmDevState *st_rw;
    PsExit *exit_host;                    // pinned: how the launch ended
    const uint32_t *keys;                 // canonical point list (empty-cluster reseed) ...
    GIdx gx;                              // ... or the index of all occupied colours
    uint64_t seed, max_iters, U;
    uint32_t K, max_skip, agg_iters, test_abort_at;
    unsigned long long timeout_ticks;
    unsigned long long *iter_ts;          // block 0's clock (100 MHz) when iteration j's centroids stood, [0] at entry; kPsTsCap entries, or null
    unsigned long long *blk_ts;           // measuring runs (CNIIC_KM_PS_BLOCK_TRACE): [block][iteration < 128][4] clock at: assign done, flushed, through the barrier, centroids stand
};

__global__ __launch_bounds__(kPsThreads) void k_rgbw_persist(PsArgs a) {
    extern __shared__ __align__(16) unsigned long long lds[];
    __shared__ uint32_t s_moved, s_cell, s_nmoved, s_reseed, s_active, s_ok, s_nx, s_nxcd;
    __shared__ uint32_t s_mlist[kMaxMovedSkip];
    __shared__ uint32_t s_nS[kPsSlots];
    __shared__ unsigned long long s_mm[4], s_evals, s_changed, s_pev;
    const uint32_t tid = threadIdx.x, K = a.K;
    const int lane = tid & 63, wid = tid >> 6;
    unsigned long long *acc = lds;
    uint2 *tab = reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(lds) + kPsOffTab);
    uint2 *Sbase = reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(lds) + kPsOffS);
    unsigned long long *wmask = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(lds) + kPsOffMask) + wid * 4;
    if (*a.ranges_fail) {   // (the same word for every block: nobody starts, nobody waits)
        if (blockIdx.x == 0 && tid == 0) { a.exit_host->status = kPsStatusRanges; __threadfence_system(); }
        return;
    }
    const PsRange rg = a.ranges[blockIdx.x];
    const uint32_t C = rg.mb1 - rg.mb0, Cres = rg.msplit - rg.mb0;
    uint32_t *cstart = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(lds) + kPsOffCell);   // [C + 1] first point of a cell, relative to the block's
    uint16_t *ccell = reinterpret_cast<uint16_t *>(cstart + C + 1);                                   // [C] cell ids
    uint32_t *recs = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(cstart) + ps_desc_bytes(C));   // [C][kPsRecWords]
    uint32_t *pts = recs + (size_t)kPsRecWords * C;                                                   // the resident points' words
    const uint32_t q0 = a.ne_start[rg.mb0], nq = a.ne_start[rg.mb1] - q0, nres = a.ne_start[rg.msplit] - q0;
    uint32_t *pkq = a.pk + q0;                   // the same positions in memory (used from nres on)
    const uint32_t *cwq = a.cweight + q0;
    for (uint32_t i = tid; i <= C; i += kPsThreads) cstart[i] = a.ne_start[rg.mb0 + i] - q0;
    for (uint32_t i = tid; i < C; i += kPsThreads) ccell[i] = (uint16_t)a.ne_cell[rg.mb0 + i];
    for (uint32_t i = tid; i < 5 * K; i += kPsThreads) acc[i] = 0ull;
    for (uint32_t i = tid; i < K; i += kPsThreads) tab[i] = a.cconst0[i];
    if (tid == 0) {
        s_moved = 0; s_evals = 0; s_nmoved = 0; s_reseed = 0; s_active = 0; s_mm[0] = s_mm[1] = s_mm[2] = s_mm[3] = 0ull; s_cell = 0;
        // census: how many blocks does my XCD hold, how many XCDs are in use?  (nothing about placement is assumed)
        const uint32_t x = ps_xcc_id();
        __hip_atomic_fetch_add(&a.bar->xblocks[x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = ps_barrier_flat(a.bar, gridDim.x, a.timeout_ticks);
        uint32_t n = 0;
        for (int i = 0; i < 8; i++) n += ps_ld(&a.bar->xblocks[i].v) != 0;
        s_nx = ps_ld(&a.bar->xblocks[x].v);
        s_nxcd = n;
        s_ok = ok;
        if (blockIdx.x == 0 && a.iter_ts) a.iter_ts[0] = wall_clock64();
    }
    __syncthreads();
    if (!s_ok) {
        if (tid == 0) { a.exit_host->status = kPsStatusAborted; __threadfence_system(); }
        return;
    }
    // the super-cells of the block's range: its cells are consecutive in super-cell-major order; the first kPsSlots get a shared list
    const uint32_t sup_first = C ? (uint32_t)ccell[0] >> kSuperShift : 0u;
    const uint32_t nsl = C ? min(((uint32_t)ccell[C - 1] >> kSuperShift) - sup_first + 1u, kPsSlots) : 0u;
    const unsigned long long lt_mask = (1ull << lane) - 1;
    // running sums of cluster k = tid (kmeans.rs: the members of every cluster, as sums): registers, the same in every block
    unsigned long long run[5] = {0, 0, 0, 0, 0};
    uint32_t moved = 0;
    unsigned long long evals = 0;
    uint32_t j = 0;              // the iteration whose assign step runs
    uint32_t nS = K;             // centroids the last update moved
    unsigned long long reseeds_total = 0, evals_total = 0;
    for (;;) {
        const bool first = j == 0;
        const bool skip_mode = !first && a.max_skip && nS <= a.max_skip;
        if (!skip_mode) {
            // ============================================================= FULL schedule: every cell's candidates anew
            if ((uint32_t)wid < nsl) {
                const uint32_t n = build_super(tab, K, sup_first + wid, lane, lt_mask, Sbase + (size_t)wid * kPsScap, kPsScap);
                if (lane == 0) s_nS[wid] = n;
            }
            __syncthreads();
            for (;;) {
                uint32_t i = 0;
                if (lane == 0) i = atomicAdd(&s_cell, 1u);
                i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
                if (i >= C) break;
                const uint32_t c = ccell[i], s = cstart[i], e = cstart[i + 1];
                uint32_t *rec = recs + (size_t)kPsRecWords * i;
                const uint32_t slot = (c >> kSuperShift) - sup_first;
                unsigned long long nm[4];
                uint32_t ncand;
                const uint32_t nSl = slot < kPsSlots ? s_nS[slot] : 0xffffffffu;
                if (nSl <= kPsScap) ncand = ps_build<false>(Sbase + (size_t)slot * kPsScap, nSl, c, lane, wmask, rec, nm);
                else ncand = ps_build<true>(tab, K, c, lane, wmask, rec, nm);
                const uint32_t cbk = cell_base_key(c);
                const bool res = i < Cres;
                if (first) {
                    uint32_t p[kSweep], cur[kSweep], wt[kSweep];
                    ps_load_first(a.ckeys, a.labels, a.cweight, q0 + s, q0 + e, lane, p, cur, wt);
                    for (uint32_t base = s; base < e; base += 64 * kSweep) {
                        uint32_t pn[kSweep], curn[kSweep], wtn[kSweep];
                        ps_load_first(a.ckeys, a.labels, a.cweight, q0 + base + 64 * kSweep, q0 + e, lane, pn, curn, wtn);
                        if (res) ps_sweep_first(p, cur, wt, base, e, lane, nm, tab, K, acc, moved, [&](uint32_t idx, uint32_t w) { pts[idx] = w; });
                        else ps_sweep_first(p, cur, wt, base, e, lane, nm, tab, K, acc, moved, [&](uint32_t idx, uint32_t w) { pkq[idx] = w; });
#pragma unroll
                        for (int u = 0; u < kSweep; u++) { p[u] = pn[u]; cur[u] = curn[u]; wt[u] = wtn[u]; }
                    }
                } else if (res) {
                    const bool agg = j <= a.agg_iters;
                    for (uint32_t base = s; base < e; base += 64 * kSweep) {
                        uint32_t wd[kSweep];
#pragma unroll
                        for (int u = 0; u < kSweep; u++) { const uint32_t idx = base + u * 64 + lane; wd[u] = idx < e ? pts[idx] : 0u; }
                        ps_sweep(wd, base, e, lane, nm, ncand, tab, K, cbk, cwq, acc, moved, agg, [&](uint32_t idx, uint32_t l) { reinterpret_cast<uint8_t *>(pts)[4 * idx + 3] = (uint8_t)l; });
                    }
                } else {
                    const bool agg = j <= a.agg_iters;
                    uint32_t wd[kSweep];
#pragma unroll
                    for (int u = 0; u < kSweep; u++) { const uint32_t idx = s + u * 64 + lane; wd[u] = idx < e ? pkq[idx] : 0u; }
                    for (uint32_t base = s; base < e; base += 64 * kSweep) {
                        uint32_t wn[kSweep];
#pragma unroll
                        for (int u = 0; u < kSweep; u++) { const uint32_t idx = base + 64 * kSweep + u * 64 + lane; wn[u] = idx < e ? pkq[idx] : 0u; }
                        ps_sweep(wd, base, e, lane, nm, ncand, tab, K, cbk, cwq, acc, moved, agg, [&](uint32_t idx, uint32_t l) { reinterpret_cast<uint8_t *>(pkq)[4 * (size_t)idx + 3] = (uint8_t)l; });
#pragma unroll
                        for (int u = 0; u < kSweep; u++) wd[u] = wn[u];
                    }
                }
                evals += (unsigned long long)(e - s) * (ncand + 1);
            }
        } else {
            // ============================================================= SKIP schedule (at most max_skip centroids moved)
            // A cell none of whose candidates moved and whose pivot still dominates every moved centroid repeats all its decisions.
            // Cells are dealt to the waves with a stride (what survives the test is clustered around the centroids that moved); eight
            // cells are tested together, lane = (cell, one of eight moved centroids).
            const uint32_t k1 = (uint32_t)lane < nS ? s_mlist[lane] : 0xffffffffu;
            const uint32_t ck1 = k1 != 0xffffffffu ? tab[k1].x : 0u;
            for (uint32_t t0 = 0; (uint32_t)wid + kPsWaves * t0 < C; t0 += 8) {
                const uint32_t ci = (uint32_t)lane >> 3, ic = (uint32_t)wid + kPsWaves * (t0 + ci);
                const bool cell_ok = ic < C;
                const uint32_t *rc = recs + (size_t)kPsRecWords * (cell_ok ? ic : 0u);
                Dominance dmv;
                dmv.set(cell_box(rc[1] & 0x7fffu), (1 << kCellShift) - 1, rc[0]);
                bool dv = false;
                for (uint32_t j0 = 0; j0 < nS; j0 += 8) {
                    const uint32_t jj = j0 + ((uint32_t)lane & 7u);
                    const bool has = jj < nS;
                    const uint32_t k = has ? s_mlist[jj] : 0u;
                    const bool in = ((rc[2 + (k >> 5)] >> (k & 31)) & 1u) != 0;
                    dv = dv | (has & (in | (dmv.worst(tab[k].x) >= 0)));   // a moved centroid matters if it was a candidate or is no longer dominated by the pivot
                }
                const unsigned long long dirty8 = __ballot(dv && cell_ok);
                if (!dirty8) continue;
#pragma unroll 1
                for (uint32_t bi = 0; bi < 8; bi++) {
                    if (!((dirty8 >> (8 * bi)) & 0xffull)) continue;
                    const uint32_t i = (uint32_t)wid + kPsWaves * (t0 + bi);
                    uint32_t *rec = recs + (size_t)kPsRecWords * i;
                    const uint32_t pvt = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[0]), cw = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[1]);
                    const uint32_t c = cw & 0x7fffu, pid = (cw >> 16) & 255u;
                    const uint32_t s = cstart[i], e = cstart[i + 1];
                    constexpr int32_t ext = (1 << kCellShift) - 1;
                    const CellBox bx = cell_box(c);
                    // A COMPLETE mask whose pivot has not moved: every centroid that has not moved keeps its verdict against it; the moved
                    // ones are tested here.  Otherwise the mask is rebuilt from the whole table with a fresh pivot (and is complete then).
                    const bool keep_pivot = (cw & kRecComplete) && ((s_mm[pid >> 6] >> (pid & 63)) & 1ull) == 0ull;
                    unsigned long long nm[4];
                    uint32_t ncand;
                    if (keep_pivot) {
                        Dominance dm;
                        dm.set(bx, ext, pvt);
                        unsigned long long f1 = __ballot(k1 != 0xffffffffu && dm.worst(ck1) >= 0);
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            const unsigned long long om = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)rec[3 + 2 * w]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[2 + 2 * w]);
                            nm[w] = om & ~s_mm[w];
                        }
                        while (f1) {
                            const int l = __builtin_ctzll(f1);
                            f1 &= f1 - 1;
                            const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)k1, l);
                            const unsigned long long b = 1ull << (k & 63);
                            const uint32_t w = (k >> 6) & 3;
                            nm[0] |= w == 0 ? b : 0ull; nm[1] |= w == 1 ? b : 0ull; nm[2] |= w == 2 ? b : 0ull; nm[3] |= w == 3 ? b : 0ull;
                        }
#pragma unroll
                        for (int w = 0; w < 4; w++) nm[w] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(nm[w] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nm[w]);
                        uint32_t wv = 0;
#pragma unroll
                        for (int t = 0; t < 8; t++)
                            if (lane == t) wv = (uint32_t)(nm[t >> 1] >> (32 * (t & 1)));
                        if (lane < 8) rec[2 + lane] = wv;
                        ncand = (uint32_t)(__popcll(nm[0]) + __popcll(nm[1]) + __popcll(nm[2]) + __popcll(nm[3]));
                    } else ncand = ps_build<true>(tab, K, c, lane, wmask, rec, nm);
                    const uint32_t cbk = cell_base_key(c);
                    if (i < Cres) {
                        for (uint32_t base = s; base < e; base += 64 * kSweep) {
                            uint32_t wd[kSweep];
#pragma unroll
                            for (int u = 0; u < kSweep; u++) { const uint32_t idx = base + u * 64 + lane; wd[u] = idx < e ? pts[idx] : 0u; }
                            ps_sweep(wd, base, e, lane, nm, ncand, tab, K, cbk, cwq, acc, moved, false, [&](uint32_t idx, uint32_t l) { reinterpret_cast<uint8_t *>(pts)[4 * idx + 3] = (uint8_t)l; });
                        }
                    } else {
                        for (uint32_t base = s; base < e; base += 64 * kSweep) {
                            uint32_t wd[kSweep];
#pragma unroll
                            for (int u = 0; u < kSweep; u++) { const uint32_t idx = base + u * 64 + lane; wd[u] = idx < e ? pkq[idx] : 0u; }
                            ps_sweep(wd, base, e, lane, nm, ncand, tab, K, cbk, cwq, acc, moved, false, [&](uint32_t idx, uint32_t l) { reinterpret_cast<uint8_t *>(pkq)[4 * (size_t)idx + 3] = (uint8_t)l; });
                        }
                    }
                    evals += (unsigned long long)(e - s) * (ncand + 1);
                }
            }
        }
        // ------------------------------------------------------------- this iteration's deltas leave the block
        moved = wave_reduce_sum(moved);
        if (lane == 0) {
            if (moved) atomicAdd(&s_moved, moved);
            if (evals) atomicAdd(&s_evals, evals);
        }
        moved = 0; evals = 0;
        __syncthreads();
        if (a.blk_ts && tid == 0 && j < 128) a.blk_ts[((size_t)blockIdx.x * 128 + j) * 4 + 0] = wall_clock64();
        unsigned long long *Pcur = a.partials + (size_t)(j % 3) * kPsPartWords;
        for (uint32_t i = tid; i < 5 * K; i += kPsThreads) {
            const unsigned long long v = acc[i];
            if (v) { __hip_atomic_fetch_add(&Pcur[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc[i] = 0ull; }
        }
        if (tid == 0) {
            if (s_moved) __hip_atomic_fetch_add(&Pcur[5 * (size_t)K], (unsigned long long)s_moved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s_evals) __hip_atomic_fetch_add(&Pcur[5 * (size_t)K + 1], s_evals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_moved = 0; s_evals = 0; s_cell = 0; s_nmoved = 0; s_reseed = 0; s_active = 0; s_mm[0] = s_mm[1] = s_mm[2] = s_mm[3] = 0ull;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's atomics have been performed before its block arrives
        __syncthreads();
        if (a.blk_ts && tid == 0 && j < 128) a.blk_ts[((size_t)blockIdx.x * 128 + j) * 4 + 1] = wall_clock64();
        if (tid == 0) {
            bool ok = ps_barrier_xcd(a.bar, ps_xcc_id(), s_nx, s_nxcd, a.timeout_ticks);
#ifdef CNIIC_TESTING
            if (a.test_abort_at && j + 1 == a.test_abort_at) { ps_st(&a.bar->abort_.v, 1u); ok = false; }   // (fault injection: CNIIC_TEST_PS_ABORT_AT)
#endif
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) {
            if (tid == 0) { a.exit_host->status = kPsStatusAborted; __threadfence_system(); }
            return;
        }
        if (a.blk_ts && tid == 0 && j < 128) a.blk_ts[((size_t)blockIdx.x * 128 + j) * 4 + 2] = wall_clock64();
        j++;
        // ------------------------------------------------------------- finish iteration j - 1: Point::mean for ColorCount (clusterc.rs:83-113)
        // + empty-cluster reseed (kmeans.rs:110-137), redundantly in every block: the sums are the same everywhere
        if (blockIdx.x == 0) {   // the buffer iteration j + 1 adds into (every block has read it: they all came through the barrier)
            unsigned long long *Pclr = a.partials + (size_t)((j + 1) % 3) * kPsPartWords;
            for (uint32_t i = tid; i < 5 * K + 2; i += kPsThreads) __hip_atomic_store(&Pclr[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < K) {
            const uint32_t k = tid;
            const size_t at[5] = {3 * (size_t)k, 3 * (size_t)k + 1, 3 * (size_t)k + 2, 3 * (size_t)K + k, 4 * (size_t)K + k};
            unsigned long long d[5];
#pragma unroll
            for (int i = 0; i < 5; i++) d[i] = ps_aread(&Pcur[at[i]]);
#pragma unroll
            for (int i = 0; i < 5; i++) run[i] += d[i];
            uint32_t ck;
            if (run[4] == 0) {
                const uint64_t ri = reseed_index(a.seed, j - 1, k, a.U);  // fake_clone of the stolen point
                ck = a.gx.bits ? gidx_select(a.gx, ri) : a.keys[ri];
                atomicAdd(&s_reseed, 1u);
            } else {
                const uint32_t r = div_floor_small(run[0], run[3]) & 255, g = div_floor_small(run[1], run[3]) & 255, b = div_floor_small(run[2], run[3]) & 255;
                ck = (r << 16) | (g << 8) | b;
                atomicAdd(&s_active, 1u);
            }
            const uint32_t oldc = tab[k].x;
            tab[k] = make_cconst(ck, k, 8);
            if (ck != oldc) {
                const uint32_t pos = atomicAdd(&s_nmoved, 1u);
                if (pos < kMaxMovedSkip) s_mlist[pos] = k;
                atomicOr(&s_mm[(k >> 6) & 3], 1ull << (k & 63));
            }
        } else if (tid == kPsThreads - 1) {
            s_changed = ps_aread(&Pcur[5 * (size_t)K]);
            s_pev = ps_aread(&Pcur[5 * (size_t)K + 1]);
        }
        __syncthreads();
        if (a.blk_ts && tid == 0 && j <= 128) a.blk_ts[((size_t)blockIdx.x * 128 + j - 1) * 4 + 3] = wall_clock64();
        const unsigned long long changed = s_changed;
        nS = s_nmoved;
        reseeds_total += s_reseed;
        evals_total += s_pev;
        const bool fin = changed == 0 || (a.max_iters && j >= a.max_iters);
        if (blockIdx.x == 0 && tid == 0) {
            KmDevState *sw = a.st_rw;
            sw->changed_ring[(j - 1) % kHistRing] = changed;
            sw->nmoved_ring[(j - 1) % kHistRing] = nS;
            if (a.iter_ts && j < kPsTsCap) a.iter_ts[j] = wall_clock64();
            if (fin) {
                sw->moved_last = changed; sw->reseeds = reseeds_total; sw->active = s_active; sw->pair_evals = evals_total; sw->iter = j; sw->done = 1;
                PsExit *x = a.exit_host;
                x->iter = j; x->moved_last = changed; x->reseeds = reseeds_total; x->active = s_active; x->pair_evals = evals_total;
            }
        }
        if (fin) break;   // converged (kmeans.rs:26-32) or the iteration cap: every block sees the same sums and leaves together
    }
    // ----------------------------------------------------------------- results
    if (blockIdx.x == 0 && tid < K) {
        const uint2 cc = tab[tid];
        a.cent_g[tid] = cc.x;
        a.cconst_g[tid] = cc;
        a.members_out[tid] = run[4];
        a.wsum_out[tid] = run[3];
    }
    // the labels of the block's points, four per thread and store where the block's range allows
    {
        const uint64_t g_lo = q0, g_hi = (uint64_t)q0 + nq;
        for (uint64_t g4 = (g_lo & ~3ull) + 4ull * tid; g4 < g_hi; g4 += 4ull * kPsThreads) {
            uint32_t lb[4];
            bool in[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint64_t g = g4 + t;
                in[t] = g >= g_lo && g < g_hi;
                const uint32_t idx = in[t] ? (uint32_t)(g - g_lo) : 0u;
                lb[t] = 0u;
                if (in[t]) { if (idx < nres) lb[t] = pts[idx] >> 24; else lb[t] = pkq[idx] >> 24; }
            }
            if (in[0] && in[3]) *reinterpret_cast<uint32_t *>(a.labels + g4) = lb[0] | (lb[1] << 8) | (lb[2] << 16) | (lb[3] << 24);
            else {
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (in[t]) a.labels[g4 + t] = (uint8_t)lb[t];
            }
        }
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        if (tid == 0) { __threadfence_system(); a.exit_host->status = kPsStatusDone; __threadfence_system(); }
    }
}

// =========================================================================== host side
static std::atomic<int> g_ps_cus_in_use[16];   // CUs promised to persistent launches in flight, per device: two such grids that do not fit the
                                               // chip TOGETHER would each hold CUs the other waits for (this process can know; another cannot: the abort word)

static int ps_cu_count(int device) {   // (hipGetDeviceProperties takes a fraction of a millisecond: once per device)
    static std::atomic<int> cached[16];
    const int d = device >= 0 && device < 16 ? device : 0;
    int n = cached[d].load();
    if (n) return n;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    cached[d].store(prop.multiProcessorCount);
    return prop.multiProcessorCount;
}

uint32_t ps_grid_for(Ctx *c, uint64_t Umax) {
    uint32_t G = (uint32_t)ps_cu_count(c->device);
    if (!G) return 0;
    G = std::max(1u, G / std::max(1u, c->ps_div));
    // (CNIIC_OPT_KM_MAX_BLOCKS: a share of the classic grid's 768 blocks -- batch encodes run several images side by side)
    if (const uint64_t mb = c->opt(CNIIC_OPT_KM_MAX_BLOCKS, "CNIIC_KM_MAX_BLOCKS", 0)) G = std::max<uint32_t>(1u, (uint32_t)((uint64_t)G * std::min<uint64_t>(mb, 768) / 768));
    if (const char *e = test_env("CNIIC_KM_PS_BLOCKS")) G = std::max(1, atoi(e));
    G = (uint32_t)std::min<uint64_t>(G, std::max<uint64_t>(1, ceil_div(Umax, 512)));   // small images: fewer blocks, not emptier ones
    return std::min<uint32_t>(G, 1024u);
}

// the set-up the persistent launch needs beyond the classic loop's (km_rgbw_create calls this once the cell list is enqueued)
int ps_prepare(KmRgbwState *s) {
    Ctx *c = s->c;
    const uint32_t G = ps_grid_for(c, s->U);
    if (!G) return CNIIC_OK;   // (no device properties: the classic loop)
    static bool attr_set[16] = {};
    if (c->device >= 0 && c->device < 16 && !attr_set[c->device]) {
        CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_rgbw_persist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPsDynBytes));
        attr_set[c->device] = true;
    }
    const uint64_t o_bar = 0, o_part = (sizeof(PsBar) + 255) & ~255ull, o_fail = o_part + ((3 * (uint64_t)kPsPartWords * 8 + 255) & ~255ull);
    const uint64_t o_rng = o_fail + 256, total = o_rng + (uint64_t)G * sizeof(PsRange);
    CNIIC_HIP_TRY(c, s->ps_arena.alloc(total));
    CNIIC_HIP_TRY(c, hipMemsetAsync(s->ps_arena.p, 0, o_rng, c->stream));
    CNIIC_HIP_TRY(c, s->ps_pk.alloc(std::max<uint64_t>(s->U, 1) * 4));
    s->ps_blocks = G;
    s->ps_o_part = o_part; s->ps_o_fail = o_fail; s->ps_o_rng = o_rng;
    (void)o_bar;
    uint8_t *a = s->ps_arena.as<uint8_t>();
    uint32_t budget = kPsDynBytes;   // (the tests shrink it: CNIIC_TEST_PS_LDS_BYTES leaves most cells' points in memory)
    if (const char *e = test_env("CNIIC_TEST_PS_LDS_BYTES")) budget = std::min<uint32_t>(kPsDynBytes, (uint32_t)atoi(e));
    hipLaunchKernelGGL(k_ps_ranges, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)s->ne_cost.as<uint32_t>(), (const uint32_t *)s->ne_start.as<uint32_t>(),
                       (const uint32_t *)s->ne_count.as<uint32_t>(), G, budget, reinterpret_cast<PsRange *>(a + o_rng), reinterpret_cast<uint32_t *>(a + o_fail));
    CNIIC_HIP_TRY(c, hipGetLastError());
    s->ps = true;
    return CNIIC_OK;
}

// The whole loop as one launch.  *ran = false: not tried (the CUs are promised to another persistent launch of this process) or given
// up without harm (the grid was not resident together in time; a block's range does not fit): the caller runs the classic loop, whose
// inputs are untouched.
int km_rgbw_run_persistent(KmRgbwState *s, bool *ran) {
    Ctx *c = s->c;
    *ran = false;
    if (!s->ps || s->ps_tried) return CNIIC_OK;
    s->ps_tried = true;   // (a second km_rgbw_run on the same state continues classically: the persistent launch starts from the initial assignment)
    const int dev = c->device >= 0 && c->device < 16 ? c->device : 0, G = (int)s->ps_blocks;
    if (g_ps_cus_in_use[dev].fetch_add(G) + G > ps_cu_count(c->device)) { g_ps_cus_in_use[dev].fetch_sub(G); return CNIIC_OK; }
    struct Release { int dev, G; ~Release() { g_ps_cus_in_use[dev].fetch_sub(G); } } release{dev, G};
    if (!c->pinned_ps) CNIIC_HIP_TRY(c, hipHostMalloc(&c->pinned_ps, 256, hipHostMallocDefault));
    PsExit *xh = static_cast<PsExit *>(c->pinned_ps);
    memset(xh, 0, sizeof *xh);
    uint8_t *ar = s->ps_arena.as<uint8_t>();
    DevBuf ts;
    const bool want_ts = s->profile || test_env("CNIIC_KM_PS_TRACE");
    if (want_ts) { CNIIC_HIP_TRY(c, ts.alloc((uint64_t)kPsTsCap * 8)); CNIIC_HIP_TRY(c, hipMemsetAsync(ts.p, 0, (uint64_t)kPsTsCap * 8, c->stream)); }
    PsArgs a{};
    a.ckeys = s->ckeys.as<uint32_t>(); a.cweight = s->cweight.as<uint32_t>(); a.labels = s->labels.as<uint8_t>();
    a.ne_cell = s->ne_cell.as<uint32_t>(); a.ne_start = s->ne_start.as<uint32_t>();
    a.ranges = reinterpret_cast<const PsRange *>(ar + s->ps_o_rng); a.ranges_fail = reinterpret_cast<const uint32_t *>(ar + s->ps_o_fail);
    a.pk = s->ps_pk.as<uint32_t>(); a.cconst0 = s->cconst.as<uint2>();
    a.partials = reinterpret_cast<unsigned long long *>(ar + s->ps_o_part); a.bar = reinterpret_cast<PsBar *>(ar);
    a.cconst_g = s->cconst.as<uint2>(); a.cent_g = s->cent.as<uint32_t>(); a.members_out = s->members_last.as<uint64_t>(); a.wsum_out = s->wsum_last.as<uint64_t>();
    a.st_rw = s->dstate.as<KmDevState>(); a.exit_host = xh;
    a.keys = s->keys; a.gx = s->gidx; a.seed = s->seed; a.max_iters = s->max_iters; a.U = s->gidx.bits ? s->gidx.U : s->U;
    a.K = s->K; a.max_skip = s->no_skip ? 0u : s->max_skip; a.agg_iters = s->agg_launches;
    a.test_abort_at = 0;
    if (const char *e = test_env("CNIIC_TEST_PS_ABORT_AT")) a.test_abort_at = (uint32_t)atoi(e);
    uint64_t tmo_ms = 2000;
    if (const char *e = test_env("CNIIC_KM_PS_TIMEOUT_MS")) tmo_ms = (uint64_t)atoll(e);
    a.timeout_ticks = tmo_ms * 100000ull;   // 100 MHz
    a.iter_ts = want_ts ? ts.as<unsigned long long>() : nullptr;
    DevBuf bts;
    const char *bt_path = test_env("CNIIC_KM_PS_BLOCK_TRACE");
    if (bt_path) { CNIIC_HIP_TRY(c, bts.alloc((uint64_t)G * 128 * 4 * 8)); CNIIC_HIP_TRY(c, hipMemsetAsync(bts.p, 0, (uint64_t)G * 128 * 4 * 8, c->stream)); a.blk_ts = bts.as<unsigned long long>(); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s->profile) { CNIIC_HIP_TRY(c, hipEventCreate(&e0)); CNIIC_HIP_TRY(c, hipEventCreate(&e1)); }
    hipExtLaunchKernelGGL(k_rgbw_persist, dim3((uint32_t)G), dim3(kPsThreads), kPsDynBytes, c->stream, e0, e1, 0, a);
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, ctx_spin_sync(c));
    const uint32_t status = xh->status;
    if (status != kPsStatusDone) {
        if (e0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
        if (test_env("CNIIC_KM_PS_REQUIRE")) return c->fail(CNIIC_ERR_HIP, "kmeans_rgbw: the persistent launch ended with status %u (CNIIC_KM_PS_REQUIRE)", status);
        // nothing the classic loop reads has been written (labels, colours, counts; the centroids are the initial ones): start over there
        return CNIIC_OK;
    }
    s->run_stats.iterations = xh->iter; s->run_stats.moved_last = xh->moved_last; s->run_stats.empty_reseeds = xh->reseeds;
    s->run_stats.active = xh->active; s->run_stats.pair_evals = xh->pair_evals;
    s->run_stats_valid = true;
    if (s->profile) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        KernelTime &kt = c->ktimes["kmeans_rgbw_persist"];
        kt.ms += ms; kt.launches += 1;
        KernelTime &ki = c->ktimes["kmeans_rgbw_persist_iters"];   // (launches: the iterations the launch ran, for per-iteration figures)
        ki.ms += ms; ki.launches += xh->iter;
    }
    if (want_ts) {
        std::vector<unsigned long long> t(kPsTsCap);
        CNIIC_HIP_TRY(c, hipMemcpy(t.data(), ts.p, (size_t)kPsTsCap * 8, hipMemcpyDeviceToHost));
        if (const char *path = test_env("CNIIC_KM_PS_TRACE")) {
            KmDevState hf;
            CNIIC_HIP_TRY(c, hipMemcpy(&hf, s->dstate.p, sizeof hf, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(path, "w")) {
                fprintf(f, "iteration,us,centroids_moved_before,points_moved\n");
                for (uint64_t i = 0; i < xh->iter && i + 1 < kPsTsCap; i++) {
                    const bool in_ring = xh->iter - i <= kHistRing;
                    const bool prev_in_ring = i >= 1 && xh->iter - (i - 1) <= kHistRing;
                    fprintf(f, "%llu,%.2f,%lld,%lld\n", (unsigned long long)i, (t[i + 1] - t[i]) / 100.0, prev_in_ring ? (long long)hf.nmoved_ring[(i - 1) % kHistRing] : -1ll,
                            in_ring ? (long long)hf.changed_ring[i % kHistRing] : -1ll);
                }
                fclose(f);
            }
        }
    }
    if (bt_path) {   // per block and iteration: microseconds in the assign step, the flush, the barrier, the update (the clock starts where the previous iteration's centroids stood)
        std::vector<unsigned long long> b((size_t)G * 128 * 4);
        CNIIC_HIP_TRY(c, hipMemcpy(b.data(), bts.p, b.size() * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(bt_path, "w")) {
            fprintf(f, "block,iteration,assign_us,flush_us,barrier_us,update_us\n");
            for (int g = 0; g < G; g++)
                for (uint64_t i = 1; i < xh->iter && i < 128; i++) {
                    const unsigned long long *r = &b[((size_t)g * 128 + i) * 4], start = b[((size_t)g * 128 + i - 1) * 4 + 3];
                    fprintf(f, "%d,%llu,%.2f,%.2f,%.2f,%.2f\n", g, (unsigned long long)i, (r[0] - start) / 100.0, (r[1] - r[0]) / 100.0, (r[2] - r[1]) / 100.0, (r[3] - r[2]) / 100.0);
                }
            fclose(f);
        }
    }
    *ran = true;
    return CNIIC_OK;
}

}  // namespace cniic
