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
constexpr uint32_t kPsAccWords = 5 * 256;              // u64 accumulators (K <= 256)
// dynamic LDS: [acc u64 5 x 256 | tab uint2 256 | ids u8 kPsSlotsMax x kPsScap | per block (ps_cell_bytes): first points, records, cell ids, work list, list slots | points u32[...]]
constexpr uint32_t kPsOffTab = kPsAccWords * 8, kPsOffS = kPsOffTab + 256 * 8;
static_assert(kPsOffCell == kPsOffS + kPsSlotsMax * kPsScap * 4, "ps_cell_bytes / kPsOffCell (kmeans_rgbw.hpp) describe this layout");

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
// two levels: the last block of an XCD to arrive reports to the top counter, the last XCD bumps the top generation -- which EVERY block
// watches (round 5, late: until then the XCDs' last blocks watched it and each released its own XCD through a word of its own: one more
// store-to-poll hop, ~0.5 us, on the path behind the slowest block of every iteration; 256 pollers of one line are no load worth the hop)
__device__ __forceinline__ bool ps_barrier_xcd(PsBar *b, uint32_t x, uint32_t nx_blocks, uint32_t nxcd, unsigned long long ticks) {
    const uint32_t tg = ps_ld(&b->topgen.v);   // (before the arrival: nobody can bump it until this block has arrived)
    if (__hip_atomic_fetch_add(&b->xcount[x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nx_blocks - 1) {
        ps_st(&b->xcount[x].v, 0u);   // (its XCD's blocks come back only after the top generation has moved, which this block's report precedes)
        if (__hip_atomic_fetch_add(&b->top.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nxcd - 1) {
            ps_st(&b->top.v, 0u);
            ps_st(&b->topgen.v, tg + 1);
            return true;
        }
    }
    return ps_spin(&b->topgen.v, tg, b, ticks);
}

// ---------------------------------------------------------------- who owns what
// The compacted list of non-empty cells is cut into G x kPsChunks consecutive chunks of equal estimated cost (the classic loop's cost
// model: ne_cost) and block b owns chunks b, G + b, 2 G + b, ...: equal work per block by construction, and a region of colour space --
// what one moved centroid dirties late in a run is a hundred or two ADJACENT cells -- is spread over as many blocks as it has chunks
// instead of landing on one or two (profiles/r05_persist_v1_block_phases.txt: with one contiguous range per block the slowest block of a
// skip iteration did 25 us of work while the mean was 2).  cb[i] = first cell of chunk i, cb[G kPsChunks] = M.
// fail: a block would own more cells than its LDS can describe (then no block starts and the classic loop runs).
__global__ __launch_bounds__(1024) void k_ps_ranges(const uint32_t *__restrict__ ne_cost, const uint32_t *__restrict__ ne_count, uint32_t G, uint32_t dyn_bytes,
                                                    uint32_t *__restrict__ cb, uint32_t *__restrict__ fail) {
    const uint32_t M = *ne_count, NC = G * kPsChunks;
    const uint64_t total = ne_cost[M];
    // (a thread's boundaries searched TOGETHER, eight at a time: one after the other the 4097 bisections of the headline image were 60
    // dependent reads a thread, 21 us in front of the launch)
    for (uint32_t g0 = threadIdx.x; g0 <= NC; g0 += 8 * blockDim.x) {
        uint32_t a[8], b[8];
        uint64_t c_lo[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t g = g0 + r * blockDim.x;
            c_lo[r] = total * min(g, NC) / NC;
            a[r] = 0; b[r] = g <= NC ? M : 0u;   // first cell whose cost prefix is >= c_lo
        }
        bool more = true;
        while (more) {
            more = false;
            uint32_t mid[8], v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) { mid[r] = (a[r] + b[r]) >> 1; v[r] = a[r] < b[r] ? ne_cost[mid[r]] : 0u; }
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (a[r] < b[r]) { if (v[r] < c_lo[r]) a[r] = mid[r] + 1; else b[r] = mid[r]; more = more || a[r] < b[r]; }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t g = g0 + r * blockDim.x;
            if (g <= NC) cb[g] = g == NC ? M : a[r];
        }
    }
    __syncthreads();   // (one block: its own stores are visible to it behind the barrier)
    for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) {
        uint32_t C = 0;
        for (uint32_t r = 0; r < kPsChunks; r++) C += cb[r * G + g + 1] - cb[r * G + g];
        if (C > kPsMaxCells || kPsOffCell + ps_cell_bytes(C) > dyn_bytes) atomicAdd(fail, 1u);
    }
}

// ---- reductions inside a ROW of 16 lanes (a DPP row: rotations stay inside it), the result in every lane of the row
template <class Op> __device__ __forceinline__ uint32_t row_allreduce(uint32_t v, Op op) {
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));  // row_ror:1
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));  // row_ror:2
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));  // row_ror:4
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));  // row_ror:8
    return v;
}
__device__ __forceinline__ uint32_t row_min(uint32_t v) { return row_allreduce(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
__device__ __forceinline__ uint32_t row_max(uint32_t v) { return row_allreduce(v, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); }
__device__ __forceinline__ uint32_t row_sum(uint32_t v) { return row_allreduce(v, [](uint32_t a, uint32_t b) { return a + b; }); }

// ---------------------------------------------------------------- candidates
// squared distance from colour key ck to the centre of the cube at low corner `lo` (packed) with half side h: |v|^2 - 2 v.c + |c|^2
__device__ __forceinline__ uint32_t ps_centre_dist(uint32_t ck, uint32_t cpk, uint32_t cc) { return dot4u8(ck, ck, cc) - 2u * dot4u8(ck, cpk, 0); }

// The super-cell lists of the block, a wave per list.  List `slot` = the centroids (ascending id) that can be nearest somewhere in
// super-cell ssup[slot]: pivot = the centroid nearest the cube's centre, kept = whoever the pivot does not dominate over the whole cube
// (Dominance, kmeans_rgbw.hpp).  An entry is id << 24 | colour: the per-cell builds read ONE word per member, all of a lane's members
// at once.  A list of more than kPsScap members is not kept: its cells build from the table.
__device__ __forceinline__ void ps_build_lists(const uint2 *tab, uint32_t K, uint32_t nslots, const uint16_t *ssup, uint32_t *s_nS, uint32_t *s_lpiv, uint32_t *Sent, int wid, int lane) {
    constexpr int32_t ext = (1 << (kCellShift + 2)) - 1;
    for (uint32_t slot = (uint32_t)wid; slot < nslots; slot += kPsWaves) {
        const CellBox bx = super_box((uint32_t)__builtin_amdgcn_readfirstlane((int)ssup[slot]));
        const uint32_t cpk = pack_rgb(bx.r0 + 16, bx.g0 + 16, bx.b0 + 16), cc = dot4u8(cpk, cpk, 0);
        uint32_t ck[4], best = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t k = 64u * r + (uint32_t)lane;
            ck[r] = k < K ? tab[k].x : 0u;
            if (k < K) best = min(best, (ps_centre_dist(ck[r], cpk, cc) << 8) | k);
        }
        const uint32_t piv = wave_reduce_min(best) & 255u;
        Dominance dm;
        dm.set(bx, ext, (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[piv].x));
        uint32_t n = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t k = 64u * r + (uint32_t)lane;
            const bool keep = k < K && dm.worst(ck[r]) >= 0;
            const unsigned long long bm = __ballot(keep);
            const uint32_t pos = n + lanes_below(bm);
            if (keep && pos < kPsScap) Sent[slot * kPsScap + pos] = (k << 24) | ck[r];
            n += (uint32_t)__popcll(bm);
        }
        if (lane == 0) { s_nS[slot] = n; s_lpiv[slot] = piv; }
    }
    __syncthreads();
}

// word 1 of a cell's record: bits 0..8 the label all its points carry (kPsUlMask: not known), bit 9 the mask holds EVERY centroid of the table
// the cell's pivot does not dominate (built from the table), bits 16..23 the cell's pivot, bits 24..31 the pivot of the super-cell list the
// mask was built from.  A mask stays a base for the incremental update of the skip schedule while the cell's pivot has not moved and
// -- unless complete -- the list's pivot has not either: what the list left out was dominated by THAT centroid over the whole super-cell.
constexpr uint32_t kPsUlMask = 0x1ffu, kPsComplete = 0x200u;

// A cell's record once its mask words stand in rec[2..9] (every lane of the cell's row calls this): the number of candidates, up to
// four of their ids as bytes of rec[10] (what a sweep reads instead of walking the mask), and the pivot's id / the complete flag in
// word 1 (the cell's common label stays).  Returns (row-uniform) whether the cell must be swept: not if ONE candidate is left and every
// point already carries it (the lone candidate beats every other centroid for every colour of the cube: nothing can move).
__device__ __forceinline__ bool ps_row_finish(uint32_t *rec, uint32_t l16, uint32_t pid, uint32_t lpid, bool complete) {
    const uint32_t w = l16 < 8 ? rec[2 + l16] : 0u;
    const uint32_t pc = (uint32_t)__popc(w);
    uint32_t inc = pc;   // inclusive scan inside the row (lanes 8..15 add nothing)
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, true);   // row_shr:1
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, true);   // row_shr:2
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, true);   // row_shr:4
    const uint32_t cnt = row_max(inc);
    uint32_t pos = inc - pc, ww = w;
    uint8_t *c4 = reinterpret_cast<uint8_t *>(rec + 10);
    while (ww && pos < 4) {
        c4[pos] = (uint8_t)(32u * l16 + (uint32_t)__builtin_ctz(ww));
        ww &= ww - 1;
        pos++;
    }
    const uint32_t r1 = rec[1];
    if (l16 == 0) rec[1] = (r1 & kPsUlMask) | (pid << 16) | (lpid << 24) | (complete ? kPsComplete : 0u);
    const uint32_t only = rec[10] & 255u;   // (behind the byte stores of this wave: the LDS serves a wave in order)
    return !(cnt == 1 && (r1 & kPsUlMask) == only);
}

// One cell's candidates from its super-cell's list, by the 16 lanes of a row: every lane takes the members l16, l16 + 16, ... (at most
// kPsScap / 16, all read at once), pivot = the member nearest the cube's centre (lowest id on ties), kept = the members the pivot does not
// dominate over the cell's cube; the mask (bit k <=> centroid k) into the cell's record.  Returns the pivot's id.
__device__ __forceinline__ uint32_t ps_row_build_list(const uint2 *tab, const uint32_t *ent, uint32_t n, uint32_t c, uint32_t l16, uint32_t *rec) {
    constexpr int32_t ext = (1 << kCellShift) - 1;
    constexpr int PER = kPsScap / 16;
    const CellBox bx = cell_box(c);
    const uint32_t cpk = pack_rgb(bx.r0 + 4, bx.g0 + 4, bx.b0 + 4), cc = dot4u8(cpk, cpk, 0);
    uint32_t en[PER];
#pragma unroll
    for (int t = 0; t < PER; t++) en[t] = l16 + 16u * t < n ? ent[l16 + 16u * t] : 0xffffffffu;
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int t = 0; t < PER; t++)
        if (l16 + 16u * t < n) best = min(best, (ps_centre_dist(en[t] & 0xffffffu, cpk, cc) << 8) | (en[t] >> 24));
    const uint32_t pid = row_min(best) & 255u;
    Dominance dm;
    dm.set(bx, ext, tab[pid].x);
    if (l16 < 8) rec[2 + l16] = 0u;
#pragma unroll
    for (int t = 0; t < PER; t++)
        if (l16 + 16u * t < n && dm.worst(en[t] & 0xffffffu) >= 0) { const uint32_t k = en[t] >> 24; atomicOr(&rec[2 + (k >> 5)], 1u << (k & 31)); }
    if (l16 == 0) rec[0] = tab[pid].x;
    return pid;
}
// ... and from the whole table (a list that did not fit; a dirty cell of the skip schedule whose mask is not complete or whose pivot moved)
__device__ __forceinline__ uint32_t ps_row_build_table(const uint2 *tab, uint32_t K, uint32_t c, uint32_t l16, uint32_t *rec) {
    constexpr int32_t ext = (1 << kCellShift) - 1;
    const CellBox bx = cell_box(c);
    const uint32_t cpk = pack_rgb(bx.r0 + 4, bx.g0 + 4, bx.b0 + 4), cc = dot4u8(cpk, cpk, 0);
    uint32_t best = 0xffffffffu;
    for (uint32_t k = l16; k < K; k += 16) best = min(best, (ps_centre_dist(tab[k].x, cpk, cc) << 8) | k);
    const uint32_t pid = row_min(best) & 255u;
    Dominance dm;
    dm.set(bx, ext, tab[pid].x);
    if (l16 < 8) rec[2 + l16] = 0u;
    for (uint32_t k = l16; k < K; k += 16)
        if (dm.worst(tab[k].x) >= 0) atomicOr(&rec[2 + (k >> 5)], 1u << (k & 31));
    if (l16 == 0) rec[0] = tab[pid].x;
    return pid;
}

// -DCNIIC_PS_PHASES: wave-clock totals per phase of k_rgbw_persist (a measuring build, never the shipped one):
// 0 lists, 1 classify, 2 draw + descriptors, 3 mask, 4 point words, 5 sweep, 6 cell tail, 7 flush .. barrier .. update, 8 cells swept;
// inside a sweep: 10 unpack + one-candidate exit, 11 table reads + scores, 12 who moves, 13 labels, 14 booking
#ifdef CNIIC_PS_PHASES
__device__ unsigned long long g_ps_phase[16];
#define PS_PHASE(i) do { const long long now_ = clock64(); ph_[i] += (unsigned long long)(now_ - t_ph); t_ph = now_; } while (0)
#define PS_COUNT(i, v) do { ph_[i] += (unsigned long long)(v); } while (0)
#define PS_PROF_PARAMS , unsigned long long (&ph_)[16], long long &t_ph
#define PS_PROF_ARGS , ph_, t_ph
#else
#define PS_PHASE(i) do {} while (0)
#define PS_COUNT(i, v) do {} while (0)
#define PS_PROF_PARAMS
#define PS_PROF_ARGS
#endif

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

// ---- the resident point word (its own format: the sweep's common path is two instructions to unpack a point)
//   bits 0..2 / 8..10 / 16..18   the colour inside its 8^3 cell, b / g / r, each in the byte lane the key has it in: key = cell base | (w & 0x070707)
//   bits 24..31                  255 - label: what the low byte of a packed score holds, so "the best candidate is not my cluster" is one compare
//   bits 3..7, 11..13            the pixel count's low five and high three bits; 255 = look it up (needed only when a point moves)
__device__ __forceinline__ uint32_t ps_pack(uint32_t key, uint32_t w, uint32_t label) {
    const uint32_t wc = min(w, 255u);
    return (key & 0x070707u) | ((wc & 31u) << 3) | ((wc >> 5) << 11) | ((255u - label) << 24);
}
__device__ __forceinline__ uint32_t ps_wt(uint32_t w) { return ((w >> 3) & 31u) | ((w >> 6) & 0xe0u); }

// the signed deltas of one mover into the block's accumulators (clusterc.rs:92-98: sums of channel x count, of counts, of members)
__device__ __forceinline__ void ps_book_move(unsigned long long *acc, uint32_t K, uint32_t pp, uint64_t w, uint32_t ol, uint32_t nl) {
    const unsigned long long rw = ((pp >> 16) & 255) * w, gw = ((pp >> 8) & 255) * w, bw = (pp & 255) * w;
    atomicAdd(&acc[3 * nl + 0], rw); atomicAdd(&acc[3 * nl + 1], gw); atomicAdd(&acc[3 * nl + 2], bw);
    atomicAdd(&acc[3 * K + nl], (unsigned long long)w); atomicAdd(&acc[4 * K + nl], 1ull);
    atomicAdd(&acc[3 * ol + 0], 0ull - rw); atomicAdd(&acc[3 * ol + 1], 0ull - gw); atomicAdd(&acc[3 * ol + 2], 0ull - bw);
    atomicAdd(&acc[3 * K + ol], 0ull - (unsigned long long)w); atomicAdd(&acc[4 * K + ol], 0ull - 1ull);
}

// the best packed score of every slot's colour over a cell's candidates: up to four ids as bytes of the record's last word (their
// table entries asked for together), more than four by walking the mask on the scalar unit
__device__ __forceinline__ void ps_scores(const uint32_t (&key)[kSweep], const unsigned long long (&nm)[4], uint32_t ncand, uint32_t cand4, const uint2 *tab, uint32_t (&best)[kSweep]) {
    if (ncand <= 4) {
        const uint2 c0 = tab[cand4 & 255u], c1 = tab[(cand4 >> 8) & 255u], c2 = tab[(cand4 >> 16) & 255u], c3 = tab[cand4 >> 24];
#pragma unroll
        for (int u = 0; u < kSweep; u++) best[u] = (dot4u8(key[u], c0.x, 0) << 9) + c0.y;
        if (ncand > 1) {
#pragma unroll
            for (int u = 0; u < kSweep; u++) best[u] = max(best[u], (dot4u8(key[u], c1.x, 0) << 9) + c1.y);
        }
        if (ncand > 2) {
#pragma unroll
            for (int u = 0; u < kSweep; u++) best[u] = max(best[u], (dot4u8(key[u], c2.x, 0) << 9) + c2.y);
        }
        if (ncand > 3) {
#pragma unroll
            for (int u = 0; u < kSweep; u++) best[u] = max(best[u], (dot4u8(key[u], c3.x, 0) << 9) + c3.y);
        }
    } else ps_best(key, nm, tab, best);
}

// One sweep of an iteration after the first: the 64 x kSweep point words wd (positions base + 64 u + lane of the block's point range,
// those below e) against the cell's candidates.  Stay unless another centroid is STRICTLY closer (kmeans.rs:375), lowest id among
// equals (the score's low byte).  The common path never looks at a point's own centroid: its score is among the candidates', so the
// best one carrying the point's label means "stays"; only points whose best candidate is somebody else compare with their own
// centroid (a tie keeps them).  A mover's label byte is rewritten in place (store_label: the words' home, LDS or memory), its signed
// deltas booked, one mover per lane and pass.  agg (the first iterations after iteration 0, where centroids still travel and whole
// cells change hands): the movers that share the first mover's (old, new) pair are summed in the wave and booked by one lane.
// Returns true if the sweep left through the one-candidate exit (every point carries the lone candidate).
template <typename StoreLabel>
__device__ __forceinline__ bool ps_sweep(const uint32_t (&wd)[kSweep], uint32_t base, uint32_t e, uint32_t s0, int lane, const unsigned long long (&nm)[4], uint32_t ncand,
                                         uint32_t cand4, const uint2 *tab, uint32_t K, uint32_t cbk, const uint32_t *__restrict__ cwq, unsigned long long *acc, uint32_t &moved,
                                         bool agg, StoreLabel store_label PS_PROF_PARAMS) {
    uint32_t key[kSweep], lc[kSweep];
    bool valid[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        key[u] = cbk | (wd[u] & 0x070707u);
        lc[u] = wd[u] >> 24;
        valid[u] = base + u * 64 + lane < e;
    }
    if (ncand == 1) {
        // More than half of the cells lie inside one cluster's region: ONE candidate, and as a rule every point already carries its
        // label; the lone candidate beats every other centroid for every colour of the cube, so nothing can move.
        const uint32_t onlyc = 255u - (cand4 & 255u);
        bool same = true;
#pragma unroll
        for (int u = 0; u < kSweep; u++) same = same & (!valid[u] | (lc[u] == onlyc));
        if (__ballot(!same) == 0ull) { PS_PHASE(10); return true; }
    }
    PS_PHASE(10);
    uint32_t best[kSweep];
    ps_scores(key, nm, ncand, cand4, tab, best);
    PS_PHASE(11);
    bool mv[kSweep], any = false;
#pragma unroll
    for (int u = 0; u < kSweep; u++) { mv[u] = valid[u] & ((best[u] & 255u) != lc[u]); any = any | mv[u]; }
    if (!__ballot(any)) { PS_PHASE(12); return false; }
    any = false;
#pragma unroll
    for (int u = 0; u < kSweep; u++) {   // (the few whose best candidate is not their own cluster: strictly closer, kmeans.rs:375?)
        const uint2 cc = tab[255u - lc[u]];
        const uint32_t kc = (dot4u8(key[u], cc.x, 0) << 9) + cc.y;
        mv[u] = mv[u] & ((best[u] >> 8) > (kc >> 8));
        any = any | mv[u];
    }
    if (!__ballot(any)) { PS_PHASE(12); return false; }
    PS_PHASE(12);
    uint32_t wt[kSweep];
    bool heavy = false;
#pragma unroll
    for (int u = 0; u < kSweep; u++) { wt[u] = ps_wt(wd[u]); heavy = heavy | (mv[u] & (wt[u] == 255u)); }
    if (__ballot(heavy)) {   // a pixel count of 255 and more is looked up (rare in a photograph)
#pragma unroll
        for (int u = 0; u < kSweep; u++)
            if (mv[u] && wt[u] == 255u) wt[u] = cwq[base + u * 64 + lane - s0];
    }
#pragma unroll
    for (int u = 0; u < kSweep; u++)
        if (mv[u]) { store_label(base + u * 64 + lane, best[u] & 255u); moved++; }
    PS_PHASE(13);
    if (agg) {
        static_assert(kSweep == 4, "four slots per lane");
        uint32_t nmv = 0;
#pragma unroll
        for (int u = 0; u < kSweep; u++) nmv += (uint32_t)__popcll(__ballot(mv[u]));
        if (nmv >= kAggMin) {
#pragma unroll 1
            for (int round = 0; round < 6; round++) {
                const unsigned long long b0 = __ballot(mv[0]), b1 = __ballot(mv[1]), b2 = __ballot(mv[2]), b3 = __ballot(mv[3]);
                if (!(b0 | b1 | b2 | b3)) return false;
                uint32_t pn, po;   // (as low bytes of scores: 255 - id)
                if (b0) { const int l = __builtin_ctzll(b0); pn = (uint32_t)__builtin_amdgcn_readlane((int)best[0], l) & 255u; po = (uint32_t)__builtin_amdgcn_readlane((int)lc[0], l); }
                else if (b1) { const int l = __builtin_ctzll(b1); pn = (uint32_t)__builtin_amdgcn_readlane((int)best[1], l) & 255u; po = (uint32_t)__builtin_amdgcn_readlane((int)lc[1], l); }
                else if (b2) { const int l = __builtin_ctzll(b2); pn = (uint32_t)__builtin_amdgcn_readlane((int)best[2], l) & 255u; po = (uint32_t)__builtin_amdgcn_readlane((int)lc[2], l); }
                else { const int l = __builtin_ctzll(b3); pn = (uint32_t)__builtin_amdgcn_readlane((int)best[3], l) & 255u; po = (uint32_t)__builtin_amdgcn_readlane((int)lc[3], l); }
                uint32_t cnt = 0;
                bool mt[kSweep];
#pragma unroll
                for (int u = 0; u < kSweep; u++) {
                    mt[u] = mv[u] && (best[u] & 255u) == pn && lc[u] == po;
                    cnt += (uint32_t)__popcll(__ballot(mt[u]));
                    mv[u] = mv[u] && !mt[u];
                }
                if (cnt < kAggMin) {   // a handful books itself, and so does everybody who is left
#pragma unroll
                    for (int u = 0; u < kSweep; u++) mv[u] = mv[u] || mt[u];
                    break;
                }
                const size_t kn = 255u - pn, ko = 255u - po;
                auto book = [&](int shift, uint32_t mask, size_t at_new, size_t at_old) {
                    unsigned long long v = 0;
#pragma unroll
                    for (int u = 0; u < kSweep; u++)
                        if (mt[u]) v += (unsigned long long)(mask ? (key[u] >> shift) & mask : 1u) * wt[u];
                    v = wave_reduce_sum64(v);
                    if (lane == 0) { atomicAdd(&acc[at_new], v); atomicAdd(&acc[at_old], 0ull - v); }
                };
                book(16, 255u, 3 * kn + 0, 3 * ko + 0);
                book(8, 255u, 3 * kn + 1, 3 * ko + 1);
                book(0, 255u, 3 * kn + 2, 3 * ko + 2);
                book(0, 0u, 3 * (size_t)K + kn, 3 * (size_t)K + ko);
                if (lane == 0) { atomicAdd(&acc[4 * K + kn], (unsigned long long)cnt); atomicAdd(&acc[4 * K + ko], 0ull - (unsigned long long)cnt); }
            }
        }
    }
    // one mover per lane and pass (as four exec-masked bodies every slot with a single mover in the wave cost the whole booking sequence)
    for (;;) {
        const bool has = mv[0] | mv[1] | mv[2] | mv[3];
        if (!__ballot(has)) break;
        if (has) {
            const int u = mv[0] ? 0 : mv[1] ? 1 : mv[2] ? 2 : 3;
            const uint32_t kk = u == 0 ? key[0] : u == 1 ? key[1] : u == 2 ? key[2] : key[3];
            const uint32_t ww = u == 0 ? wt[0] : u == 1 ? wt[1] : u == 2 ? wt[2] : wt[3];
            const uint32_t bb = u == 0 ? best[0] : u == 1 ? best[1] : u == 2 ? best[2] : best[3];
            const uint32_t oo = u == 0 ? lc[0] : u == 1 ? lc[1] : u == 2 ? lc[2] : lc[3];
            ps_book_move(acc, K, kk, ww, 255u - oo, 255u - (bb & 255u));
            mv[0] = mv[0] & (u != 0); mv[1] = mv[1] & (u != 1); mv[2] = mv[2] & (u != 2); mv[3] = false | (mv[3] & (u != 3));
        }
    }
    PS_PHASE(14);
    return false;
}

// The sweep of iteration 0 over the point words the block loaded at entry (colour, pixel count, the initial label of init_assignment,
// kmeans.rs:61-78): EVERY point adds to the sums of the cluster it ends in (the running sums start at zero).  A sweep lies inside one
// 8^3 cell and its points join one, two, three clusters: round by round, the cluster of the first point still to be booked, every
// point that joins it summed in the wave, one lane adds the totals.
template <typename StoreLabel>
__device__ __forceinline__ void ps_sweep_first(const uint32_t (&wd)[kSweep], uint32_t base, uint32_t e, uint32_t s0, int lane, const unsigned long long (&nm)[4], uint32_t ncand,
                                               uint32_t cand4, const uint2 *tab, uint32_t K, uint32_t cbk, const uint32_t *__restrict__ cwq, unsigned long long *acc, uint32_t &moved,
                                               StoreLabel store_label) {
    uint32_t p[kSweep], lc[kSweep], wt[kSweep];
    bool heavy = false;
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        p[u] = cbk | (wd[u] & 0x070707u);
        lc[u] = wd[u] >> 24;
        wt[u] = ps_wt(wd[u]);
        heavy = heavy | (wt[u] == 255u);
    }
    if (__ballot(heavy)) {   // a pixel count of 255 and more is looked up
#pragma unroll
        for (int u = 0; u < kSweep; u++)
            if (wt[u] == 255u && base + u * 64 + lane < e) wt[u] = cwq[base + u * 64 + lane - s0];
    }
    uint32_t best[kSweep];
    ps_scores(p, nm, ncand, cand4, tab, best);
    uint32_t nl[kSweep];   // 255 - the cluster a point ends in
    bool rem[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        const uint32_t idx = base + u * 64 + lane;
        nl[u] = lc[u];
        rem[u] = idx < e;
        if (rem[u] && (best[u] & 255u) != lc[u]) {
            const uint2 cc = tab[255u - lc[u]];
            const uint32_t kc = (dot4u8(p[u], cc.x, 0) << 9) + cc.y;
            if ((best[u] >> 8) > (kc >> 8)) { nl[u] = best[u] & 255u; moved++; store_label(idx, nl[u]); }  // strictly closer (kmeans.rs:375)
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
        const size_t kn = 255u - pn;
        auto book = [&](int shift, uint32_t mask, size_t at) {
            unsigned long long v = 0;
#pragma unroll
            for (int u = 0; u < kSweep; u++)
                if ((mbits >> u) & 1u) v += (unsigned long long)(mask ? (p[u] >> shift) & mask : 1u) * wt[u];
            v = wave_reduce_sum64(v);
            if (lane == 0) atomicAdd(&acc[at], v);
        };
        book(16, 255u, 3 * kn + 0);
        book(8, 255u, 3 * kn + 1);
        book(0, 255u, 3 * kn + 2);
        book(0, 0u, 3 * (size_t)K + kn);
        if (lane == 0) atomicAdd(&acc[4 * K + kn], (unsigned long long)cnt);
    }
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        if (rem[u]) {
            const uint32_t pp = p[u], n_ = 255u - nl[u];
            const uint64_t w = wt[u];
            atomicAdd(&acc[3 * n_ + 0], ((pp >> 16) & 255) * w); atomicAdd(&acc[3 * n_ + 1], ((pp >> 8) & 255) * w); atomicAdd(&acc[3 * n_ + 2], (pp & 255) * w);
            atomicAdd(&acc[3 * K + n_], (unsigned long long)w); atomicAdd(&acc[4 * K + n_], 1ull);
        }
    }
}

// What only the launch's last instructions (or a rare branch) need sits in pinned host memory behind one pointer: as kernel arguments
// those thirty scalar registers would be alive across the whole loop (the first build spilled 136 scalar registers into vector ones and
// three of those to scratch).
struct PsCold {
    PsExit exit;                          // how the launch ended (the host reads it)
    uint2 *cconst_g;                      // results, written by block 0 on a regular exit
    uint32_t *cent_g;
    uint64_t *members_out, *wsum_out;
    const uint32_t *keys;                 // canonical point list (empty-cluster reseed) ...
    GIdx gx;                              // ... or the index of all occupied colours
    uint64_t seed, U;
};
struct PsArgs {
    const uint32_t *ckeys, *cweight;      // cell-major colours and pixel counts
    uint8_t *labels;                      // cell-major labels: the initial assignment on entry, the result on a regular exit
    const uint32_t *ne_cell, *ne_start;   // compacted non-empty cells: id, first position
    const uint32_t *cb;                   // chunk boundaries (k_ps_ranges)
    const uint32_t *ranges_fail;
    uint32_t *pk;                         // packed words of the points that do not fit their block's LDS, by cell-major position
    const uint2 *cconst0;                 // the initial centroids (k_rgbw_init_cent)
    unsigned long long *partials;         // 3 x kPsPartWords, zero on entry
    PsBar *bar;                           // zero on entry
    KmDevState *st_rw;
    PsCold *cold;                         // pinned
    uint64_t max_iters;
    uint32_t K, max_skip, agg_iters, test_abort_at, lds_budget;
    uint32_t clean_skip;                  // full schedule: cells none of whose old or new candidates moved are not swept (0: A/B measurements, CNIIC_KM_PS_CLEANSKIP)
    unsigned long long timeout_ticks;
    unsigned long long *iter_ts;          // block 0's clock (100 MHz) when iteration j's centroids stood, [0] at entry; kPsTsCap entries, or null
    unsigned long long *blk_ts;           // measuring runs (CNIIC_KM_PS_BLOCK_TRACE): [block][iteration < 128][4] clock at: assign done, flushed, through the barrier, centroids stand
};

__global__ __launch_bounds__(kPsThreads) void k_rgbw_persist(PsArgs a) {
    extern __shared__ __align__(16) unsigned long long lds[];
    __shared__ uint32_t s_moved, s_nmoved, s_reseed, s_active, s_ok, s_nx, s_nxcd, s_qn, s_qhead, s_Cres, s_nslots;
    __shared__ uint32_t s_mlist[kMaxMovedSkip];
    __shared__ unsigned long long s_mm[4], s_evals, s_changed, s_pev;
    __shared__ uint32_t s_nS[kPsSlotsMax], s_lpiv[kPsSlotsMax], s_mcol[kMaxMovedSkip];
    __shared__ uint16_t s_ssup[kPsSlotsMax];
    __shared__ uint32_t s_cbase[kPsChunks + 1], s_cm0[kPsChunks], s_scan[kPsThreads / 64];
    const uint32_t tid = threadIdx.x, K = a.K, G = gridDim.x;
    const int lane = tid & 63, wid = tid >> 6;
    unsigned long long *acc = lds;
    uint2 *tab = reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(lds) + kPsOffTab);
    uint32_t *Sent = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(lds) + kPsOffS);
    if (*a.ranges_fail) {   // (the same word for every block: nobody starts, nobody waits)
        if (blockIdx.x == 0 && tid == 0) { a.cold->exit.status = kPsStatusRanges; __threadfence_system(); }
        return;
    }
    // ---- the block's cells: chunks b, G + b, 2 G + b, ... of the compacted list, in that order
    if (tid < kPsChunks) {
        const uint32_t m0 = a.cb[tid * G + blockIdx.x], m1 = a.cb[tid * G + blockIdx.x + 1];
        s_cm0[tid] = m0;
        s_cbase[tid + 1] = m1 - m0;
    }
    if (tid == 0) {
        s_cbase[0] = 0;
        s_moved = 0; s_evals = 0; s_nmoved = 0; s_reseed = 0; s_active = 0; s_mm[0] = s_mm[1] = s_mm[2] = s_mm[3] = 0ull; s_qn = 0; s_qhead = 0; s_Cres = 0; s_nslots = 0;
    }
    __syncthreads();
    if (tid == 0)
        for (uint32_t r = 0; r < kPsChunks; r++) s_cbase[r + 1] += s_cbase[r];
    __syncthreads();
    const uint32_t C = s_cbase[kPsChunks], Cr = (C + 3u) & ~3u;
    uint8_t *cellb = reinterpret_cast<uint8_t *>(lds) + kPsOffCell;
    uint32_t *cstart = reinterpret_cast<uint32_t *>(cellb);                     // [Cr + 4] first point of a cell in the block's own numbering ([C]: all its points)
    uint32_t *gstart = cstart + Cr + 4;                                          // [Cr] ... and in the cell-major arrays
    uint32_t *recs = gstart + Cr;                                                // [Cr][kPsRecWords] skip records
    uint16_t *ccell = reinterpret_cast<uint16_t *>(recs + (size_t)kPsRecWords * Cr);   // [Cr] cell ids
    uint16_t *queue = ccell + Cr;                                                // [Cr] cells to sweep in this iteration
    uint8_t *cslot = reinterpret_cast<uint8_t *>(queue + Cr);                    // [Cr] which shared list (255: none, the table)
    uint32_t *pts = reinterpret_cast<uint32_t *>(cslot + Cr);                    // the resident points' words
    const uint32_t cap = (a.lds_budget - kPsOffCell - ps_cell_bytes(C)) / 4u;    // (k_ps_ranges made sure the budget covers the descriptors)
    {   // two cells per thread: 2 tid and 2 tid + 1 (a block owns at most kPsMaxCells = 2 x kPsThreads)
        uint32_t np[2] = {0, 0}, cid[2] = {0, 0}, rr[2] = {0, 0};
        bool mine[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t i = 2 * tid + q;
            mine[q] = i < C;
            if (mine[q]) {
                uint32_t r = 0;
                while (i >= s_cbase[r + 1]) r++;
                rr[q] = r;
                const uint32_t m = s_cm0[r] + (i - s_cbase[r]);
                cid[q] = a.ne_cell[m];
                const uint32_t gs = a.ne_start[m];
                np[q] = a.ne_start[m + 1] - gs;
                ccell[i] = (uint16_t)cid[q];
                gstart[i] = gs;
                uint32_t *rec = recs + (size_t)kPsRecWords * i;
                rec[0] = 0u; rec[1] = kPsUlMask; rec[10] = 0u;   // (no pivot yet; the cell's common label: unknown)
            }
        }
        const uint32_t off = block_exclusive_scan<kPsThreads>(np[0] + np[1], s_scan);
        if (mine[0]) cstart[2 * tid] = off;
        if (mine[1]) cstart[2 * tid + 1] = off + np[0];
        if (tid == (C ? (C - 1) / 2 : 0)) cstart[C] = C ? off + np[0] + np[1] : 0u;
        __syncthreads();
        // the distinct super-cells of the range, in order: a shared list each (the first kPsSlotsMax of them)
        bool flag[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t i = 2 * tid + q;
            flag[q] = mine[q] && (i == s_cbase[rr[q]] || ((uint32_t)ccell[i - 1] >> kSuperShift) != (cid[q] >> kSuperShift));
        }
        const uint32_t before = block_exclusive_scan<kPsThreads>((flag[0] ? 1u : 0u) + (flag[1] ? 1u : 0u), s_scan);
        uint32_t run_f = before;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t i = 2 * tid + q;
            run_f += flag[q] ? 1u : 0u;
            if (mine[q]) {
                const uint32_t sidx = run_f - 1u;
                cslot[i] = sidx < kPsSlotsMax ? (uint8_t)sidx : (uint8_t)255;
                if (flag[q] && sidx < kPsSlotsMax) s_ssup[sidx] = (uint16_t)(cid[q] >> kSuperShift);
                if (i == C - 1) s_nslots = min(sidx + 1u, kPsSlotsMax);
                if (cstart[i + 1] <= cap && (i + 1 == C || cstart[i + 2] > cap)) s_Cres = i + 1;   // the cells whose points fit the LDS behind the descriptors
            }
        }
    }
    for (uint32_t i = tid; i < 5 * K; i += kPsThreads) acc[i] = 0ull;
    for (uint32_t i = tid; i < K; i += kPsThreads) tab[i] = a.cconst0[i];
    __syncthreads();
    // ---- the block's points: colour, pixel count and the initial label (init_assignment, kmeans.rs:61-78) of every point of its cells,
    // as one packed word each, into LDS (the cells that fit) or the packed array in memory; a wave per cell, four loads in flight per array
    {
        const uint32_t Cres0 = s_Cres;
        for (uint32_t i = wid; i < C; i += kPsWaves) {
            const uint32_t s = cstart[i], n = cstart[i + 1] - s, gs = gstart[i];
            const uint32_t cbk = cell_base_key(ccell[i]);
            (void)cbk;
            for (uint32_t t0 = 0; t0 < n; t0 += 256) {
                uint32_t kk[4], ww[4], ll[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t t = t0 + u * 64 + lane;
                    const bool in = t < n;
                    kk[u] = in ? a.ckeys[gs + t] : 0u;
                    ww[u] = in ? a.cweight[gs + t] : 0u;
                    ll[u] = in ? (uint32_t)a.labels[gs + t] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t t = t0 + u * 64 + lane;
                    if (t < n) {
                        const uint32_t w = ps_pack(kk[u], ww[u], ll[u]);
                        if (i < Cres0) pts[s + t] = w; else a.pk[gs + t] = w;
                    }
                }
            }
        }
    }
    if (tid == 0) {
        // census: how many blocks does my XCD hold, how many XCDs are in use?  (nothing about placement is assumed)
        const uint32_t x = ps_xcc_id();
        __hip_atomic_fetch_add(&a.bar->xblocks[x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = ps_barrier_flat(a.bar, G, a.timeout_ticks);
        uint32_t n = 0;
        for (int i = 0; i < 8; i++) n += ps_ld(&a.bar->xblocks[i].v) != 0;
        s_nx = ps_ld(&a.bar->xblocks[x].v);
        s_nxcd = n;
        s_ok = ok;
        if (blockIdx.x == 0 && a.iter_ts) a.iter_ts[0] = wall_clock64();
    }
    __syncthreads();
    if (!s_ok) {
        if (tid == 0) { a.cold->exit.status = kPsStatusAborted; __threadfence_system(); }
        return;
    }
    const uint32_t Cres = s_Cres, nslots = s_nslots;
    if (a.blk_ts && tid == 0) { a.blk_ts[((size_t)blockIdx.x * 128 + 0) * 8 + 7] = C; a.blk_ts[((size_t)blockIdx.x * 128 + 1) * 8 + 7] = cstart[C]; a.blk_ts[((size_t)blockIdx.x * 128 + 2) * 8 + 7] = nslots; }
    const uint32_t row = (uint32_t)lane >> 4, l16 = (uint32_t)lane & 15u;
    // running sums of cluster k = tid (kmeans.rs: the members of every cluster, as sums): registers, the same in every block
    unsigned long long run[5] = {0, 0, 0, 0, 0};
    uint32_t moved = 0;
    unsigned long long evals = 0;
    uint32_t j = 0;              // the iteration whose assign step runs
    uint32_t nS = K;             // centroids the last update moved
    unsigned long long reseeds_total = 0, evals_total = 0;
#ifdef CNIIC_PS_PHASES
    long long t_ph = clock64();
    unsigned long long ph_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        const bool first = j == 0;
        const bool skip_mode = !first && a.max_skip && nS <= a.max_skip;
        PS_PHASE(7);
        if (!skip_mode) {
            // ============================================================= FULL schedule: every cell's candidates anew
            ps_build_lists(tab, K, nslots, s_ssup, s_nS, s_lpiv, Sent, wid, lane);
            if (a.blk_ts && tid == 0 && j < 128) a.blk_ts[((size_t)blockIdx.x * 128 + j) * 8 + 4] = wall_clock64();
            PS_PHASE(0);
            for (uint32_t i0 = (uint32_t)wid * 4; i0 < C; i0 += kPsWaves * 4) {   // a row of 16 lanes per cell
                const uint32_t i = i0 + row;
                if (i < C) {
                    const uint32_t c = ccell[i], slot = cslot[i];
                    const uint32_t nl = slot != 255u ? s_nS[slot] : 0xffffffffu;
                    uint32_t *rec = recs + (size_t)kPsRecWords * i;
                    const uint32_t w_old = l16 < 8 ? rec[2 + l16] : 0u;   // the mask of the iteration before (whose sweep, or whose reasons for none, left every label inside it)
                    bool needs;
                    if (nl <= kPsScap) {
                        const uint32_t pid = ps_row_build_list(tab, Sent + slot * kPsScap, nl, c, l16, rec);
                        needs = ps_row_finish(rec, l16, pid, s_lpiv[slot], false);
                    } else {
                        const uint32_t pid = ps_row_build_table(tab, K, c, l16, rec);
                        needs = ps_row_finish(rec, l16, pid, 0u, true);
                    }
                    // A cell none of whose candidates -- old or new -- moved repeats every decision: its points' labels are among the old
                    // candidates, every centroid that can win now is among the new ones, and all of them stand where they stood when the
                    // labels were given.  (Half of the swept cells of iterations 15 .. 25, where a quarter to a half of the centroids move.)
                    if (!first && a.clean_skip) {
                        const uint32_t mm32 = l16 < 8 ? (uint32_t)(s_mm[l16 >> 1] >> (32u * (l16 & 1u))) : 0u;
                        const uint32_t w_new = l16 < 8 ? rec[2 + l16] : 0u;
                        needs = needs && row_max(((w_old | w_new) & mm32) ? 1u : 0u) != 0u;
                    }
                    if (l16 == 0 && (needs || first)) queue[atomicAdd(&s_qn, 1u)] = (uint16_t)i;
                }
            }
        } else {
            // ============================================================= SKIP schedule (at most max_skip centroids moved)
            // A cell none of whose candidates moved and whose pivot still dominates every moved centroid repeats all its decisions.  The
            // test, a row of 16 lanes per cell; what fails it goes on the list and is worked on by a whole wave below.
            uint32_t mk[4], mc[4];   // the lane's share of the moved centroids: ids and colours
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t jj = l16 + 16u * t;
                mk[t] = jj < nS ? s_mlist[jj] : 0xffffffffu;
                mc[t] = jj < nS ? s_mcol[jj] : 0u;
            }
            for (uint32_t i0 = (uint32_t)wid * 4; i0 < C; i0 += kPsWaves * 4) {
                const uint32_t i = i0 + row;
                const bool ok = i < C;
                const uint32_t *rec = recs + (size_t)kPsRecWords * (ok ? i : 0u);
                const uint32_t r1 = rec[1], lp = r1 >> 24, c = ccell[ok ? i : 0u];
                Dominance dm;
                dm.set(cell_box(c), (1 << kCellShift) - 1, rec[0]);
                bool dv = !(r1 & kPsComplete) && ((s_mm[lp >> 6] >> (lp & 63)) & 1ull) != 0ull;   // the list the mask came from lost its pivot
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (mk[t] != 0xffffffffu) {
                        const bool in = ((rec[2 + (mk[t] >> 5)] >> (mk[t] & 31)) & 1u) != 0;
                        dv = dv | in | (dm.worst(mc[t]) >= 0);   // a moved centroid matters if it was a candidate or is no longer dominated by the pivot
                    }
                }
                const unsigned long long bm = __ballot(dv && ok);
                if (l16 == 0 && ((bm >> (16 * row)) & 0xffffull)) queue[atomicAdd(&s_qn, 1u)] = (uint16_t)i;
            }
        }
        __syncthreads();   // the work list is complete
        if (a.blk_ts && tid == 0 && j < 128) { a.blk_ts[((size_t)blockIdx.x * 128 + j) * 8 + 5] = wall_clock64(); a.blk_ts[((size_t)blockIdx.x * 128 + j) * 8 + 6] = s_qn; }
        PS_PHASE(1);
        // ------------------------------------------------------------- sweeps: list entry wid, wid + 16, ... is this wave's
        {
            const uint32_t qn = s_qn;
            const bool agg = !first && j <= a.agg_iters;
            for (;;) {   // the waves draw cells from the list (what a cell costs depends on how many of its points move)
                uint32_t qi = 0;
                if (lane == 0) qi = atomicAdd(&s_qhead, 1u);
                qi = (uint32_t)__builtin_amdgcn_readfirstlane((int)qi);
                if (qi >= qn) break;
                const uint32_t i = queue[qi];
                const uint32_t s = cstart[i], e = cstart[i + 1], gs = gstart[i], c = ccell[i];
                uint32_t rw = lane < (int)kPsRecWords ? recs[(size_t)kPsRecWords * i + lane] : 0u;
                if (skip_mode) {
                    // ---- a dirty cell of the skip schedule: its mask brought up to date by the whole wave.  A mask whose pivot (and, unless
                    // complete, whose list's pivot) has not moved keeps every verdict on the centroids that have not moved: the moved ones
                    // are tested, one per lane.  Otherwise: from the whole table with a fresh pivot, position = id, the ballots ARE the mask.
                    constexpr int32_t ext = (1 << kCellShift) - 1;
                    const uint32_t r1o = (uint32_t)__builtin_amdgcn_readlane((int)rw, 1), pido = (r1o >> 16) & 255u, lpo = r1o >> 24;
                    const bool keep = ((s_mm[pido >> 6] >> (pido & 63)) & 1ull) == 0ull && ((r1o & kPsComplete) || ((s_mm[lpo >> 6] >> (lpo & 63)) & 1ull) == 0ull);
                    const CellBox bx = cell_box(c);
                    unsigned long long um[4];
                    uint32_t npid = pido, npv = (uint32_t)__builtin_amdgcn_readlane((int)rw, 0);
                    bool complete = (r1o & kPsComplete) != 0;
                    if (keep) {
                        Dominance dm;
                        dm.set(bx, ext, npv);
                        const uint32_t k1 = (uint32_t)lane < nS ? s_mlist[lane] : 0xffffffffu;
                        const uint32_t ck1 = (uint32_t)lane < nS ? s_mcol[lane] : 0u;
                        unsigned long long f1 = __ballot(k1 != 0xffffffffu && dm.worst(ck1) >= 0);
#pragma unroll
                        for (int w = 0; w < 4; w++)
                            um[w] = (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)rw, 3 + 2 * w) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)rw, 2 + 2 * w)) & ~s_mm[w];
                        while (f1) {
                            const int l = __builtin_ctzll(f1);
                            f1 &= f1 - 1;
                            const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)k1, l);
                            const unsigned long long bit = 1ull << (k & 63);
                            const uint32_t w = (k >> 6) & 3;
                            um[0] |= w == 0 ? bit : 0ull; um[1] |= w == 1 ? bit : 0ull; um[2] |= w == 2 ? bit : 0ull; um[3] |= w == 3 ? bit : 0ull;
                        }
                    } else {
                        npid = nearest_to_centre(tab, K, bx, ext, lane);
                        npv = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[npid].x);
                        Dominance dm;
                        dm.set(bx, ext, npv);
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            const uint32_t k = 64 * w + lane;
                            um[w] = __ballot(k < K && dm.worst(tab[k < K ? k : 0].x) >= 0);
                        }
                        complete = true;
                    }
#pragma unroll
                    for (int w = 0; w < 4; w++) um[w] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(um[w] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)um[w]);
                    // up to four candidate ids as bytes (scalar unit), the record's words rebuilt in the lanes that hold them
                    uint32_t c4 = 0, got = 0;
                    {
                        unsigned long long t4[4] = {um[0], um[1], um[2], um[3]};
#pragma unroll
                        for (int w = 0; w < 4; w++)
                            while (t4[w] && got < 4) { c4 |= (64u * w + (uint32_t)__builtin_ctzll(t4[w])) << (8 * got); t4[w] &= t4[w] - 1; got++; }
                    }
                    const uint32_t r1n = (r1o & kPsUlMask) | (npid << 16) | (complete ? kPsComplete : (r1o & 0xff000000u));
                    uint32_t nw = rw;
#pragma unroll
                    for (int t = 0; t < 8; t++)
                        if (lane == 2 + t) nw = (uint32_t)(um[t >> 1] >> (32 * (t & 1)));
                    if (lane == 0) nw = npv;
                    if (lane == 1) nw = r1n;
                    if (lane == 10) nw = c4;
                    rw = nw;
                    if (lane < (int)kPsRecWords) recs[(size_t)kPsRecWords * i + lane] = rw;
                    const uint32_t ncd = (uint32_t)(__popcll(um[0]) + __popcll(um[1]) + __popcll(um[2]) + __popcll(um[3]));
                    if (ncd == 1 && (r1o & kPsUlMask) == (c4 & 255u)) continue;   // one candidate, and every point carries it: nothing can move
                }
                uint32_t wd[kSweep];
                if (i < Cres) {
#pragma unroll
                    for (int u = 0; u < kSweep; u++) { const uint32_t idx = s + u * 64 + lane; wd[u] = idx < e ? pts[idx] : 0u; }
                } else {
#pragma unroll
                    for (int u = 0; u < kSweep; u++) { const uint32_t idx = s + u * 64 + lane; wd[u] = idx < e ? a.pk[gs + (idx - s)] : 0u; }
                }
                PS_PHASE(2);
                PS_COUNT(8, 1);
                unsigned long long nm[4];
#pragma unroll
                for (int w = 0; w < 4; w++)
                    nm[w] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)rw, 3 + 2 * w) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)rw, 2 + 2 * w);
                const uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)rw, 1), cand4 = (uint32_t)__builtin_amdgcn_readlane((int)rw, 10);
                const uint32_t ncand = (uint32_t)(__popcll(nm[0]) + __popcll(nm[1]) + __popcll(nm[2]) + __popcll(nm[3]));
                const uint32_t cbk = cell_base_key(c);
                const bool res = i < Cres;
                const uint32_t *cwc = a.cweight + gs;
                uint32_t *pkc = a.pk + gs;
                uint32_t *rec = recs + (size_t)kPsRecWords * i;
                bool uniform = !first;   // every sweep of the cell left through the one-candidate exit: all its points carry that candidate
                PS_PHASE(3);
                for (uint32_t base = s; base < e; base += 64 * kSweep) {
                    uint32_t w2[kSweep] = {0, 0, 0, 0};
                    if (base + 64 * kSweep < e) {   // (a cell of more than 256 points: its next sweep's words now)
                        if (res) {
#pragma unroll
                            for (int u = 0; u < kSweep; u++) { const uint32_t idx = base + 64 * kSweep + u * 64 + lane; w2[u] = idx < e ? pts[idx] : 0u; }
                        } else {
#pragma unroll
                            for (int u = 0; u < kSweep; u++) { const uint32_t idx = base + 64 * kSweep + u * 64 + lane; w2[u] = idx < e ? pkc[idx - s] : 0u; }
                        }
                    }
                    PS_PHASE(4);
                    if (res) {
                        auto st = [&](uint32_t idx, uint32_t l) { reinterpret_cast<uint8_t *>(pts)[4 * idx + 3] = (uint8_t)l; };
                        if (first) ps_sweep_first(wd, base, e, s, lane, nm, ncand, cand4, tab, K, cbk, cwc, acc, moved, st);
                        else uniform = ps_sweep(wd, base, e, s, lane, nm, ncand, cand4, tab, K, cbk, cwc, acc, moved, agg, st PS_PROF_ARGS) && uniform;
                    } else {
                        auto st = [&](uint32_t idx, uint32_t l) { reinterpret_cast<uint8_t *>(pkc)[4 * (size_t)(idx - s) + 3] = (uint8_t)l; };
                        if (first) ps_sweep_first(wd, base, e, s, lane, nm, ncand, cand4, tab, K, cbk, cwc, acc, moved, st);
                        else uniform = ps_sweep(wd, base, e, s, lane, nm, ncand, cand4, tab, K, cbk, cwc, acc, moved, agg, st PS_PROF_ARGS) && uniform;
                    }
                    PS_PHASE(5);
#pragma unroll
                    for (int u = 0; u < kSweep; u++) wd[u] = w2[u];
                }
                if (lane == 0) rec[1] = (r1 & ~kPsUlMask) | (uniform && ncand == 1 ? cand4 & 255u : kPsUlMask);
                evals += (unsigned long long)(e - s) * (ncand + 1);
                PS_PHASE(6);
            }
        }
        PS_PHASE(9);
        // ------------------------------------------------------------- this iteration's deltas leave the block
        moved = wave_reduce_sum(moved);
        if (lane == 0) {
            if (moved) atomicAdd(&s_moved, moved);
            if (evals) atomicAdd(&s_evals, evals);
        }
        moved = 0; evals = 0;
        __syncthreads();
        if (a.blk_ts && tid == 0 && j < 128) a.blk_ts[((size_t)blockIdx.x * 128 + j) * 8 + 0] = wall_clock64();
        unsigned long long *Pcur = a.partials + (size_t)(j % 3) * kPsPartWords;
        for (uint32_t i = tid; i < 5 * K; i += kPsThreads) {
            const unsigned long long v = acc[i];
            if (v) { __hip_atomic_fetch_add(&Pcur[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc[i] = 0ull; }
        }
        if (tid == 0) {
            if (s_moved) __hip_atomic_fetch_add(&Pcur[5 * (size_t)K], (unsigned long long)s_moved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s_evals) __hip_atomic_fetch_add(&Pcur[5 * (size_t)K + 1], s_evals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_moved = 0; s_evals = 0; s_qn = 0; s_qhead = 0; s_nmoved = 0; s_reseed = 0; s_active = 0; s_mm[0] = s_mm[1] = s_mm[2] = s_mm[3] = 0ull;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's atomics have been performed before its block arrives
        __syncthreads();
        if (a.blk_ts && tid == 0 && j < 128) a.blk_ts[((size_t)blockIdx.x * 128 + j) * 8 + 1] = wall_clock64();
        if (tid == 0) {
            bool ok = ps_barrier_xcd(a.bar, ps_xcc_id(), s_nx, s_nxcd, a.timeout_ticks);
#ifdef CNIIC_TESTING
            if (a.test_abort_at && j + 1 == a.test_abort_at) { ps_st(&a.bar->abort_.v, 1u); ok = false; }   // (fault injection: CNIIC_TEST_PS_ABORT_AT)
#endif
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) {
            if (tid == 0) { a.cold->exit.status = kPsStatusAborted; __threadfence_system(); }
            return;
        }
        if (a.blk_ts && tid == 0 && j < 128) a.blk_ts[((size_t)blockIdx.x * 128 + j) * 8 + 2] = wall_clock64();
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
                const PsCold *cd = a.cold;
                const uint64_t ri = reseed_index(cd->seed, j - 1, k, cd->U);  // fake_clone of the stolen point
                const GIdx gx = cd->gx;
                ck = gx.bits ? gidx_select(gx, ri) : cd->keys[ri];
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
                if (pos < kMaxMovedSkip) { s_mlist[pos] = k; s_mcol[pos] = ck; }
                atomicOr(&s_mm[(k >> 6) & 3], 1ull << (k & 63));
            }
        } else if (tid == kPsThreads - 1) {
            s_changed = ps_aread(&Pcur[5 * (size_t)K]);
            s_pev = ps_aread(&Pcur[5 * (size_t)K + 1]);
        }
        __syncthreads();
        if (a.blk_ts && tid == 0 && j <= 128) a.blk_ts[((size_t)blockIdx.x * 128 + j - 1) * 8 + 3] = wall_clock64();
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
                PsExit *x = &a.cold->exit;
                x->iter = j; x->moved_last = changed; x->reseeds = reseeds_total; x->active = s_active; x->pair_evals = evals_total;
            }
        }
        if (fin) break;   // converged (kmeans.rs:26-32) or the iteration cap: every block sees the same sums and leaves together
    }
#ifdef CNIIC_PS_PHASES
    if (lane == 0)
        for (int i = 0; i < 16; i++)
            if (ph_[i]) atomicAdd(&g_ps_phase[i], ph_[i]);
#endif
    // ----------------------------------------------------------------- results
    if (blockIdx.x == 0 && tid < K) {
        const PsCold *cd = a.cold;
        const uint2 cc = tab[tid];
        cd->cent_g[tid] = cc.x;
        cd->cconst_g[tid] = cc;
        cd->members_out[tid] = run[4];
        cd->wsum_out[tid] = run[3];
    }
    for (uint32_t i = wid; i < C; i += kPsWaves) {   // the labels of the block's points, a wave per cell
        const uint32_t s = cstart[i], n = cstart[i + 1] - s, gs = gstart[i];
        if (i < Cres) { for (uint32_t t = lane; t < n; t += 64) a.labels[gs + t] = (uint8_t)(255u - (pts[s + t] >> 24)); }
        else { for (uint32_t t = lane; t < n; t += 64) a.labels[gs + t] = (uint8_t)(255u - (a.pk[gs + t] >> 24)); }
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        if (tid == 0) { __threadfence_system(); a.cold->exit.status = kPsStatusDone; __threadfence_system(); }
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
// a launch that ended: what the context learns from it (a grid that could not get the CUs in time costs seconds: do not try again at once)
static void ps_learn(Ctx *c, uint32_t status, bool injected) {
    if (status == kPsStatusDone) { c->ps_backoff = 0; c->ps_backoff_left = 0; return; }
    if (status != kPsStatusAborted || injected) return;   // (a range that does not fit is found out before any block waits; the tests' aborts are no news)
    c->ps_backoff = std::min<uint32_t>(256u, std::max<uint32_t>(4u, 2u * c->ps_backoff));
    c->ps_backoff_left = c->ps_backoff;
}

int ps_prepare(KmRgbwState *s) {
    Ctx *c = s->c;
    if (c->ps_backoff_left && !test_env("CNIIC_KM_PS_REQUIRE")) { c->ps_backoff_left--; return CNIIC_OK; }   // (see ps_learn: the launches meanwhile; the tests' REQUIRE always tries)
    const uint32_t G = ps_grid_for(c, s->U);
    if (!G) return CNIIC_OK;   // (no device properties: the classic loop)
    static bool attr_set[16] = {};
    if (c->device >= 0 && c->device < 16 && !attr_set[c->device]) {
        CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_rgbw_persist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPsDynBytes));
        attr_set[c->device] = true;
    }
    const uint64_t o_bar = 0, o_part = (sizeof(PsBar) + 255) & ~255ull, o_fail = o_part + ((3 * (uint64_t)kPsPartWords * 8 + 255) & ~255ull);
    const uint64_t o_rng = o_fail + 256, total = o_rng + ((uint64_t)G * kPsChunks + 1) * 4;
    CNIIC_HIP_TRY(c, s->ps_arena.alloc(total));
    CNIIC_HIP_TRY(c, hipMemsetAsync(s->ps_arena.p, 0, o_rng, c->stream));
    CNIIC_HIP_TRY(c, s->ps_pk.alloc(std::max<uint64_t>(s->U, 1) * 4));
    s->ps_blocks = G;
    s->ps_o_part = o_part; s->ps_o_fail = o_fail; s->ps_o_rng = o_rng;
    (void)o_bar;
    uint8_t *a = s->ps_arena.as<uint8_t>();
    uint32_t budget = kPsDynBytes;   // (the tests shrink it: CNIIC_TEST_PS_LDS_BYTES leaves most cells' points in memory)
    if (const char *e = test_env("CNIIC_TEST_PS_LDS_BYTES")) budget = std::min<uint32_t>(kPsDynBytes, (uint32_t)atoi(e));
    s->ps_budget = budget;
    hipLaunchKernelGGL(k_ps_ranges, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)s->ne_cost.as<uint32_t>(), (const uint32_t *)s->ne_count.as<uint32_t>(), G, budget,
                       reinterpret_cast<uint32_t *>(a + o_rng), reinterpret_cast<uint32_t *>(a + o_fail));
    CNIIC_HIP_TRY(c, hipGetLastError());
    s->ps = true;
    return CNIIC_OK;
}

// The whole loop as one launch.  *ran = false: not tried (the CUs are promised to another persistent launch of this process) or given
// up without harm (the grid was not resident together in time; a block's range does not fit): the caller runs the classic loop, whose
// inputs are untouched.
int km_rgbw_run_persistent(KmRgbwState *s, bool *ran, bool may_defer) {
    Ctx *c = s->c;
    *ran = false;
    if (!s->ps || s->ps_tried) return CNIIC_OK;
    s->ps_tried = true;   // (a second km_rgbw_run on the same state continues classically: the persistent launch starts from the initial assignment)
    const int dev = c->device >= 0 && c->device < 16 ? c->device : 0, G = (int)s->ps_blocks;
    if (g_ps_cus_in_use[dev].fetch_add(G) + G > ps_cu_count(c->device)) { g_ps_cus_in_use[dev].fetch_sub(G); return CNIIC_OK; }
    std::shared_ptr<void> release(nullptr, [dev, G](void *) { g_ps_cus_in_use[dev].fetch_sub(G); });   // (given back when this goes -- or, deferred, when the state's copy does)
    if (!c->pinned_ps) CNIIC_HIP_TRY(c, hipHostMalloc(&c->pinned_ps, 1024, hipHostMallocDefault));
    static_assert(sizeof(PsCold) <= 1024, "the pinned block holds a PsCold");
    PsCold *cold = static_cast<PsCold *>(c->pinned_ps);
    memset(cold, 0, sizeof *cold);
    PsExit *xh = &cold->exit;
    uint8_t *ar = s->ps_arena.as<uint8_t>();
    DevBuf ts;
    const bool want_ts = s->profile || test_env("CNIIC_KM_PS_TRACE");
    if (want_ts) { CNIIC_HIP_TRY(c, ts.alloc((uint64_t)kPsTsCap * 8)); CNIIC_HIP_TRY(c, hipMemsetAsync(ts.p, 0, (uint64_t)kPsTsCap * 8, c->stream)); }
    PsArgs a{};
    a.ckeys = s->ckeys.as<uint32_t>(); a.cweight = s->cweight.as<uint32_t>(); a.labels = s->labels.as<uint8_t>();
    a.ne_cell = s->ne_cell.as<uint32_t>(); a.ne_start = s->ne_start.as<uint32_t>();
    a.cb = reinterpret_cast<const uint32_t *>(ar + s->ps_o_rng); a.ranges_fail = reinterpret_cast<const uint32_t *>(ar + s->ps_o_fail);
    a.lds_budget = s->ps_budget;
    a.pk = s->ps_pk.as<uint32_t>(); a.cconst0 = s->cconst.as<uint2>();
    a.partials = reinterpret_cast<unsigned long long *>(ar + s->ps_o_part); a.bar = reinterpret_cast<PsBar *>(ar);
    cold->cconst_g = s->cconst.as<uint2>(); cold->cent_g = s->cent.as<uint32_t>(); cold->members_out = s->members_last.as<uint64_t>(); cold->wsum_out = s->wsum_last.as<uint64_t>();
    cold->keys = s->keys; cold->gx = s->gidx; cold->seed = s->seed; cold->U = s->gidx.bits ? s->gidx.U : s->U;
    a.st_rw = s->dstate.as<KmDevState>(); a.cold = cold; a.max_iters = s->max_iters;
    a.K = s->K; a.max_skip = s->no_skip ? 0u : s->max_skip; a.agg_iters = s->agg_launches;
    a.test_abort_at = 0;
    if (const char *e = test_env("CNIIC_TEST_PS_ABORT_AT")) a.test_abort_at = (uint32_t)atoi(e);
    a.clean_skip = 1;
    if (const char *e = test_env("CNIIC_KM_PS_CLEANSKIP")) a.clean_skip = atoi(e) ? 1u : 0u;
    uint64_t tmo_ms = 2000;
    if (const char *e = test_env("CNIIC_KM_PS_TIMEOUT_MS")) tmo_ms = (uint64_t)atoll(e);
    a.timeout_ticks = tmo_ms * 100000ull;   // 100 MHz
    a.iter_ts = want_ts ? ts.as<unsigned long long>() : nullptr;
    DevBuf bts;
    const char *bt_path = test_env("CNIIC_KM_PS_BLOCK_TRACE");
    if (bt_path) { CNIIC_HIP_TRY(c, bts.alloc((uint64_t)G * 128 * 8 * 8)); CNIIC_HIP_TRY(c, hipMemsetAsync(bts.p, 0, (uint64_t)G * 128 * 8 * 8, c->stream)); a.blk_ts = bts.as<unsigned long long>(); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s->profile) { CNIIC_HIP_TRY(c, hipEventCreate(&e0)); CNIIC_HIP_TRY(c, hipEventCreate(&e1)); }
    hipExtLaunchKernelGGL(k_rgbw_persist, dim3((uint32_t)G), dim3(kPsThreads), kPsDynBytes, c->stream, e0, e1, 0, a);
    CNIIC_HIP_TRY(c, hipGetLastError());
    // The caller of an encode goes straight on to the labels of the pixels and to fetching the result block (cc_finish): nothing of
    // that needs the host to have seen how the launch ended, and the look cost the stream 40 us (a wait, a copy, the next launch's
    // way to the GPU).  The verdict is read where the result block is (km_rgbw_result_end -> km_rgbw_persistent_verdict).
    if (may_defer && !s->profile && !want_ts && !bt_path) {
        s->ps_pending = true;
        s->ps_hold = release;
        *ran = true;
        return CNIIC_OK;
    }
    CNIIC_HIP_TRY(c, ctx_spin_sync(c));
    const uint32_t status = xh->status;
    ps_learn(c, status, a.test_abort_at != 0);
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
        std::vector<unsigned long long> b((size_t)G * 128 * 8);
        CNIIC_HIP_TRY(c, hipMemcpy(b.data(), bts.p, b.size() * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(bt_path, "w")) {
            fprintf(f, "# block cells points lists:");
            for (int g = 0; g < G; g++) fprintf(f, " %d %llu %llu %llu;", g, b[((size_t)g * 128 + 0) * 8 + 7], b[((size_t)g * 128 + 1) * 8 + 7], b[((size_t)g * 128 + 2) * 8 + 7]);
            fprintf(f, "\n");
            fprintf(f, "block,iteration,assign_us,flush_us,barrier_us,update_us,lists_us,classify_us,sweeps_us,cells_swept\n");
            for (int g = 0; g < G; g++)
                for (uint64_t i = 1; i < xh->iter && i < 128; i++) {
                    const unsigned long long *r = &b[((size_t)g * 128 + i) * 8], start = b[((size_t)g * 128 + i - 1) * 8 + 3];
                    const unsigned long long l_end = r[4] ? r[4] : start;   // (skip schedule: no lists)
                    fprintf(f, "%d,%llu,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f,%llu\n", g, (unsigned long long)i, (r[0] - start) / 100.0, (r[1] - r[0]) / 100.0, (r[2] - r[1]) / 100.0, (r[3] - r[2]) / 100.0,
                            (l_end - start) / 100.0, (r[5] - l_end) / 100.0, (r[0] - r[5]) / 100.0, r[6]);
                }
            fclose(f);
        }
    }
#ifdef CNIIC_PS_PHASES
    {   // (a measuring build: run it with the per-iteration trace or KM_PROFILE, which wait for the launch)
        unsigned long long ph[16], zero[16] = {0};
        CNIIC_HIP_TRY(c, hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ps_phase), sizeof ph));
        CNIIC_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_ps_phase), zero, sizeof zero));
        const double w = (double)G * kPsWaves;
        fprintf(stderr, "persist phases (wave clocks per wave, %llu iterations): lists %.0f classify %.0f | draw+desc %.0f mask %.0f words %.0f sweep %.0f tail %.0f wait-for-others %.0f | sync+update %.0f | cells swept per wave %.1f"
                        " | in the sweeps: unpack+exit %.0f scores %.0f movers? %.0f labels %.0f booking %.0f\n",
                (unsigned long long)xh->iter, ph[0] / w, ph[1] / w, ph[2] / w, ph[3] / w, ph[4] / w, ph[5] / w, ph[6] / w, ph[9] / w, ph[7] / w, ph[8] / w, ph[10] / w, ph[11] / w, ph[12] / w, ph[13] / w, ph[14] / w);
    }
#endif
    *ran = true;
    return CNIIC_OK;
}

// How the deferred launch ended (the stream has been waited for).  Done: its statistics.  Given up: the loop of launches, now, from the
// inputs the persistent launch has not touched -- *retry tells the caller to fetch the result again.
int km_rgbw_persistent_verdict(KmRgbwState *s, bool *retry) {
    Ctx *c = s->c;
    *retry = false;
    s->ps_pending = false;
    s->ps_hold.reset();
    const PsExit *xh = &static_cast<const PsCold *>(c->pinned_ps)->exit;
    ps_learn(c, xh->status, test_env("CNIIC_TEST_PS_ABORT_AT") != nullptr);
    if (xh->status == kPsStatusDone) {
        s->run_stats.iterations = xh->iter; s->run_stats.moved_last = xh->moved_last; s->run_stats.empty_reseeds = xh->reseeds;
        s->run_stats.active = xh->active; s->run_stats.pair_evals = xh->pair_evals;
        s->run_stats_valid = true;
        return CNIIC_OK;
    }
    if (test_env("CNIIC_KM_PS_REQUIRE")) return c->fail(CNIIC_ERR_HIP, "kmeans_rgbw: the persistent launch ended with status %u (CNIIC_KM_PS_REQUIRE)", xh->status);
    CNIIC_TRY(km_rgbw_run(s, nullptr, false));   // (ps_tried is set: the launches)
    *retry = true;
    return CNIIC_OK;
}

}  // namespace cniic
