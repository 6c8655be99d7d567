// kmeans_rgbw.hpp -- what the two K-means-on-colours translation units share (k_kmeans_rgbw.hip: the launch-per-iteration
// kernels, the set-up and the host side; k_kmeans_persist.hip: the whole loop as ONE launch with the points resident in LDS):
// the state, the packed distance key, the exact cube-against-pivot pruning tests and the packed point word.
// (reference: src/kmeans.rs:21-143, 330-416 with Point = ColorCount, src/codec/clusterc.rs:68-114, src/geom.rs:8-24)
#pragma once
#include <cstdlib>
#include <memory>
#include <vector>

#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

constexpr uint32_t kBias = 1u << 18;  // > max |c|^2 = 195075
constexpr int kPPT = 8;               // colours per thread per sweep (brute kernel)
constexpr int kAssignThreads = 256;
constexpr uint32_t kMaxBlocks = 512;
#ifndef CNIIC_CELL_WAVES
#define CNIIC_CELL_WAVES 8
#endif
constexpr int kCellWaves = CNIIC_CELL_WAVES;                  // waves per block (narrow labels); they share the block's cell range
// LDS strips of a wave: the super-cell list (scap) and the cell's candidates (ccap).  K <= 256: half the table and the whole table -- nothing
// can overflow.  Larger K (u16 labels): both capped (round 4; whole-table strips left a block TWO waves at K = 2048, 0.27 ms an iteration against
// 0.07 at K = 1024): a longer list falls back to the table, a longer strip too (the table IS a candidate list: ascending ids, the same records).
__host__ __device__ constexpr uint32_t km_scap(uint32_t K) { return K <= 256 ? (K + 1) / 2 : (K + 1) / 2 < 512u ? (K + 1) / 2 : 512u; }
__host__ __device__ constexpr uint32_t km_ccap(uint32_t K) { return K <= 256 ? K : 256u; }
constexpr uint32_t kCellWavesBig = 12;                        // ... and in the settled part of a run (launch_assign)
constexpr uint32_t kCellBlocks = 256 * (kCellWaves == 4 ? 6 : kCellWaves == 6 ? 4 : kCellWaves == 8 ? 3 : 2);  // every block resident at once (LDS, K <= 256)
constexpr int kSweep = 4;                  // points per lane per sweep (cells kernel)
// The full schedule's split of the cells into ranges of equal estimated cost: a cell costs its candidate build plus one sweep
// per 256 points (a sweep of 3 points takes as long as one of 256) -- 2 : 1 measured on the headline encode (assign launches
// 1.87 ms with "1024 + points", 1.825 with 512 + 256 per sweep; 384 / 640 / 768 + 256: 1.87 / 1.84 / 1.85).
constexpr uint32_t kCellFixedCost = 512;  // per cell (CNIIC_CELL_COST)
constexpr uint32_t kCellSweepCost = 256;  // per sweep of 64 x kSweep points (CNIIC_CELL_SWEEP_COST; 0: the cell's points count instead)

struct KmRgbwState {
    Ctx *c = nullptr;
    uint64_t U = 0, lo = 0, hi = 0, seed = 0, max_iters = 0;
    uint32_t K = 0, Kpad = 0, idbits = 8, nblocks = 1;
    bool wide = false;   // u16 labels
    bool big = false;    // K > 2048: k_rgbw_assign_big (k_kmeans_wide.hip)
    bool cells = true;   // cell-pruned assign (default) vs brute force
    bool profile = false; // per-launch event timing of the assign kernel (CNIIC_KM_PROFILE)
    bool no_skip = false; // CNIIC_KM_NO_SKIP: always run the full schedule (A/B measurement)
    const uint32_t *keys = nullptr, *weight = nullptr;  // device, canonical order [0,U)
    DevBuf labels;       // canonical-order labels of [lo,hi) (brute path) / cell-major labels of [0,U) (cells path)
    DevBuf cconst, slabs, partials_own, dstate, cent, members_last, wsum_last;  // the last four are views into resblk
    DevBuf arena;        // every zero-initialised buffer of the state in one allocation (resblk, partials_own, running, fused_*, moved_list are views)
    DevBuf resblk;       // [KmDevState | cent u32[K] | members u64[K] | wsum u64[K]]: one copy brings the result to the host
    uint64_t res_cent = 0, res_members = 0, res_wsum = 0, res_bytes = 0;
    std::unique_ptr<LaggedPoll> lagged;  // cniic_cc_poll_lagged
    DevBuf ckeys, cweight, crank, cell_start, running, ne_cell, ne_start, ne_cost, ne_count, wfirst;
    DevBuf cell_rec, moved_list;  // skip schedule state
    uint32_t wfirst_waves = 0;    // != 0: wfirst has not been computed yet for this many waves (ensure_wave_ranges)
    GIdx gidx{nullptr, nullptr, 0};  // several GPUs: the points are this rank's share of gidx.U colours
    DevBuf fused_partials, fused_running, fused_cent;  // km_rgbw_run with the update folded into the assign launches (3 / 2 / 2 buffers)
    bool fused = false;
    cniic_kmeans_stats run_stats{};  // the statistics km_rgbw_run ended on
    bool run_stats_valid = false;
    uint32_t big_blocks_from = 10;  // launches from this one on run in blocks of kCellWavesBig waves (CNIIC_KM_BIG_BLOCKS_FROM; a huge value: never)
    uint32_t agg_launches = 3;  // launches 1 .. agg_launches book their movers round by round (CNIIC_KM_AGG_LAUNCHES)
    uint32_t max_skip = 64;  // (= kMaxMovedSkip) skip schedule when at most this many centroids moved (CNIIC_KM_MAXSKIP)
    long fail_at = -1;           // fault injection for the multi-rank tests (CNIIC_TEST_FAIL_AT_LAUNCH), read once as well
    uint32_t shard = 0, nshards = 1;
    uint64_t *partials = nullptr;  // device: 5K+2 words (per-iteration sums or deltas)
    // the loop as ONE launch (k_kmeans_persist.hip): K <= 256, one shard, no communicator
    bool ps = false, ps_tried = false;
    bool ps_pending = false;     // the persistent launch is in the stream and nobody has looked at how it ended yet (km_rgbw_run with may_defer): km_rgbw_result_end does
    std::shared_ptr<void> ps_hold;   // the CUs promised to that launch, given back when the verdict is in (or the state goes)
    uint32_t ps_blocks = 0;
    DevBuf ps_arena;             // [PsBar | 3 x kPsPartWords sums | fail word | PsRange[ps_blocks]]
    DevBuf ps_pk;                // packed words of the points that do not fit their block's LDS
    uint64_t ps_o_part = 0, ps_o_fail = 0, ps_o_rng = 0;
    uint32_t ps_budget = 0;      // bytes of dynamic LDS the block ranges were made for
};

// ---- k_kmeans_persist.hip: the LDS of a block and what the launch shares with its set-up
#ifndef CNIIC_PS_CHUNKS
#define CNIIC_PS_CHUNKS 24
#endif
constexpr uint32_t kPsChunks = CNIIC_PS_CHUNKS;         // chunks of the cell list a block owns (interleaved with the other blocks').  24 by measurement (16 / 20 / 24 / 28 / 32 / 48 on
                                                        // six images, profiles/r05_persist_chunks_probe.txt: finer chunks spread a moved centroid's dirty cells over more blocks; beyond ~28 a block's
                                                        // cells lie in more super-cells than it has shared lists)
#ifndef CNIIC_PS_SLOTS
#define CNIIC_PS_SLOTS 32
#endif
constexpr uint32_t kPsSlotsMax = CNIIC_PS_SLOTS;                    // shared super-cell lists of a block (cells of further super-cells build from the table)
constexpr uint32_t kPsScap = 96;                        // members a shared list holds (a longer list: its cells build from the table)
constexpr uint32_t kPsRecWords = 11;                    // a cell's record: pivot colour, common label | pivot id << 16 | candidates << 24 | kRecComplete, 8 mask words, four candidate ids
constexpr uint32_t kPsMaxCells = 2048;                  // cells a block may own (two per thread of its set-up)
constexpr uint32_t kPsOffCell = 5 * 256 * 8 + 256 * 8 + kPsSlotsMax * kPsScap * 4;   // accumulators, table, the shared lists (id << 24 | colour)
constexpr uint32_t kPsDynBytes = 160 * 1024 - 3072;     // the launch's dynamic LDS (the kernel's static variables take the rest)
constexpr uint32_t kPsPartWords = 5 * 256 + 8;          // u64 words of one buffer of sums (5K + 2, padded)
constexpr uint32_t kPsTsCap = 1024;                     // iterations whose end block 0 timestamps
// per cell (C rounded up to a multiple of 4): first point u32 (own numbering; + 4 words), first point u32 (cell-major), record, id u16, work list u16, list slot u8
__host__ __device__ constexpr uint32_t ps_cell_bytes(uint32_t C) { return 16u + ((C + 3u) & ~3u) * (4u + 4u + kPsRecWords * 4u + 2u + 2u + 1u); }
struct alignas(128) PsLine { uint32_t v; uint32_t pad[31]; };
struct PsBar { PsLine xcount[8], xgen[8], xblocks[8], top, topgen, count, gen, abort_; };   // every counter on a line of its own
constexpr uint32_t kPsStatusDone = 1, kPsStatusAborted = 2, kPsStatusRanges = 3;
struct PsExit { uint32_t status, pad; uint64_t iter, moved_last, reseeds, active, pair_evals; };   // pinned: how the launch ended
void launch_rgbw_assign_big(Ctx *c, const uint32_t *ckeys, const uint32_t *cweight, uint64_t U, uint32_t K, const uint32_t *cent, uint16_t *labels,
                            unsigned long long *partials, const KmDevState *st);
int ps_prepare(KmRgbwState *s);
int km_rgbw_run_persistent(KmRgbwState *s, bool *ran, bool may_defer);
int km_rgbw_persistent_verdict(KmRgbwState *s, bool *retry);   // for km_rgbw_result_end: how a deferred launch ended
constexpr uint32_t kAggMin = 16;  // points that must share the first mover's (old, new) pair for a round of aggregated booking to be worth it


__device__ __forceinline__ uint32_t dot4u8(uint32_t a, uint32_t b, uint32_t acc) {
    return __builtin_amdgcn_udot4(a, b, acc, false);
}

// (packed centroid key, const term) for cluster k
__device__ __forceinline__ uint2 make_cconst(uint32_t ckey, uint32_t k, uint32_t idbits) {
    uint32_t h = dot4u8(ckey, ckey, 0);
    uint32_t idmask = (1u << idbits) - 1;
    return make_uint2(ckey, ((kBias - h) << idbits) | (idmask - k));
}

struct CellBox { int32_t r0, g0, b0; };  // low corner of a cube of colours
__device__ __forceinline__ CellBox super_box(uint32_t sup) {
    return CellBox{(int32_t)((sup / (kSupersPerDim * kSupersPerDim)) << (kCellShift + 2)),
                   (int32_t)(((sup / kSupersPerDim) % kSupersPerDim) << (kCellShift + 2)),
                   (int32_t)((sup % kSupersPerDim) << (kCellShift + 2))};
}
__device__ __forceinline__ CellBox cell_box(uint32_t c) {
    const CellBox sb = super_box(c >> kSuperShift);
    return CellBox{sb.r0 + (int32_t)(((c >> 4) & 3) << kCellShift), sb.g0 + (int32_t)(((c >> 2) & 3) << kCellShift),
                   sb.b0 + (int32_t)((c & 3) << kCellShift)};
}

constexpr uint32_t kRecComplete = 0x80000000u;  // word 1 of a record: the mask holds EVERY centroid its pivot does not dominate (skip schedule, K <= 256)
__host__ __device__ constexpr uint32_t cell_rec_words(uint32_t MW) { return (2 + 2 * MW + 15) & ~15u; }  // u32 words of a cell's skip record (CellState below)
constexpr uint32_t kMaxMovedSkip = 64;  // skip schedule when at most this many centroids moved (one per lane of the test; measured on the
                                        // headline encode: 128 -> 1.92 ms of assign launches, 96 -> 1.90, 64 -> 1.88, 40 -> 1.88: above ~60 moved
                                        // centroids a third of the cells are dirty and dealing them round-robin costs more than the full schedule's ranges)

// Both tests of the pruning on PACKED colour bytes (round 4; until then three field extractions, three products and their sums each: 10 and 21
// vector instructions, a third of a candidate build).  r | g | b in bytes 2, 1, 0 of a key, byte 3 zero; a cube is aligned, so lo + ext <= 255.
__device__ __forceinline__ uint32_t pack_rgb(int32_t r, int32_t g, int32_t b) { return ((uint32_t)r << 16) | ((uint32_t)g << 8) | (uint32_t)b; }

// squared distance from colour key ck to the centre of the cube with low corner bx and side ext + 1:
// |v - c|^2 = v.v - 2 v.c + c.c, three byte dot products (two of them per candidate)
__device__ __forceinline__ uint32_t centre_dist(uint32_t ck, const CellBox &bx, int32_t ext) {
    const int32_t h = (ext + 1) >> 1;
    const uint32_t c = pack_rgb(bx.r0 + h, bx.g0 + h, bx.b0 + h);
    return dot4u8(ck, ck, dot4u8(c, c, 0)) - 2u * dot4u8(ck, c, 0);
}

// a pivot centroid p against one cube [lo, lo + ext]^3.  max over the cube of d(x, p) - d(x, v) -- v can be nearest (or tie) somewhere in the
// cube only if it is >= 0 -- is, per channel with d = p - v, max(d (v + p - 2 lo), d (v + p - 2 hi)) = d (v + p - 2 lo) + 2 ext max(0, -d);
// summed: (p.p - 2 p.lo - ext sum(p)) - v.v + v.lo + v.hi + ext sad(p, v)      [sum max(0, -d) = (sad(p, v) - sum(p) + sum(v)) / 2]
// -- a constant of the pivot, three byte dot products, one sum of absolute differences and one multiply-add per candidate.
struct Dominance {
    uint32_t ppk, lopk, hipk;
    int32_t cp, ext;
    __device__ __forceinline__ void set(const CellBox &bx, int32_t e, uint32_t pivot) {
        ppk = pivot & 0xffffffu;
        ext = e;
        lopk = pack_rgb(bx.r0, bx.g0, bx.b0);
        hipk = pack_rgb(bx.r0 + e, bx.g0 + e, bx.b0 + e);
        cp = (int32_t)dot4u8(ppk, ppk, 0) - 2 * (int32_t)dot4u8(ppk, lopk, 0) - e * (int32_t)dot4u8(ppk, 0x010101u, 0);
    }
    __device__ __forceinline__ int32_t worst(uint32_t ck) const {
        const uint32_t v = ck;   // (a colour key: byte 3 is zero)
        return cp - (int32_t)dot4u8(v, v, 0) + (int32_t)dot4u8(v, hipk, dot4u8(v, lopk, 0)) + __mul24(ext, (int32_t)__builtin_amdgcn_sad_u8(ppk, v, 0u));
    }
};

__device__ __forceinline__ uint32_t wave_all_min(uint32_t v) { return wave_reduce_min(v); }  // (DPP: the result is in every lane)

// position in `list` (n entries, ascending cluster id) of the centroid nearest the cube centre; lowest position on ties
// (ONE reduction over distance << 12 | position: a distance is below 3 * 255^2 < 2^18 and a list holds at most 4096 entries; until round 4 two
// reductions, first the distance, then the position among the lanes that had it -- a sixth of a candidate build's vector instructions)
__device__ __forceinline__ uint32_t nearest_to_centre(const uint2 *list, uint32_t n, const CellBox &bx, int32_t ext, int lane) {
    uint32_t bd = 0xfffffu, be = 4095u;
    for (uint32_t e = lane; e < n; e += 64) {
        const uint32_t d = centre_dist(list[e].x, bx, ext);
        if (d < bd) { bd = d; be = e; }
    }
    return wave_all_min((bd << 12) | be) & 4095u;
}

// S = the centroids of `tab` (ascending id) that can be nearest somewhere in super-cell `sup`; returns |S| >= 1
// set bits of a ballot below this lane: v_mbcnt (two instructions, and no 64-bit lane mask kept in registers)
__device__ __forceinline__ uint32_t lanes_below(unsigned long long bm) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
}

__device__ __forceinline__ uint32_t build_super(const uint2 *tab, uint32_t K, uint32_t sup, int lane, unsigned long long lt_mask,
                                                uint2 *S, uint32_t cap) {
    constexpr int32_t ext = (1 << (kCellShift + 2)) - 1;
    const CellBox bx = super_box(sup);
    Dominance dm;
    dm.set(bx, ext, tab[nearest_to_centre(tab, K, bx, ext, lane)].x);
    uint32_t n = 0;
    for (uint32_t k0 = 0; k0 < K; k0 += 64) {
        const uint32_t k = k0 + lane;
        uint2 cc = make_uint2(0u, 0u);
        bool keep = false;
        if (k < K) { cc = tab[k]; keep = dm.worst(cc.x) >= 0; }
        const unsigned long long bm = __ballot(keep);
        const uint32_t pos = n + lanes_below(bm);
        if (keep && pos < cap) S[pos] = cc;  // a longer list is not kept: the caller falls back to the whole table
        n += (uint32_t)__popcll(bm);
    }
    __builtin_amdgcn_wave_barrier();
    return n;
}

// ---- the packed point word of the resident points (k_kmeans_persist.hip)
__device__ __forceinline__ uint32_t pk_make(uint32_t key, uint32_t w, uint32_t label) {
    return (((key >> 16) & 7u) << 6) | (((key >> 8) & 7u) << 3) | (key & 7u) | (min(w, 255u) << 16) | (label << 24);
}
__device__ __forceinline__ uint32_t pk_key(uint32_t pw, uint32_t cell_base) {   // cell_base = the cell's low corner r0 << 16 | g0 << 8 | b0
    return cell_base | ((pw & 0x1c0u) << 10) | ((pw & 0x38u) << 5) | (pw & 7u);
}
__device__ __forceinline__ uint32_t cell_base_key(uint32_t c) {
    const CellBox b = cell_box(c);
    return ((uint32_t)b.r0 << 16) | ((uint32_t)b.g0 << 8) | (uint32_t)b.b0;
}

}  // namespace cniic
