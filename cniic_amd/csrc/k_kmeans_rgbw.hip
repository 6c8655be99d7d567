// k_kmeans_rgbw.hip -- kmeans::cluster::<ColorCount> on gfx950
// (reference: src/kmeans.rs:21-143, 330-416 with Point = ColorCount, src/codec/clusterc.rs:68-114,
//  distance src/geom.rs:8-24).
//
// Points are the image's DISTINCT colours with pixel-count weights (clusterc.rs:21-28), one
// packed u32 key + one u32 weight + one label byte each: 4 + 4 + 1 read + 1 written = 10 B per
// colour per iteration ("dedup form").
//
// Assign is exact Lloyd under the reference's rules (stay unless another centroid is STRICTLY
// closer, kmeans.rs:350-378; lowest id among equidistant minima).  Every comparison is between
// distances from the SAME point, so |p|^2 cancels and the kernels maximise
//     g_k = 2 p.c_k - |c_k|^2          (p.c_k = one v_dot4_u32_u8)
// packed with the cluster id into one u32, so arg-max + lowest-id tie-break is one v_max_u32:
//     key_k = ((g_k + BIAS) << IDBITS) | (IDMASK - k)  =  (dot << (IDBITS+1)) + const_k.
// MFMA is deliberately unused: the inner dimension is 3 and the arg-max dominates.
//
// Two assign kernels, identical results:
//   * k_rgbw_assign        brute force over all K centroids (3 VALU per pair; constants arrive by
//                          scalar loads).  VALU-bound.  Used by the single-step ABI and for A/B.
//   * k_rgbw_assign_cells  colour space is cut into 8x8x8 cells, grouped 4x4x4 into super-cells,
//                          and the points are kept in cell-major order.  For a cube B and a pivot
//                          centroid c*, d(p,c*) - d(p,c_k) = sum_dim (c*_d - k_d)(c*_d + k_d - 2 p_d)
//                          is linear in p: its maximum over B is a sum of per-axis maxima at the
//                          cube faces.  If that maximum is negative c_k cannot be nearest - or tied
//                          - for any colour of B and is dropped.  A wave runs the test once per
//                          super-cell over all K (list S, pivot = centroid nearest the cube centre)
//                          and once per cell over S only.  This is the reference's
//                          triangle-inequality pruning (kmeans.rs:355-370) applied to a box of
//                          points and one pivot instead of one point against a neighbour list; it
//                          is exact at every iteration (the reference's truncated lists are not).
//
// Centroid update is exact u64 integer arithmetic (clusterc.rs:92-105).  The cells path keeps
// RUNNING per-cluster sums and feeds them signed deltas from the points that moved (integer adds
// commute, so the result equals a full re-accumulation); per-block LDS accumulators (ds_add_u64),
// non-zero entries flushed with global u64 atomics.  Sums are independent of block count, launch
// order and GPU count.
#include <cstdlib>
#include <memory>
#include <vector>


#include <hip/hip_ext.h>

#include "kmeans_rgbw.hpp"

namespace cniic {


// ---------------------------------------------------------------- init (kmeans.rs:61-108)
__global__ void k_rgbw_init_cent(const uint32_t *__restrict__ keys, uint64_t U, uint32_t K, uint32_t Kpad, uint32_t idbits,
                                 uint2 *__restrict__ cconst, uint32_t *__restrict__ cent, GIdx gx, uint32_t *__restrict__ cent_copy,
                                 uint32_t *__restrict__ moved_list, const uint64_t *__restrict__ U_dev) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (U_dev) U = max(*U_dev, (uint64_t)K);  // the count is still on its way to the host (fewer points than clusters: refused there)
    if (k == 0 && moved_list) moved_list[0] = K;  // before the first update every centroid counts as moved
    if (k < K) {
        // init_centroids (kmeans.rs:101-108): first element of chunk k
        uint64_t ppc = U / K;
        uint64_t first = (k < K - 1) ? U - ((uint64_t)k + 1) * ppc : 0;
        uint32_t ck = gx.bits ? gidx_select(gx, first) : keys[first];
        cent[k] = ck;
        if (cent_copy) cent_copy[k] = ck;
        cconst[k] = make_cconst(ck, k, idbits);
    } else if (k < Kpad) {
        cconst[k] = make_uint2(0u, 0u);  // padding: key 0 never wins
    }
}

// labels[i - lo] = init label of canonical index rank[i] (or i itself when rank == nullptr)
template <typename LabelT>
__global__ void k_rgbw_init_labels(const uint32_t *__restrict__ rank, uint64_t U, uint64_t lo, uint64_t hi, uint32_t K,
                                   LabelT *__restrict__ labels) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride)
        labels[i - lo] = (LabelT)init_label(rank ? rank[i] : i, U, K);  // kmeans.rs:61-78
}

// ---------------------------------------------------------------- brute-force assign + full sums
// slab row layout (u64 words): [3k+d] sum of channel d * weight, [3K+k] sum of weights,
// [4K+k] member count, [5K] moved count, [5K+1] pair evaluations.
template <typename LabelT, int IDBITS>
__global__ __launch_bounds__(kAssignThreads) void k_rgbw_assign(
    const uint32_t *__restrict__ keys, const uint32_t *__restrict__ weight, uint64_t lo, uint64_t hi,
    uint32_t K, uint32_t Kpad, const uint2 *__restrict__ cconst, LabelT *__restrict__ labels,
    uint64_t *__restrict__ slabs, const KmDevState *__restrict__ st) {
    extern __shared__ __align__(16) unsigned long long lds[];  // [5K] accumulators, then uint2[K] table
    if (st->done) return;
    unsigned long long *acc = lds;
    uint2 *tab = reinterpret_cast<uint2 *>(lds + 5 * (size_t)K);
    for (uint32_t i = threadIdx.x; i < 5 * K; i += kAssignThreads) acc[i] = 0ull;
    for (uint32_t i = threadIdx.x; i < K; i += kAssignThreads) tab[i] = cconst[i];
    __syncthreads();

    constexpr uint32_t IDMASK = (1u << IDBITS) - 1;
    const uint64_t n = hi - lo;
    const uint64_t sweep = (uint64_t)gridDim.x * kAssignThreads * kPPT;
    uint32_t moved = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * kAssignThreads * kPPT; base < n; base += sweep) {
        uint32_t p[kPPT], best[kPPT];
        bool valid[kPPT];
#pragma unroll
        for (int j = 0; j < kPPT; j++) {
            uint64_t i = base + (uint64_t)j * kAssignThreads + threadIdx.x;
            valid[j] = i < n;
            p[j] = valid[j] ? keys[lo + i] : 0u;
            best[j] = 0u;
        }
        // brute-force sweep over the (padded) centroid table; k is wave-uniform -> scalar loads
#pragma unroll 4
        for (uint32_t k = 0; k < Kpad; k++) {
            const uint2 cc = cconst[k];
#pragma unroll
            for (int j = 0; j < kPPT; j++) {
                uint32_t key = (dot4u8(p[j], cc.x, 0) << (IDBITS + 1)) + cc.y;
                best[j] = max(best[j], key);
            }
        }
#pragma unroll
        for (int j = 0; j < kPPT; j++) {
            if (!valid[j]) continue;
            uint64_t i = base + (uint64_t)j * kAssignThreads + threadIdx.x;
            uint32_t cur = labels[i];
            uint2 cc = tab[cur];
            uint32_t kcur = (dot4u8(p[j], cc.x, 0) << (IDBITS + 1)) + cc.y;
            uint32_t nl = cur;
            if ((best[j] >> IDBITS) > (kcur >> IDBITS)) {  // strictly closer (kmeans.rs:375)
                nl = IDMASK - (best[j] & IDMASK);
                labels[i] = (LabelT)nl;
                moved++;
            }
            uint64_t w = weight[lo + i];
            uint32_t r = (p[j] >> 16) & 255, g = (p[j] >> 8) & 255, b = p[j] & 255;
            atomicAdd(&acc[3 * nl + 0], (unsigned long long)(r * w));  // clusterc.rs:92-98
            atomicAdd(&acc[3 * nl + 1], (unsigned long long)(g * w));
            atomicAdd(&acc[3 * nl + 2], (unsigned long long)(b * w));
            atomicAdd(&acc[3 * K + nl], (unsigned long long)w);
            atomicAdd(&acc[4 * K + nl], 1ull);
        }
    }
    moved = block_reduce_sum<kAssignThreads>(moved);
    __syncthreads();
    uint64_t *row = slabs + (size_t)blockIdx.x * (5 * (size_t)K + 2);
    for (uint32_t i = threadIdx.x; i < 5 * K; i += kAssignThreads) row[i] = acc[i];
    if (threadIdx.x == 0) {
        row[5 * (size_t)K] = moved;
        uint64_t per_block = 0;  // points this block visited x K
        for (uint64_t base = (uint64_t)blockIdx.x * kAssignThreads * kPPT; base < n; base += sweep)
            per_block += min((uint64_t)kAssignThreads * kPPT, n - base);
        row[5 * (size_t)K + 1] = per_block * K;
    }
}

// grid (ceil(W/64), R): each block sums a stripe of rows for 64 columns, then one atomic per column.
__global__ __launch_bounds__(256) void k_slab_reduce(const uint64_t *__restrict__ slabs, uint32_t nrows, uint32_t W,
                                                     uint64_t *__restrict__ partials,
                                                     const KmDevState *__restrict__ st) {
    if (st->done) return;
    __shared__ unsigned long long sh[4][64];
    const uint32_t col = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t rl = threadIdx.x >> 6;  // 0..3
    unsigned long long s = 0;
    if (col < W)
        for (uint32_t r = blockIdx.y * 4 + rl; r < nrows; r += gridDim.y * 4) s += slabs[(size_t)r * W + col];
    sh[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < W) {
        s = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        if (s) atomicAdd(reinterpret_cast<unsigned long long *>(partials) + col, s);
    }
}

// ---------------------------------------------------------------- cell-major point order
// wave-aggregated histogram of cell ids (sorted input -> one or two atomics per wave)
__global__ __launch_bounds__(256) void k_cell_count(const uint32_t *__restrict__ keys, uint64_t U,
                                                    uint32_t *__restrict__ cell_count) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < U; base += stride) {
        const uint64_t i = base + threadIdx.x;
        bool active = i < U;
        const uint32_t cell = active ? cell_of(keys[i]) : 0;
        for (;;) {
            unsigned long long m = __ballot(active);
            if (!m) break;
            const int leader = __ffsll((long long)m) - 1;
            const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)cell, leader);
            const bool same = active && cell == lc;
            const unsigned long long ms = __ballot(same);
            if (lane == leader) atomicAdd(&cell_count[lc], (uint32_t)__popcll(ms));
            active = active && !same;
        }
    }
}

// Exclusive scan of the cell counts -> cell_start[kNumCells + 1] (+ cursor copy), and the compacted list of
// NON-EMPTY cells: ne_cell[m] = cell id, ne_start[m] = its first position, ne_cost[m] = start + m * fixed_cost,
// ne_start[M] = U, *ne_count = M.  Blocks walk the compacted list, never the empty cells.
// Two launches of kNumCells / 64 one-wave blocks (a single 1024-thread block took 57 us: ~10^5 scattered store
// requests from one CU): per 64 cells (points, non-empty cells), then every wave adds up the groups before its own.
constexpr uint32_t kCellGroups = kNumCells / 64;
// (tot[kCellGroups + g]: the group's sweeps, ceil(points / 256) per cell -- the full schedule's cost model counts them)
__global__ __launch_bounds__(64) void k_cell_totals(const uint32_t *__restrict__ cell_count, uint2 *__restrict__ tot) {
    const uint32_t v = cell_count[blockIdx.x * 64 + threadIdx.x];
    const uint32_t sum = wave_reduce_sum(v), sweeps = wave_reduce_sum((v + 64 * kSweep - 1) / (64 * kSweep));
    const uint32_t ne = (uint32_t)__popcll(__ballot(v != 0));
    if (threadIdx.x == 0) { tot[blockIdx.x] = make_uint2(sum, ne); tot[kCellGroups + blockIdx.x] = make_uint2(sweeps, 0u); }
}
__global__ __launch_bounds__(64) void k_cell_scan(const uint32_t *__restrict__ cell_count, const uint2 *__restrict__ tot,
                                                  uint32_t *__restrict__ cell_start, uint32_t *__restrict__ cursor,
                                                  uint32_t *__restrict__ ne_cell, uint32_t *__restrict__ ne_start,
                                                  uint32_t *__restrict__ ne_cost, uint32_t *__restrict__ ne_count, uint32_t fixed_cost,
                                                  uint32_t sweep_cost) {
    // cost of a cell in the full schedule's split: fixed_cost + points if sweep_cost == 0, else fixed_cost + sweep_cost * sweeps
    const uint32_t lane = threadIdx.x, grp = blockIdx.x;
    uint32_t ps = 0, pn = 0, pw = 0;
    for (uint32_t g = lane; g < grp; g += 64) { const uint2 t = tot[g]; ps += t.x; pn += t.y; pw += tot[kCellGroups + g].x; }
    const uint32_t base = wave_reduce_sum(ps), nbase = wave_reduce_sum(pn), wbase = wave_reduce_sum(pw);
    const uint32_t cell = grp * 64 + lane, v = cell_count[cell], sw = (v + 64 * kSweep - 1) / (64 * kSweep);
    const uint32_t start = base + wave_inclusive_scan(v) - v, swb = wbase + wave_inclusive_scan(sw) - sw;
    cell_start[cell] = start;
    if (cursor) cursor[cell] = start;
    const unsigned long long nzm = __ballot(v != 0);
    if (v) {
        const uint32_t m = nbase + (uint32_t)__popcll(nzm & ((1ull << lane) - 1ull));
        ne_cell[m] = cell; ne_start[m] = start; ne_cost[m] = (sweep_cost ? swb * sweep_cost : start) + m * fixed_cost;
    }
    if (grp == kCellGroups - 1 && lane == 63) {
        const uint32_t U = start + v, M = nbase + (uint32_t)__popcll(nzm);
        cell_start[kNumCells] = U;
        ne_start[M] = U;
        ne_cost[M] = (sweep_cost ? (swb + sw) * sweep_cost : U) + M * fixed_cost;
        *ne_count = M;
    }
}

// ---- cell-major order straight from the dense colour table (codec path).  After the canonical
// compaction table[key] = rank + 1 for every colour that occurs.  One 512-thread block per cell walks the
// cell's colours t = (r_lo, g_lo, b_lo) in ascending order, 512 at a time.
constexpr uint32_t kCellColours = 1u << (3 * kCellShift);
__device__ __forceinline__ uint32_t cell_key(uint32_t cell, uint32_t t) {
    constexpr uint32_t lm = (1u << kCellShift) - 1;
    const CellBox bx = cell_box(cell);
    const uint32_t r = (uint32_t)bx.r0 | (t >> (2 * kCellShift));
    const uint32_t g = (uint32_t)bx.g0 | ((t >> kCellShift) & lm);
    const uint32_t b = (uint32_t)bx.b0 | (t & lm);
    return (r << 16) | (g << 8) | b;
}
__global__ __launch_bounds__(512) void k_cells_count_tbl(const uint32_t *__restrict__ table, uint32_t *__restrict__ cell_count) {
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < kCellColours; t += 512) mine += table[cell_key(blockIdx.x, t)] != 0;
    const uint32_t n = block_reduce_sum<512>(mine);
    if (threadIdx.x == 0) cell_count[blockIdx.x] = n;
}
template <typename LabelT>
__global__ __launch_bounds__(512) void k_cells_write_tbl(const uint32_t *__restrict__ table, const uint32_t *__restrict__ weight,
                                                         const uint32_t *__restrict__ cell_start, uint64_t U, uint32_t K,
                                                         uint32_t *__restrict__ ckeys, uint32_t *__restrict__ cweight,
                                                         uint32_t *__restrict__ crank, LabelT *__restrict__ labels, GIdx gx) {
    __shared__ uint32_t wsum[512 / 64];
    __shared__ uint32_t s_run;
    const uint32_t s = cell_start[blockIdx.x], e = cell_start[blockIdx.x + 1];
    if (s == e) return;
    if (threadIdx.x == 0) s_run = s;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < kCellColours; t0 += 512) {
        const uint32_t key = cell_key(blockIdx.x, t0 + threadIdx.x);
        const uint32_t v = table[key];
        const uint32_t run = s_run;
        const uint32_t pos = run + block_exclusive_scan<512>(v != 0, wsum);  // (its barriers: every thread has read s_run)
        if (v) {
            const uint32_t rank = v - 1;
            ckeys[pos] = key;
            cweight[pos] = weight[rank];
            crank[pos] = rank;
            labels[pos] = (LabelT)init_label(gx.bits ? gidx_rank(gx, key) : rank, U, K);  // init_assignment kmeans.rs:61-78
        }
        if (threadIdx.x == 511) s_run = pos + (v != 0);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_cell_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ weight,
                                                      uint64_t U, uint32_t *__restrict__ cursor,
                                                      uint32_t *__restrict__ ckeys, uint32_t *__restrict__ cweight,
                                                      uint32_t *__restrict__ crank) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < U; base += stride) {
        const uint64_t i = base + threadIdx.x;
        bool active = i < U;
        const uint32_t key = active ? keys[i] : 0;
        const uint32_t cell = cell_of(key);
        uint32_t pos = 0;
        for (;;) {
            unsigned long long m = __ballot(active);
            if (!m) break;
            const int leader = __ffsll((long long)m) - 1;
            const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)cell, leader);
            const bool same = active && cell == lc;
            const unsigned long long ms = __ballot(same);
            uint32_t b = 0;
            if (lane == leader) b = atomicAdd(&cursor[lc], (uint32_t)__popcll(ms));
            b = (uint32_t)__builtin_amdgcn_readlane((int)b, leader);
            if (same) pos = b + (uint32_t)__popcll(ms & ((1ull << lane) - 1));
            active = active && !same;
        }
        if (i < U) {
            ckeys[pos] = key;
            cweight[pos] = weight[i];
            crank[pos] = (uint32_t)i;
        }
    }
}

// ---------------------------------------------------------------- cell-pruned assign, delta sums
// One WAVE per cell: the 64 lanes bound all K centroids against the cell's cube (K/64 per lane),
// reduce min ub across the wave, compact the candidate list into the wave's own LDS strip with
// ballot prefixes, then sweep the cell's points 64 x kSweep at a time.  No block barrier inside
// the loop, so sparse and dense cells cost what they contain.  partials receives SIGNED deltas
// (two's complement u64) of the points that moved (full sums at iteration 0).
//
// Two schedules, same per-cell work:
//  * FULL (many centroids moved): cells are dealt to waves in contiguous runs of equal COST
//    (cost = kCellFixedCost + points, prefix-summed at setup; per-wave first cell precomputed by
//    k_wave_ranges).  A wave's points are contiguous in memory, so the loads of the next sweep -
//    of this cell or of the next one - are always in flight while the current sweep computes.
//  * SKIP (at most kMaxMovedSkip centroids changed in the last update - the long tail of Lloyd,
//    centroids are integers and stop moving one by one): every cell remembers T and the bitmask
//    of its candidate set.  If no moved centroid is in that mask and none of them comes within T
//    of the cell, then T, the candidate set and every candidate's value are what they were, so
//    every point of the cell repeats last iteration's decision: the cell is skipped after
//    |moved| bound evaluations.  Cells are dealt round-robin because the surviving work is
//    clustered around the centroids that moved.
__global__ __launch_bounds__(256) void k_wave_ranges(const uint32_t *__restrict__ ne_cost, const uint32_t *__restrict__ ne_count,
                                                     uint32_t G, uint32_t *__restrict__ wfirst) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > G) return;
    const uint32_t M = *ne_count;
    const uint64_t total = ne_cost[M];
    const uint64_t c_lo = total * g / G;
    uint32_t a = 0, b = M;  // first cell whose cost prefix is >= c_lo (prefix strictly increasing, ne_cost[0] = 0)
    while (a < b) { uint32_t mid = (a + b) >> 1; if (ne_cost[mid] < c_lo) a = mid + 1; else b = mid; }
    wfirst[g] = g == G ? M : a;
}


// candidates of cell c = the members of `list` that can be nearest somewhere in the cell's cube, into the
// wave's strip (ascending id).  The pivot's colour and the candidate bitmask (bit k <=> centroid k) are
// stored for the skip test.
// (ccap: the strip's capacity; more candidates than that are counted -- and recorded in the mask -- but not written: the caller sweeps
// against the whole table then, see km_ccap)
template <int IDBITS>
__device__ __forceinline__ uint32_t build_candidates(const uint2 *list, uint32_t n, uint32_t c, int lane, unsigned long long lt_mask,
                                                     uint2 *cand, unsigned long long *wmask, uint32_t *cell_rec, uint32_t m,
                                                     uint32_t MW, uint32_t ccap = 0xffffffffu) {
    constexpr uint32_t IDMASK = (1u << IDBITS) - 1;
    constexpr int32_t ext = (1 << kCellShift) - 1;
    const CellBox bx = cell_box(c);
    const uint2 pvc = list[nearest_to_centre(list, n, bx, ext, lane)];
    const uint32_t pv = pvc.x, pid = IDMASK - (pvc.y & IDMASK);
    Dominance dm;
    dm.set(bx, ext, pv);
    for (uint32_t i = lane; i < MW; i += 64) wmask[i] = 0ull;
    __builtin_amdgcn_wave_barrier();
    uint32_t ncand = 0;
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {
        const uint32_t e = e0 + lane;
        uint2 cc = make_uint2(0u, 0u);
        bool keep = false;
        if (e < n) { cc = list[e]; keep = dm.worst(cc.x) >= 0; }
        const unsigned long long bm = __ballot(keep);
        if (keep) {
            const uint32_t pos = ncand + lanes_below(bm);
            if (IDBITS == 8 || pos < ccap) cand[pos] = cc;
            const uint32_t k = IDMASK - (cc.y & IDMASK);
            atomicOr(&wmask[k >> 6], 1ull << (k & 63));
        }
        ncand += (uint32_t)__popcll(bm);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t *rec = cell_rec + (size_t)m * cell_rec_words(MW);
    // (the lane's index made afresh and pinned: or "cell_rec + 8 lane + 8" is computed once per launch and two registers hold it
    // across the cell loop -- or are spilled, and a scratch reload per cell sits in front of the loads in flight)
    uint32_t i_first = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(i_first));
    for (uint32_t i = i_first; i < MW; i += 64) *reinterpret_cast<unsigned long long *>(rec + 2 + 2 * i) = wmask[i];
    if (lane == 0) *reinterpret_cast<uint2 *>(rec) = make_uint2(pv, c | (pid << 16));  // (cell ids are 15 bits, cluster ids at most 11; bit 31: kRecComplete)
    return ncand;
}

// The 64 x kSweep points from `from` on (those below `end`; the others read as 0) as BUFFER loads: the range check is the hardware's, and
// one load is one vector instruction with an immediate offset -- as flat loads every one of the twelve cost an add, a compare, a select
// and a 64-bit address (about 60 of a sweep's 120 vector instructions; round 4).  from / end are wave-uniform.
#ifndef CNIIC_RGBW_BUFLOADS
#define CNIIC_RGBW_BUFLOADS 1
#endif
template <typename LabelT, bool NOWT = false>
__device__ __forceinline__ void load_points(const uint32_t *__restrict__ ckeys, const LabelT *__restrict__ labels, const uint32_t *__restrict__ cweight,
                                            uint32_t from, uint32_t end, int lane, uint32_t (&p)[kSweep], uint32_t (&cur)[kSweep], uint32_t (&wt)[kSweep]) {
#if CNIIC_RGBW_BUFLOADS
    // (the descriptors keep the ARRAY's base -- scalar registers the kernel holds anyway -- and end at `end`; the lane's offset carries `from`:
    // a descriptor per range start cost nine more live scalar registers and spilled)
    const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)from), e = (uint32_t)__builtin_amdgcn_readfirstlane((int)end);
    const uint32_t lim = e > f ? e : 0u;   // (an empty range: no record at all)
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(ckeys), 0, (int)(lim * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<LabelT *>(labels), 0, (int)(lim * (uint32_t)sizeof(LabelT)), 0x00020000);
    const uint32_t q0 = f + (uint32_t)lane;
    const uint32_t v4 = q0 * 4u, vl = q0 * (uint32_t)sizeof(LabelT);
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        p[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rk, (int)(v4 + u * 256u), 0, 0);
        if constexpr (sizeof(LabelT) == 1) cur[u] = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rl, (int)(vl + u * 64u), 0, 0);
        else cur[u] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rl, (int)(vl + u * 128u), 0, 0);
    }
    if constexpr (!NOWT) {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(cweight), 0, (int)(lim * 4u), 0x00020000);
#pragma unroll
        for (int u = 0; u < kSweep; u++) wt[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rw, (int)(v4 + u * 256u), 0, 0);
    } else {
#pragma unroll
        for (int u = 0; u < kSweep; u++) wt[u] = 0u;
    }
#else
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        const uint32_t q = from + u * 64 + lane;
        p[u] = q < end ? ckeys[q] : 0u;
        cur[u] = q < end ? (uint32_t)labels[q] : 0u;
        wt[u] = (!NOWT && q < end) ? cweight[q] : 0u;
    }
#endif
}

// one sweep: the 64 x kSweep points starting at `base` (those < e) against the candidate strip.
// AGG (the full schedule, where centroids still travel): when a centroid shifts, whole cells change hands -- every lane of the
// sweep moves from the same old cluster to the same new one, and ten LDS atomics per point on the same ten words run one lane
// at a time (~300 cycles per instruction, 12000 per sweep).  So, round by round: the (old, new) pair of the first point still to
// be booked, every point that shares it summed in the wave (DPP), one lane adds the totals.  A cell that changes hands is one
// round, a cell split between two new owners two; a handful of movers, and whatever is left after six rounds, go point by point.
// (Not in the skip schedule: few points move per sweep there.  The kernel must not spill a single register for this: a
// scratch segment costs every full-schedule launch 20 us)
template <typename LabelT, int IDBITS, bool ALLWRITE = false, bool AGG = false>  // ALLWRITE: the labels in memory are stale: write every one
__device__ __forceinline__ void sweep_points(const uint32_t (&p)[kSweep], const uint32_t (&cur)[kSweep], const uint32_t (&wt)[kSweep],
                                             uint32_t base, uint32_t e, int lane,
                                             const uint2 *cand, uint32_t ncand, const uint2 *tab, uint32_t K, bool first,
                                             LabelT *__restrict__ labels, unsigned long long *acc, uint32_t &moved,
                                             const uint32_t *__restrict__ cweight = nullptr) {
    constexpr uint32_t IDMASK = (1u << IDBITS) - 1;
    if (!first && !ALLWRITE && ncand == 1) {
        // More than half of the cells lie inside one cluster's region: ONE candidate, and every point already carries its label.
        // Nothing can move then (the lone candidate beats every other centroid for every colour of the cube): the sweep is over
        // after four compares and a ballot instead of ~90 instructions of scores and label checks.
        const uint32_t only = IDMASK - (cand[0].y & IDMASK);
        bool same = true;   // (bitwise: the short-circuit form compiled to four nested exec-masked branches)
#pragma unroll
        for (int u = 0; u < kSweep; u++) same = same & ((base + u * 64 + lane >= e) | (cur[u] == only));
        if (__ballot(!same) == 0ull) return;
    }
    uint32_t best[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; u++) best[u] = 0;
    for (uint32_t j = 0; j < ncand; j++) {
        const uint2 cc = cand[j];  // same address in every lane: LDS broadcast
#pragma unroll
        for (int u = 0; u < kSweep; u++) best[u] = max(best[u], (dot4u8(p[u], cc.x, 0) << (IDBITS + 1)) + cc.y);
    }
    if constexpr (AGG) {
        static_assert(kSweep == 4, "four slots per lane");
        uint32_t nl[kSweep];
        bool rem[kSweep], anym = false;
#pragma unroll
        for (int u = 0; u < kSweep; u++) {
            const uint32_t q = base + u * 64 + lane;
            nl[u] = cur[u];
            rem[u] = false;
            if (q < e) {
                const uint2 cc = tab[cur[u]];
                const uint32_t kcur = (dot4u8(p[u], cc.x, 0) << (IDBITS + 1)) + cc.y;
                rem[u] = (best[u] >> IDBITS) > (kcur >> IDBITS);  // strictly closer (kmeans.rs:375)
                if (rem[u]) { nl[u] = IDMASK - (best[u] & IDMASK); labels[q] = (LabelT)nl[u]; moved++; }
            }
            anym = anym || rem[u];
        }
        if (!__ballot(anym)) return;
        uint32_t nmv = 0;
#pragma unroll
        for (int u = 0; u < kSweep; u++) nmv += (uint32_t)__popcll(__ballot(rem[u]));
        if (nmv >= kAggMin) {
#pragma unroll 1
            for (int round = 0; round < 6; round++) {
                const unsigned long long b0 = __ballot(rem[0]), b1 = __ballot(rem[1]), b2 = __ballot(rem[2]), b3 = __ballot(rem[3]);
                if (!(b0 | b1 | b2 | b3)) return;
                uint32_t pn, po;
                if (b0) { const int l = __builtin_ctzll(b0); pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[0], l); po = (uint32_t)__builtin_amdgcn_readlane((int)cur[0], l); }
                else if (b1) { const int l = __builtin_ctzll(b1); pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[1], l); po = (uint32_t)__builtin_amdgcn_readlane((int)cur[1], l); }
                else if (b2) { const int l = __builtin_ctzll(b2); pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[2], l); po = (uint32_t)__builtin_amdgcn_readlane((int)cur[2], l); }
                else { const int l = __builtin_ctzll(b3); pn = (uint32_t)__builtin_amdgcn_readlane((int)nl[3], l); po = (uint32_t)__builtin_amdgcn_readlane((int)cur[3], l); }
                uint32_t cnt = 0;
                bool mt[kSweep];  // (lane masks in scalar registers)
#pragma unroll
                for (int u = 0; u < kSweep; u++) {
                    mt[u] = rem[u] && nl[u] == pn && cur[u] == po;
                    cnt += (uint32_t)__popcll(__ballot(mt[u]));
                    rem[u] = rem[u] && !mt[u];
                }
                if (cnt < kAggMin) {  // a round costs ~130 instructions: it pays for a pair that many points share, not for a handful --
                                      // these few book themselves, and so does everybody who is left
#pragma unroll
                    for (int u = 0; u < kSweep; u++) rem[u] = rem[u] || mt[u];
                    break;
                }
                // (one sum at a time: four 64-bit sums held together spill registers)
                auto book = [&](int shift, uint32_t mask, size_t at_new, size_t at_old) {
                    unsigned long long v = 0;
#pragma unroll
                    for (int u = 0; u < kSweep; u++)
                        if (mt[u]) v += (unsigned long long)(mask ? (p[u] >> shift) & mask : 1u) * wt[u];
                    v = wave_reduce_sum64(v);
                    if (lane == 0) { atomicAdd(&acc[at_new], v); atomicAdd(&acc[at_old], 0ull - v); }
                };
                book(16, 255u, 3 * (size_t)pn + 0, 3 * (size_t)po + 0);
                book(8, 255u, 3 * (size_t)pn + 1, 3 * (size_t)po + 1);
                book(0, 255u, 3 * (size_t)pn + 2, 3 * (size_t)po + 2);
                book(0, 0u, 3 * (size_t)K + pn, 3 * (size_t)K + po);  // (mask 0: the factor is 1, the sum of the weights)
                if (lane == 0) { atomicAdd(&acc[4 * K + pn], (unsigned long long)cnt); atomicAdd(&acc[4 * K + po], 0ull - (unsigned long long)cnt); }
            }
        }
#pragma unroll
        for (int u = 0; u < kSweep; u++) {
            if (rem[u]) {
                const uint32_t ol = cur[u], pp = p[u], n_ = nl[u];
                const uint64_t w = wt[u];
                const unsigned long long rw = ((pp >> 16) & 255) * w, gw = ((pp >> 8) & 255) * w, bw = (pp & 255) * w;
                atomicAdd(&acc[3 * n_ + 0], rw); atomicAdd(&acc[3 * n_ + 1], gw); atomicAdd(&acc[3 * n_ + 2], bw);
                atomicAdd(&acc[3 * K + n_], (unsigned long long)w); atomicAdd(&acc[4 * K + n_], 1ull);
                atomicAdd(&acc[3 * ol + 0], 0ull - rw); atomicAdd(&acc[3 * ol + 1], 0ull - gw); atomicAdd(&acc[3 * ol + 2], 0ull - bw);
                atomicAdd(&acc[3 * K + ol], 0ull - (unsigned long long)w); atomicAdd(&acc[4 * K + ol], 0ull - 1ull);
            }
        }
        return;
    }
    // every slot's current centroid first (lanes past the cell's end hold point 0 of cluster 0: harmless), the four products before anything
    // that depends on one, ONE branch for the sweeps in which nothing moves -- as four "if (q < e)" bodies each slot saved and restored the
    // exec mask and waited three cycles behind its product
    bool mvs[kSweep], any = ALLWRITE || first;
#pragma unroll
    for (int u0 = 0; u0 < kSweep; u0 += 2) {   // (two slots at a time: four held six more registers than the body has)
        const uint2 ca = tab[cur[u0]], cb = tab[cur[u0 + 1]];
        const uint32_t da = dot4u8(p[u0], ca.x, 0), db = dot4u8(p[u0 + 1], cb.x, 0);
        const uint32_t ka = (da << (IDBITS + 1)) + ca.y, kb = (db << (IDBITS + 1)) + cb.y;
        mvs[u0] = (base + u0 * 64 + lane < e) & ((best[u0] >> IDBITS) > (ka >> IDBITS));  // strictly closer (kmeans.rs:375)
        mvs[u0 + 1] = (base + (u0 + 1) * 64 + lane < e) & ((best[u0 + 1] >> IDBITS) > (kb >> IDBITS));
        any = any | mvs[u0] | mvs[u0 + 1];
    }
    if (!__ballot(any)) return;
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        const uint32_t q = base + u * 64 + lane;
        if (q < e) {
            const bool mv = mvs[u];
            const uint32_t ol = cur[u], pp = p[u];
            const uint32_t nl = mv ? IDMASK - (best[u] & IDMASK) : ol;
            if (mv || ALLWRITE) labels[q] = (LabelT)nl;
            if (mv) moved++;
            if (mv || first) {
                const uint64_t w = wt[u];  // (loaded with the key: a gather here stalls every sweep that moves a point -- NOTES.md D)
                const unsigned long long rw = ((pp >> 16) & 255) * w, gw = ((pp >> 8) & 255) * w, bw = (pp & 255) * w;
                atomicAdd(&acc[3 * nl + 0], rw);
                atomicAdd(&acc[3 * nl + 1], gw);
                atomicAdd(&acc[3 * nl + 2], bw);
                atomicAdd(&acc[3 * K + nl], (unsigned long long)w);
                atomicAdd(&acc[4 * K + nl], 1ull);
                if (!first) {
                    atomicAdd(&acc[3 * ol + 0], 0ull - rw);
                    atomicAdd(&acc[3 * ol + 1], 0ull - gw);
                    atomicAdd(&acc[3 * ol + 2], 0ull - bw);
                    atomicAdd(&acc[3 * K + ol], 0ull - (unsigned long long)w);
                    atomicAdd(&acc[4 * K + ol], 0ull - 1ull);
                }
            }
        }
    }
}

// one sweep with the candidates given as a bitmask of cluster ids (K <= 256) instead of a list: the set bits are walked on the
// scalar unit, each candidate read from the block's table (no candidate list, no compaction)
template <typename LabelT, int IDBITS>
__device__ __forceinline__ void sweep_points_mask(const uint32_t (&p)[kSweep], const uint32_t (&cur)[kSweep], const uint32_t (&wt)[kSweep],
                                                  uint32_t base, uint32_t e, int lane, const unsigned long long (&nm)[4],
                                                  const uint2 *tab, uint32_t K, LabelT *__restrict__ labels, unsigned long long *acc, uint32_t &moved,
                                                  const uint32_t *__restrict__ cweight = nullptr) {
    constexpr uint32_t IDMASK = (1u << IDBITS) - 1;
    uint32_t best[kSweep];
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
            for (int u = 0; u < kSweep; u++) best[u] = max(best[u], (dot4u8(p[u], cc.x, 0) << (IDBITS + 1)) + cc.y);
        }
    }
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        const uint32_t q = base + u * 64 + lane;
        if (q < e) {
            const uint2 cc = tab[cur[u]];
            const uint32_t kcur = (dot4u8(p[u], cc.x, 0) << (IDBITS + 1)) + cc.y;
            const bool mv = (best[u] >> IDBITS) > (kcur >> IDBITS);  // strictly closer (kmeans.rs:375)
            if (mv) {
                const uint32_t ol = cur[u], pp = p[u], nl = IDMASK - (best[u] & IDMASK);
                labels[q] = (LabelT)nl;
                moved++;
                const uint64_t w = wt[u];
                const unsigned long long rw = ((pp >> 16) & 255) * w, gw = ((pp >> 8) & 255) * w, bw = (pp & 255) * w;
                atomicAdd(&acc[3 * nl + 0], rw); atomicAdd(&acc[3 * nl + 1], gw); atomicAdd(&acc[3 * nl + 2], bw);
                atomicAdd(&acc[3 * K + nl], (unsigned long long)w); atomicAdd(&acc[4 * K + nl], 1ull);
                atomicAdd(&acc[3 * ol + 0], 0ull - rw); atomicAdd(&acc[3 * ol + 1], 0ull - gw); atomicAdd(&acc[3 * ol + 2], 0ull - bw);
                atomicAdd(&acc[3 * K + ol], 0ull - (unsigned long long)w); atomicAdd(&acc[4 * K + ol], 0ull - 1ull);
            }
        }
    }
}

// the same sweep at iteration 0, where every point adds to the sums of its cluster (kept apart from
// sweep_points: sharing the code cost the later iterations 30 % through the register allocation)
template <typename LabelT, int IDBITS, bool ROUNDS = false>
__device__ __forceinline__ void sweep_points_first(const uint32_t (&p)[kSweep], const uint32_t (&cur)[kSweep], const uint32_t (&wt)[kSweep],
                                             uint32_t base, uint32_t e, int lane,
                                             const uint2 *cand, uint32_t ncand, const uint2 *tab, uint32_t K,
                                             LabelT *__restrict__ labels, unsigned long long *acc, uint32_t &moved) {
    constexpr uint32_t IDMASK = (1u << IDBITS) - 1;
    uint32_t best[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; u++) best[u] = 0;
    for (uint32_t j = 0; j < ncand; j++) {
        const uint2 cc = cand[j];  // same address in every lane: LDS broadcast
#pragma unroll
        for (int u = 0; u < kSweep; u++) best[u] = max(best[u], (dot4u8(p[u], cc.x, 0) << (IDBITS + 1)) + cc.y);
    }
    uint32_t nl[kSweep];
    bool mvd[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        const uint32_t q = base + u * 64 + lane;
        nl[u] = cur[u];
        mvd[u] = false;
        if (q < e) {
            const uint2 cc = tab[cur[u]];
            const uint32_t kcur = (dot4u8(p[u], cc.x, 0) << (IDBITS + 1)) + cc.y;
            mvd[u] = (best[u] >> IDBITS) > (kcur >> IDBITS);  // strictly closer (kmeans.rs:375)
            if (mvd[u]) { nl[u] = IDMASK - (best[u] & IDMASK); labels[q] = (LabelT)nl[u]; moved++; }
        }
    }
    bool rem[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; u++) rem[u] = base + u * 64 + lane < e;
    if constexpr (ROUNDS) {
        // Iteration 0: every point adds to the sums of its cluster.  A sweep lies inside one 8^3 cell and its points join one,
        // two, three clusters: round by round, the cluster of the first point still to be booked, every point that joins it
        // summed in the wave (DPP), one lane adds the totals -- ten LDS atomics per point on the same few words run one lane at
        // a time instead (~300 cycles an instruction), and the CU's LDS did little else during this launch.  Whatever is left
        // after six rounds goes point by point.
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
            // (one sum at a time: four 64-bit sums held together spill registers)
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
            book(0, 0u, 3 * (size_t)K + pn);  // (mask 0: the factor is 1, the sum of the weights)
            if (lane == 0) atomicAdd(&acc[4 * K + pn], (unsigned long long)cnt);
        }
    } else {
        // Iteration 0: every point adds to the sums of its cluster.  A sweep lies inside one 8^3 cell, so as a
        // rule all its points join the same cluster: one wave reduction and five LDS atomics instead of five
        // 64-way colliding atomics per slot.
        const uint32_t l0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nl[0]);  // lane 0, slot 0 is always a point of the sweep
        bool same = true;
        unsigned long long rw = 0, gw = 0, bw = 0, ww = 0, cn = 0;
#pragma unroll
        for (int u = 0; u < kSweep; u++) {
            if (rem[u]) {
                same = same && nl[u] == l0;
                const uint64_t w = wt[u];
                rw += ((p[u] >> 16) & 255) * w; gw += ((p[u] >> 8) & 255) * w; bw += (p[u] & 255) * w; ww += w; cn += 1;
            }
        }
        if (__ballot(!same) == 0ull) {
            rw = wave_reduce_sum64(rw); gw = wave_reduce_sum64(gw); bw = wave_reduce_sum64(bw);
            ww = wave_reduce_sum64(ww); cn = wave_reduce_sum64(cn);
            if (lane == 0) {
                atomicAdd(&acc[3 * l0 + 0], rw); atomicAdd(&acc[3 * l0 + 1], gw); atomicAdd(&acc[3 * l0 + 2], bw);
                atomicAdd(&acc[3 * K + l0], ww); atomicAdd(&acc[4 * K + l0], cn);
            }
            return;
        }
    }
#pragma unroll
    for (int u = 0; u < kSweep; u++) {
        if (rem[u]) {
            const uint32_t pp = p[u], n_ = nl[u];
            const uint64_t w = wt[u];  // loaded with the key: a gather here would stall every sweep that moves a point
            const unsigned long long rw = ((pp >> 16) & 255) * w, gw = ((pp >> 8) & 255) * w, bw = (pp & 255) * w;
            atomicAdd(&acc[3 * n_ + 0], rw);
            atomicAdd(&acc[3 * n_ + 1], gw);
            atomicAdd(&acc[3 * n_ + 2], bw);
            atomicAdd(&acc[3 * K + n_], (unsigned long long)w);
            atomicAdd(&acc[4 * K + n_], 1ull);
        }
    }
}

// Per non-empty cell, carried between iterations for the skip schedule: ONE record of cell_rec_words(MW) u32 --
// [0] colour of the pivot of the last candidate build, [1] the cell's id, [2 + w] word w of the candidate bitmask --
// 64 bytes for K <= 256, so that the skip test of a cell is one coalesced load (lane <-> word) that can be
// requested several cells ahead; pivot, cell id and mask words then come out of the register by lane reads.
// (Three separate arrays cost the test two dependent round trips to memory per cell: 16 % of the encode.)
struct CellState {
    uint32_t *rec;          // [M][cell_rec_words(MW)]
    const uint32_t *moved;  // [0] = number of centroids changed by the last update, then their ids
    uint32_t max_moved;     // skip schedule when moved[0] <= max_moved (0 disables it)
};

// Centroid update folded into the next assign launch (km_rgbw_run, K <= 256): launch j first finishes iteration
// j-1 -- every block redundantly turns (running sums + the sums of launch j-1) into the K centroids it needs in
// LDS anyway (K divisions, 10 KiB of L2 reads), block 0 also writes the global state -- and then assigns.  That
// removes one dependent kernel (4.8 us + the gap before it) per iteration.  The sums are triple-buffered: launch
// j adds into buffer j % 3, reads (j - 1) % 3 and clears (j + 1) % 3; the running sums ping-pong.
struct FusedUpdate {
    uint32_t on;          // 0: an update kernel ran in between (centroids in cconst, moved list in CellState)
    uint32_t launch_no;   // j
    uint32_t idbits;
    uint64_t max_iters, seed, U;
    const uint32_t *keys;                        // canonical point list (empty-cluster reseed)
    GIdx gx;                                     // ... or the index of all occupied colours (several GPUs)
    const unsigned long long *partials_prev;     // sums of launch j - 1 (all-reduced when there are several GPUs)
    unsigned long long *partials_clear;          // the buffer launch j + 1 adds into
    const unsigned long long *running_prev;
    unsigned long long *running_new;
    const uint32_t *cent_prev;                   // centroids of launch j - 1 (to tell which ones moved)
    uint32_t *cent_new;                          // block 0 writes: this launch's centroids for launch j + 1 to compare with,
    uint2 *cconst_g;                             // and the result block (centroids, members, weights, state)
    uint32_t *cent_g;
    uint64_t *members_out, *wsum_out;
    KmDevState *st_rw;
    KmDevState *st_host;                         // pinned host copy of the scalar state (lagged polling without a copy kernel), or null
    PollRec *st_ring;                            // ... or the ring of per-launch records (loops with collectives), or null
};

// this launch's record of the scalar state (block 0, thread 0): fields first, the launch number last
__device__ __forceinline__ void poll_record(PollRec *ring, uint32_t launch_no, const KmDevState *sw) {
    PollRec *r = ring + launch_no % kPollRing;
    r->iter = sw->iter; r->done = sw->done; r->moved_last = sw->moved_last; r->reseeds = sw->reseeds; r->active = sw->active; r->pair_evals = sw->pair_evals;
    __threadfence_system();
    r->seq = launch_no;
}

// FIRSTK: -1 = everything found out on the device (the loops without the folded-in update).  The loop with the folded-in update
// knows which launch it enqueues: 1 = the launch of iteration 0 (alone that body has registers to spare for booking its points
// round by round, sweep_points_first), 2 = one of the next three (the same body as 0 plus the rounds for movers that share an
// (old, new) pair, sweep_points<AGG>: whole cells change hands while the centroids still travel; the rounds' code costs the
// launches that do not need it 4-5 us each, so only these get it), 0 = a later one.
template <typename LabelT, int IDBITS, int WAVES, int FIRSTK = -1>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_rgbw_assign_cells(
    const uint32_t *__restrict__ ckeys, const uint32_t *__restrict__ cweight, const uint32_t *__restrict__ ne_cell,
    const uint32_t *__restrict__ ne_start, const uint32_t *__restrict__ wfirst,
    uint32_t shard, uint32_t K, const uint2 *__restrict__ cconst, LabelT *__restrict__ labels,
    unsigned long long *__restrict__ partials, const KmDevState *__restrict__ st, CellState cs, FusedUpdate fz) {
    extern __shared__ __align__(16) unsigned long long lds[];  // [5K] deltas | uint2 tab[K] | WAVES x (uint2 S[(K+1)/2], uint2 cand[K]) | WAVES x u64 mask[MW]
    __shared__ uint32_t s_moved, s_cell, s_nmoved, s_reseed, s_active;
    __shared__ uint32_t s_mlist[kMaxMovedSkip];
    __shared__ uint32_t s_nS[WAVES];        // full schedule: lengths of the block's shared super-cell lists
    __shared__ unsigned long long s_mm[4];  // K <= 256: bit k <=> centroid k moved in the last update
    __shared__ unsigned long long s_evals;
    const uint32_t done = st->done;  // acted on below, once the set-up loads are on their way: a launch after convergence costs one round trip, not two
    constexpr int THREADS = WAVES * 64;
    unsigned long long *acc = lds;
    uint2 *tab = reinterpret_cast<uint2 *>(lds + 5 * (size_t)K);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t MW = (K + 63) >> 6;  // mask words per cell
    const uint32_t scap = km_scap(K);  // super-cell list capacity (it rarely holds more than a quarter of the table)
    const uint32_t ccap = km_ccap(K);  // ... and the candidate strip's
    uint2 *S = tab + K + (size_t)wid * (scap + ccap);
    uint2 *cand = S + scap;
    unsigned long long *wmask = reinterpret_cast<unsigned long long *>(tab + K + (size_t)WAVES * (scap + ccap)) + (size_t)wid * MW;
    const uint32_t gw0 = shard * gridDim.x * WAVES;          // first wave index of this shard
    // Iteration 0 accumulates the FULL sums of the new assignment (the running sums start at zero:
    // the initial chunk assignment, kmeans.rs:61-78, only matters through the labels); later
    // iterations add/subtract only the points that moved.
    const bool first = FIRSTK >= 0 ? FIRSTK == 1 : (fz.on ? fz.launch_no == 0 : st->iter == 0);
    uint32_t nS = fz.on ? K : cs.moved[0];
    const uint32_t *mlist = cs.moved + 1;  // ids of the centroids the last update changed
    const unsigned long long lt_mask = (1ull << lane) - 1;
    uint32_t moved = 0;
    unsigned long long evals = 0;
    for (uint32_t i = threadIdx.x; i < 5 * K; i += THREADS) acc[i] = 0ull;
    if (!fz.on || first)
        for (uint32_t i = threadIdx.x; i < K; i += THREADS) tab[i] = cconst[i];
    if (threadIdx.x == 0) { s_moved = 0; s_evals = 0; s_nmoved = 0; s_reseed = 0; s_active = 0; s_mm[0] = s_mm[1] = s_mm[2] = s_mm[3] = 0ull; }
    // fused update: one cluster per thread (K <= 256 <= THREADS); its ten sums and its old centroid are requested
    // together with the set-up loads above, before anything is waited for
    const bool upd = fz.on && !first;
    const uint32_t uk = threadIdx.x;
    unsigned long long v[5] = {0, 0, 0, 0, 0}, changed = 0, pev = 0;
    uint32_t oldc = 0;
    if (upd) {
        changed = fz.partials_prev[5 * (size_t)K];
        pev = fz.partials_prev[5 * (size_t)K + 1];
        if (uk < K) {
            const size_t at[5] = {3 * (size_t)uk, 3 * (size_t)uk + 1, 3 * (size_t)uk + 2, 3 * (size_t)K + uk, 4 * (size_t)K + uk};
#pragma unroll
            for (int i = 0; i < 5; i++) v[i] = fz.partials_prev[at[i]] + fz.running_prev[at[i]];
            oldc = fz.cent_prev[uk];
        }
    }
    if (done) {  // a launch past convergence: nothing but its state record (the host may ask for exactly this launch)
        if (fz.on && fz.st_ring && blockIdx.x == 0 && threadIdx.x == 0) poll_record(fz.st_ring, fz.launch_no, fz.st_rw);
        return;
    }
    __syncthreads();
    if (upd) {
        // ---- finish iteration j - 1: Point::mean for ColorCount (clusterc.rs:83-113) + empty-cluster reseed (kmeans.rs:110-137)
        const uint32_t j = fz.launch_no;
        if (uk < K) {
            const uint32_t k = uk;
            const size_t at[5] = {3 * (size_t)k, 3 * (size_t)k + 1, 3 * (size_t)k + 2, 3 * (size_t)K + k, 4 * (size_t)K + k};
            uint32_t ck;
            if (v[4] == 0) {
                const uint64_t ri = reseed_index(fz.seed, j - 1, k, fz.U);  // fake_clone of the stolen point
                ck = fz.gx.bits ? gidx_select(fz.gx, ri) : fz.keys[ri];
                atomicAdd(&s_reseed, 1u);
            } else {
                // (sums of channel * weight over at most 2^32 pixels: exact through a double quotient)
                const uint32_t r = div_floor_small(v[0], v[3]) & 255, g = div_floor_small(v[1], v[3]) & 255, b = div_floor_small(v[2], v[3]) & 255;
                ck = (r << 16) | (g << 8) | b;
                atomicAdd(&s_active, 1u);
            }
            const uint2 cc = make_cconst(ck, k, fz.idbits);
            tab[k] = cc;
            if (ck != oldc) {
                const uint32_t pos = atomicAdd(&s_nmoved, 1u);
                if (pos < kMaxMovedSkip) s_mlist[pos] = k;
                atomicOr(&s_mm[(k >> 6) & 3], 1ull << (k & 63));
            }
            if (blockIdx.x == 0) {
#pragma unroll
                for (int i = 0; i < 5; i++) fz.running_new[at[i]] = v[i];
                fz.cent_new[k] = ck;
                fz.cent_g[k] = ck;
                fz.cconst_g[k] = cc;
                fz.members_out[k] = v[4];
                fz.wsum_out[k] = v[3];
            }
        }
        if (blockIdx.x == 0)
            for (uint32_t i = threadIdx.x; i < 5 * K + 2; i += THREADS) fz.partials_clear[i] = 0ull;
        __syncthreads();
        const bool fin = changed == 0 || (fz.max_iters && j >= fz.max_iters);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            KmDevState *sw = fz.st_rw;
            sw->changed_ring[(j - 1) % kHistRing] = changed;
            sw->nmoved_ring[(j - 1) % kHistRing] = s_nmoved;
            sw->moved_last = changed;
            sw->reseeds += s_reseed;
            sw->active = s_active;
            sw->pair_evals += pev;
            sw->iter = j;
            if (fin) sw->done = 1;
            if (fz.st_host) {  // the host polls this after the batch's event: scalars first, the flag last
                KmDevState *hs = fz.st_host;
                hs->moved_last = changed; hs->reseeds = sw->reseeds; hs->active = s_active; hs->pair_evals = sw->pair_evals; hs->iter = j;
                __threadfence_system();
                if (fin) hs->done = 1;
            }
            if (fz.st_ring) poll_record(fz.st_ring, j, sw);
        }
        if (fin) return;  // converged (or the iteration cap): nothing to assign
        nS = s_nmoved;
        mlist = s_mlist;
    } else if (fz.on && blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < 5 * K + 2; i += THREADS) fz.partials_clear[i] = 0ull;
        if (fz.st_ring && threadIdx.x == 0) poll_record(fz.st_ring, fz.launch_no, fz.st_rw);  // (launch 0: nothing finished yet)
    }
    const bool skip_mode = !first && nS <= cs.max_moved;

    if (!skip_mode) {
        // ================================================================= FULL schedule
        // The block owns the cells [mb0, mb1) (equal-cost ranges, k_wave_ranges) and its waves draw them one
        // at a time from an LDS counter: what a cell costs depends on how many of its points move, which no
        // static split predicts (with one fixed range per wave the slowest wave ran 2x the mean).
        const uint32_t bg = gw0 + blockIdx.x * WAVES;
        const uint32_t mb0 = wfirst[bg], mb1 = wfirst[bg + WAVES];
        // The super-cell lists S of the block's range, ONCE per block: its cells are consecutive in super-cell-major order, so they
        // lie in one to three super-cells; wave w builds the list of super-cell sup_first + w into its own strip and every wave
        // reads the strip it needs.  (Until round 3 every wave built S for every super-cell it touched: 8 K builds a launch for 512
        // super-cells, ~15 % of the full schedule's VALU instructions.)  A range
        // that spans more super-cells than the block has waves (a sparse image) keeps the private builds.
        uint32_t sup_first = 0, nsl = 0;
        if (mb1 > mb0) {
            const uint32_t cf = ne_cell[mb0] >> kSuperShift, cl = ne_cell[mb1 - 1] >> kSuperShift;
            sup_first = cf;
            if (cl - cf < (uint32_t)WAVES) nsl = cl - cf + 1;
        }
        if ((uint32_t)wid < nsl) {
            const uint32_t n = build_super(tab, K, sup_first + wid, lane, lt_mask, S, scap);
            if (lane == 0) s_nS[wid] = n;
        }
        if (threadIdx.x == 0) s_cell = mb0;
        __syncthreads();
        auto draw = [&]() -> uint32_t {
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(&s_cell, 1u);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        };
        uint32_t m = draw();
        uint32_t s = 0, e = 0, c = 0;
        uint32_t p[kSweep], cur[kSweep], wt[kSweep];
        if (m < mb1) {
            s = ne_start[m]; e = ne_start[m + 1]; c = ne_cell[m];
            load_points<LabelT>(ckeys, labels, cweight, s, e, lane, p, cur, wt);
        }
        uint32_t sup = 0xffffffffu, nSup = 0;
        while (m < mb1) {
            // the next cell of this wave: its descriptors now, the loads of its first sweep during this cell's last sweep
            const uint32_t mn = draw();
            uint32_t s_next = 0, e_next = 0, c_next = 0;
            if (mn < mb1) { s_next = ne_start[mn]; e_next = ne_start[mn + 1]; c_next = ne_cell[mn]; }
            const uint2 *Sl = S;
            if (nsl) {
                const uint32_t slot = (c >> kSuperShift) - sup_first;
                Sl = tab + K + (size_t)slot * (scap + ccap);
                nSup = s_nS[slot];
            } else if ((c >> kSuperShift) != sup) { sup = c >> kSuperShift; nSup = build_super(tab, K, sup, lane, lt_mask, S, scap); }
            uint32_t ncand = nSup <= scap ? build_candidates<IDBITS>(Sl, nSup, c, lane, lt_mask, cand, wmask, cs.rec, m, MW, ccap)
                                      : build_candidates<IDBITS>(tab, K, c, lane, lt_mask, cand, wmask, cs.rec, m, MW, ccap);
            const uint2 *cl = cand;   // (u16 labels: a strip that overflowed -- the table is the candidate list then)
            if (IDBITS != 8 && ncand > ccap) { cl = tab; ncand = K; }
            for (uint32_t base = s; base < e; base += 64 * kSweep) {
                const bool more = base + 64 * kSweep < e;
                const uint32_t nts = more ? base + 64 * kSweep : s_next;
                const uint32_t nte = more ? e : e_next;
                uint32_t pn[kSweep], curn[kSweep], wtn[kSweep];
                load_points<LabelT>(ckeys, labels, cweight, nts, nte, lane, pn, curn, wtn);
                if (first) sweep_points_first<LabelT, IDBITS, FIRSTK == 1>(p, cur, wt, base, e, lane, cl, ncand, tab, K, labels, acc, moved);
                else sweep_points<LabelT, IDBITS, false, FIRSTK == 2>(p, cur, wt, base, e, lane, cl, ncand, tab, K, false, labels, acc, moved);
#pragma unroll
                for (int u = 0; u < kSweep; u++) { p[u] = pn[u]; cur[u] = curn[u]; wt[u] = wtn[u]; }
            }
            evals += (unsigned long long)(e - s) * (ncand + 1);
            __builtin_amdgcn_wave_barrier();  // the strip is rewritten for the next cell
            m = mn; s = s_next; e = e_next; c = c_next;
        }
    } else {
        // ================================================================= SKIP schedule
        // this shard's cells: [m_lo, m_hi); the centroids that moved, one per lane
        static_assert(kMaxMovedSkip <= 64, "one moved centroid per lane");
        const uint32_t m_lo = wfirst[gw0], m_hi = wfirst[gw0 + gridDim.x * WAVES];
        const uint32_t k1 = (uint32_t)lane < nS ? mlist[lane] : 0xffffffffu;
        const uint32_t ck1 = k1 != 0xffffffffu ? tab[k1].x : 0u;
        if (IDBITS == 8 && !upd) {  // (an update kernel ran in between: only its list is at hand) the moved ids as a bitmask
            if (wid == 0 && k1 != 0xffffffffu) atomicOr(&s_mm[(k1 >> 6) & 3], 1ull << (k1 & 63));
            __syncthreads();
        }
        // cells are dealt round-robin: what survives the skip test is clustered around the centroids that
        // moved, and striding spreads those clusters over all waves (a shared atomic queue would
        // saturate: one word serves ~90 dequeues/us)
        const uint32_t nwaves = gridDim.x * WAVES;
        {
            // The records of this wave's next kRecBatch cells are requested together -- lanes 16 i .. 16 i + 15 of one
            // register hold cell i's record (K <= 256: 16 words), four cells per load instruction -- and read out lane
            // by lane: one round trip to memory per batch, not per cell (most cells turn out clean and cost nothing else).
            constexpr uint32_t kRecBatch = IDBITS == 8 ? 8 : 1;
            const uint32_t RW = cell_rec_words(MW);
            const uint32_t m_first = m_lo + blockIdx.x * WAVES + wid;
            for (uint32_t mb = m_first; mb < m_hi; mb += kRecBatch * nwaves) {
              uint32_t rA = 0, rB = 0;
              if (IDBITS == 8) {
                  const uint32_t ma = mb + ((uint32_t)lane >> 4) * nwaves, mc = ma + 4 * nwaves;
                  if (ma < m_hi) rA = cs.rec[(size_t)ma * 16 + (lane & 15)];
                  if (mc < m_hi) rB = cs.rec[(size_t)mc * 16 + (lane & 15)];
              } else {  // a record of up to 80 words (K <= 2048): one cell at a time
                  const uint32_t *rec = cs.rec + (size_t)mb * RW;
                  if ((uint32_t)lane < RW) rA = rec[lane];
                  if (64 + (uint32_t)lane < RW) rB = rec[64 + lane];
              }
              // K <= 256: the batch's eight cells are tested TOGETHER, lane = (cell, one of eight moved centroids), ceil(nS / 8) steps --
              // one step while few centroids move (the long tail of the run); a wave-wide test per cell cost ~30 instructions for
              // each of the eight cells however few had moved.  dirty8: bits 8 i .. 8 i + 7 belong to cell i of the batch.
              unsigned long long dirty8 = 0;
              if (IDBITS == 8) {
                  const uint32_t ci = (uint32_t)lane >> 3;
                  const int src0 = (int)((ci & 3) * 16);
                  const uint32_t pvA = (uint32_t)__shfl((int)rA, src0, 64), pvB = (uint32_t)__shfl((int)rB, src0, 64);
                  const uint32_t ccA = (uint32_t)__shfl((int)rA, src0 + 1, 64), ccB = (uint32_t)__shfl((int)rB, src0 + 1, 64);
                  const uint32_t pv = ci < 4 ? pvA : pvB, cc = ci < 4 ? ccA : ccB;
                  const bool cell_ok = mb + ci * nwaves < m_hi;
                  Dominance dmv;
                  dmv.set(cell_box(cc & 0xffffu), (1 << kCellShift) - 1, pv);
                  bool dv = false;
                  for (uint32_t j0 = 0; j0 < nS; j0 += 8) {
                      const uint32_t j = j0 + ((uint32_t)lane & 7u);
                      const bool has = j < nS;
                      const uint32_t k = has ? mlist[j] : 0u;
                      const uint32_t ck = tab[k].x;
                      const int wl = src0 + 2 + (int)((k >> 5) & 7);
                      const uint32_t wA = (uint32_t)__shfl((int)rA, wl, 64), wB = (uint32_t)__shfl((int)rB, wl, 64);
                      const bool in = (((ci < 4 ? wA : wB) >> (k & 31)) & 1u) != 0;
                      // a moved centroid matters if it was a candidate or is no longer dominated by the pivot
                      dv = dv || (has && (in || dmv.worst(ck) >= 0));
                  }
                  dirty8 = __ballot(dv && cell_ok);
                  if (!dirty8) continue;
              }
              if constexpr (IDBITS == 8) {
                // the moved ids as a bitmask laid out like a register of records (lanes 16 i + 2 .. 16 i + 9 = the eight mask words)
                const uint32_t jw = ((uint32_t)lane & 15u) - 2u;
                const uint32_t mmv = jw < 8u ? reinterpret_cast<const uint32_t *>(s_mm)[jw] : 0u;
#pragma unroll 1
                for (uint32_t bi = 0; bi < kRecBatch; bi++) {
                    const uint32_t m = mb + bi * nwaves;
                    if (m >= m_hi) break;
                    if (!((dirty8 >> (8 * bi)) & 0xffull)) continue;  // nothing that matters to this cell changed: every label repeats
                    const uint32_t r = bi < 4 ? rA : rB;
                    const int l0 = (int)((bi & 3) * 16);
                    const uint32_t pvt = (uint32_t)__builtin_amdgcn_readlane((int)r, l0);
                    const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)r, l0 + 1);
                    const uint32_t c = cw & 0xffffu, pid = (cw >> 16) & 0x7fffu;
                    const uint32_t s = ne_start[m], e = ne_start[m + 1];
                    uint32_t p[kSweep], cur[kSweep], wt[kSweep];
                    load_points<LabelT>(ckeys, labels, cweight, s, e, lane, p, cur, wt);
                    // A mask is COMPLETE if it holds every centroid of the table that the cell's pivot does not dominate (the full
                    // schedule builds from the super-cell's list instead: what that list left out was beaten by ANOTHER centroid,
                    // which may have moved since).  While the pivot of a complete mask stands where it stood, every centroid that
                    // has not moved keeps its verdict against it; the moved ones are tested here: new mask = old mask without the
                    // moved ids + the moved ids that pass.  Otherwise the mask is rebuilt from the whole table with a fresh pivot,
                    // one ballot per 64 ids -- and is complete from then on.  Either way there is no candidate list: the sweep
                    // walks the mask on the scalar unit.
                    constexpr int32_t ext = (1 << kCellShift) - 1;
                    const CellBox bx = cell_box(c);
                    const bool keep_pivot = (cw & kRecComplete) && (((uint32_t)__builtin_amdgcn_readlane((int)mmv, 2 + (int)((pid >> 5) & 7)) >> (pid & 31)) & 1u) == 0u;
                    unsigned long long nm[4];
                    Dominance dm;
                    if (keep_pivot) {
                        dm.set(bx, ext, pvt);
                        unsigned long long f1 = __ballot(k1 != 0xffffffffu && dm.worst(ck1) >= 0);
                        const uint32_t nmv = r & ~mmv;
#pragma unroll
                        for (int w = 0; w < 4; w++)
                            nm[w] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)nmv, l0 + 3 + 2 * w) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)nmv, l0 + 2 + 2 * w);
                        while (f1) {
                            const int l = __builtin_ctzll(f1);
                            f1 &= f1 - 1;
                            const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)k1, l);
                            const unsigned long long b = 1ull << (k & 63);
                            const uint32_t w = (k >> 6) & 3;
                            nm[0] |= w == 0 ? b : 0ull; nm[1] |= w == 1 ? b : 0ull; nm[2] |= w == 2 ? b : 0ull; nm[3] |= w == 3 ? b : 0ull;
                        }
                    } else {
                        const uint32_t npid = nearest_to_centre(tab, K, bx, ext, lane);  // (the table is in id order)
                        const uint32_t npv = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[npid].x);
                        dm.set(bx, ext, npv);
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            const uint32_t k = 64 * w + lane;
                            nm[w] = __ballot(k < K && dm.worst(tab[k < K ? k : 0].x) >= 0);
                        }
                        if (lane == 0) *reinterpret_cast<uint2 *>(cs.rec + (size_t)m * 16) = make_uint2(npv, c | (npid << 16) | kRecComplete);
                    }
                    {   // the lanes that hold the cell's mask words write the ones that changed
                        uint32_t wv = 0;
#pragma unroll
                        for (int t = 0; t < 8; t++)
                            if (jw == (uint32_t)t) wv = (uint32_t)(nm[t >> 1] >> (32 * (t & 1)));
                        if ((lane >> 4) == (int)(bi & 3) && jw < 8u && wv != r) cs.rec[(size_t)m * 16 + 2 + jw] = wv;
                    }
                    const uint32_t ncand = (uint32_t)(__popcll(nm[0]) + __popcll(nm[1]) + __popcll(nm[2]) + __popcll(nm[3]));
                    for (uint32_t base = s; base < e; base += 64 * kSweep) {
                        uint32_t pn[kSweep], curn[kSweep], wtn[kSweep];
                        load_points<LabelT>(ckeys, labels, cweight, base + 64 * kSweep, e, lane, pn, curn, wtn);
                        sweep_points_mask<LabelT, IDBITS>(p, cur, wt, base, e, lane, nm, tab, K, labels, acc, moved);
#pragma unroll
                        for (int u = 0; u < kSweep; u++) { p[u] = pn[u]; cur[u] = curn[u]; wt[u] = wtn[u]; }
                    }
                    evals += (unsigned long long)(e - s) * (ncand + 1);
                }
              } else {
              for (uint32_t bi = 0; bi < kRecBatch; bi++) {
                const uint32_t m = mb + bi * nwaves;
                if (m >= m_hi) break;
                uint32_t pvt, c;
                bool in1 = false;
                {
                    pvt = (uint32_t)__builtin_amdgcn_readlane((int)rA, 0);
                    c = (uint32_t)__builtin_amdgcn_readlane((int)rA, 1) & 0xffffu;
                    const uint32_t i1 = 2 + ((k1 & 0x7ffu) >> 5);
                    // (both shuffles by every lane: a lane that sits out a shuffle is read as nothing by the others)
                    const uint32_t a1 = (uint32_t)__shfl((int)rA, (int)(i1 & 63), 64), b1 = (uint32_t)__shfl((int)rB, (int)(i1 & 63), 64);
                    const uint32_t w1 = i1 < 64 ? a1 : b1;
                    in1 = ((w1 >> (k1 & 31)) & 1u) != 0;
                }
                if (IDBITS != 8) {
                    Dominance dm;
                    dm.set(cell_box(c), (1 << kCellShift) - 1, pvt);
                    // a moved centroid matters if it was a candidate or is no longer dominated by the pivot
                    bool dirty = false;
                    if (k1 != 0xffffffffu) dirty = in1 || dm.worst(ck1) >= 0;
                    if (!__ballot(dirty)) continue;  // nothing that matters to this cell changed: every label repeats
                }
                const uint32_t s = ne_start[m], e = ne_start[m + 1];
                uint32_t p[kSweep], cur[kSweep], wt[kSweep];
                load_points<LabelT>(ckeys, labels, cweight, s, e, lane, p, cur, wt);
                // few cells survive and they are dealt round-robin: straight from the table, no super-cell list
                uint32_t ncand = build_candidates<IDBITS>(tab, K, c, lane, lt_mask, cand, wmask, cs.rec, m, MW, ccap);
                const uint2 *cl = cand;
                if (IDBITS != 8 && ncand > ccap) { cl = tab; ncand = K; }
                for (uint32_t base = s; base < e; base += 64 * kSweep) {
                    uint32_t pn[kSweep], curn[kSweep], wtn[kSweep];
                    load_points<LabelT>(ckeys, labels, cweight, base + 64 * kSweep, e, lane, pn, curn, wtn);
                    sweep_points<LabelT, IDBITS>(p, cur, wt, base, e, lane, cl, ncand, tab, K, false, labels, acc, moved);
#pragma unroll
                    for (int u = 0; u < kSweep; u++) { p[u] = pn[u]; cur[u] = curn[u]; wt[u] = wtn[u]; }
                }
                evals += (unsigned long long)(e - s) * (ncand + 1);
                __builtin_amdgcn_wave_barrier();
              }
              }
            }
        }
    }
    moved = wave_reduce_sum(moved);
    if (lane == 0) {
        if (moved) atomicAdd(&s_moved, moved);
        if (evals) atomicAdd(&s_evals, evals);
    }
    __syncthreads();
    uint32_t i_first = threadIdx.x;
    asm volatile("" : "+v"(i_first));  // (or the flush's addresses are computed before the cell loops and held in two registers across them)
    for (uint32_t i = i_first; i < 5 * K; i += THREADS)
        if (acc[i]) atomicAdd(&partials[i], acc[i]);
    if (threadIdx.x == 0) {
        if (s_moved) atomicAdd(&partials[5 * (size_t)K], (unsigned long long)s_moved);
        if (s_evals) atomicAdd(&partials[5 * (size_t)K + 1], s_evals);
    }
}

// ---------------------------------------------------------------- centroid update
// Point::mean for ColorCount (clusterc.rs:83-113) + empty-cluster reseed (kmeans.rs:110-137).
// mode 0: partials hold the full sums of this iteration (brute path)
// mode 1: partials hold deltas (full sums at iteration 0); running += partials first (cells path)
__global__ __launch_bounds__(256) void k_rgbw_update(uint64_t *__restrict__ partials, uint64_t *__restrict__ running,
                                                     int mode, const uint32_t *__restrict__ keys,
                                                     uint64_t U, uint32_t K, uint32_t idbits, uint64_t seed,
                                                     uint64_t max_iters, uint2 *__restrict__ cconst,
                                                     uint32_t *__restrict__ cent, uint64_t *__restrict__ members_out,
                                                     uint64_t *__restrict__ wsum_out,
                                                     uint32_t *__restrict__ moved_list,
                                                     KmDevState *__restrict__ st, GIdx gx) {
    // everything this launch reads is requested before the `done` flag is looked at: one memory round trip
    const uint32_t done = st->done;
    const uint64_t iter = st->iter;
    const uint64_t changed = partials[5 * (size_t)K];
    const uint64_t evals = partials[5 * (size_t)K + 1];
    __shared__ uint32_t s_reseed, s_active, s_nmoved;
    if (threadIdx.x == 0) { s_reseed = 0; s_active = 0; s_nmoved = 0; }
    __syncthreads();
    if (done) return;
    for (uint32_t k = threadIdx.x; k < K; k += blockDim.x) {
        // the cluster's five sums: this iteration's partials, on top of the running sums in delta mode
        const size_t at[5] = {3 * (size_t)k, 3 * (size_t)k + 1, 3 * (size_t)k + 2, 3 * (size_t)K + k, 4 * (size_t)K + k};
        uint64_t v[5];
#pragma unroll
        for (int i = 0; i < 5; i++) { v[i] = partials[at[i]]; partials[at[i]] = 0; }  // zero: ready for the next iteration
        if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 5; i++) { v[i] += running[at[i]]; running[at[i]] = v[i]; }
        }
        const uint64_t members = v[4], w = v[3];
        members_out[k] = members;
        wsum_out[k] = w;
        uint32_t ck;
        if (members == 0) {
            const uint64_t ri = reseed_index(seed, iter, k, U);  // fake_clone of the stolen point
            ck = gx.bits ? gidx_select(gx, ri) : keys[ri];
            atomicAdd(&s_reseed, 1u);
        } else {
            const uint32_t r = div_floor_small(v[0], w) & 255, g = div_floor_small(v[1], w) & 255, b = div_floor_small(v[2], w) & 255;
            ck = (r << 16) | (g << 8) | b;
            atomicAdd(&s_active, 1u);
        }
        if (moved_list && ck != cent[k]) moved_list[1 + atomicAdd(&s_nmoved, 1u)] = k;  // centroids that changed value
        cent[k] = ck;
        cconst[k] = make_cconst(ck, k, idbits);
    }
    __syncthreads();
    if (threadIdx.x == 0 && moved_list) moved_list[0] = s_nmoved;
    if (threadIdx.x == 0) {
        partials[5 * (size_t)K] = 0;
        partials[5 * (size_t)K + 1] = 0;
        st->changed_ring[iter % kHistRing] = changed;
        st->nmoved_ring[iter % kHistRing] = s_nmoved;
        st->moved_last = changed;
        st->reseeds += s_reseed;
        st->active = s_active;
        st->pair_evals += evals;
        st->iter = iter + 1;
        if (changed == 0 || (max_iters && iter + 1 >= max_iters)) st->done = 1;
    }
}

template <typename LabelT>
__global__ void k_widen_labels(const LabelT *__restrict__ in, const uint32_t *__restrict__ rank, uint32_t *__restrict__ out,
                               uint64_t lo, uint64_t hi) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride) {
        if (rank) out[rank[i]] = in[i];   // cell-major -> canonical order
        else out[i - lo] = in[i - lo];
    }
}
template <typename LabelT>
__global__ void k_narrow_labels(const uint32_t *__restrict__ in, LabelT *__restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (LabelT)in[i];
}
// canonical-order u8/u16 labels from cell-major ones (for the remap / bit-pack stages)
template <typename LabelT>
__global__ void k_labels_to_canonical(const LabelT *__restrict__ in, const uint32_t *__restrict__ rank, LabelT *__restrict__ out,
                                      uint64_t U) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride) out[rank[i]] = in[i];
}
__global__ void k_set_cconst(const uint32_t *__restrict__ cent, uint32_t K, uint32_t Kpad, uint32_t idbits,
                             uint2 *__restrict__ cconst) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) cconst[k] = make_cconst(cent[k], k, idbits);
    else if (k < Kpad) cconst[k] = make_uint2(0u, 0u);
}

// =========================================================================== host side
static inline uint32_t grid_1d(uint64_t n, uint32_t cap = 2048) {
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, 256), 1), cap);
}

#define KM_ALLOC(buf, bytes)                                                                       \
    do {                                                                                           \
        hipError_t _e = (buf).alloc(bytes);                                                        \
        if (_e != hipSuccess) { delete s; return c->fail(CNIIC_ERR_HIP, "kmeans_rgbw: hipMalloc(%llu) failed: %s", \
                                                        (unsigned long long)(bytes), hipGetErrorString(_e)); } \
    } while (0)

int km_rgbw_create(Ctx *c, const uint32_t *keys_d, const uint32_t *weight_d, uint64_t U, uint32_t shard,
                   uint32_t nshards, uint32_t K, const cniic_kmeans_opts *opts, void *partials_dev,
                   const uint32_t *rank_table_d, KmRgbwState **out, const uint32_t *cell_count_d, const void *gbits_d,
                   const uint32_t *gprefix_d, uint64_t Ug, bool points_follow, const uint64_t *points_dev) {
    if (K == 0 || U == 0 || nshards == 0 || shard >= nshards) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_rgbw: bad sizes");
    if (points_follow && (!gbits_d || !cell_count_d || nshards != 1 || (opts && (opts->flags & CNIIC_KM_BRUTE_FORCE))))
        return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_rgbw: points written by the caller need the cell counts and the colour index");
    if (!points_follow && gbits_d && (!rank_table_d || nshards != 1 || (opts && (opts->flags & CNIIC_KM_BRUTE_FORCE))))
        return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_rgbw: a share of a larger point list needs the table-driven cells path, unsharded");
    const uint64_t lo = U * shard / nshards, hi = U * (shard + 1) / nshards;  // brute path: equal point slices
    const uint64_t Ulist = gbits_d ? Ug : U;  // length of the reference's point list (init chunks, reseed index)
    if (points_dev && !points_follow) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_rgbw: a device-side point count needs points_follow");
    if (!points_dev && Ulist / K == 0) return c->fail(CNIIC_ERR_TOO_FEW_POINTS, "kmeans: %llu points for %u clusters (src/kmeans.rs:68)",
                                       (unsigned long long)Ulist, K);
    // K > 2048: the every-centroid kernel with its sums in memory (k_kmeans_wide.hip); u16 labels end at 65535
    if (K > 65535) return c->fail(CNIIC_ERR_UNSUPPORTED, "kmeans_rgbw: K=%u > 65535 not supported", K);
    if (K > 2048 && (nshards != 1 || (opts && (opts->flags & CNIIC_KM_BRUTE_FORCE))))
        return c->fail(CNIIC_ERR_UNSUPPORTED, "kmeans_rgbw: K=%u > 2048 runs unsharded on the cell-major arrays only", K);
    if (U >= (1ull << 32)) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_rgbw: too many points");
    auto *s = new KmRgbwState();
    s->c = c; s->U = U; s->lo = lo; s->hi = hi; s->K = K; s->shard = shard; s->nshards = nshards;
    s->Kpad = (K + 3) & ~3u;
    s->wide = K > 256;
    s->big = K > 2048;
    s->idbits = s->wide ? 12 : 8;
    s->seed = (opts && opts->seed) ? opts->seed : kDefaultSeed;
    s->max_iters = opts ? opts->max_iters : 0;
    s->cells = !(opts && (opts->flags & CNIIC_KM_BRUTE_FORCE));
    s->profile = opts && (opts->flags & CNIIC_KM_PROFILE);
    s->no_skip = opts && (opts->flags & CNIIC_KM_NO_SKIP);
    if (const char *al = test_env("CNIIC_KM_AGG_LAUNCHES")) s->agg_launches = (uint32_t)atoi(al);
    if (const char *bb = test_env("CNIIC_KM_BIG_BLOCKS_FROM")) s->big_blocks_from = (uint32_t)atoi(bb);
    if (const char *ms = test_env("CNIIC_KM_MAXSKIP")) s->max_skip = std::min<uint32_t>((uint32_t)atoi(ms), kMaxMovedSkip);
    // (every knob of the loop is read here, once: getenv() per launch raced with tools that set variables between contexts)
    if (const char *fa = test_env("CNIIC_TEST_FAIL_AT_LAUNCH")) s->fail_at = atol(fa);
    s->keys = keys_d; s->weight = weight_d;
    s->gidx = GIdx{static_cast<const unsigned long long *>(gbits_d), gprefix_d, Ug};
    const uint64_t n = hi - lo;
    const uint64_t W = 5 * (uint64_t)K + 2;
    const uint64_t lab_bytes = s->wide ? 2 : 1;
    KM_ALLOC(s->cconst, (uint64_t)s->Kpad * 8);
    s->res_cent = (sizeof(KmDevState) + 255) & ~255ull;
    s->res_members = s->res_cent + (((uint64_t)K * 4 + 7) & ~7ull);
    s->res_wsum = s->res_members + (uint64_t)K * 8;
    s->res_bytes = s->res_wsum + (uint64_t)K * 8;
    // one zeroed arena (one fill instead of seven): result block | own partials | running sums | the fused loop's
    // buffers | moved list; the centroid kernel below writes the non-zero parts
    s->fused = s->cells && !s->wide && !(test_env("CNIIC_KM_UNFUSED") && atoi(test_env("CNIIC_KM_UNFUSED")));
    {
        auto up = [](uint64_t x) { return (x + 255) & ~255ull; };
        const uint64_t o_part = up(s->res_bytes), o_run = o_part + up(partials_dev ? 0 : W * 8), o_fp = o_run + up(s->cells ? W * 8 : 0);
        const uint64_t o_fr = o_fp + up(s->fused ? 3 * W * 8 : 0), o_fc = o_fr + up(s->fused ? 2 * W * 8 : 0);
        const uint64_t o_mv = o_fc + up(s->fused ? 2 * (uint64_t)K * 4 : 0), total = o_mv + up(s->cells ? ((uint64_t)K + 1) * 4 : 0);
        KM_ALLOC(s->arena, total);
        (void)hipMemsetAsync(s->arena.p, 0, total, c->stream);
        uint8_t *a = s->arena.as<uint8_t>();
        s->resblk.view(a, s->res_bytes);
        if (partials_dev) {
            s->partials = reinterpret_cast<uint64_t *>(partials_dev);
            (void)hipMemsetAsync(s->partials, 0, W * 8, c->stream);
        } else {
            s->partials_own.view(a + o_part, W * 8);
            s->partials = s->partials_own.as<uint64_t>();
        }
        if (s->cells) {
            s->running.view(a + o_run, W * 8);
            s->moved_list.view(a + o_mv, ((uint64_t)K + 1) * 4);
        }
        if (s->fused) {
            s->fused_partials.view(a + o_fp, 3 * W * 8);
            s->fused_running.view(a + o_fr, 2 * W * 8);
            s->fused_cent.view(a + o_fc, 2 * (uint64_t)K * 4);
        }
    }
    s->dstate.view(s->resblk.p, sizeof(KmDevState));
    s->cent.view(static_cast<uint8_t *>(s->resblk.p) + s->res_cent, (uint64_t)K * 4);
    s->members_last.view(static_cast<uint8_t *>(s->resblk.p) + s->res_members, (uint64_t)K * 8);
    s->wsum_last.view(static_cast<uint8_t *>(s->resblk.p) + s->res_wsum, (uint64_t)K * 8);
    hipLaunchKernelGGL(k_rgbw_init_cent, dim3(ceil_div(s->Kpad, 256)), dim3(256), 0, c->stream, keys_d, Ulist, K, s->Kpad, s->idbits,
                       s->cconst.as<uint2>(), s->cent.as<uint32_t>(), s->gidx, s->fused ? s->fused_cent.as<uint32_t>() : (uint32_t *)nullptr,
                       s->cells ? s->moved_list.as<uint32_t>() : (uint32_t *)nullptr, points_dev);
    if (s->cells) {
        // cell-major copy of the whole point list (every rank keeps all U points and works on [lo,hi))
        s->nblocks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(U / nshards, s->wide ? 1024 : 64), 1), s->wide ? 4096u : kCellBlocks);  // same on every shard
        // (CNIIC_OPT_KM_MAX_BLOCKS; batch encodes: a smaller grid per image so that several images' launches share the machine -- a multiple of 3,
        // which the regrouping into 12-wave blocks wants)
        if (const uint64_t mb = c->opt(CNIIC_OPT_KM_MAX_BLOCKS, "CNIIC_KM_MAX_BLOCKS", 0)) {
            const uint32_t cap = std::max(3u, (uint32_t)std::min<uint64_t>(mb, kCellBlocks) / 3 * 3);
            if (!s->wide && nshards == 1 && s->nblocks > cap) s->nblocks = cap;
        }
        if (s->wide) s->nblocks = (s->nblocks + 7) & ~7u;  // (wide: a wave each; launch_assign groups them by up to eight)
        KM_ALLOC(s->labels, std::max<uint64_t>(U, 1) * lab_bytes);
        KM_ALLOC(s->ckeys, U * 4);
        KM_ALLOC(s->cweight, U * 4);
        if (!points_follow) KM_ALLOC(s->crank, U * 4);
        KM_ALLOC(s->cell_start, ((uint64_t)kNumCells + 1) * 4);
        KM_ALLOC(s->ne_cell, (uint64_t)kNumCells * 4);
        KM_ALLOC(s->ne_start, ((uint64_t)kNumCells + 1) * 4);
        KM_ALLOC(s->ne_cost, ((uint64_t)kNumCells + 1) * 4);
        KM_ALLOC(s->ne_count, 4);
        const uint32_t G = s->nblocks * (s->wide ? 1u : (uint32_t)kCellWaves) * nshards;   // waves over all shards
        KM_ALLOC(s->wfirst, ((uint64_t)G + 1) * 4);
        if (!s->big) KM_ALLOC(s->cell_rec, (uint64_t)kNumCells * cell_rec_words((K + 63) / 64) * 4);
        DevBuf count, cursor, cell_tot;
        KM_ALLOC(count, (uint64_t)kNumCells * 4);
        KM_ALLOC(cell_tot, (uint64_t)kCellGroups * 8 * 2);
        uint32_t fixed_cost = kCellFixedCost, sweep_cost = kCellSweepCost;
        if (const char *ev = test_env("CNIIC_CELL_COST")) fixed_cost = (uint32_t)atoi(ev);  // tuning knobs
        if (const char *ev = test_env("CNIIC_CELL_SWEEP_COST")) sweep_cost = (uint32_t)atoi(ev);
        if (points_follow) {
            // the caller's partition (k_points.hip) writes ckeys / cweight / labels itself, from cell_start
            hipLaunchKernelGGL(k_cell_totals, dim3(kCellGroups), dim3(64), 0, c->stream, cell_count_d, cell_tot.as<uint2>());
            hipLaunchKernelGGL(k_cell_scan, dim3(kCellGroups), dim3(64), 0, c->stream, cell_count_d, (const uint2 *)cell_tot.as<uint2>(), s->cell_start.as<uint32_t>(),
                               (uint32_t *)nullptr, s->ne_cell.as<uint32_t>(), s->ne_start.as<uint32_t>(), s->ne_cost.as<uint32_t>(),
                               s->ne_count.as<uint32_t>(), fixed_cost, sweep_cost);
        } else if (rank_table_d) {
            // codec path: the dense colour table (key -> canonical rank + 1) is walked cell by cell
            const uint32_t *cnt = cell_count_d;  // counted by the compaction on its way, else one more walk of the table
            if (!cnt) {
                hipLaunchKernelGGL(k_cells_count_tbl, dim3(kNumCells), dim3(512), 0, c->stream, rank_table_d, count.as<uint32_t>());
                cnt = count.as<uint32_t>();
            }
            hipLaunchKernelGGL(k_cell_totals, dim3(kCellGroups), dim3(64), 0, c->stream, cnt, cell_tot.as<uint2>());
            hipLaunchKernelGGL(k_cell_scan, dim3(kCellGroups), dim3(64), 0, c->stream, cnt, (const uint2 *)cell_tot.as<uint2>(), s->cell_start.as<uint32_t>(),
                               (uint32_t *)nullptr, s->ne_cell.as<uint32_t>(), s->ne_start.as<uint32_t>(), s->ne_cost.as<uint32_t>(),
                               s->ne_count.as<uint32_t>(), fixed_cost, sweep_cost);
            if (s->wide)
                hipLaunchKernelGGL(k_cells_write_tbl<uint16_t>, dim3(kNumCells), dim3(512), 0, c->stream, rank_table_d, weight_d,
                                   s->cell_start.as<uint32_t>(), Ulist, K, s->ckeys.as<uint32_t>(), s->cweight.as<uint32_t>(),
                                   s->crank.as<uint32_t>(), s->labels.as<uint16_t>(), s->gidx);
            else
                hipLaunchKernelGGL(k_cells_write_tbl<uint8_t>, dim3(kNumCells), dim3(512), 0, c->stream, rank_table_d, weight_d,
                                   s->cell_start.as<uint32_t>(), Ulist, K, s->ckeys.as<uint32_t>(), s->cweight.as<uint32_t>(),
                                   s->crank.as<uint32_t>(), s->labels.as<uint8_t>(), s->gidx);
        } else {
            // generic path (any point order): counting sort of the point list by cell
            KM_ALLOC(cursor, (uint64_t)kNumCells * 4);
            (void)hipMemsetAsync(count.p, 0, (uint64_t)kNumCells * 4, c->stream);
            const uint32_t g = grid_1d(U);
            hipLaunchKernelGGL(k_cell_count, dim3(g), dim3(256), 0, c->stream, keys_d, U, count.as<uint32_t>());
            hipLaunchKernelGGL(k_cell_totals, dim3(kCellGroups), dim3(64), 0, c->stream, count.as<uint32_t>(), cell_tot.as<uint2>());
            hipLaunchKernelGGL(k_cell_scan, dim3(kCellGroups), dim3(64), 0, c->stream, count.as<uint32_t>(), (const uint2 *)cell_tot.as<uint2>(), s->cell_start.as<uint32_t>(),
                               cursor.as<uint32_t>(), s->ne_cell.as<uint32_t>(), s->ne_start.as<uint32_t>(), s->ne_cost.as<uint32_t>(),
                               s->ne_count.as<uint32_t>(), fixed_cost, sweep_cost);
            hipLaunchKernelGGL(k_cell_scatter, dim3(g), dim3(256), 0, c->stream, keys_d, weight_d, U, cursor.as<uint32_t>(),
                               s->ckeys.as<uint32_t>(), s->cweight.as<uint32_t>(), s->crank.as<uint32_t>());
            // init_assignment (kmeans.rs:61-78) by canonical rank
            if (s->wide)
                hipLaunchKernelGGL(k_rgbw_init_labels<uint16_t>, dim3(grid_1d(U)), dim3(256), 0, c->stream, s->crank.as<uint32_t>(), U,
                                   (uint64_t)0, U, K, s->labels.as<uint16_t>());
            else
                hipLaunchKernelGGL(k_rgbw_init_labels<uint8_t>, dim3(grid_1d(U)), dim3(256), 0, c->stream, s->crank.as<uint32_t>(), U,
                                   (uint64_t)0, U, K, s->labels.as<uint8_t>());
        }
        s->wfirst_waves = G;   // (the waves' ranges are the launches' business: made when the first of them is enqueued -- ensure_wave_ranges -- not in front of a persistent launch)
        // the loop as one persistent launch (k_kmeans_persist.hip): its block ranges, barrier words and sums
        // (not for the worker contexts of a batch encode: eight persistent launches side by side each hold an eighth of the CUs for a
        // whole run while the other stages of their neighbours' encodes wait for a CU -- 2.6 ms a frame against 0.67 with launches)
        if (s->fused && nshards == 1 && c->ps_div <= 1 && c->opt(CNIIC_OPT_KM_LOOP, "CNIIC_KM_LOOP", 0) == 0) {
            const int rc_ps = ps_prepare(s);
            if (rc_ps != CNIIC_OK) { delete s; return rc_ps; }
        }
        // (count / cursor go back to the context's pool here; it hands them out again in stream order)
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { delete s; return c->fail(CNIIC_ERR_HIP, "kmeans_rgbw setup: %s", hipGetErrorString(e)); }
    } else {
        s->nblocks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, (uint64_t)kAssignThreads * kPPT), 1), kMaxBlocks);
        KM_ALLOC(s->labels, std::max<uint64_t>(n, 1) * lab_bytes);
        KM_ALLOC(s->slabs, (uint64_t)s->nblocks * W * 8);
        if (s->wide)
            hipLaunchKernelGGL(k_rgbw_init_labels<uint16_t>, dim3(grid_1d(n)), dim3(256), 0, c->stream, (const uint32_t *)nullptr, U, lo,
                               hi, K, s->labels.as<uint16_t>());
        else
            hipLaunchKernelGGL(k_rgbw_init_labels<uint8_t>, dim3(grid_1d(n)), dim3(256), 0, c->stream, (const uint32_t *)nullptr, U, lo,
                               hi, K, s->labels.as<uint8_t>());
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { delete s; return c->fail(CNIIC_ERR_HIP, "kmeans_rgbw init launch: %s", hipGetErrorString(e)); }
    *out = s;
    return CNIIC_OK;
}

void km_rgbw_destroy(KmRgbwState *s) { delete s; }

static int launch_update(KmRgbwState *s, int mode) {
    Ctx *c = s->c;
    hipLaunchKernelGGL(k_rgbw_update, dim3(1), dim3(256), 0, c->stream, s->partials, s->running.as<uint64_t>(), mode, s->keys,
                       s->gidx.bits ? s->gidx.U : s->U,
                       s->K, s->idbits, s->seed, s->max_iters, s->cconst.as<uint2>(), s->cent.as<uint32_t>(),
                       s->members_last.as<uint64_t>(), s->wsum_last.as<uint64_t>(), s->cells ? s->moved_list.as<uint32_t>() : nullptr,
                       s->dstate.as<KmDevState>(), s->gidx);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// Kept for ABI stability: the first assign produces full sums, nothing to fold in beforehand.
int km_rgbw_fold_initial(KmRgbwState *) { return CNIIC_OK; }

// Brute path only: explicit centroids + labels (single-step ABI).
int km_rgbw_set_state(KmRgbwState *s, const uint8_t *centroids_h, const uint32_t *labels_d_u32) {
    Ctx *c = s->c;
    if (s->cells) return c->fail(CNIIC_ERR_BAD_ARG, "km_rgbw_set_state needs the brute-force state");
    std::vector<uint32_t> ck(s->K);
    for (uint32_t k = 0; k < s->K; k++)
        ck[k] = ((uint32_t)centroids_h[3 * k] << 16) | ((uint32_t)centroids_h[3 * k + 1] << 8) | centroids_h[3 * k + 2];
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s->cent.p, ck.data(), (size_t)s->K * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_set_cconst, dim3(ceil_div(s->Kpad, 256)), dim3(256), 0, c->stream, s->cent.as<uint32_t>(), s->K,
                       s->Kpad, s->idbits, s->cconst.as<uint2>());
    const uint64_t n = s->hi - s->lo;
    if (n) {
        if (s->wide)
            hipLaunchKernelGGL(k_narrow_labels<uint16_t>, dim3(grid_1d(n)), dim3(256), 0, c->stream, labels_d_u32, s->labels.as<uint16_t>(), n);
        else
            hipLaunchKernelGGL(k_narrow_labels<uint8_t>, dim3(grid_1d(n)), dim3(256), 0, c->stream, labels_d_u32, s->labels.as<uint8_t>(), n);
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));  // ck is a stack-lifetime source
    return CNIIC_OK;
}

// ev_start / ev_stop (profiling): events attached to the dispatch itself (hipExtLaunchKernelGGL), i.e. the kernel's
// own begin and end as a profiler sees them, not an event pair around it (which adds ~4 us of dispatch per launch)
// fused: the update of the previous iteration runs in this launch's prologue and the sums go to part_fused
static void ensure_wave_ranges(KmRgbwState *s) {
    if (!s->wfirst_waves) return;
    hipLaunchKernelGGL(k_wave_ranges, dim3(ceil_div((uint64_t)s->wfirst_waves + 1, 256)), dim3(256), 0, s->c->stream, s->ne_cost.as<uint32_t>(),
                       s->ne_count.as<uint32_t>(), s->wfirst_waves, s->wfirst.as<uint32_t>());
    s->wfirst_waves = 0;
}

static void launch_assign(KmRgbwState *s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const FusedUpdate *fused = nullptr,
                          unsigned long long *part_fused = nullptr) {
    Ctx *c = s->c;
    if (s->cells) ensure_wave_ranges(s);
    FusedUpdate fz{};
    if (fused) fz = *fused;
    const KmDevState *st = s->dstate.as<KmDevState>();
    if (s->cells) {
        auto *part = fused ? part_fused : reinterpret_cast<unsigned long long *>(s->partials);
        CellState cs{s->cell_rec.as<uint32_t>(), s->moved_list.as<uint32_t>(), s->no_skip ? 0u : s->max_skip};
        if (s->big) {   // K > 2048: every colour against every centroid, sums in memory (k_kmeans_wide.hip)
            launch_rgbw_assign_big(c, s->ckeys.as<uint32_t>(), s->cweight.as<uint32_t>(), s->U, s->K, s->cent.as<uint32_t>(), s->labels.as<uint16_t>(), part, st);
        } else if (s->wide) {  // K up to 2048: as many waves per block (8, 4, 2, 1) as leave room for the block's sums and table and every
            // wave's candidate strip in 150 KiB of LDS -- one wave per block (K = 512: five waves per CU, K = 2048: one) took 0.10
            // and 0.69 ms per iteration at 4096^2 against 0.03 at K = 256.  The ranges are per wave: any grouping that divides them.
            auto lds_for = [&](uint32_t wv) {
                return (size_t)s->K * (5 * 8 + 8) + (size_t)wv * (km_scap(s->K) + km_ccap(s->K)) * 8 + (size_t)wv * ((s->K + 63) / 64) * 8;
            };
            uint32_t wv = 8;
            while (wv > 1 && (lds_for(wv) > 150 * 1024 || s->nblocks % wv)) wv >>= 1;
            const size_t lds = lds_for(wv);
            auto kern = wv == 8 ? k_rgbw_assign_cells<uint16_t, 12, 8> : wv == 4 ? k_rgbw_assign_cells<uint16_t, 12, 4>
                        : wv == 2 ? k_rgbw_assign_cells<uint16_t, 12, 2> : k_rgbw_assign_cells<uint16_t, 12, 1>;
            hipLaunchKernelGGL(kern, dim3(s->nblocks / wv), dim3(64 * wv), lds, c->stream,
                               (const uint32_t *)s->ckeys.as<uint32_t>(), (const uint32_t *)s->cweight.as<uint32_t>(), (const uint32_t *)s->ne_cell.as<uint32_t>(),
                               (const uint32_t *)s->ne_start.as<uint32_t>(), (const uint32_t *)s->wfirst.as<uint32_t>(), s->shard, s->K,
                               (const uint2 *)s->cconst.as<uint2>(), s->labels.as<uint16_t>(), part, st, cs, fz);
        } else {
            // The same wave ranges in blocks of 12 waves, two per CU, once the run has settled (from launch big_blocks_from):
            // measured on the headline encode, the first ten launches are 5-13 us faster in blocks of 8 (three per CU), every later
            // one 1-3 us faster in blocks of 12.  (The ranges are per wave, so regrouping them needs nothing but a multiple of 12.)
            const bool big = kCellWaves == 8 && fz.on && fz.launch_no >= s->big_blocks_from && (s->nblocks * (uint32_t)kCellWaves) % kCellWavesBig == 0;
            const uint32_t wpb = big ? kCellWavesBig : (uint32_t)kCellWaves, nblk = s->nblocks * (uint32_t)kCellWaves / wpb;
            const size_t lds = (size_t)s->K * (5 * 8 + 8) + (size_t)wpb * (km_scap(s->K) + km_ccap(s->K)) * 8 + (size_t)wpb * ((s->K + 63) / 64) * 8;
            auto kern = big ? k_rgbw_assign_cells<uint8_t, 8, kCellWavesBig, 0>
                        : !fz.on ? k_rgbw_assign_cells<uint8_t, 8, kCellWaves, -1>
                        : fz.launch_no == 0 ? k_rgbw_assign_cells<uint8_t, 8, kCellWaves, 1>
                        : fz.launch_no <= s->agg_launches ? k_rgbw_assign_cells<uint8_t, 8, kCellWaves, 2>
                        : k_rgbw_assign_cells<uint8_t, 8, kCellWaves, 0>;
            hipExtLaunchKernelGGL(kern, dim3(nblk), dim3(64 * wpb), (uint32_t)lds,
                                  c->stream, ev_start, ev_stop, 0, (const uint32_t *)s->ckeys.as<uint32_t>(),
                                  (const uint32_t *)s->cweight.as<uint32_t>(), (const uint32_t *)s->ne_cell.as<uint32_t>(),
                                  (const uint32_t *)s->ne_start.as<uint32_t>(), (const uint32_t *)s->wfirst.as<uint32_t>(), s->shard, s->K,
                                  (const uint2 *)s->cconst.as<uint2>(), s->labels.as<uint8_t>(),
                                  part, st, cs, fz);
        }
        return;
    }
    const size_t lds = (size_t)s->K * (5 * 8 + 8);
    if (s->wide)
        hipLaunchKernelGGL((k_rgbw_assign<uint16_t, 12>), dim3(s->nblocks), dim3(kAssignThreads), lds, c->stream, s->keys,
                           s->weight, s->lo, s->hi, s->K, s->Kpad, s->cconst.as<uint2>(), s->labels.as<uint16_t>(),
                           s->slabs.as<uint64_t>(), st);
    else
        hipLaunchKernelGGL((k_rgbw_assign<uint8_t, 8>), dim3(s->nblocks), dim3(kAssignThreads), lds, c->stream, s->keys,
                           s->weight, s->lo, s->hi, s->K, s->Kpad, s->cconst.as<uint2>(), s->labels.as<uint8_t>(),
                           s->slabs.as<uint64_t>(), st);
}

int km_rgbw_assign(KmRgbwState *s) {
    Ctx *c = s->c;
    launch_assign(s);
    if (!s->cells) {
        const uint32_t W = 5 * s->K + 2;
        const uint32_t ry = std::max(1u, std::min(16u, s->nblocks / 4));
        hipLaunchKernelGGL(k_slab_reduce, dim3(ceil_div(W, 64), ry), dim3(256), 0, c->stream, s->slabs.as<uint64_t>(),
                           s->nblocks, W, s->partials, s->dstate.as<KmDevState>());
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int km_rgbw_update(KmRgbwState *s) { return launch_update(s, s->cells ? 1 : 0); }

static int read_state(KmRgbwState *s, KmDevState *h) {
    Ctx *c = s->c;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(h, s->dstate.p, sizeof(KmDevState), hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

// the statistics of the last km_rgbw_run without touching the stream (false: no run has finished on this state)
bool km_rgbw_run_stats(KmRgbwState *s, cniic_kmeans_stats *st) {
    if (!s->run_stats_valid) return false;
    *st = s->run_stats;
    return true;
}

int km_rgbw_poll(KmRgbwState *s, cniic_kmeans_stats *st, uint32_t *done) {
    KmDevState h;
    CNIIC_TRY(read_state(s, &h));
    st->iterations = h.iter; st->moved_last = h.moved_last; st->empty_reseeds = h.reseeds; st->active = h.active; st->pair_evals = h.pair_evals;
    *done = h.done;
    return CNIIC_OK;
}

// The state as of the PREVIOUS call (have = false on the first one): the copy enqueued by this call is waited for
// by the next, so a caller that polls after every batch of iterations never makes the GPU wait for the host.
int km_rgbw_poll_lagged(KmRgbwState *s, cniic_kmeans_stats *st, uint32_t *done, uint32_t *have) {
    if (!s->lagged) {   // kept only once prepare() has succeeded: a poll refused by the owner guard must not leave a half-made object
        auto lp = std::make_unique<LaggedPoll>(s->c, s->dstate.p);   // behind that the next call would use without owning the context's slots
        CNIIC_TRY(lp->prepare());
        s->lagged = std::move(lp);
    }
    KmDevState h{};
    bool got = false;
    CNIIC_TRY(s->lagged->after_batch(&h, &got));
    *have = got ? 1u : 0u;
    *done = got ? h.done : 0u;
    if (got) { st->iterations = h.iter; st->moved_last = h.moved_last; st->empty_reseeds = h.reseeds; st->active = h.active; st->pair_evals = h.pair_evals; }
    return CNIIC_OK;
}

int km_rgbw_poll_changed(KmRgbwState *s, uint64_t *changed) {
    KmDevState h;
    CNIIC_TRY(read_state(s, &h));
    *changed = h.moved_last;
    return CNIIC_OK;
}

// per-launch HIP event pairs (profiling only): sum of the assign kernel's own durations
struct LaunchTimer {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    ~LaunchTimer() { for (auto e : ev) (void)hipEventDestroy(e); }
    hipEvent_t next() {
        if (used == ev.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev.push_back(e); }
        return ev[used++];
    }
    double total_ms() {
        double t = 0;
        for (size_t i = 0; i + 1 < used; i += 2) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); t += ms; }
        return t;
    }
};

// Whole loop on one GPU: iterations are enqueued in batches with no host round trip inside a
// batch; kernels of iterations past convergence exit on the device-side `done` flag, so the
// result is exactly that of the reference's `while changed_assignment` loop (kmeans.rs:26-32).
// With CNIIC_KM_PROFILE every assign launch is bracketed by HIP events on the ctx stream and the
// summed kernel time is reported as "kmeans_rgbw_assign" (launch count = iterations).
static int km_rgbw_run_loop(KmRgbwState *s, Comm *cm, bool may_defer);

// With a communicator, a failure on this rank (a launch error, a collective that fails, a missing state record) must not
// strand the peers in their next all-reduce: the communicator is aborted before the error is returned (comm.cpp).
int km_rgbw_run(KmRgbwState *s, Comm *cm, bool may_defer) {
    const int rc = km_rgbw_run_loop(s, cm, may_defer);
    if (rc != CNIIC_OK && cm) comm_abort(cm);
    return rc;
}

static int km_rgbw_run_loop(KmRgbwState *s, Comm *cm, bool may_defer) {
    Ctx *c = s->c;
    int batch = test_env("CNIIC_KM_BATCH") ? atoi(test_env("CNIIC_KM_BATCH")) : cm ? 2 : 8;  // iterations enqueued between two looks at the state; with collectives an iteration past convergence still
                                   // pays a full all-reduce, so fewer are in flight (and each is long enough for the host to keep up)
    const bool batch_fixed = test_env("CNIIC_KM_BATCH") != nullptr || cm != nullptr;
    KmDevState h;
    LaunchTimer lt;
    LaggedPoll poll(c, s->dstate.p);
    CNIIC_TRY(poll.prepare());
    poll.watch = cm;
    [[maybe_unused]] const long fail_at = s->fail_at;  // fault injection (testing build only): this rank fails before enqueuing launch n
    KmDevState *st_host = nullptr;
    // The mapped slot shows the host a state AT LEAST as new as the batch it asks about -- how much newer depends on timing.
    // Alone that only ends the loop a little earlier; with collectives every rank must leave after the SAME batch (a rank
    // that enqueues one more all-reduce than its peers waits for them forever), so there the state is the in-stream copy
    // taken at a fixed place of the sequence, identical on all ranks.
    if (s->fused && !cm) CNIIC_TRY(poll.mapped_slot(&st_host));
    PollRec *st_ring = nullptr;
    if (s->fused && cm) CNIIC_TRY(poll.ring_slot(&st_ring));  // (unfused loops with collectives: the in-stream copy)
    ScopedKernelTimer timer(c, "kmeans_rgbw_iter", s->profile);  // (its stop() synchronises: profiling runs only)
    const uint64_t W = 5 * (uint64_t)s->K + 2;
    uint32_t launch_no = 0;
    if (!cm && s->ps && s->fused) {   // one GPU, K <= 256: the whole loop as ONE launch (k_kmeans_persist.hip)
        bool ran = false;
        CNIIC_TRY(km_rgbw_run_persistent(s, &ran, may_defer));
        if (ran) { timer.stop(s->run_stats.iterations); return CNIIC_OK; }
    }
    for (;;) {
        for (int b = 0; b < batch; b++) {
            if (s->fused) {
                // assign j with update j - 1 in its prologue; the sums are triple-buffered, running sums and the
                // centroids to compare with ping-pong (see FusedUpdate)
                const uint32_t j = launch_no++;
#ifdef CNIIC_TESTING
                if (fail_at >= 0 && (long)j == fail_at) return c->fail(CNIIC_ERR_HIP, "injected failure before launch %u (CNIIC_TEST_FAIL_AT_LAUNCH)", j);
#endif
                auto *P = s->fused_partials.as<unsigned long long>();
                auto *Rn = s->fused_running.as<unsigned long long>();
                auto *Cn = s->fused_cent.as<uint32_t>();
                FusedUpdate fz{};
                fz.on = 1; fz.launch_no = j; fz.idbits = s->idbits; fz.max_iters = s->max_iters; fz.seed = s->seed; fz.U = s->gidx.bits ? s->gidx.U : s->U; fz.gx = s->gidx;
                fz.keys = s->keys;
                fz.partials_prev = P + ((j + 2) % 3) * W;
                fz.partials_clear = P + ((j + 1) % 3) * W;
                fz.running_prev = Rn + ((j + 1) % 2) * W;
                fz.running_new = Rn + (j % 2) * W;
                fz.cent_prev = Cn + ((j + 1) % 2) * (size_t)s->K;
                fz.cent_new = Cn + (j % 2) * (size_t)s->K;
                fz.cconst_g = s->cconst.as<uint2>();
                fz.cent_g = s->cent.as<uint32_t>();
                fz.members_out = s->members_last.as<uint64_t>();
                fz.wsum_out = s->wsum_last.as<uint64_t>();
                fz.st_rw = s->dstate.as<KmDevState>();
                fz.st_host = st_host;
                fz.st_ring = st_ring;
                unsigned long long *cur = P + (j % 3) * W;
                if (s->profile) { hipEvent_t ea = lt.next(), eb = lt.next(); launch_assign(s, ea, eb, &fz, cur); }
                else launch_assign(s, nullptr, nullptr, &fz, cur);
                if (cm) CNIIC_TRY(comm_all_reduce(cm, cur, W, 2));  // the sums of all shards, before launch j + 1 reads them
                continue;
            }
            if (s->profile && s->cells && !s->wide) {
                hipEvent_t a = lt.next(), b = lt.next();
                launch_assign(s, a, b);
            } else {
                if (s->profile) (void)hipEventRecord(lt.next(), c->stream);
                launch_assign(s);
                if (s->profile) (void)hipEventRecord(lt.next(), c->stream);
            }
            if (!s->cells) {
                const uint32_t W = 5 * s->K + 2;
                const uint32_t ry = std::max(1u, std::min(16u, s->nblocks / 4));
                hipLaunchKernelGGL(k_slab_reduce, dim3(ceil_div(W, 64), ry), dim3(256), 0, c->stream, s->slabs.as<uint64_t>(),
                                   s->nblocks, W, s->partials, s->dstate.as<KmDevState>());
            }
            // several GPUs: the K partial sums (and the moved count) of all shards, summed in place on this stream;
            // integer sums, so every rank updates to identical centroids and sees the same convergence flag
            if (cm) CNIIC_TRY(comm_all_reduce(cm, s->partials, 5 * (uint64_t)s->K + 2, 2));
            CNIIC_TRY(km_rgbw_update(s));
        }
        CNIIC_HIP_TRY(c, hipGetLastError());
        bool have = false;
        CNIIC_TRY(poll.after_batch(&h, &have, launch_no ? launch_no - 1 : 0));
        if (have && h.done) break;
        // The tail of a run (a few thousand points still moving) is launches of ~11 us that do little: four in flight keep the GPU fed
        // as well as eight, and up to eight fewer launches past convergence (4 us each) are paid at the end.
        if (have && !batch_fixed && h.iter > 8) batch = h.moved_last < 20000 ? 4 : 8;
    }
    // (the state the loop ended on is final -- launches past convergence change nothing: callers that only want the statistics
    // need not wait for those launches, km_rgbw_run_stats)
    s->run_stats.iterations = h.iter; s->run_stats.moved_last = h.moved_last; s->run_stats.empty_reseeds = h.reseeds;
    s->run_stats.active = h.active; s->run_stats.pair_evals = h.pair_evals;
    s->run_stats_valid = true;
    timer.stop(h.iter);
    if (s->profile) {
        // every launch that was issued counts, including the (at most batch-1) launches after
        // convergence that exit on the device-side flag: the same population rocprofv3 --stats averages
        KernelTime &kt = c->ktimes["kmeans_rgbw_assign"];
        kt.ms += lt.total_ms();
        kt.launches += lt.used / 2;
        // launches that did an iteration's work, apart from the (at most batch - 1) that found the loop converged
        const size_t working = std::min<size_t>(lt.used / 2, (size_t)h.iter + (s->fused ? 1 : 0));
        double wms = 0;
        for (size_t i = 0; i < working; i++) { float ms = 0.f; (void)hipEventElapsedTime(&ms, lt.ev[2 * i], lt.ev[2 * i + 1]); wms += ms; }
        KernelTime &kw = c->ktimes["kmeans_rgbw_assign_working"];
        kw.ms += wms;
        kw.launches += working;
        // per class of launch: iteration 0 (every point adds to the sums), full schedule (more than kMaxMovedSkip centroids moved in
        // the update before it), skip schedule, the launch that only finishes the last iteration, launches past convergence
        KmDevState hf;
        CNIIC_TRY(read_state(s, &hf));  // (the polled copy carries the scalars only)
        const char *names[5] = {"first", "full", "skip", "final-update", "no-op"};
        double cms[5] = {0, 0, 0, 0, 0};
        uint64_t cn[5] = {0, 0, 0, 0, 0};
        FILE *f = nullptr;
        if (const char *path = test_env("CNIIC_KM_LAUNCH_TRACE")) f = fopen(path, "w");  // one line per launch: number, duration, class, what the iteration before it moved
        if (f) fprintf(f, "launch,us,class,centroids_moved_before,points_moved\n");
        for (size_t i = 0; i < lt.used / 2; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, lt.ev[2 * i], lt.ev[2 * i + 1]);
            const bool prev_in_ring = i >= 1 && i - 1 < hf.iter && hf.iter - (i - 1) <= kHistRing;
            const long long nmv = prev_in_ring ? (long long)hf.nmoved_ring[(i - 1) % kHistRing] : -1;
            const bool in_ring = i < hf.iter && hf.iter - i <= kHistRing;
            int cls = i >= working ? 4 : i == 0 ? 0 : (s->fused && i == (size_t)hf.iter) ? 3 : (nmv >= 0 && nmv <= (long long)s->max_skip && !s->no_skip) ? 2 : 1;
            cms[cls] += ms; cn[cls]++;
            if (f) fprintf(f, "%zu,%.2f,%s,%lld,%lld\n", i, ms * 1e3, names[cls], nmv, in_ring ? (long long)hf.changed_ring[i % kHistRing] : -1ll);
        }
        if (f) fclose(f);
        for (int k = 0; k < 5; k++) {
            KernelTime &kc = c->ktimes[std::string("kmeans_rgbw_assign_") + names[k]];
            kc.ms += cms[k]; kc.launches += cn[k];
        }
    }
    return CNIIC_OK;
}

// Average duration of the assign kernel alone over `reps` launches on the current state (HIP event
// pair per launch on the ctx stream).  Labels may move towards the fixed point of the current
// centroids; the partial sums are left inconsistent, so the state must be discarded afterwards.
int km_rgbw_time_assign(KmRgbwState *s, int reps, double *ms_per_launch) {
    Ctx *c = s->c;
    LaunchTimer lt;
    for (int i = 0; i <= reps; i++) {  // launch 0 is a warm-up
        hipEvent_t a = lt.next(), b = lt.next();
        CNIIC_HIP_TRY(c, hipEventRecord(a, c->stream));
        launch_assign(s);
        CNIIC_HIP_TRY(c, hipEventRecord(b, c->stream));
    }
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    CNIIC_HIP_TRY(c, hipGetLastError());
    double t = 0;
    for (size_t i = 2; i + 1 < lt.used; i += 2) { float ms = 0.f; (void)hipEventElapsedTime(&ms, lt.ev[i], lt.ev[i + 1]); t += ms; }
    *ms_per_launch = t / reps;
    return CNIIC_OK;
}

int km_rgbw_partials(KmRgbwState *s, uint64_t *sums_h, uint64_t *wsum_h, uint64_t *members_h, uint64_t *changed_h) {
    Ctx *c = s->c;
    const size_t K = s->K;
    std::vector<uint64_t> p(5 * K + 2);
    CNIIC_HIP_TRY(c, hipMemcpyAsync(p.data(), s->partials, p.size() * 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (sums_h) memcpy(sums_h, p.data(), 3 * K * 8);
    if (wsum_h) memcpy(wsum_h, p.data() + 3 * K, K * 8);
    if (members_h) memcpy(members_h, p.data() + 4 * K, K * 8);
    if (changed_h) *changed_h = p[5 * K];
    return CNIIC_OK;
}

void *km_rgbw_partials_dev(KmRgbwState *s) { return s->partials; }
const uint32_t *km_rgbw_centroids_dev(KmRgbwState *s) { return s->cent.as<uint32_t>(); }  // [K] 0xRRGGBB, as of the last update
bool km_rgbw_is_wide(KmRgbwState *s) { return s->wide; }

// u8/u16 labels in CANONICAL point order on the device (dst holds U entries)
int km_rgbw_labels_canonical(KmRgbwState *s, void *dst_d) {
    Ctx *c = s->c;
    if (s->cells) {
        if (s->wide)
            hipLaunchKernelGGL(k_labels_to_canonical<uint16_t>, dim3(grid_1d(s->U)), dim3(256), 0, c->stream, s->labels.as<uint16_t>(),
                               s->crank.as<uint32_t>(), reinterpret_cast<uint16_t *>(dst_d), s->U);
        else
            hipLaunchKernelGGL(k_labels_to_canonical<uint8_t>, dim3(grid_1d(s->U)), dim3(256), 0, c->stream, s->labels.as<uint8_t>(),
                               s->crank.as<uint32_t>(), reinterpret_cast<uint8_t *>(dst_d), s->U);
        CNIIC_HIP_TRY(c, hipGetLastError());
        return CNIIC_OK;
    }
    CNIIC_HIP_TRY(c, hipMemcpyAsync(dst_d, s->labels.p, (s->hi - s->lo) * (s->wide ? 2 : 1), hipMemcpyDeviceToDevice, c->stream));
    return CNIIC_OK;
}

// Sharded runs: copy this shard's labels (cell-major positions of its waves' cells) into dst,
// zero elsewhere; the caller sums the exports of all ranks (each position has one owner).
template <typename LabelT>
__global__ void k_export_labels(const LabelT *__restrict__ labels, const uint32_t *__restrict__ ne_start,
                                const uint32_t *__restrict__ wfirst, uint32_t g_lo, uint32_t g_hi, uint64_t U,
                                LabelT *__restrict__ dst) {
    const uint32_t q_lo = ne_start[wfirst[g_lo]], q_hi = ne_start[wfirst[g_hi]];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride)
        dst[i] = (i >= q_lo && i < q_hi) ? labels[i] : (LabelT)0;
}

int km_rgbw_export_labels(KmRgbwState *s, void *dst_d) {
    Ctx *c = s->c;
    if (!s->cells) return c->fail(CNIIC_ERR_BAD_ARG, "export_labels needs the cells path");
    ensure_wave_ranges(s);
    const uint32_t wpb = s->wide ? 1u : (uint32_t)kCellWaves;
    const uint32_t g_lo = s->shard * s->nblocks * wpb, g_hi = (s->shard + 1) * s->nblocks * wpb;
    if (s->wide)
        hipLaunchKernelGGL(k_export_labels<uint16_t>, dim3(grid_1d(s->U)), dim3(256), 0, c->stream, s->labels.as<uint16_t>(),
                           s->ne_start.as<uint32_t>(), s->wfirst.as<uint32_t>(), g_lo, g_hi, s->U, reinterpret_cast<uint16_t *>(dst_d));
    else
        hipLaunchKernelGGL(k_export_labels<uint8_t>, dim3(grid_1d(s->U)), dim3(256), 0, c->stream, s->labels.as<uint8_t>(),
                           s->ne_start.as<uint32_t>(), s->wfirst.as<uint32_t>(), g_lo, g_hi, s->U, reinterpret_cast<uint8_t *>(dst_d));
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int km_rgbw_import_labels(KmRgbwState *s, const void *src_d) {
    Ctx *c = s->c;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s->labels.p, src_d, s->U * (s->wide ? 2 : 1), hipMemcpyDeviceToDevice, c->stream));
    return CNIIC_OK;
}

uint64_t km_rgbw_points(KmRgbwState *s) { return s->U; }

// a state created with an upper bound (points_dev): the number of points, now that the host knows it
int km_rgbw_set_points(KmRgbwState *s, uint64_t U, uint64_t Ulist) {
    if (Ulist == 0) Ulist = U;
    if (U == 0 || U > s->U || Ulist / s->K == 0)
        return s->c->fail(CNIIC_ERR_TOO_FEW_POINTS, "kmeans: %llu points for %u clusters (src/kmeans.rs:68)", (unsigned long long)Ulist, s->K);
    s->U = U; s->hi = U;
    if (s->gidx.bits) s->gidx.U = Ulist;
    return CNIIC_OK;
}

// the cell-major arrays (cells path): first position of every cell, colours, pixel counts
void km_rgbw_cell_arrays(KmRgbwState *s, uint32_t **cell_start_d, uint32_t **ckeys_d, uint32_t **cweight_d) {
    *cell_start_d = s->cell_start.as<uint32_t>();
    *ckeys_d = s->ckeys.as<uint32_t>();
    *cweight_d = s->cweight.as<uint32_t>();
}

// cell-major label array of all U points (cells path): the buffer ranks all-gather over
void *km_rgbw_labels_internal(KmRgbwState *s, uint64_t *elem_bytes) {
    if (elem_bytes) *elem_bytes = s->wide ? 2 : 1;
    return s->labels.p;
}

// labels_d_u32: cells path -> all U labels in canonical order (only [lo,hi) of the internal order
// are meaningful unless the caller all-gathered the internal array); brute path -> slice [lo,hi).
int km_rgbw_result(KmRgbwState *s, uint8_t *centroids_h, uint32_t *labels_d_u32, uint64_t *members_h,
                   uint64_t *wsum_h, cniic_kmeans_stats *stats) {
    Ctx *c = s->c;
    const uint64_t n = s->hi - s->lo;
    if (labels_d_u32 && n) {
        const uint32_t *rank = s->cells ? s->crank.as<uint32_t>() : nullptr;
        const uint64_t a = s->cells ? 0 : s->lo, b = s->cells ? s->U : s->hi;
        if (s->wide)
            hipLaunchKernelGGL(k_widen_labels<uint16_t>, dim3(grid_1d(b - a)), dim3(256), 0, c->stream, s->labels.as<uint16_t>(), rank, labels_d_u32, a, b);
        else
            hipLaunchKernelGGL(k_widen_labels<uint8_t>, dim3(grid_1d(b - a)), dim3(256), 0, c->stream, s->labels.as<uint8_t>(), rank, labels_d_u32, a, b);
        CNIIC_HIP_TRY(c, hipGetLastError());
    }
    CNIIC_TRY(km_rgbw_result_begin(s));
    int rc = km_rgbw_result_end(s, centroids_h, members_h, wsum_h, stats);
    if (rc == kKmRetry) {   // (only after a deferred run, which this file's callers of km_rgbw_result do not ask for)
        CNIIC_TRY(km_rgbw_result_begin(s));
        rc = km_rgbw_result_end(s, centroids_h, members_h, wsum_h, stats);
    }
    return rc;
}

// state, centroids, members, weight sums are one block: one copy into pinned memory, one event
int km_rgbw_result_begin(KmRgbwState *s) {
    Ctx *c = s->c;
    if (c->pinned_res_bytes < s->res_bytes) {
        if (c->pinned_res) CNIIC_HIP_TRY(c, hipHostFree(c->pinned_res));
        c->pinned_res = nullptr; c->pinned_res_bytes = 0;
        CNIIC_HIP_TRY(c, hipHostMalloc(&c->pinned_res, s->res_bytes, hipHostMallocDefault));
        c->pinned_res_bytes = s->res_bytes;
    }
    if (!c->res_ev) CNIIC_HIP_TRY(c, hipEventCreateWithFlags(&c->res_ev, hipEventDisableTiming));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(c->pinned_res, s->resblk.p, s->res_bytes, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipEventRecord(c->res_ev, c->stream));
    return CNIIC_OK;
}

int km_rgbw_result_end(KmRgbwState *s, uint8_t *centroids_h, uint64_t *members_h, uint64_t *wsum_h, cniic_kmeans_stats *stats) {
    Ctx *c = s->c;
    CNIIC_HIP_TRY(c, hipEventSynchronize(c->res_ev));
    if (s->ps_pending) {   // the persistent launch nobody waited for: did it run to the end?  If not, the loop of launches has run by now (from the untouched inputs)
        bool retry = false;
        CNIIC_TRY(km_rgbw_persistent_verdict(s, &retry));
        if (retry) return kKmRetry;
    }
    const uint8_t *blk = static_cast<const uint8_t *>(c->pinned_res);
    KmDevState h;
    memcpy(&h, blk, sizeof h);
    const uint32_t *ck = reinterpret_cast<const uint32_t *>(blk + s->res_cent);
    if (centroids_h)
        for (uint32_t k = 0; k < s->K; k++) {
            centroids_h[3 * k] = (uint8_t)(ck[k] >> 16); centroids_h[3 * k + 1] = (uint8_t)(ck[k] >> 8); centroids_h[3 * k + 2] = (uint8_t)ck[k];
        }
    if (members_h) memcpy(members_h, blk + s->res_members, (size_t)s->K * 8);
    if (wsum_h) memcpy(wsum_h, blk + s->res_wsum, (size_t)s->K * 8);
    if (stats) {
        stats->iterations = h.iter;
        stats->moved_last = h.moved_last;
        stats->empty_reseeds = h.reseeds;
        stats->active = h.active;
        stats->pair_evals = h.pair_evals;
    }
    return CNIIC_OK;
}

}  // namespace cniic
