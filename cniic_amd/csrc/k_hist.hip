// k_hist.hip -- utils::count_freqs (reference src/utils.rs:4-16) on gfx950.
//
// The reference counts symbols in a HashMap.  Here the alphabet is a packed integer key
// (24-bit RGB, 27-bit signed delta), so the map is a DENSE u32 table resident in HBM
// (64 MiB / 512 MiB; sized for 288 GB parts), filled with no-return global atomics and then
// compacted to ascending (key,count) pairs by a 3-phase scan.  The compaction leaves
// rank+1 in every occupied bin, which later stages (remap, Huffman bit-pack) use as the
// symbol -> code-table index.
//
// Roofline: the fill pass is HBM-bound on its input (3 B/px RGB or 4 B/symbol); the table clear
// and compaction stream the table twice (reported separately in DESIGN.md).
#include <cstdlib>

#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

// ---------------------------------------------------------------- fill
// 16 pixels (48 B) per thread per step: three 16-B loads, keys extracted in registers.
__global__ __launch_bounds__(256) void k_hist_rgb(const uint8_t *__restrict__ rgb, uint64_t npx,
                                                  uint32_t *__restrict__ table) {
    const uint64_t ngroups = npx / 16;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint4 *v = reinterpret_cast<const uint4 *>(rgb);
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        uint32_t key[16];
        load16px_keys(v + 3 * g, key);
#pragma unroll
        for (int i = 0; i < 16; i++) atomic_count(table, key[i]);
    }
    // tail (< 16 px) by the first threads of block 0
    if (blockIdx.x == 0) {
        uint64_t i = ngroups * 16 + threadIdx.x;
        if (i < npx) atomic_count(table, rgb_key(rgb + 3 * i));
    }
}

// ---------------------------------------------------------------- fill by partition (large images)
// A random atomic on the 64 MiB table moves a whole cache line to the atomic unit and back: ~27 G
// atomics/s however they are scoped (measured: device scope and XCD-local scope cost the same), i.e.
// 630 us for 4096^2 pixels.  Instead the pixels are first PARTITIONED by the top 12 bits of their key
// (4096 buckets = the 4096-entry slices of the table) with LDS counters and cursors, then every bucket's
// 12-bit remainders are counted in an LDS histogram and the slice is written once:
//   k_part_count    per-block bucket counts (LDS atomics)            reads 3 B/px
//   k_part_colscan  per-bucket exclusive prefix over the blocks      8 MiB
//   k_part_starts   bucket starts + work items of the last pass      16 KiB
//   k_part_scatter  remainder (u16) of every pixel to its bucket     reads 3 B/px, writes 2 B/px
//   k_part_hist     LDS histogram per bucket (or per 16 Ki entries of a crowded one) -> table slice
// Counts are exact and order-free, so the result is the table the atomics would have produced.
constexpr int kPartBits = 12;
constexpr uint32_t kPartBuckets = 1u << (24 - kPartBits);   // 4096
constexpr uint32_t kPartBins = 1u << kPartBits;              // 4096 table entries per bucket
constexpr uint32_t kPartBlocks = 512;                        // blocks of the two passes over the image
constexpr uint32_t kPartSub = 16384;                         // entries per work item of k_part_hist
constexpr int kPartCols = 32;                                // buckets per block of k_part_colscan

// block b owns the 16-pixel groups [b gpb, (b+1) gpb); the last block also owns the < 16 tail pixels
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_part_pass(const uint8_t *__restrict__ rgb, uint64_t npx, uint64_t ngroups, uint64_t gpb,
                                                   uint32_t *__restrict__ counts, const uint32_t *__restrict__ start,
                                                   uint16_t *__restrict__ payload) {
    __shared__ uint32_t cur[kPartBuckets];
    uint32_t *mine = counts + (size_t)blockIdx.x * kPartBuckets;
    for (uint32_t k = threadIdx.x; k < kPartBuckets; k += 256) cur[k] = SCATTER ? start[k] + mine[k] : 0u;
    __syncthreads();
    const uint64_t g0 = blockIdx.x * gpb, g1 = min(g0 + gpb, ngroups);
    const uint4 *v = reinterpret_cast<const uint4 *>(rgb);
    for (uint64_t g = g0 + threadIdx.x; g < g1; g += 256) {
        uint32_t key[16];
        load16px_keys(v + 3 * g, key);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (SCATTER) payload[atomicAdd(&cur[key[i] >> kPartBits], 1u)] = (uint16_t)(key[i] & (kPartBins - 1));
            else atomicAdd(&cur[key[i] >> kPartBits], 1u);
        }
    }
    if (blockIdx.x == gridDim.x - 1) {
        const uint64_t i = ngroups * 16 + threadIdx.x;
        if (i < npx) {
            const uint32_t key = rgb_key(rgb + 3 * i);
            if (SCATTER) payload[atomicAdd(&cur[key >> kPartBits], 1u)] = (uint16_t)(key & (kPartBins - 1));
            else atomicAdd(&cur[key >> kPartBits], 1u);
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < kPartBuckets; k += 256) mine[k] = cur[k];
    }
}

// counts[b][k] -> sum over b' < b of counts[b'][k]; total[k] = column sum.  One block per kPartCols buckets.
__global__ __launch_bounds__(256) void k_part_colscan(uint32_t *__restrict__ counts, uint32_t *__restrict__ total) {
    __shared__ uint32_t tile[kPartBlocks][kPartCols];        // 64 KiB
    __shared__ uint32_t part[256 / kPartCols][kPartCols];
    const uint32_t k0 = blockIdx.x * kPartCols;
    for (uint32_t idx = threadIdx.x; idx < kPartBlocks * kPartCols; idx += 256)
        tile[idx / kPartCols][idx % kPartCols] = counts[(size_t)(idx / kPartCols) * kPartBuckets + k0 + idx % kPartCols];
    __syncthreads();
    constexpr uint32_t segs = 256 / kPartCols, rows = kPartBlocks / segs;   // 8 segments of 64 rows per column
    const uint32_t j = threadIdx.x % kPartCols, sgm = threadIdx.x / kPartCols;
    uint32_t sum = 0;
    for (uint32_t r = sgm * rows; r < (sgm + 1) * rows; r++) sum += tile[r][j];
    part[sgm][j] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t q = 0; q < sgm; q++) run += part[q][j];
    for (uint32_t r = sgm * rows; r < (sgm + 1) * rows; r++) { const uint32_t c = tile[r][j]; tile[r][j] = run; run += c; }
    if (sgm == segs - 1) total[k0 + j] = run;
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < kPartBlocks * kPartCols; idx += 256)
        counts[(size_t)(idx / kPartCols) * kPartBuckets + k0 + idx % kPartCols] = tile[idx / kPartCols][idx % kPartCols];
}

// exclusive scans over the buckets: start[k] (entries) and item[k] (work items of k_part_hist); [kPartBuckets] = totals
__global__ __launch_bounds__(1024) void k_part_starts(const uint32_t *__restrict__ total, uint32_t *__restrict__ start,
                                                      uint32_t *__restrict__ item) {
    __shared__ uint32_t wsum[1024 / 64];
    constexpr uint32_t per = kPartBuckets / 1024;
    uint32_t t[per], e = 0, w = 0;
#pragma unroll
    for (uint32_t i = 0; i < per; i++) { t[i] = total[threadIdx.x * per + i]; e += t[i]; w += (t[i] + kPartSub - 1) / kPartSub; }
    uint32_t es = block_exclusive_scan<1024>(e, wsum);
    uint32_t ws = block_exclusive_scan<1024>(w, wsum);
#pragma unroll
    for (uint32_t i = 0; i < per; i++) {
        start[threadIdx.x * per + i] = es; item[threadIdx.x * per + i] = ws;
        es += t[i]; ws += (t[i] + kPartSub - 1) / kPartSub;
    }
    if (threadIdx.x == 1023) { start[kPartBuckets] = es; item[kPartBuckets] = ws; }
}

__global__ __launch_bounds__(256) void k_part_hist(const uint16_t *__restrict__ payload, const uint32_t *__restrict__ start,
                                                   const uint32_t *__restrict__ item, uint32_t *__restrict__ table) {
    __shared__ uint32_t hist[kPartBins];
    const uint32_t it = blockIdx.x;
    if (it >= item[kPartBuckets]) return;
    uint32_t a = 0, b = kPartBuckets;  // last bucket whose first item is <= it (buckets without entries have no items)
    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (item[mid] <= it) a = mid; else b = mid; }
    const uint32_t k = a, nsub = item[k + 1] - item[k];
    const uint32_t e0 = start[k] + (it - item[k]) * kPartSub, e1 = min(e0 + kPartSub, start[k + 1]);
    for (uint32_t i = threadIdx.x; i < kPartBins; i += 256) hist[i] = 0;
    __syncthreads();
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += 256) atomicAdd(&hist[payload[e]], 1u);
    __syncthreads();
    uint32_t *slice = table + ((size_t)k << kPartBits);
    for (uint32_t i = threadIdx.x; i < kPartBins; i += 256) {
        const uint32_t cnt = hist[i];
        if (cnt) { if (nsub == 1) slice[i] = cnt; else atomicAdd(&slice[i], cnt); }
    }
}

// unaligned base pointer: plain per-pixel byte loads
__global__ __launch_bounds__(256) void k_hist_rgb_bytes(const uint8_t *__restrict__ rgb, uint64_t npx,
                                                        uint32_t *__restrict__ table) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride)
        atomic_count(table, rgb_key(rgb + 3 * i));
}

__global__ __launch_bounds__(256) void k_hist_syms(const uint32_t *__restrict__ syms, uint64_t n,
                                                   uint32_t *__restrict__ table, uint32_t mask) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        atomic_count(table, syms[i] & mask);
}

int hist_rgb_dense(Ctx *c, const uint8_t *rgb_d, uint64_t npx, uint32_t *table_d) {
    if (npx == 0) return CNIIC_OK;
    static const bool partition = !(test_env("CNIIC_HIST_PARTITION") && atoi(test_env("CNIIC_HIST_PARTITION")) == 0);
    if ((reinterpret_cast<uintptr_t>(rgb_d) & 15) == 0 && partition && npx >= (1u << 20) && npx < (1ull << 32)) {
        const uint64_t groups = npx / 16, gpb = ceil_div(groups, kPartBlocks);
        DevBuf counts, total, start, item, payload;
        CNIIC_HIP_TRY(c, counts.alloc((uint64_t)kPartBlocks * kPartBuckets * 4));
        CNIIC_HIP_TRY(c, total.alloc((uint64_t)kPartBuckets * 4));
        CNIIC_HIP_TRY(c, start.alloc(((uint64_t)kPartBuckets + 1) * 4));
        CNIIC_HIP_TRY(c, item.alloc(((uint64_t)kPartBuckets + 1) * 4));
        CNIIC_HIP_TRY(c, payload.alloc(npx * 2));
        hipLaunchKernelGGL(k_part_pass<false>, dim3(kPartBlocks), dim3(256), 0, c->stream, rgb_d, npx, groups, gpb,
                           counts.as<uint32_t>(), (const uint32_t *)nullptr, (uint16_t *)nullptr);
        hipLaunchKernelGGL(k_part_colscan, dim3(kPartBuckets / kPartCols), dim3(256), 0, c->stream, counts.as<uint32_t>(),
                           total.as<uint32_t>());
        hipLaunchKernelGGL(k_part_starts, dim3(1), dim3(1024), 0, c->stream, total.as<uint32_t>(), start.as<uint32_t>(),
                           item.as<uint32_t>());
        hipLaunchKernelGGL(k_part_pass<true>, dim3(kPartBlocks), dim3(256), 0, c->stream, rgb_d, npx, groups, gpb,
                           counts.as<uint32_t>(), start.as<uint32_t>(), payload.as<uint16_t>());
        const uint32_t max_items = kPartBuckets + (uint32_t)(npx / kPartSub);  // blocks beyond the item count exit at once
        hipLaunchKernelGGL(k_part_hist, dim3(max_items), dim3(256), 0, c->stream, payload.as<uint16_t>(), start.as<uint32_t>(),
                           item.as<uint32_t>(), table_d);
    } else if ((reinterpret_cast<uintptr_t>(rgb_d) & 15) == 0) {
        uint64_t groups = npx / 16;
        uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(groups, 256), 1), 256 * 8);
        hipLaunchKernelGGL(k_hist_rgb, dim3(grid), dim3(256), 0, c->stream, rgb_d, npx, table_d);
    } else {
        uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(npx, 256), 256 * 16);
        hipLaunchKernelGGL(k_hist_rgb_bytes, dim3(grid), dim3(256), 0, c->stream, rgb_d, npx, table_d);
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int hist_syms_dense(Ctx *c, const uint32_t *syms_d, uint64_t n, uint32_t *table_d, uint32_t bits) {
    if (n == 0) return CNIIC_OK;
    uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(n, 256), 256 * 16);
    hipLaunchKernelGGL(k_hist_syms, dim3(grid), dim3(256), 0, c->stream, syms_d, n, table_d, (1u << bits) - 1);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---------------------------------------------------------------- compaction (3-phase scan)
constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 16;
constexpr int kScanChunk = kScanThreads * kScanPerThread;  // 4096 table entries per block

// cell_count (24-bit RGB tables only, may be null): occupied bins per colour-space cell of the K-means that
// follows.  A 4096-key chunk is one r, 16 g and all 256 b values: 2 x 32 cells, counted in LDS first.
__global__ __launch_bounds__(kScanThreads) void k_compact_count(const uint32_t *__restrict__ table,
                                                                uint32_t *__restrict__ blocksum, uint32_t *__restrict__ blockmax,
                                                                uint32_t *__restrict__ cell_count, const uint8_t *__restrict__ pages, uint64_t *__restrict__ total) {
    __shared__ uint32_t s_cell[64];
    __shared__ uint32_t s_max;
    if (blockIdx.x == 0 && threadIdx.x == 0) total[1] = 0;   // (the scan after this kernel collects the largest count there)
    if (pages && !pages[blockIdx.x]) {  // a page nothing was counted into
        if (threadIdx.x == 0) { blocksum[blockIdx.x] = 0; blockmax[blockIdx.x] = 0; }
        return;
    }
    if (threadIdx.x == 0) s_max = 0;
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk;
    const uint4 *v = reinterpret_cast<const uint4 *>(table + base);
    if (cell_count) {
        if (threadIdx.x < 64) s_cell[threadIdx.x] = 0;
        __syncthreads();
    }
    uint32_t nz = 0, mx = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread / 4; j++) {
        const uint32_t quad = j * kScanThreads + threadIdx.x;  // keys base + 4 quad .. + 3: one cell
        const uint4 q = v[quad];
        const uint32_t c4 = (q.x != 0) + (q.y != 0) + (q.z != 0) + (q.w != 0);
        nz += c4;
        mx = max(max(mx, max(q.x, q.y)), max(q.z, q.w));
        if (cell_count && c4) atomicAdd(&s_cell[((quad >> 6) >> kCellShift) * 32 + (((4 * quad) & 255) >> kCellShift)], c4);
    }
    mx = wave_reduce_max(mx);
    if (!cell_count) __syncthreads();   // (s_max = 0 above; with cell_count the barrier after s_cell's clearing did it)
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&s_max, mx);
    nz = block_reduce_sum<kScanThreads>(nz);  // (its barriers also complete s_cell and s_max)
    if (threadIdx.x == 0) { blocksum[blockIdx.x] = nz; blockmax[blockIdx.x] = s_max; }
    if (cell_count && threadIdx.x < 64 && s_cell[threadIdx.x]) {
        const uint32_t g = (threadIdx.x >> 5) << kCellShift, b = (threadIdx.x & 31) << kCellShift;
        atomicAdd(&cell_count[cell_of((uint32_t)base + (g << 8) + b)], s_cell[threadIdx.x]);
    }
}

// exclusive scan of nblocks sums in place; total -> *total.  Blocks of 1024 sums scan their own part and leave their
// total, then every block adds the totals before it (one block over the 32768 sums of a 2^27-bin table took 54 us).
__global__ __launch_bounds__(1024) void k_compact_scan_local(uint32_t *__restrict__ blocksum, const uint32_t *__restrict__ blockmax, uint32_t nblocks,
                                                             uint32_t *__restrict__ parttot, uint64_t *__restrict__ total) {
    __shared__ uint32_t wsum[1024 / 64];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = i < nblocks ? blocksum[i] : 0u;
    {   // the largest count of the table: total[1] (zeroed by the caller); one atomic per wave that has something to say, 16 per block
        const uint32_t m = wave_reduce_max(i < nblocks ? blockmax[i] : 0u);
        if ((threadIdx.x & 63) == 0 && m) atomicMax(reinterpret_cast<unsigned long long *>(total + 1), (unsigned long long)m);
    }
    const uint32_t ex = block_exclusive_scan<1024>(v, wsum);
    if (i < nblocks) blocksum[i] = ex;
    if (threadIdx.x == 1023) {
        parttot[blockIdx.x] = ex + v;
        if (gridDim.x == 1) *total = (uint64_t)ex + v;
    }
}
__global__ __launch_bounds__(1024) void k_compact_scan_add(uint32_t *__restrict__ blocksum, uint32_t nblocks,
                                                           const uint32_t *__restrict__ parttot, uint64_t *__restrict__ total) {
    __shared__ uint32_t s_before;
    uint32_t mine = 0;  // (a table's occupied bins fit 32 bits)
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 1024) mine += parttot[b];
    if (threadIdx.x == 0) s_before = 0;
    __syncthreads();
    mine = wave_reduce_sum(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_before, mine);
    __syncthreads();
    const uint32_t before = s_before;
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (i < nblocks) blocksum[i] += before;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = (uint64_t)before + parttot[blockIdx.x];
}

__global__ __launch_bounds__(kScanThreads) void k_compact_write(uint32_t *__restrict__ table,
                                                                const uint32_t *__restrict__ blockoff,
                                                                uint32_t *__restrict__ keys,
                                                                uint64_t *__restrict__ counts,
                                                                uint32_t *__restrict__ weights, const uint8_t *__restrict__ pages) {
    __shared__ uint32_t wsum[kScanThreads / 64];
    if (pages && !pages[blockIdx.x]) return;
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanPerThread;
    uint4 *v = reinterpret_cast<uint4 *>(table + base);
    uint32_t val[kScanPerThread];
    uint32_t nz = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread / 4; j++) {
        uint4 q = v[j];
        val[4 * j] = q.x; val[4 * j + 1] = q.y; val[4 * j + 2] = q.z; val[4 * j + 3] = q.w;
        nz += (q.x != 0) + (q.y != 0) + (q.z != 0) + (q.w != 0);
    }
    uint32_t excl = block_exclusive_scan<kScanThreads>(nz, wsum);
    uint32_t rank = blockoff[blockIdx.x] + excl;
    bool dirty = false;
#pragma unroll
    for (int i = 0; i < kScanPerThread; i++) {
        if (val[i] != 0) {
            if (keys) keys[rank] = (uint32_t)(base + i);
            if (counts) counts[rank] = val[i];
            if (weights) weights[rank] = val[i];
            val[i] = rank + 1;
            rank++;
            dirty = true;
        }
    }
    if (dirty) {
#pragma unroll
        for (int j = 0; j < kScanPerThread / 4; j++)
            v[j] = make_uint4(val[4 * j], val[4 * j + 1], val[4 * j + 2], val[4 * j + 3]);
    }
}

// phase A+B: number of occupied bins (host out-param; stream synced) + per-block rank offsets
int hist_compact_count(Ctx *c, const uint32_t *table_d, uint32_t bits, CompactPlan *plan, uint32_t *cell_count_d, const uint8_t *pages_d) {
    static_assert(kScanChunk == (1 << kPageShift), "a page flag per block of the compaction");
    plan->pages = pages_d;
    const uint64_t entries = 1ull << bits;
    const uint32_t nblocks = (uint32_t)(entries / kScanChunk);
    DevBuf tot;
    CNIIC_HIP_TRY(c, plan->blockoff.alloc((uint64_t)nblocks * 4));
    CNIIC_HIP_TRY(c, plan->blockmax.alloc((uint64_t)nblocks * 4));
    CNIIC_HIP_TRY(c, tot.alloc(16));
    hipLaunchKernelGGL(k_compact_count, dim3(nblocks), dim3(kScanThreads), 0, c->stream, table_d, plan->blockoff.as<uint32_t>(), plan->blockmax.as<uint32_t>(),
                       bits == 24 ? cell_count_d : nullptr, pages_d, tot.as<uint64_t>());
    {
        const uint32_t nparts = (nblocks + 1023) / 1024;
        DevBuf parttot;
        CNIIC_HIP_TRY(c, parttot.alloc((uint64_t)nparts * 4));
        hipLaunchKernelGGL(k_compact_scan_local, dim3(nparts), dim3(1024), 0, c->stream, plan->blockoff.as<uint32_t>(), (const uint32_t *)plan->blockmax.as<uint32_t>(), nblocks,
                           parttot.as<uint32_t>(), tot.as<uint64_t>());
        if (nparts > 1)
            hipLaunchKernelGGL(k_compact_scan_add, dim3(nparts), dim3(1024), 0, c->stream, plan->blockoff.as<uint32_t>(), nblocks,
                               (const uint32_t *)parttot.as<uint32_t>(), tot.as<uint64_t>());
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint64_t total[2] = {0, 0};
    CNIIC_HIP_TRY(c, hipMemcpyAsync(total, tot.p, 16, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, pages_d ? ctx_spin_sync(c) : hipStreamSynchronize(c->stream));
    plan->n_unique = total[0];
    plan->max_count = total[1];
    plan->bits = bits;
    return CNIIC_OK;
}

// phase C: ascending (key, count) pairs; every occupied bin of the table becomes rank+1.
// keys_d / counts_d (u64) / weights_d (u32) are optional and must hold plan->n_unique entries.
int hist_compact_write(Ctx *c, uint32_t *table_d, const CompactPlan *plan, uint32_t *keys_d, uint64_t *counts_d,
                       uint32_t *weights_d) {
    const uint32_t nblocks = (uint32_t)((1ull << plan->bits) / kScanChunk);
    hipLaunchKernelGGL(k_compact_write, dim3(nblocks), dim3(kScanThreads), 0, c->stream, table_d,
                       plan->blockoff.as<uint32_t>(), keys_d, counts_d, weights_d, plan->pages);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---------------------------------------------------------------- occupancy across ranks (shared-palette cluster-colors)
// occ: one nibble per 24-bit key (8 keys per u32 word, key k in bits 4 (k & 7) of word k >> 3), 1 where the key
// occurs.  Summed over <= 15 ranks by an ordinary all-reduce the nibbles cannot carry; non-zero = occupied somewhere.
__global__ __launch_bounds__(256) void k_occ_pack(const uint32_t *__restrict__ table, uint32_t *__restrict__ occ) {
    const uint32_t wd = blockIdx.x * 256 + threadIdx.x;  // 2^21 words
    const uint4 a = reinterpret_cast<const uint4 *>(table)[2 * (size_t)wd], b = reinterpret_cast<const uint4 *>(table)[2 * (size_t)wd + 1];
    occ[wd] = (a.x != 0) | ((a.y != 0) << 4) | ((a.z != 0) << 8) | ((a.w != 0) << 12) | ((b.x != 0) << 16) | ((b.y != 0) << 20) |
              ((b.z != 0) << 24) | ((uint32_t)(b.w != 0) << 28);
}
// bits[w] = occupancy of keys 64 w .. 64 w + 63 from 8 nibble words; wprefix[w] = occupied keys before word w INSIDE
// this block's 1024 words, blocktot[block] = occupied keys of the block (k_gidx_finish adds the blocks before)
__global__ __launch_bounds__(256) void k_gidx_words(const uint32_t *__restrict__ occ, unsigned long long *__restrict__ bits,
                                                    uint32_t *__restrict__ wprefix, uint32_t *__restrict__ blocktot) {
    __shared__ uint32_t wsum[256 / 64];
    const uint32_t w0 = (blockIdx.x * 256 + threadIdx.x) * 4;  // 4 consecutive words per thread
    uint32_t cnt[4], mine = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned long long m = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t v = occ[8 * (size_t)(w0 + q) + j];
#pragma unroll
            for (int i = 0; i < 8; i++)
                if ((v >> (4 * i)) & 15u) m |= 1ull << (8 * j + i);
        }
        bits[w0 + q] = m;
        cnt[q] = (uint32_t)__popcll(m);
        mine += cnt[q];
    }
    uint32_t run = block_exclusive_scan<256>(mine, wsum);
#pragma unroll
    for (int q = 0; q < 4; q++) { wprefix[w0 + q] = run; run += cnt[q]; }
    if (threadIdx.x == 255) blocktot[blockIdx.x] = run;
}
// 256 blocks of k_gidx_words: every block's words get the occupied keys of the blocks before; *total = all of them
__global__ __launch_bounds__(256) void k_gidx_finish(uint32_t *__restrict__ wprefix, const uint32_t *__restrict__ blocktot,
                                                     uint64_t *__restrict__ total) {
    __shared__ uint32_t wsum[256 / 64];
    __shared__ uint32_t boff[256];
    const uint32_t t = blocktot[threadIdx.x];
    const uint32_t ex = block_exclusive_scan<256>(t, wsum);
    boff[threadIdx.x] = ex;
    if (blockIdx.x == 0 && threadIdx.x == 255) *total = (uint64_t)ex + t;
    __syncthreads();
    const uint32_t add = boff[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < 1024; i += 256) wprefix[blockIdx.x * 1024 + i] += add;
}

int gidx_finish(Ctx *c, uint32_t *wprefix_d, const uint32_t *blocktot_d, uint64_t *total_d) {
    hipLaunchKernelGGL(k_gidx_finish, dim3(256), dim3(256), 0, c->stream, wprefix_d, blocktot_d, total_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int occupancy_pack(Ctx *c, const uint32_t *table_d, uint32_t *occ_d) {
    hipLaunchKernelGGL(k_occ_pack, dim3((1u << 21) / 256), dim3(256), 0, c->stream, table_d, occ_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// occ_d: the summed nibbles -> bitmap + popcount prefix per word; *U_h = occupied keys (stream synced)
// total_keep (optional): the count also stays on the device there, and nothing waits -- the caller takes it from *U_h after
// synchronising the stream or an event of its own
int gidx_build(Ctx *c, const uint32_t *occ_d, DevBuf &bits, DevBuf &wprefix, uint64_t *U_h, DevBuf *total_keep) {
    DevBuf tot_own;
    DevBuf &tot = total_keep ? *total_keep : tot_own;
    CNIIC_HIP_TRY(c, bits.alloc((1ull << 18) * 8));
    CNIIC_HIP_TRY(c, wprefix.alloc((1ull << 18) * 4));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    DevBuf blocktot;
    CNIIC_HIP_TRY(c, blocktot.alloc(256 * 4));
    hipLaunchKernelGGL(k_gidx_words, dim3(256), dim3(256), 0, c->stream, occ_d, bits.as<unsigned long long>(), wprefix.as<uint32_t>(),
                       blocktot.as<uint32_t>());
    hipLaunchKernelGGL(k_gidx_finish, dim3(256), dim3(256), 0, c->stream, wprefix.as<uint32_t>(), (const uint32_t *)blocktot.as<uint32_t>(),
                       tot.as<uint64_t>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipMemcpyAsync(U_h, tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    if (!total_keep) CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

int dense_table(Ctx *c, uint32_t bits, uint32_t **table_d) {
    const uint64_t bytes = (1ull << bits) * 4;
    if (c->dense.bytes < bytes) CNIIC_HIP_TRY(c, c->dense.alloc(bytes));
    CNIIC_HIP_TRY(c, hipMemsetAsync(c->dense.p, 0, bytes, c->stream));
    *table_d = c->dense.as<uint32_t>();
    return CNIIC_OK;
}

}  // namespace cniic
