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
        for (int i = 0; i < 16; i++) atomicAdd(&table[key[i]], 1u);
    }
    // tail (< 16 px) by the first threads of block 0
    if (blockIdx.x == 0) {
        uint64_t i = ngroups * 16 + threadIdx.x;
        if (i < npx) atomicAdd(&table[rgb_key(rgb + 3 * i)], 1u);
    }
}

// unaligned base pointer: plain per-pixel byte loads
__global__ __launch_bounds__(256) void k_hist_rgb_bytes(const uint8_t *__restrict__ rgb, uint64_t npx,
                                                        uint32_t *__restrict__ table) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride)
        atomicAdd(&table[rgb_key(rgb + 3 * i)], 1u);
}

__global__ __launch_bounds__(256) void k_hist_syms(const uint32_t *__restrict__ syms, uint64_t n,
                                                   uint32_t *__restrict__ table, uint32_t mask) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        atomicAdd(&table[syms[i] & mask], 1u);
}

int hist_rgb_dense(Ctx *c, const uint8_t *rgb_d, uint64_t npx, uint32_t *table_d) {
    if (npx == 0) return CNIIC_OK;
    if ((reinterpret_cast<uintptr_t>(rgb_d) & 15) == 0) {
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

__global__ __launch_bounds__(kScanThreads) void k_compact_count(const uint32_t *__restrict__ table,
                                                                uint32_t *__restrict__ blocksum) {
    const uint4 *v = reinterpret_cast<const uint4 *>(table + (uint64_t)blockIdx.x * kScanChunk);
    uint32_t nz = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread / 4; j++) {
        uint4 q = v[j * kScanThreads + threadIdx.x];
        nz += (q.x != 0) + (q.y != 0) + (q.z != 0) + (q.w != 0);
    }
    nz = block_reduce_sum<kScanThreads>(nz);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = nz;
}

// single block: exclusive scan of nblocks sums in place; total -> *total
__global__ __launch_bounds__(1024) void k_compact_scan(uint32_t *__restrict__ blocksum, uint32_t nblocks,
                                                       uint64_t *__restrict__ total) {
    __shared__ uint32_t sh[1024];
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = min(lo + per, nblocks);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += blocksum[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t add = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - s;  // exclusive prefix of this thread's range
    for (uint32_t i = lo; i < hi; i++) {
        uint32_t v = blocksum[i];
        blocksum[i] = run;
        run += v;
    }
    if (threadIdx.x == 1023) *total = sh[1023];
}

__global__ __launch_bounds__(kScanThreads) void k_compact_write(uint32_t *__restrict__ table,
                                                                const uint32_t *__restrict__ blockoff,
                                                                uint32_t *__restrict__ keys,
                                                                uint64_t *__restrict__ counts,
                                                                uint32_t *__restrict__ weights) {
    __shared__ uint32_t wsum[kScanThreads / 64];
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
int hist_compact_count(Ctx *c, const uint32_t *table_d, uint32_t bits, CompactPlan *plan) {
    const uint64_t entries = 1ull << bits;
    const uint32_t nblocks = (uint32_t)(entries / kScanChunk);
    DevBuf tot;
    CNIIC_HIP_TRY(c, plan->blockoff.alloc((uint64_t)nblocks * 4));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    hipLaunchKernelGGL(k_compact_count, dim3(nblocks), dim3(kScanThreads), 0, c->stream, table_d, plan->blockoff.as<uint32_t>());
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, c->stream, plan->blockoff.as<uint32_t>(), nblocks, tot.as<uint64_t>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint64_t total = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&total, tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    plan->n_unique = total;
    plan->bits = bits;
    return CNIIC_OK;
}

// phase C: ascending (key, count) pairs; every occupied bin of the table becomes rank+1.
// keys_d / counts_d (u64) / weights_d (u32) are optional and must hold plan->n_unique entries.
int hist_compact_write(Ctx *c, uint32_t *table_d, const CompactPlan *plan, uint32_t *keys_d, uint64_t *counts_d,
                       uint32_t *weights_d) {
    const uint32_t nblocks = (uint32_t)((1ull << plan->bits) / kScanChunk);
    hipLaunchKernelGGL(k_compact_write, dim3(nblocks), dim3(kScanThreads), 0, c->stream, table_d,
                       plan->blockoff.as<uint32_t>(), keys_d, counts_d, weights_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
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
