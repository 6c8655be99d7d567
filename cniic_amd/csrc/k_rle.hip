// k_rle.hip -- Hilbert { compress: RLE(0.0) } on gfx950: exact run-length coding along the Hilbert scan
// (reference: src/codec/hilbertc.rs:12-98; rle_exact = AbstractRle + Exact, :100-196).
//
// A run is a colour and every following equal colour, up to RepCount::MAX = 255 elements; the element that
// would be the 256th starts the next run (:128-137).  So inside a maximal segment of equal colours starting
// at position s the runs start at s, s + 255, s + 510, ...: position i starts a run iff (i - s(i)) % 255 == 0
// with s(i) = the last position <= i whose colour differs from its predecessor (position 0 counts).
//
//   k_rle_last      per 4096-position chunk: its last segment start                      reads 3 B/px
//   k_rle_carry_*   exclusive max-scan over the chunks (1024 per block, then the blocks before)
//   k_rle_flags     run-start flags (16 per thread, one u16) + runs per chunk            reads 3 B/px
//   k_rle_offsets   exclusive sum-scan of the runs per chunk (single block) + total
//   k_rle_records   (count:u8, colour: u64 len = 3 + 3 bytes) = three aligned words per run, behind the header: a run ends
//                   where the next flag is                                               reads the flags + 3 B per run
#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

constexpr int kRleThreads = 256;
constexpr int kRlePer = 16;
constexpr uint32_t kRleChunk = kRleThreads * kRlePer;  // 4096
constexpr uint32_t kRleMaxRun = 255;                    // RepCount::MAX (hilbertc.rs:23,130)

// this thread's 16 positions: bit j of the result <=> position base + j differs from its predecessor (or is 0)
// (every thread of the block calls it: the fast path takes the predecessor's last colour from the lane below)
__device__ __forceinline__ uint32_t segment_starts(const uint8_t *__restrict__ lin, uint64_t n, uint64_t base) {
    uint32_t m = 0;
    // A whole chunk inside the image, 16-byte aligned: a thread's 16 pixels are three 16-byte loads, and the colour in front of them is
    // the lane below's last one (lane 0 of a wave reads its own) -- 51 single-byte loads per thread held the two passes over the
    // linearised image at 1.3 TB/s each.
    const uint64_t cbase = (uint64_t)blockIdx.x * kRleChunk;
    if (cbase + kRleChunk <= n && (reinterpret_cast<uintptr_t>(lin) & 15) == 0) {   // (block-uniform)
        uint32_t key[kRlePer];
        load16px_keys(reinterpret_cast<const uint4 *>(lin + 3 * base), key);
        uint32_t prev = wave_prev_lane(key[kRlePer - 1], 0u);
        if ((threadIdx.x & 63) == 0) prev = base ? rgb_key(lin + 3 * (base - 1)) : 0xffffffffu;  // no colour has this key
#pragma unroll
        for (int j = 0; j < kRlePer; j++) {
            if (key[j] != prev) m |= 1u << j;
            prev = key[j];
        }
        return m;
    }
    if (base >= n) return 0;
    uint32_t prev = base ? rgb_key(lin + 3 * (base - 1)) : 0xffffffffu;  // no colour has this key
#pragma unroll
    for (int j = 0; j < kRlePer; j++) {
        if (base + j < n) {
            const uint32_t k = rgb_key(lin + 3 * (base + j));
            if (k != prev) m |= 1u << j;
            prev = k;
        }
    }
    return m;
}

// inclusive max-scan over the 256 threads of a block, then exclusive (own value left out)
__device__ __forceinline__ uint64_t block_exclusive_max(uint64_t v, uint64_t *wmax) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t inc = wave_inclusive_scan64<true>(v);
    if (lane == 63) wmax[wid] = inc;
    __syncthreads();
    uint64_t pre = 0;
    for (int i = 0; i < wid; i++) pre = max(pre, wmax[i]);
    uint64_t ex = __shfl_up(inc, 1, 64);
    if (lane == 0) ex = 0;
    __syncthreads();
    return max(pre, ex);
}

// positions are stored + 1 so that 0 means "none"
__global__ __launch_bounds__(kRleThreads) void k_rle_last(const uint8_t *__restrict__ lin, uint64_t n, uint64_t *__restrict__ chunk_last) {
    __shared__ uint64_t wmax[kRleThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kRleChunk + (uint64_t)threadIdx.x * kRlePer;
    const uint32_t m = segment_starts(lin, n, base);
    uint64_t last = m ? base + (31 - __clz((int)m)) + 1 : 0;
    last = wave_reduce_max64(last);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = last;
    __syncthreads();
    if (threadIdx.x == 0) chunk_last[blockIdx.x] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
}

// carry[c] = max over chunks before c (0 = none).  Up to 1024 chunks: one block; more (a 16384^2 image has 65536): every
// 1024-chunk block scans its own part and leaves its maximum, then every block takes in the maxima before it (a single block
// walking 64 values per thread took 0.12 ms there).
__global__ __launch_bounds__(1024) void k_rle_carry_local(const uint64_t *__restrict__ chunk_last, uint32_t nchunks, uint64_t *__restrict__ carry,
                                                          uint64_t *__restrict__ blockmax) {
    __shared__ uint64_t wmax[1024 / 64];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint64_t v = i < nchunks ? chunk_last[i] : 0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t inc = wave_inclusive_scan64<true>(v);
    if (lane == 63) wmax[wid] = inc;
    __syncthreads();
    uint64_t pre = 0;
    for (int k = 0; k < wid; k++) pre = max(pre, wmax[k]);
    uint64_t ex = __shfl_up(inc, 1, 64);
    if (lane == 0) ex = 0;
    if (i < nchunks) carry[i] = max(pre, ex);
    if (threadIdx.x == 1023) blockmax[blockIdx.x] = max(pre, inc);
}
__global__ __launch_bounds__(1024) void k_rle_carry_add(uint32_t nchunks, uint64_t *__restrict__ carry, const uint64_t *__restrict__ blockmax) {
    __shared__ unsigned long long s_before;
    unsigned long long mine = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 1024) mine = max(mine, (unsigned long long)blockmax[b]);
    if (threadIdx.x == 0) s_before = 0;
    __syncthreads();
    mine = wave_reduce_max64(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicMax(&s_before, mine);
    __syncthreads();
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (i < nchunks) carry[i] = max(carry[i], (uint64_t)s_before);
}

__global__ __launch_bounds__(kRleThreads) void k_rle_flags(const uint8_t *__restrict__ lin, uint64_t n, const uint64_t *__restrict__ carry,
                                                           uint16_t *__restrict__ flags, uint32_t *__restrict__ chunk_runs) {
    __shared__ uint64_t wmax[kRleThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kRleChunk + (uint64_t)threadIdx.x * kRlePer;
    const uint32_t m = segment_starts(lin, n, base);
    const uint64_t mine = m ? base + (31 - __clz((int)m)) + 1 : 0;
    uint64_t seg = max(block_exclusive_max(mine, wmax), carry[blockIdx.x]);  // (+1) start of the segment running into this thread
    uint32_t f = 0;
#pragma unroll
    for (int j = 0; j < kRlePer; j++) {
        if (base + j < n) {
            if ((m >> j) & 1u) seg = base + j + 1;
            if ((base + j + 1 - seg) % kRleMaxRun == 0) f |= 1u << j;
        }
    }
    flags[(size_t)blockIdx.x * kRleThreads + threadIdx.x] = (uint16_t)f;
    const uint32_t runs = block_reduce_sum<kRleThreads>((uint32_t)__popc(f));
    if (threadIdx.x == 0) chunk_runs[blockIdx.x] = runs;
}

// single block: run_off[c] = runs in the chunks before c; *total = all runs
__global__ __launch_bounds__(1024) void k_rle_offsets(const uint32_t *__restrict__ chunk_runs, uint32_t nchunks, uint64_t *__restrict__ run_off,
                                                      uint64_t *__restrict__ total) {
    __shared__ uint64_t sh[1024];
    const uint32_t per = (nchunks + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, nchunks);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += chunk_runs[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint64_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t run = sh[threadIdx.x] - s;
    for (uint32_t i = lo; i < hi; i++) { run_off[i] = run; run += chunk_runs[i]; }
    if (threadIdx.x == 1023) *total = sh[1023];
}

// The records of a chunk's runs, straight from the flags (a list of run starts in between -- 8 B written and read per run,
// 4.3 GB at 16384^2 on a noisy image -- took 1.2 + 1.35 ms there).  A run ends where the next one starts: a later bit of the
// thread's own flags, else the first bit of a following thread's -- at most 16 threads on, a run has at most 255 elements --
// else the end of the image.
// record = count:u8 | len:u64 = 3 | r g b  = 12 bytes = words { count | 3 << 8, 0, r << 8 | g << 16 | b << 24 }
__global__ __launch_bounds__(kRleThreads) void k_rle_records(const uint8_t *__restrict__ lin, uint64_t n, const uint16_t *__restrict__ flags,
                                                             const uint64_t *__restrict__ run_off, uint32_t nchunks, uint32_t *__restrict__ out_words) {
    __shared__ uint32_t wsum[kRleThreads / 64];
    __shared__ uint16_t s_f[kRleThreads + 16];
    __shared__ uint32_t s_rec[3 * kRleChunk];  // the chunk's records, written out as whole rows of words (a thread's records are
                                               // 192 bytes from the next thread's: stored one by one the kernel took 2.3 ms)
    __shared__ uint32_t s_total;
    const uint32_t f = flags[(size_t)blockIdx.x * kRleThreads + threadIdx.x];
    s_f[threadIdx.x] = (uint16_t)f;
    if (threadIdx.x < 16) s_f[kRleThreads + threadIdx.x] = blockIdx.x + 1 < nchunks ? flags[(size_t)(blockIdx.x + 1) * kRleThreads + threadIdx.x] : (uint16_t)0;
    const uint32_t mine = (uint32_t)__popc(f);
    uint32_t r = block_exclusive_scan<kRleThreads>(mine, wsum);  // (its barriers also complete s_f)
    if (threadIdx.x == kRleThreads - 1) s_total = r + mine;
    if (f) {
        const uint64_t base = (uint64_t)blockIdx.x * kRleChunk + (uint64_t)threadIdx.x * kRlePer;
        // where the run that is open at the end of this thread's positions ends
        uint64_t after = n;
        for (uint32_t k = 1; k <= 16; k++) {
            const uint32_t g = s_f[threadIdx.x + k];
            if (g) { after = base + (uint64_t)k * kRlePer + (uint32_t)(__ffs((int)g) - 1); break; }
        }
        if (after > n) after = n;
        for (uint32_t m = f; m; r++) {
            const uint32_t j = (uint32_t)(__ffs((int)m) - 1);
            m &= m - 1;
            const uint64_t s = base + j, e = m ? base + (uint32_t)(__ffs((int)m) - 1) : after;
            const uint8_t *p = lin + 3 * s;
            s_rec[3 * r] = (uint32_t)(e - s) | (3u << 8);
            s_rec[3 * r + 1] = 0u;
            s_rec[3 * r + 2] = ((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24);
        }
    }
    __syncthreads();
    uint32_t *o = out_words + 3 * run_off[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < 3 * s_total; i += kRleThreads) o[i] = s_rec[i];
}

// lin_d: the image in Hilbert order (3 B/px).  Phase 1 counts the runs (host out-param, stream synced) and keeps
// its scratch in `plan`; phase 2 writes the records at out_words (device, 4-byte aligned).
int rle_plan(Ctx *c, const uint8_t *lin_d, uint64_t n, RlePlan *plan) {
    plan->n = n;
    plan->nruns = 0;
    if (n == 0) return CNIIC_OK;
    const uint64_t nchunks64 = ceil_div(n, kRleChunk);
    if (nchunks64 > 0x7fffffffull) return c->fail(CNIIC_ERR_BAD_ARG, "hilbert-rle: image too large");
    const uint32_t nchunks = (uint32_t)nchunks64;
    plan->nchunks = nchunks;
    DevBuf chunk_last, carry, chunk_runs, tot;
    CNIIC_HIP_TRY(c, chunk_last.alloc((uint64_t)nchunks * 8));
    CNIIC_HIP_TRY(c, carry.alloc((uint64_t)nchunks * 8));
    CNIIC_HIP_TRY(c, chunk_runs.alloc((uint64_t)nchunks * 4));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    CNIIC_HIP_TRY(c, plan->flags.alloc((uint64_t)nchunks * kRleThreads * 2));
    CNIIC_HIP_TRY(c, plan->run_off.alloc((uint64_t)nchunks * 8));
    hipLaunchKernelGGL(k_rle_last, dim3(nchunks), dim3(kRleThreads), 0, c->stream, lin_d, n, chunk_last.as<uint64_t>());
    {
        const uint32_t nb = (nchunks + 1023) / 1024;
        DevBuf blockmax;
        CNIIC_HIP_TRY(c, blockmax.alloc((uint64_t)nb * 8));
        hipLaunchKernelGGL(k_rle_carry_local, dim3(nb), dim3(1024), 0, c->stream, chunk_last.as<uint64_t>(), nchunks, carry.as<uint64_t>(),
                           blockmax.as<uint64_t>());
        if (nb > 1)
            hipLaunchKernelGGL(k_rle_carry_add, dim3(nb), dim3(1024), 0, c->stream, nchunks, carry.as<uint64_t>(), (const uint64_t *)blockmax.as<uint64_t>());
    }
    hipLaunchKernelGGL(k_rle_flags, dim3(nchunks), dim3(kRleThreads), 0, c->stream, lin_d, n, carry.as<uint64_t>(),
                       plan->flags.as<uint16_t>(), chunk_runs.as<uint32_t>());
    if (nchunks > 1024)  // (one block walking 64 chunks per thread took 0.11 ms at 16384^2)
        CNIIC_TRY(pack_scan(c, chunk_runs.as<uint32_t>(), nchunks, plan->run_off.as<uint64_t>(), tot.as<uint64_t>()));
    else
        hipLaunchKernelGGL(k_rle_offsets, dim3(1), dim3(1024), 0, c->stream, chunk_runs.as<uint32_t>(), nchunks, plan->run_off.as<uint64_t>(),
                           tot.as<uint64_t>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint64_t total = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&total, tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    plan->nruns = total;
    return CNIIC_OK;
}

int rle_emit(Ctx *c, const uint8_t *lin_d, const RlePlan *plan, uint32_t *out_words_d) {
    if (plan->nruns == 0) return CNIIC_OK;
    hipLaunchKernelGGL(k_rle_records, dim3(plan->nchunks), dim3(kRleThreads), 0, c->stream, lin_d, plan->n, (const uint16_t *)plan->flags.as<uint16_t>(),
                       (const uint64_t *)plan->run_off.as<uint64_t>(), plan->nchunks, out_words_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---------------------------------------------------------------- RleDecoder (hilbertc.rs:304-337) as a scan + search
// rec: the run records (12 bytes each) on the device.  The decoder reads records until w*h colours are out
// (it is zipped with hilbert::iter, :58-61), so a record is READ iff it starts before colour n; a zero count
// (assert!, :327) or a colour that is not 3 bytes long (unwrap, :328) in a record that is read is an error.
// Round 3: the decode was one block scanning all R counts (1.2 s for the 2.7 10^8 records of a noisy 16384^2 image) and a binary search over
// all R offsets for every pixel (49 ms at 4096^2).  Now:
//   k_rle_dec_counts  1024 records per block (three aligned words each): the count's exclusive sum inside the block (u32), the block's
//                     total, and the lowest-numbered record that is malformed or has count 0 (atomicMin; a well-formed stream has none)
//   pack_scan         over the block totals -> the blocks' first colour index (u64)
//   k_rle_dec_verdict the stream is bad iff that first bad record starts before colour n (it would be READ); the total with it
//   k_rle_dec_expand  a block per 4096 colours: the first record of its stretch by a two-level search (blocks, then inside one), the up
//                     to 4097 records that cover the stretch into LDS (start relative to the stretch, colour), every thread its 16
//                     consecutive colours: one search in LDS, then a walk; 48 bytes out as three 16-byte stores.
constexpr uint32_t kRdBlock = 1024, kRdStretch = 4096, kRdThreads = 256, kRdPer = kRdStretch / kRdThreads;
__global__ __launch_bounds__(kRdBlock) void k_rle_dec_counts(const uint32_t *__restrict__ recw, uint64_t R, uint32_t *__restrict__ offl, uint32_t *__restrict__ blocksum,
                                                             unsigned long long *__restrict__ first_bad) {
    __shared__ uint32_t wsum[kRdBlock / 64];
    const uint64_t r = (uint64_t)blockIdx.x * kRdBlock + threadIdx.x;
    uint32_t cnt = 0;
    if (r < R) {
        const uint32_t w0 = recw[3 * r], w1 = recw[3 * r + 1], w2 = recw[3 * r + 2];
        cnt = w0 & 255u;
        // count > 0 (assert!, hilbertc.rs:327); the colour's length is the u64 3 (unwrap, :328): bytes 1..8 = 3, 0, 0, 0, 0, 0, 0, 0
        if (cnt == 0 || (w0 >> 8) != 3u || w1 != 0u || (w2 & 255u) != 0u) atomicMin(first_bad, (unsigned long long)r);
    }
    const uint32_t ex = block_exclusive_scan<kRdBlock>(cnt, wsum);
    if (r < R) offl[r] = ex;
    if (threadIdx.x == kRdBlock - 1) blocksum[blockIdx.x] = ex + cnt;
}
__global__ void k_rle_dec_verdict(const unsigned long long *__restrict__ first_bad, const uint32_t *__restrict__ offl, const uint64_t *__restrict__ blockoff, uint64_t R,
                                  uint64_t n, const uint64_t *__restrict__ total, uint64_t *__restrict__ out /* [0] total, [1] bad */) {
    const unsigned long long fb = *first_bad;
    out[0] = *total;
    out[1] = (fb < R && blockoff[fb / kRdBlock] + offl[fb] < n) ? 1u : 0u;   // a bad record is an error only if the decoder gets to read it
}
__global__ __launch_bounds__(kRdThreads) void k_rle_dec_expand(const uint32_t *__restrict__ recw, uint64_t R, const uint32_t *__restrict__ offl,
                                                               const uint64_t *__restrict__ blockoff, uint32_t nb, uint64_t total, uint64_t n, uint8_t *__restrict__ lin) {
    __shared__ uint32_t s_rel[kRdStretch + 2], s_col[kRdStretch + 2];
    __shared__ unsigned long long s_r0;
    const uint64_t c0 = (uint64_t)blockIdx.x * kRdStretch;
    if (threadIdx.x == 0 && c0 < total) {
        uint32_t a = 0, b = nb;   // last block whose first colour index is <= c0 (block 0's is 0)
        while (b - a > 1) { const uint32_t m = a + (b - a) / 2; if (blockoff[m] <= c0) a = m; else b = m; }
        const uint64_t base = blockoff[a], rb = (uint64_t)a * kRdBlock;
        uint32_t lo = 0, hi = (uint32_t)min<uint64_t>(kRdBlock, R - rb);   // ... and the last record in it that starts at or before c0
        while (hi - lo > 1) { const uint32_t m = lo + (hi - lo) / 2; if (base + offl[rb + m] <= c0) lo = m; else hi = m; }
        s_r0 = rb + lo;
    }
    __syncthreads();
    if (c0 < total) {
        const uint64_t r0 = s_r0;
        for (uint32_t k = threadIdx.x; k < kRdStretch + 2; k += kRdThreads) {
            const uint64_t r = r0 + k;
            uint32_t rel = kRdStretch, col = 0;
            if (r < R) {
                const uint64_t at = blockoff[r / kRdBlock] + offl[r];
                rel = at <= c0 ? 0u : (uint32_t)min<uint64_t>(at - c0, kRdStretch);
                col = recw[3 * r + 2] >> 8;   // r | g << 8 | b << 16
            }
            s_rel[k] = rel; s_col[k] = col;
        }
    }
    __syncthreads();
    const uint32_t p0 = threadIdx.x * kRdPer;
    if (c0 + p0 >= n) return;
    uint32_t px[kRdPer];
    uint32_t k = 0;
    if (c0 < total) {   // last entry that starts at or before this thread's first colour (entry 0 starts at 0)
        uint32_t lo = 0, hi = kRdStretch + 1;
        while (hi - lo > 1) { const uint32_t m = lo + (hi - lo) / 2; if (s_rel[m] <= p0) lo = m; else hi = m; }
        k = lo;
    }
#pragma unroll
    for (uint32_t j = 0; j < kRdPer; j++) {
        const uint32_t p = p0 + j;
        uint32_t col = 0;
        // Only colours INSIDE the image are looked up (colours past the stream stay zero: ImageBuffer::new).  Records that start at or
        // after colour n are never read by the decoder and may therefore have count 0 (k_rle_dec_verdict accepts them, as the reference
        // does): a thread that straddles n must not walk over them -- thousands of zero-count records at start == n would keep
        // s_rel[k + 1] <= p for every staged entry.  Inside the image every record has count >= 1, so at most p + 1 staged
        // entries start at or before p; the bound on k is the belt to those braces.
        if (c0 + p < total && c0 + p < n) {
            while (k + 1 < kRdStretch + 2 && s_rel[k + 1] <= p) k++;
            col = s_col[k];
        }
        px[j] = col;
    }
    uint8_t *o = lin + 3 * (c0 + p0);
    if (c0 + p0 + kRdPer <= n && (reinterpret_cast<uintptr_t>(lin) & 15) == 0) {
        uint32_t w[12];
#pragma unroll
        for (int g = 0; g < 4; g++) {   // 4 colours (r | g << 8 | b << 16) -> 3 words of r g b r g b ...
            const uint32_t a = px[4 * g], b = px[4 * g + 1], c2 = px[4 * g + 2], d = px[4 * g + 3];
            w[3 * g] = a | (b << 24);
            w[3 * g + 1] = (b >> 8) | (c2 << 16);
            w[3 * g + 2] = (c2 >> 16) | (d << 8);
        }
        uint4 *o4 = reinterpret_cast<uint4 *>(o);
        o4[0] = make_uint4(w[0], w[1], w[2], w[3]); o4[1] = make_uint4(w[4], w[5], w[6], w[7]); o4[2] = make_uint4(w[8], w[9], w[10], w[11]);
    } else {
        for (uint32_t j = 0; j < kRdPer && c0 + p0 + j < n; j++) { o[3 * j] = (uint8_t)px[j]; o[3 * j + 1] = (uint8_t)(px[j] >> 8); o[3 * j + 2] = (uint8_t)(px[j] >> 16); }
    }
}

// rec_d: R complete records on the device, 4-byte aligned -> lin_d: n colours in scan order.  *status: 0 ok, 1 a record that is read is
// bad or (tail_bytes != 0 and the complete records do not reach n colours: the next record is cut)
int rle_expand_dev(Ctx *c, const uint8_t *rec_d, uint64_t R, uint64_t tail_bytes, uint64_t n, uint8_t *lin_d, int *status) {
    *status = 0;
    if (n == 0) return CNIIC_OK;
    if (reinterpret_cast<uintptr_t>(rec_d) & 3) return c->fail(CNIIC_ERR_BAD_ARG, "hilbert-rle: the records must be 4-byte aligned");
    if (R > 0xffffffffull) return c->fail(CNIIC_ERR_BAD_ARG, "hilbert-rle: stream too long");
    const uint32_t *recw = reinterpret_cast<const uint32_t *>(rec_d);
    uint64_t total = 0;
    DevBuf offl, blocksum, blockoff, small;
    const uint32_t nb = (uint32_t)std::max<uint64_t>(ceil_div(R, (uint64_t)kRdBlock), 1);
    CNIIC_HIP_TRY(c, offl.alloc(std::max<uint64_t>(R, 1) * 4));
    CNIIC_HIP_TRY(c, blocksum.alloc((uint64_t)nb * 4));
    CNIIC_HIP_TRY(c, blockoff.alloc((uint64_t)nb * 8));
    CNIIC_HIP_TRY(c, small.alloc(64));   // [0] first bad record, [1] total, [2] total (out), [3] bad (out)
    uint64_t *sm = small.as<uint64_t>();
    static const uint64_t init[4] = {~0ull, 0, 0, 0};
    CNIIC_HIP_TRY(c, hipMemcpyAsync(sm, init, 32, hipMemcpyHostToDevice, c->stream));
    if (R == 0) CNIIC_HIP_TRY(c, hipMemsetAsync(blocksum.p, 0, 4, c->stream));
    else
        hipLaunchKernelGGL(k_rle_dec_counts, dim3(nb), dim3(kRdBlock), 0, c->stream, recw, R, offl.as<uint32_t>(), blocksum.as<uint32_t>(),
                           reinterpret_cast<unsigned long long *>(sm));
    CNIIC_TRY(pack_scan(c, blocksum.as<uint32_t>(), nb, blockoff.as<uint64_t>(), sm + 1));
    hipLaunchKernelGGL(k_rle_dec_verdict, dim3(1), dim3(1), 0, c->stream, reinterpret_cast<const unsigned long long *>(sm), (const uint32_t *)offl.as<uint32_t>(),
                       (const uint64_t *)blockoff.as<uint64_t>(), R, n, (const uint64_t *)(sm + 1), sm + 2);
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint64_t res[2] = {0, 0};
    CNIIC_HIP_TRY(c, hipMemcpyAsync(res, sm + 2, 16, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    total = res[0];
    if (res[1]) { *status = 1; return CNIIC_OK; }
    if (total < n && tail_bytes) { *status = 1; return CNIIC_OK; }  // the decoder starts one more record and runs out of bytes
    hipLaunchKernelGGL(k_rle_dec_expand, dim3((uint32_t)ceil_div(n, (uint64_t)kRdStretch)), dim3(kRdThreads), 0, c->stream, recw, R, (const uint32_t *)offl.as<uint32_t>(),
                       (const uint64_t *)blockoff.as<uint64_t>(), nb, total, n, lin_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

}  // namespace cniic
