// k_huff.hip -- second pass of huf::encode_all (reference src/huf.rs:36-41): every symbol's code
// is appended MSB-first to a bit stream that is zero-padded to a byte (src/bit.rs:209-254).
//
// The reference pushes bits one symbol at a time through IoBitWriter.  Here the stream position of
// every symbol is a prefix sum of code lengths, so the pack is three launches:
//   1. per-chunk bit totals (4096 symbols per 256-thread block)
//   2. single-block exclusive scan of the chunk totals (u64 offsets)
//   3. per-chunk pack: block scan of the per-thread bit counts, codes OR-ed into an LDS image of
//      the chunk aligned to the 32-bit word grid of the output, whole words stored big-endian;
//      the (at most two) words shared with neighbouring chunks go out with global atomicOr.
// Symbols reach their code through the rank table left behind by the histogram compaction
// (key -> rank+1), then len[rank] / code[rank].
#include <mutex>

#include <algorithm>
#include <vector>

#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

constexpr int kPackThreads = 256;
constexpr int kPackPer = 16;
constexpr int kPackChunk = kPackThreads * kPackPer;  // symbols per block

enum { SRC_RGB = 0, SRC_KEYS = 1, SRC_SYM16 = 2, SRC_RANKS = 3 };

// fetch the ranks of this thread's 16 consecutive symbols (0xffffffff past the end)
template <int SRC>
__device__ __forceinline__ void fetch_ranks(const void *__restrict__ src, uint64_t n, uint64_t first,
                                            const uint32_t *__restrict__ rank_table, uint32_t rank[kPackPer]) {
    if (SRC == SRC_RGB) {
        const uint8_t *rgb = reinterpret_cast<const uint8_t *>(src);
        if (first + kPackPer <= n && ((reinterpret_cast<uintptr_t>(rgb) & 15) == 0)) {
            uint32_t key[16];
            load16px_keys(reinterpret_cast<const uint4 *>(rgb + 3 * first), key);
#pragma unroll
            for (int i = 0; i < kPackPer; i++) rank[i] = rank_table[key[i]] - 1;
        } else {
#pragma unroll
            for (int i = 0; i < kPackPer; i++)
                rank[i] = (first + i < n) ? rank_table[rgb_key(rgb + 3 * (first + i))] - 1 : 0xffffffffu;
        }
    } else if (SRC == SRC_KEYS) {
        const uint32_t *keys = reinterpret_cast<const uint32_t *>(src);
#pragma unroll
        for (int i = 0; i < kPackPer; i++) rank[i] = (first + i < n) ? rank_table[keys[first + i]] - 1 : 0xffffffffu;
    } else if (SRC == SRC_RANKS) {  // the stream of rank + 1 left by k_rank_stream
        const uint32_t *rk = reinterpret_cast<const uint32_t *>(src);
        if (first + kPackPer <= n && ((reinterpret_cast<uintptr_t>(rk) & 15) == 0)) {
            const uint4 *v = reinterpret_cast<const uint4 *>(rk + first);
#pragma unroll
            for (int j = 0; j < 4; j++) { const uint4 q = v[j]; rank[4 * j] = q.x - 1; rank[4 * j + 1] = q.y - 1; rank[4 * j + 2] = q.z - 1; rank[4 * j + 3] = q.w - 1; }
        } else {
#pragma unroll
            for (int i = 0; i < kPackPer; i++) rank[i] = (first + i < n) ? rk[first + i] - 1 : 0xffffffffu;
        }
    } else {
        const uint16_t *sym = reinterpret_cast<const uint16_t *>(src);
#pragma unroll
        for (int i = 0; i < kPackPer; i++) rank[i] = (first + i < n) ? (uint32_t)sym[first + i] : 0xffffffffu;
    }
}

// rank + 1 of every symbol as a linear stream: the one random read per symbol into the dense table, done while the
// host builds the tree (it needs no code); the pack passes then read the stream and the small per-rank tables
template <int SRC>
__global__ __launch_bounds__(kPackThreads) void k_rank_stream(const void *__restrict__ src, uint64_t n,
                                                              const uint32_t *__restrict__ rank_table, uint32_t *__restrict__ out,
                                                              uint32_t plus) {
    const uint64_t first = (uint64_t)blockIdx.x * kPackChunk + (uint64_t)threadIdx.x * kPackPer;
    uint32_t rank[kPackPer];
    fetch_ranks<SRC>(src, n, first, rank_table, rank);
    if (first + kPackPer <= n && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
        uint4 *o = reinterpret_cast<uint4 *>(out + first);
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = make_uint4(rank[4 * j] + plus, rank[4 * j + 1] + plus, rank[4 * j + 2] + plus, rank[4 * j + 3] + plus);
    } else {
        for (int i = 0; i < kPackPer; i++)
            if (first + i < n) out[first + i] = rank[i] + plus;
    }
}

template <int SRC>
__global__ __launch_bounds__(kPackThreads) void k_pack_count(const void *__restrict__ src, uint64_t n,
                                                             const uint32_t *__restrict__ rank_table,
                                                             const uint8_t *__restrict__ len,
                                                             uint32_t *__restrict__ chunk_bits) {
    const uint64_t first = (uint64_t)blockIdx.x * kPackChunk + (uint64_t)threadIdx.x * kPackPer;
    uint32_t rank[kPackPer];
    fetch_ranks<SRC>(src, n, first, rank_table, rank);
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < kPackPer; i++)
        if (rank[i] != 0xffffffffu) bits += len[rank[i]];
    bits = block_reduce_sum<kPackThreads>(bits);
    if (threadIdx.x == 0) chunk_bits[blockIdx.x] = bits;
}

// chunk_off[i] = exclusive prefix (u64) of chunk_bits; total -> *total_bits.  Up to 1024 chunks: one block.  More
// (a 16384^2 image has 65536): every 1024-chunk block scans its own part and leaves its sum, then every block adds the
// sums before it (a single block walking 64 values per thread took 115 us there).
__global__ __launch_bounds__(1024) void k_pack_scan_local(const uint32_t *__restrict__ chunk_bits, uint32_t nchunks,
                                                          uint64_t *__restrict__ chunk_off, uint64_t *__restrict__ blocktot,
                                                          uint64_t *__restrict__ total_bits) {
    __shared__ uint32_t wsum[1024 / 64];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = i < nchunks ? chunk_bits[i] : 0u;  // <= 4096 symbols x 64 bits: a block's sum fits 32 bits
    const uint32_t ex = block_exclusive_scan<1024>(v, wsum);
    if (i < nchunks) chunk_off[i] = ex;
    if (threadIdx.x == 1023) {
        blocktot[blockIdx.x] = (uint64_t)ex + v;
        if (gridDim.x == 1) *total_bits = (uint64_t)ex + v;
    }
}
__global__ __launch_bounds__(1024) void k_pack_scan_add(uint32_t nchunks, uint64_t *__restrict__ chunk_off,
                                                        const uint64_t *__restrict__ blocktot, uint64_t *__restrict__ total_bits) {
    __shared__ unsigned long long s_before;
    unsigned long long mine = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 1024) mine += blocktot[b];
    if (threadIdx.x == 0) s_before = 0;
    __syncthreads();
    mine = wave_reduce_sum64(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_before, mine);
    __syncthreads();
    const unsigned long long before = s_before;
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (i < nchunks) chunk_off[i] += before;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_bits = before + blocktot[blockIdx.x];
}
// ... and up to 8192 chunks in ONE block, eight consecutive values per thread (round 4: the leaves' sort of a `delta` alphabet scans 3584
// counters per pass -- two launches of 2 us of work each, a third of the pass's wall time with the gaps in front of them).
// (8192 values of at most 2^18: the block's sum fits 32 bits.)
__global__ __launch_bounds__(1024) void k_pack_scan_small(const uint32_t *__restrict__ chunk_bits, uint32_t nchunks,
                                                          uint64_t *__restrict__ chunk_off, uint64_t *__restrict__ total_bits) {
    __shared__ uint32_t wsum[1024 / 64];
    const uint32_t i0 = threadIdx.x * 8;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { v[j] = i0 + j < nchunks ? chunk_bits[i0 + j] : 0u; sum += v[j]; }
    uint32_t run = block_exclusive_scan<1024>(sum, wsum);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (i0 + j < nchunks) chunk_off[i0 + j] = run;
        run += v[j];
    }
    if (threadIdx.x == 1023) *total_bits = run;
}
// (only for callers whose values are known to be small: the sort's per-block digit counts, at most kSortBlock each)
static int pack_scan_counts(Ctx *c, const uint32_t *counts_d, uint32_t n, uint64_t *off_d, uint64_t *total_d) {
    if (n > 1024 && n <= 8192) {
        hipLaunchKernelGGL(k_pack_scan_small, dim3(1), dim3(1024), 0, c->stream, counts_d, n, off_d, total_d);
        CNIIC_HIP_TRY(c, hipGetLastError());
        return CNIIC_OK;
    }
    return pack_scan(c, counts_d, n, off_d, total_d);
}
int pack_scan(Ctx *c, const uint32_t *chunk_bits_d, uint32_t nchunks, uint64_t *chunk_off_d, uint64_t *total_d) {
    const uint32_t nb = (nchunks + 1023) / 1024;
    DevBuf blocktot;
    CNIIC_HIP_TRY(c, blocktot.alloc((uint64_t)nb * 8));
    hipLaunchKernelGGL(k_pack_scan_local, dim3(nb), dim3(1024), 0, c->stream, chunk_bits_d, nchunks, chunk_off_d, blocktot.as<uint64_t>(), total_d);
    if (nb > 1)
        hipLaunchKernelGGL(k_pack_scan_add, dim3(nb), dim3(1024), 0, c->stream, nchunks, chunk_off_d, (const uint64_t *)blocktot.as<uint64_t>(), total_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// tests: CNIIC_TEST_PACK_IMG_WORDS caps the packs' LDS bit image so that chunks take the direct-to-memory route
uint32_t pack_img_cap() {
    const char *e = test_env("CNIIC_TEST_PACK_IMG_WORDS");
    return e ? (uint32_t)atoi(e) : 0xffffffffu;
}

template <int SRC>
__global__ __launch_bounds__(kPackThreads) void k_pack_write(const void *__restrict__ src, uint64_t n,
                                                             const uint32_t *__restrict__ rank_table,
                                                             const uint8_t *__restrict__ len,
                                                             const uint64_t *__restrict__ code,
                                                             const uint64_t *__restrict__ chunk_off,
                                                             uint32_t *__restrict__ out_words, uint64_t bit_base, uint32_t img_cap) {
    // the chunk's bit image: 32 bits per symbol on average fit (the codes of 10^5 delta symbols average 14, of 7 M colours 23);
    // a chunk of rarer symbols goes to memory piece by piece instead.  Sized for the worst case (64 bits per symbol, 32 KiB,
    // all of it cleared for every chunk) the array held the kernel at four blocks per CU.
    constexpr uint32_t kImgWords = kPackChunk + 2;
    __shared__ uint32_t img[kImgWords];
    __shared__ uint32_t wsum[kPackThreads / 64];
    __shared__ uint32_t s_total;
    const uint64_t first = (uint64_t)blockIdx.x * kPackChunk + (uint64_t)threadIdx.x * kPackPer;
    uint32_t rank[kPackPer];
    fetch_ranks<SRC>(src, n, first, rank_table, rank);
    uint32_t l[kPackPer];
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < kPackPer; i++) {
        l[i] = rank[i] != 0xffffffffu ? len[rank[i]] : 0;
        bits += l[i];
    }
    uint32_t excl = block_exclusive_scan<kPackThreads>(bits, wsum);
    const uint64_t g0 = bit_base + chunk_off[blockIdx.x];  // global bit offset of the chunk
    const uint32_t skew = (uint32_t)(g0 & 31);          // chunk image is aligned to the output word grid
    uint32_t pos = skew + excl;
    if (threadIdx.x == kPackThreads - 1) s_total = excl + bits;
    __syncthreads();
    const uint32_t total = s_total;
    if (total == 0) return;                             // (zero-length codes: a single-symbol alphabet, huf.rs:140-142)
    const uint32_t nwords = (skew + total + 31) >> 5;
    const uint64_t w0 = g0 >> 5;
    if (nwords + 2 > min(kImgWords, img_cap)) {
#pragma unroll 1
        for (int i = 0; i < kPackPer; i++) {
            if (l[i]) pack_put<true>(out_words + w0, pos, l[i], code[rank[i]]);  // L significant bits, first stream bit = bit L-1
            pos += l[i];
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < nwords + 2; i += kPackThreads) img[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPackPer; i++) {
        if (l[i]) pack_put<false>(img, pos, l[i], code[rank[i]]);
        pos += l[i];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nwords; i += kPackThreads) {
        uint32_t v = __builtin_bswap32(img[i]);          // MSB-first bit order -> big-endian bytes
        if (v == 0) continue;
        if (i == 0 || i == nwords - 1) atomicOr(&out_words[w0 + i], v);   // shared with a neighbour chunk
        else out_words[w0 + i] = v;
    }
}

// ---------------------------------------------------------------- cluster-colors fast path
// Symbols are cluster labels (<= K distinct): the pixels become a label stream (pass 0), whose code
// lengths are summed per chunk (pass 1) and which is packed from an LDS (len, code) table (pass 2).
// pass 0 (needs no code table, so it runs while the host builds the tree): every pixel's cluster label
// through the dense colour -> label table, ONE random read per pixel, stored as a linear label stream
template <typename LabelT>
__global__ __launch_bounds__(kPackThreads) void k_pixel_labels(const uint8_t *__restrict__ rgb, uint64_t n,
                                                               const LabelT *__restrict__ key2label, LabelT *__restrict__ pixlab) {
    const uint64_t first = (uint64_t)blockIdx.x * kPackChunk + (uint64_t)threadIdx.x * kPackPer;
    if (first + kPackPer <= n && ((reinterpret_cast<uintptr_t>(rgb) & 15) == 0)) {
        uint32_t key[16];
        load16px_keys(reinterpret_cast<const uint4 *>(rgb + 3 * first), key);
        LabelT lab[16];
#pragma unroll
        for (int i = 0; i < 16; i++) lab[i] = key2label[key[i]];
        if (sizeof(LabelT) == 1) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = (uint32_t)lab[4 * j] | ((uint32_t)lab[4 * j + 1] << 8) | ((uint32_t)lab[4 * j + 2] << 16) | ((uint32_t)lab[4 * j + 3] << 24);
            *reinterpret_cast<uint4 *>(pixlab + first) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) pixlab[first + i] = lab[i];
        }
    } else {
        for (int i = 0; i < kPackPer; i++)
            if (first + i < n) pixlab[first + i] = key2label[rgb_key(rgb + 3 * (first + i))];
    }
}

// pass 1: code lengths of a chunk's labels, from an LDS table
template <typename LabelT>
__global__ __launch_bounds__(kPackThreads) void k_pack_count_lab(const LabelT *__restrict__ pixlab, uint64_t n, uint32_t K,
                                                                 const uint8_t *__restrict__ clen,
                                                                 uint32_t *__restrict__ chunk_bits, uint64_t lab_stride = 0) {
    extern __shared__ uint8_t s_len[];  // [K]
    // (a batch of frames: blockIdx.y = frame, its labels lab_stride elements further, its own code table and chunk row)
    pixlab += (size_t)blockIdx.y * lab_stride;
    clen += (size_t)blockIdx.y * K;
    chunk_bits += (size_t)blockIdx.y * gridDim.x;
    for (uint32_t i = threadIdx.x; i < K; i += kPackThreads) s_len[i] = clen[i];
    __syncthreads();
    const uint64_t first = (uint64_t)blockIdx.x * kPackChunk + (uint64_t)threadIdx.x * kPackPer;
    uint32_t bits = 0;
    if (sizeof(LabelT) == 1 && first + kPackPer <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(pixlab + first);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; i++) bits += s_len[(w[i >> 2] >> (8 * (i & 3))) & 255];
    } else {
        for (int i = 0; i < kPackPer; i++)
            if (first + i < n) bits += s_len[pixlab[first + i]];
    }
    bits = block_reduce_sum<kPackThreads>(bits);
    if (threadIdx.x == 0) chunk_bits[blockIdx.x] = bits;
}

template <typename LabelT>
__global__ __launch_bounds__(kPackThreads) void k_pack_write_lab(const LabelT *__restrict__ pixlab, uint64_t n, uint32_t K,
                                                                 const uint8_t *__restrict__ clen,
                                                                 const uint64_t *__restrict__ ccode,
                                                                 const uint64_t *__restrict__ chunk_off,
                                                                 uint32_t *__restrict__ out_words, uint64_t bit_base, uint64_t lab_stride = 0,
                                                                 uint64_t out_stride_words = 0, const uint64_t *__restrict__ bit_base_frames = nullptr,
                                                                 uint32_t img_cap = 0xffffffffu /* tests: a smaller image forces the direct route */) {
    extern __shared__ __align__(8) unsigned long long s_tab[];  // [K] codes, then [K] lens (bytes)
    pixlab += (size_t)blockIdx.y * lab_stride;   // (a batch of frames: blockIdx.y = frame)
    clen += (size_t)blockIdx.y * K;
    ccode += (size_t)blockIdx.y * K;
    chunk_off += (size_t)blockIdx.y * gridDim.x;
    out_words += (size_t)blockIdx.y * out_stride_words;
    if (bit_base_frames) bit_base = bit_base_frames[blockIdx.y];
    // the chunk's bit image: 16 bits per symbol on average fit (a palette's codes average 8); a chunk of rarer symbols goes to
    // memory piece by piece instead (below).  Sized for the worst case (64 bits per symbol, 32 KiB) the array held the kernel at
    // four blocks per CU.
    constexpr uint32_t kImgWords = kPackChunk / 2 + 2;
    __shared__ uint32_t img[kImgWords];
    __shared__ uint32_t wsum[kPackThreads / 64];
    __shared__ uint32_t s_total;
    uint8_t *s_len = reinterpret_cast<uint8_t *>(s_tab + K);
    for (uint32_t i = threadIdx.x; i < K; i += kPackThreads) { s_tab[i] = ccode[i]; s_len[i] = clen[i]; }
    __syncthreads();
    const uint64_t first = (uint64_t)blockIdx.x * kPackChunk + (uint64_t)threadIdx.x * kPackPer;
    uint32_t lab[kPackPer], l[kPackPer];
    uint32_t bits = 0;
    if (sizeof(LabelT) == 1 && first + kPackPer <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(pixlab + first);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; i++) lab[i] = (w[i >> 2] >> (8 * (i & 3))) & 255;
    } else {
#pragma unroll
        for (int i = 0; i < kPackPer; i++) lab[i] = first + i < n ? (uint32_t)pixlab[first + i] : 0xffffffffu;
    }
#pragma unroll
    for (int i = 0; i < kPackPer; i++) { l[i] = lab[i] != 0xffffffffu ? s_len[lab[i]] : 0; bits += l[i]; }
    uint32_t excl = block_exclusive_scan<kPackThreads>(bits, wsum);
    const uint64_t g0 = bit_base + chunk_off[blockIdx.x];
    const uint32_t skew = (uint32_t)(g0 & 31);
    uint32_t pos = skew + excl;
    // the image is cleared only as far as this chunk's bits reach (the array is sized for 64 bits per symbol; a palette's codes
    // average 8: clearing all of it was 8 LDS bytes per symbol)
    if (threadIdx.x == kPackThreads - 1) s_total = excl + bits;
    __syncthreads();
    const uint32_t nw_all = (skew + s_total + 31) >> 5;
    if (nw_all + 2 > min(kImgWords, img_cap)) {  // too many bits for the image: straight to memory (the output is pre-zeroed; words are big-endian bit order)
        const uint64_t w0 = g0 >> 5;
#pragma unroll 1
        for (int i = 0; i < kPackPer; i++) {
            const uint32_t L = l[i];
            if (L == 0) continue;
            const uint64_t cd = s_tab[lab[i]];
            const uint32_t w = pos >> 5, b = pos & 31, room = 32 - b;
            if (L <= room) {
                atomicOr(&out_words[w0 + w], __builtin_bswap32((uint32_t)(cd << (room - L))));
            } else {
                const uint32_t rem = L - room;
                atomicOr(&out_words[w0 + w], __builtin_bswap32((uint32_t)(cd >> rem)));
                if (rem <= 32) atomicOr(&out_words[w0 + w + 1], __builtin_bswap32((uint32_t)(cd << (32 - rem))));
                else { atomicOr(&out_words[w0 + w + 1], __builtin_bswap32((uint32_t)(cd >> (rem - 32)))); atomicOr(&out_words[w0 + w + 2], __builtin_bswap32((uint32_t)(cd << (64 - rem)))); }
            }
            pos += L;
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < nw_all + 2; i += kPackThreads) img[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPackPer; i++) {
        const uint32_t L = l[i];
        if (L == 0) continue;
        const uint64_t cd = s_tab[lab[i]];
        const uint32_t w = pos >> 5, b = pos & 31, room = 32 - b;
        if (L <= room) {
            atomicOr(&img[w], (uint32_t)(cd << (room - L)));
        } else {
            const uint32_t rem = L - room;
            atomicOr(&img[w], (uint32_t)(cd >> rem));
            if (rem <= 32) atomicOr(&img[w + 1], (uint32_t)(cd << (32 - rem)));
            else { atomicOr(&img[w + 1], (uint32_t)(cd >> (rem - 32))); atomicOr(&img[w + 2], (uint32_t)(cd << (64 - rem))); }
        }
        pos += L;
    }
    __syncthreads();
    const uint32_t total = s_total;
    if (total == 0) return;
    const uint32_t nwords = (skew + total + 31) >> 5;
    const uint64_t w0 = g0 >> 5;
    for (uint32_t i = threadIdx.x; i < nwords; i += kPackThreads) {
        const uint32_t v = __builtin_bswap32(img[i]);
        if (v == 0) continue;
        if (i == 0 || i == nwords - 1) atomicOr(&out_words[w0 + i], v);
        else out_words[w0 + i] = v;
    }
}

// key2label_d: LabelT[2^24] dense colour -> cluster label; pixlab_d receives n labels (+16 bytes of slack)
int pixel_labels(Ctx *c, const uint8_t *rgb_d, uint64_t n, const void *key2label_d, bool wide, void *pixlab_d) {
    if (n == 0) return CNIIC_OK;
    const uint32_t nchunks = (uint32_t)ceil_div(n, kPackChunk);
    if (wide)
        hipLaunchKernelGGL(k_pixel_labels<uint16_t>, dim3(nchunks), dim3(kPackThreads), 0, c->stream, rgb_d, n,
                           reinterpret_cast<const uint16_t *>(key2label_d), reinterpret_cast<uint16_t *>(pixlab_d));
    else
        hipLaunchKernelGGL(k_pixel_labels<uint8_t>, dim3(nchunks), dim3(kPackThreads), 0, c->stream, rgb_d, n,
                           reinterpret_cast<const uint8_t *>(key2label_d), reinterpret_cast<uint8_t *>(pixlab_d));
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// Packs the label stream directly at bit_base of out_d (a 4-byte aligned, pre-zeroed buffer; the bytes
// before bit_base may already hold the header).
int huff_pack_labels(Ctx *c, const void *pixlab_d, uint64_t n, bool wide, uint32_t K,
                     const uint8_t *clen_d, const uint64_t *ccode_d, uint8_t *out_d, uint64_t bit_base, uint64_t *nbits_h) {
    *nbits_h = 0;
    if (n == 0) return CNIIC_OK;
    if (reinterpret_cast<uintptr_t>(out_d) & 3) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: output must be 4-byte aligned");
    const uint32_t nchunks = (uint32_t)ceil_div(n, kPackChunk);
    DevBuf cb, co, tot;
    CNIIC_HIP_TRY(c, cb.alloc((uint64_t)nchunks * 4));
    CNIIC_HIP_TRY(c, co.alloc((uint64_t)nchunks * 8));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    if (wide)
        hipLaunchKernelGGL(k_pack_count_lab<uint16_t>, dim3(nchunks), dim3(kPackThreads), K, c->stream,
                           reinterpret_cast<const uint16_t *>(pixlab_d), n, K, clen_d, cb.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_pack_count_lab<uint8_t>, dim3(nchunks), dim3(kPackThreads), K, c->stream,
                           reinterpret_cast<const uint8_t *>(pixlab_d), n, K, clen_d, cb.as<uint32_t>());
    CNIIC_TRY(pack_scan(c, cb.as<uint32_t>(), nchunks, co.as<uint64_t>(), tot.as<uint64_t>()));
    if (wide)
        hipLaunchKernelGGL(k_pack_write_lab<uint16_t>, dim3(nchunks), dim3(kPackThreads), (size_t)K * 9 + 8, c->stream,
                           reinterpret_cast<const uint16_t *>(pixlab_d), n, K, clen_d, ccode_d, co.as<uint64_t>(),
                           reinterpret_cast<uint32_t *>(out_d), bit_base, (uint64_t)0, (uint64_t)0, (const uint64_t *)nullptr, pack_img_cap());
    else
        hipLaunchKernelGGL(k_pack_write_lab<uint8_t>, dim3(nchunks), dim3(kPackThreads), (size_t)K * 9 + 8, c->stream,
                           reinterpret_cast<const uint8_t *>(pixlab_d), n, K, clen_d, ccode_d, co.as<uint64_t>(),
                           reinterpret_cast<uint32_t *>(out_d), bit_base, (uint64_t)0, (uint64_t)0, (const uint64_t *)nullptr, pack_img_cap());
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint64_t total = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&total, tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *nbits_h = total;
    return CNIIC_OK;
}

// ---- a batch of frames coded with one palette (north_star config 4): per-frame label histograms, and the label pack
// of every frame enqueued back to back (no host round trip per frame)
template <typename LabelT>
__global__ __launch_bounds__(256) void k_frame_label_hist(const LabelT *__restrict__ pixlab, uint64_t npf, uint64_t lab_stride, uint32_t K,
                                                          uint32_t *__restrict__ out /* [frames][K] */) {
    extern __shared__ uint32_t s_hist[];  // [4][K]: one histogram per wave (a frame's pixels fall into few clusters: four times fewer collisions)
    for (uint32_t i = threadIdx.x; i < 4 * K; i += 256) s_hist[i] = 0;
    __syncthreads();
    const LabelT *base = pixlab + (size_t)blockIdx.y * lab_stride;
    uint32_t *hw = s_hist + (threadIdx.x >> 6) * K;
    const uint64_t per = ((npf + gridDim.x - 1) / gridDim.x + 15) & ~15ull;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < npf ? lo + per : npf;
    if (sizeof(LabelT) == 1 && (reinterpret_cast<uintptr_t>(base) & 15) == 0) {  // 16 labels per load
        for (uint64_t i = lo + (uint64_t)threadIdx.x * 16; i < hi; i += 256 * 16) {
            if (i + 16 <= hi) {
                const uint4 v = *reinterpret_cast<const uint4 *>(base + i);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 16; j++) atomicAdd(&hw[(w[j >> 2] >> (8 * (j & 3))) & 255], 1u);
            } else {
                for (uint64_t q = i; q < hi; q++) atomicAdd(&hw[base[q]], 1u);
            }
        }
    } else {
        for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&hw[base[i]], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < K; i += 256) {
        const uint32_t v = s_hist[i] + s_hist[K + i] + s_hist[2 * K + i] + s_hist[3 * K + i];
        if (v) atomicAdd(&out[(size_t)blockIdx.y * K + i], v);
    }
}

int frame_label_hist(Ctx *c, const void *pixlab_d, uint64_t npf, uint64_t lab_stride, uint32_t frames, bool wide, uint32_t K, uint32_t *out_d) {
    if (!frames || !npf) return CNIIC_OK;
    CNIIC_HIP_TRY(c, hipMemsetAsync(out_d, 0, (uint64_t)frames * K * 4, c->stream));
    const uint32_t bx = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(npf, 1u << 16), 1), 64);
    if (wide)
        hipLaunchKernelGGL(k_frame_label_hist<uint16_t>, dim3(bx, frames), dim3(256), (size_t)K * 16, c->stream,
                           reinterpret_cast<const uint16_t *>(pixlab_d), npf, lab_stride, K, out_d);
    else
        hipLaunchKernelGGL(k_frame_label_hist<uint8_t>, dim3(bx, frames), dim3(256), (size_t)K * 16, c->stream,
                           reinterpret_cast<const uint8_t *>(pixlab_d), npf, lab_stride, K, out_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---- the Huffman code of every frame of a batch, on the GPU (K <= 256): one 256-thread block per frame does what the host did
// per frame with std::sort + huff_build_tree + huff_codes + huff_serialize_tree (codec.cpp cc_finish_frames; huff_host.cpp) --
// 128 frames took 16 host threads 0.6 ms between two stretches of GPU work.  Same rules, so the same bytes:
//   symbols   the frame's clusters by centroid COLOUR, ascending (two clusters with one mean are one symbol; huf.rs:30, clusterc.rs:43-53)
//   tree      leaves sorted by (count, symbol), two-queue merge: the rarest first, a leaf before a branch among equals
//   codes     Bit::Zero = left, Bit::One = right from the root (huf.rs:125-135); longer than 64 bits is an error
//   header    u32 w, u32 h, then the trie in pre-order: 0 + u64 3 + r g b for a leaf, 1 + left + right for a branch (huf.rs:305-321)
// Out: the header at out + f stride, the code table of the frame's clusters, the header's bit length, the payload's bit count.
__device__ __forceinline__ void bitonic_sort_256(unsigned long long *a, uint32_t tid) {
    for (uint32_t k = 2; k <= 256; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            const uint32_t x = tid ^ j;
            if (x > tid) {
                const unsigned long long u = a[tid], v = a[x];
                const bool up = (tid & k) == 0;
                if ((u > v) == up) { a[tid] = v; a[x] = u; }
            }
        }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_frame_trees(const uint32_t *__restrict__ cnt /* [F][K] */, const uint32_t *__restrict__ cent /* [K] 0xRRGGBB */,
                                                     uint32_t K, uint32_t w, uint32_t h, uint8_t *__restrict__ out, uint64_t stride,
                                                     uint8_t *__restrict__ clen /* [F][K] */, unsigned long long *__restrict__ ccode /* [F][K] */,
                                                     unsigned long long *__restrict__ bit_base /* [F] */, unsigned long long *__restrict__ nbits /* [F] */,
                                                     unsigned long long *__restrict__ lens /* [F] */, uint32_t *__restrict__ err) {
    __shared__ unsigned long long s_sort[256];
    __shared__ uint32_t s_key[256], s_cnt[256], s_bfreq[256];
    __shared__ unsigned short s_symk[256], s_left[256], s_right[256], s_parent[512], s_size[512];
    __shared__ unsigned char s_side[512], s_len[256];
    __shared__ unsigned long long s_code[256], s_bits;
    __shared__ uint32_t s_wsum[4], s_n;
    const uint32_t tid = threadIdx.x, f = blockIdx.x;
    const uint32_t *fc = cnt + (size_t)f * K;
    const uint32_t myc = tid < K ? fc[tid] : 0u;
    // ---- symbols: (colour, cluster) sorted; heads of runs of one colour are the symbols
    s_sort[tid] = myc ? (((unsigned long long)(cent[tid] & 0xffffffu) << 32) | tid) : ~0ull;
    s_cnt[tid] = 0;
    if (tid == 0) s_bits = 0;
    bitonic_sort_256(s_sort, tid);
    const unsigned long long me = s_sort[tid];
    const bool valid = me != ~0ull;
    const uint32_t key = (uint32_t)(me >> 32), k = (uint32_t)me & 0xffffu;
    const bool head = valid && (tid == 0 || (uint32_t)(s_sort[tid - 1] >> 32) != key);
    uint32_t inc = wave_inclusive_scan(head ? 1u : 0u);
    if ((tid & 63) == 63) s_wsum[tid >> 6] = inc;
    __syncthreads();
    for (uint32_t i = 0; i < (tid >> 6); i++) inc += s_wsum[i];
    const uint32_t si = inc - 1;  // symbol of this (colour, cluster) entry
    if (tid == 255) s_n = inc;
    if (valid) {
        s_symk[k] = (unsigned short)si;
        if (head) s_key[si] = key;
        atomicAdd(&s_cnt[si], fc[k]);
    }
    __syncthreads();
    const uint32_t n = s_n;
    if (n == 0) {  // (a frame without pixels cannot happen: npf > 0)
        if (tid == 0) atomicOr(err, 1u);
        return;
    }
    // ---- leaves by (count, symbol)
    s_sort[tid] = tid < n ? (((unsigned long long)s_cnt[tid] << 32) | tid) : ~0ull;
    bitonic_sort_256(s_sort, tid);
    // ---- two-queue merge (build_tree_u32), one thread; sizes in bytes of the serialised subtrees on the way
    if (tid < n) s_size[tid] = 12;
    __syncthreads();
    if (tid == 0) {
        uint32_t li = 0, bi = 0, made = 0;
        while (made + 1 < n) {
            uint32_t node[2], fr[2];
            for (int q = 0; q < 2; q++) {
                const uint32_t lc = li < n ? (uint32_t)(s_sort[li] >> 32) : 0u;
                if (li < n && (bi >= made || lc <= s_bfreq[bi])) { fr[q] = lc; node[q] = (uint32_t)s_sort[li] & 0xffffu; li++; }
                else { fr[q] = s_bfreq[bi]; node[q] = n + bi; bi++; }
            }
            s_left[made] = (unsigned short)node[0];
            s_right[made] = (unsigned short)node[1];
            s_bfreq[made] = fr[0] + fr[1];
            s_parent[node[0]] = (unsigned short)(n + made); s_side[node[0]] = 0;
            s_parent[node[1]] = (unsigned short)(n + made); s_side[node[1]] = 1;
            s_size[n + made] = (unsigned short)(1 + s_size[node[0]] + s_size[node[1]]);
            made++;
        }
    }
    __syncthreads();
    const uint32_t root = n > 1 ? 2 * n - 2 : 0, nnodes = 2 * n - 1;
    uint8_t *o = out + (size_t)f * stride;
    const unsigned long long hbytes = 8ull + 12ull * n + (n - 1);
    const bool fits = hbytes <= stride;  // (the caller checks the whole stream against the stride once the lengths are back; the header alone must not overrun it)
    if (!fits && tid == 0) atomicOr(err, 4u);
    if (fits && tid < 8) o[tid] = (uint8_t)((tid < 4 ? w : h) >> (8 * (tid & 3)));
    // ---- every node walks up to the root: its offset in the pre-order stream, and for a leaf its code
    for (uint32_t v = tid; v < nnodes; v += 256) {
        uint32_t off = 0, depth = 0, cur = v;
        unsigned long long code = 0;
        while (cur != root) {
            const uint32_t p = s_parent[cur], sd = s_side[cur];
            off += 1 + (sd ? s_size[s_left[p - n]] : 0u);
            if (v < n) { if (depth < 64) code |= (unsigned long long)sd << depth; depth++; }
            cur = p;
        }
        uint8_t *b = o + 8 + off;
        if (v < n) {
            if (depth > 64) atomicOr(err, 2u);
            s_len[v] = (unsigned char)(depth > 64 ? 0 : depth);
            s_code[v] = code;
            atomicAdd(&s_bits, (unsigned long long)s_cnt[v] * depth);
            const uint32_t ky = s_key[v];
            if (fits) {
                b[0] = 0; b[1] = 3; b[2] = b[3] = b[4] = b[5] = b[6] = b[7] = b[8] = 0;
                b[9] = (uint8_t)(ky >> 16); b[10] = (uint8_t)(ky >> 8); b[11] = (uint8_t)ky;
            }
        } else if (fits) {
            b[0] = 1;
        }
    }
    __syncthreads();
    if (tid < K) {
        const uint32_t sy = s_symk[tid];
        clen[(size_t)f * K + tid] = myc ? s_len[sy] : (unsigned char)0;
        ccode[(size_t)f * K + tid] = myc ? s_code[sy] : 0ull;
    }
    if (tid == 0) {
        bit_base[f] = hbytes * 8;
        nbits[f] = s_bits;
        lens[f] = hbytes + (s_bits + 7) / 8;
    }
}

int frame_trees(Ctx *c, const uint32_t *cnt_d, const uint32_t *cent_d, uint32_t frames, uint32_t K, uint32_t w, uint32_t h, uint8_t *out_d, uint64_t stride,
                uint8_t *clen_d, uint64_t *ccode_d, uint64_t *bit_base_d, uint64_t *nbits_d, uint64_t *lens_d, uint32_t *err_d) {
    if (K > 256) return c->fail(CNIIC_ERR_BAD_ARG, "frame_trees: K <= 256");
    hipLaunchKernelGGL(k_frame_trees, dim3(frames), dim3(256), 0, c->stream, cnt_d, cent_d, K, w, h, out_d, stride, clen_d,
                       reinterpret_cast<unsigned long long *>(ccode_d), reinterpret_cast<unsigned long long *>(bit_base_d),
                       reinterpret_cast<unsigned long long *>(nbits_d), reinterpret_cast<unsigned long long *>(lens_d), err_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// exclusive scan (u64) of each frame's row of chunk totals: one 1024-thread block per frame
__global__ __launch_bounds__(1024) void k_pack_scan_frames(const uint32_t *__restrict__ chunk_bits, uint32_t nchunks, uint64_t *__restrict__ chunk_off,
                                                           uint64_t *__restrict__ totals) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_carry;
    const uint32_t *in = chunk_bits + (size_t)blockIdx.x * nchunks;
    uint64_t *out = chunk_off + (size_t)blockIdx.x * nchunks;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nchunks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < nchunks ? in[i] : 0ull;
        const unsigned long long inc = wave_inclusive_scan64<false>(v);
        if (lane == 63) s_w[wid] = inc;
        __syncthreads();
        unsigned long long pre = s_carry;
        for (int k = 0; k < wid; k++) pre += s_w[k];
        if (i < nchunks) out[i] = pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = s_carry;
}

// huff_pack_labels for `frames` frames of npf labels each (frame f starts at element f lab_stride, 16-byte aligned): frame f
// uses the code table at clen_d + f K / ccode_d + f K and writes behind bit_base[f] of out_d + f stride.  Three launches
// for the whole batch (grid.y = frame; one launch triple per frame was 2 ms for 128 frames, two thirds of it launch
// overhead); one synchronisation at the end brings totals_h[f] (bits packed per frame).
int huff_pack_labels_frames(Ctx *c, const void *pixlab_d, uint64_t npf, uint64_t lab_stride, uint32_t frames, bool wide, uint32_t K, const uint8_t *clen_d,
                            const uint64_t *ccode_d, uint8_t *out_d, uint64_t stride, const uint64_t *bit_base_h, uint64_t *totals_h,
                            const uint64_t *bit_base_d) {
    if (!frames || !npf) return CNIIC_OK;
    if ((reinterpret_cast<uintptr_t>(out_d) & 3) || (stride & 3)) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: output and stride must be 4-byte aligned");
    if (frames > 65535) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: at most 65535 frames per batch");
    const uint32_t nchunks = (uint32_t)ceil_div(npf, kPackChunk);
    DevBuf cb, co, tot, bb;
    CNIIC_HIP_TRY(c, cb.alloc((uint64_t)frames * nchunks * 4));
    CNIIC_HIP_TRY(c, co.alloc((uint64_t)frames * nchunks * 8));
    CNIIC_HIP_TRY(c, tot.alloc((uint64_t)frames * 8));
    if (!bit_base_d) {
        CNIIC_HIP_TRY(c, bb.alloc((uint64_t)frames * 8));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(bb.p, bit_base_h, (size_t)frames * 8, hipMemcpyHostToDevice, c->stream));
        bit_base_d = bb.as<uint64_t>();
    }
    const dim3 grid(nchunks, frames);
    if (wide)
        hipLaunchKernelGGL(k_pack_count_lab<uint16_t>, grid, dim3(kPackThreads), K, c->stream, reinterpret_cast<const uint16_t *>(pixlab_d), npf, K, clen_d,
                           cb.as<uint32_t>(), lab_stride);
    else
        hipLaunchKernelGGL(k_pack_count_lab<uint8_t>, grid, dim3(kPackThreads), K, c->stream, reinterpret_cast<const uint8_t *>(pixlab_d), npf, K, clen_d,
                           cb.as<uint32_t>(), lab_stride);
    hipLaunchKernelGGL(k_pack_scan_frames, dim3(frames), dim3(1024), 0, c->stream, (const uint32_t *)cb.as<uint32_t>(), nchunks, co.as<uint64_t>(), tot.as<uint64_t>());
    if (wide)
        hipLaunchKernelGGL(k_pack_write_lab<uint16_t>, grid, dim3(kPackThreads), (size_t)K * 9 + 8, c->stream, reinterpret_cast<const uint16_t *>(pixlab_d), npf, K,
                           clen_d, ccode_d, (const uint64_t *)co.as<uint64_t>(), reinterpret_cast<uint32_t *>(out_d), (uint64_t)0, lab_stride, stride / 4,
                           bit_base_d, pack_img_cap());
    else
        hipLaunchKernelGGL(k_pack_write_lab<uint8_t>, grid, dim3(kPackThreads), (size_t)K * 9 + 8, c->stream, reinterpret_cast<const uint8_t *>(pixlab_d), npf, K,
                           clen_d, ccode_d, (const uint64_t *)co.as<uint64_t>(), reinterpret_cast<uint32_t *>(out_d), (uint64_t)0, lab_stride, stride / 4,
                           bit_base_d, pack_img_cap());
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipMemcpyAsync(totals_h, tot.p, (size_t)frames * 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

// interior words of a chunk that are all-zero must still be written: the output is pre-zeroed.

template <int SRC>
static int pack_impl(Ctx *c, const void *src_d, uint64_t n, const uint32_t *rank_table_d, const uint8_t *len_d,
                     const uint64_t *code_d, uint8_t *out_d, uint64_t bit_base, uint64_t *nbits_h) {
    *nbits_h = 0;
    if (n == 0) return CNIIC_OK;
    if (reinterpret_cast<uintptr_t>(out_d) & 3) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: output must be 4-byte aligned");
    const uint64_t nchunks64 = ceil_div(n, kPackChunk);
    if (nchunks64 > 0x7fffffffull) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: too many symbols");
    const uint32_t nchunks = (uint32_t)nchunks64;
    DevBuf cb, co, tot;
    CNIIC_HIP_TRY(c, cb.alloc((uint64_t)nchunks * 4));
    CNIIC_HIP_TRY(c, co.alloc((uint64_t)nchunks * 8));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    hipLaunchKernelGGL(k_pack_count<SRC>, dim3(nchunks), dim3(kPackThreads), 0, c->stream, src_d, n, rank_table_d, len_d, cb.as<uint32_t>());
    CNIIC_TRY(pack_scan(c, cb.as<uint32_t>(), nchunks, co.as<uint64_t>(), tot.as<uint64_t>()));
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint64_t total = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&total, tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *nbits_h = total;
    hipLaunchKernelGGL(k_pack_write<SRC>, dim3(nchunks), dim3(kPackThreads), 0, c->stream, src_d, n, rank_table_d, len_d, code_d,
                       co.as<uint64_t>(), reinterpret_cast<uint32_t *>(out_d), bit_base, pack_img_cap());
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));  // cb/co/tot are released on return
    return CNIIC_OK;
}


// ---------------------------------------------------------------- generic fast path (codes <= 26 bits inline)
// The dense symbol table is overwritten with (len << 26 | code) per symbol key, so a symbol costs
// ONE random read; pass 1 leaves that word per symbol in a linear array (in place over the symbol
// stream when it is ours), pass 2 streams it back.
__global__ __launch_bounds__(256) void k_fill_code32(const uint32_t *__restrict__ keys, const uint8_t *__restrict__ len,
                                                     const uint64_t *__restrict__ code, uint64_t U, uint32_t *__restrict__ table) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride)
        table[keys ? keys[i] : (uint32_t)i] = len[i] <= 26 ? ((uint32_t)len[i] << 26) | (uint32_t)code[i]
                                      : (kEscape << 26) | (uint32_t)i;  // long (rare) code: escape to len[rank] / code[rank]
}

template <int SRC>
__global__ __launch_bounds__(kPackThreads) void k_pack_count32(const void *__restrict__ src, uint64_t n,
                                                               const uint32_t *__restrict__ code_table,
                                                               const uint8_t *__restrict__ len,
                                                               uint32_t *__restrict__ packed, uint32_t *__restrict__ chunk_bits) {
    const uint64_t first = (uint64_t)blockIdx.x * kPackChunk + (uint64_t)threadIdx.x * kPackPer;
    uint32_t key[kPackPer];
    bool full = first + kPackPer <= n;
    if (SRC == SRC_RGB) {
        const uint8_t *rgb = reinterpret_cast<const uint8_t *>(src);
        if (full && ((reinterpret_cast<uintptr_t>(rgb) & 15) == 0)) load16px_keys(reinterpret_cast<const uint4 *>(rgb + 3 * first), key);
        else
            for (int i = 0; i < kPackPer; i++) key[i] = first + i < n ? rgb_key(rgb + 3 * (first + i)) : 0u;
    } else {
        const uint32_t *k = reinterpret_cast<const uint32_t *>(src);
        if (full && ((reinterpret_cast<uintptr_t>(k) & 15) == 0)) {
            const uint4 *v = reinterpret_cast<const uint4 *>(k + first);
#pragma unroll
            for (int j = 0; j < 4; j++) { uint4 q = v[j]; key[4 * j] = q.x; key[4 * j + 1] = q.y; key[4 * j + 2] = q.z; key[4 * j + 3] = q.w; }
        } else {
            for (int i = 0; i < kPackPer; i++) key[i] = first + i < n ? k[first + i] : 0u;
        }
    }
    uint32_t v[kPackPer];
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < kPackPer; i++) {
        v[i] = first + i < n ? code_table[key[i]] : 0u;
        const uint32_t L = v[i] >> 26;
        bits += L == kEscape ? (uint32_t)len[v[i] & 0x3ffffffu] : L;
    }
    if (full && ((reinterpret_cast<uintptr_t>(packed) & 15) == 0)) {
        uint4 *o = reinterpret_cast<uint4 *>(packed + first);
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = make_uint4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    } else {
        for (int i = 0; i < kPackPer; i++)
            if (first + i < n) packed[first + i] = v[i];
    }
    bits = block_reduce_sum<kPackThreads>(bits);
    if (threadIdx.x == 0) chunk_bits[blockIdx.x] = bits;
}

__global__ __launch_bounds__(kPackThreads) void k_pack_write32(const uint32_t *__restrict__ packed, uint64_t n,
                                                               const uint8_t *__restrict__ len, const uint64_t *__restrict__ code,
                                                               const uint64_t *__restrict__ chunk_off,
                                                               uint32_t *__restrict__ out_words, uint64_t bit_base, uint32_t img_cap) {
    constexpr uint32_t kImgWords = kPackChunk + 2;  // (32 bits per symbol on average; see k_pack_write)
    __shared__ uint32_t img[kImgWords];
    __shared__ uint32_t wsum[kPackThreads / 64];
    __shared__ uint32_t s_total;
    const uint64_t first = (uint64_t)blockIdx.x * kPackChunk + (uint64_t)threadIdx.x * kPackPer;
    uint32_t v[kPackPer];
    if (first + kPackPer <= n && ((reinterpret_cast<uintptr_t>(packed) & 15) == 0)) {
        const uint4 *p = reinterpret_cast<const uint4 *>(packed + first);
#pragma unroll
        for (int j = 0; j < 4; j++) { uint4 q = p[j]; v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w; }
    } else {
#pragma unroll
        for (int i = 0; i < kPackPer; i++) v[i] = first + i < n ? packed[first + i] : 0u;
    }
    uint32_t bits = 0;
    uint32_t l[kPackPer];
#pragma unroll
    for (int i = 0; i < kPackPer; i++) {
        const uint32_t L = v[i] >> 26;
        l[i] = L == kEscape ? (uint32_t)len[v[i] & 0x3ffffffu] : L;
        bits += l[i];
    }
    const uint32_t excl = block_exclusive_scan<kPackThreads>(bits, wsum);
    const uint64_t g0 = bit_base + chunk_off[blockIdx.x];
    const uint32_t skew = (uint32_t)(g0 & 31);
    uint32_t pos = skew + excl;
    if (threadIdx.x == kPackThreads - 1) s_total = excl + bits;
    __syncthreads();
    const uint32_t total = s_total;
    if (total == 0) return;
    const uint32_t nwords = (skew + total + 31) >> 5;
    const uint64_t w0 = g0 >> 5;
    auto code_of = [&](int i) -> uint64_t { return (v[i] >> 26) == kEscape ? code[v[i] & 0x3ffffffu] : (uint64_t)(v[i] & 0x3ffffffu); };
    if (nwords + 2 > min(kImgWords, img_cap)) {
#pragma unroll 1
        for (int i = 0; i < kPackPer; i++) {
            if (l[i]) pack_put<true>(out_words + w0, pos, l[i], code_of(i));
            pos += l[i];
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < nwords + 2; i += kPackThreads) img[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPackPer; i++) {
        if (l[i]) pack_put<false>(img, pos, l[i], code_of(i));
        pos += l[i];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nwords; i += kPackThreads) {
        const uint32_t o = __builtin_bswap32(img[i]);
        if (o == 0) continue;
        if (i == 0 || i == nwords - 1) atomicOr(&out_words[w0 + i], o);
        else out_words[w0 + i] = o;
    }
}

// ---------------------------------------------------------------- the leaves of a large alphabet, sorted by (count, key)
// huf.rs:58-117 `build` takes the two rarest subtrees again and again; with the leaves sorted by count (ties: ascending key,
// DESIGN.md 2 D1) that is a two-queue merge on the host, and the sort -- 26 ms of one core for 6.8 M colours -- is a stable
// LSD radix sort here: 8 bits a pass over count << 32 | rank (the compaction's order is ascending key, so rank order = key
// order and a stable sort by count keeps it inside equal counts).  A pass: digit counts per 4096-element block, one scan
// over [digit][block], then every block places its elements -- 16 rounds of 256 in element order, the rank among equal
// digits from ballots inside a wave, wave totals and running digit counters in LDS across waves and rounds.
constexpr int kSortThreads = 256, kSortPer = 16, kSortBlock = kSortThreads * kSortPer;
__global__ __launch_bounds__(256) void k_leaf_init(const uint64_t *__restrict__ counts, uint32_t n, uint64_t *__restrict__ leaf) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) leaf[i] = (counts[i] << 32) | i;
}
__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const uint64_t *__restrict__ src, uint32_t n, uint32_t shift, uint32_t nblocks,
                                                            uint32_t *__restrict__ blockhist /* [256][nblocks] */) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSortPer; j++) {
        const uint32_t i = blockIdx.x * kSortBlock + j * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(src[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    blockhist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint64_t *__restrict__ src, uint32_t n, uint32_t shift, uint32_t nblocks,
                                                               const uint64_t *__restrict__ blockoff /* [256][nblocks], scanned */,
                                                               uint64_t *__restrict__ dst) {
    __shared__ uint32_t base[256];            // where the block's next element of a digit goes (the low 32 bits: n < 2^32)
    __shared__ uint32_t wcnt[4][256];         // this round's elements per wave and digit
    const uint32_t wave = threadIdx.x >> 6;
    base[threadIdx.x] = (uint32_t)blockoff[(size_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; w++) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    for (int j = 0; j < kSortPer; j++) {
        const uint32_t i = blockIdx.x * kSortBlock + j * kSortThreads + threadIdx.x;
        const bool ok = i < n;
        const uint64_t v = ok ? src[i] : 0;
        const uint32_t dg = (uint32_t)(v >> shift) & 255u;
        // the lanes of the wave with the same digit
        unsigned long long peers = __builtin_amdgcn_ballot_w64(ok);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64((dg >> b) & 1u);
            peers &= ((dg >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        if (ok && before == 0) wcnt[wave][dg] = (uint32_t)__popcll(peers);  // (the first of its peers)
        __syncthreads();
        if (ok) {
            uint32_t pos = base[dg] + before;
            for (uint32_t w = 0; w < wave; w++) pos += wcnt[w][dg];
            dst[pos] = v;
        }
        __syncthreads();
        base[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
#pragma unroll
        for (int w = 0; w < 4; w++) wcnt[w][threadIdx.x] = 0;
        __syncthreads();
    }
}

// counts_d: n counts below 2^32 in ascending key order -> sorted_d: count << 32 | rank, ascending (count, rank); max_count: a
// bound on the counts (the number of symbols).  tmp_d: n u64 of scratch.  The result may end in either buffer: *out_d tells.
// stable LSD radix sort of n u64 values in buf_a by their bits lo_bit .. 63 (8 bits a pass); the result ends in either buffer: *out_d
int huff_sort_u64(Ctx *c, uint64_t *buf_a, uint64_t *buf_b, uint32_t n, uint32_t lo_bit, uint64_t **out_d) {
    const uint32_t nblocks = ceil_div(n, (uint32_t)kSortBlock);
    DevBuf hist, off, tot;
    CNIIC_HIP_TRY(c, hist.alloc((uint64_t)256 * nblocks * 4));
    CNIIC_HIP_TRY(c, off.alloc((uint64_t)256 * nblocks * 8));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    uint64_t *src = buf_a, *dst = buf_b;
    for (uint32_t shift = lo_bit & ~7u; shift < 64; shift += 8) {
        hipLaunchKernelGGL(k_sort_hist, dim3(nblocks), dim3(kSortThreads), 0, c->stream, (const uint64_t *)src, n, shift, nblocks, hist.as<uint32_t>());
        CNIIC_TRY(pack_scan_counts(c, hist.as<uint32_t>(), 256 * nblocks, off.as<uint64_t>(), tot.as<uint64_t>()));
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks), dim3(kSortThreads), 0, c->stream, (const uint64_t *)src, n, shift, nblocks,
                           (const uint64_t *)off.as<uint64_t>(), dst);
        std::swap(src, dst);
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    *out_d = src;
    return CNIIC_OK;
}

int huff_sort_leaves_dev(Ctx *c, const uint64_t *counts_d, uint32_t n, uint64_t max_count, uint64_t *buf_a, uint64_t *buf_b, uint64_t **out_d) {
    const uint32_t nblocks = ceil_div(n, (uint32_t)kSortBlock);
    DevBuf hist, off, tot;
    CNIIC_HIP_TRY(c, hist.alloc((uint64_t)256 * nblocks * 4));
    CNIIC_HIP_TRY(c, off.alloc((uint64_t)256 * nblocks * 8));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    hipLaunchKernelGGL(k_leaf_init, dim3(ceil_div(n, 256u)), dim3(256), 0, c->stream, counts_d, n, buf_a);
    uint64_t *src = buf_a, *dst = buf_b;
    for (uint32_t shift = 32; shift < 64 && (max_count >> (shift - 32)) != 0; shift += 8) {
        hipLaunchKernelGGL(k_sort_hist, dim3(nblocks), dim3(kSortThreads), 0, c->stream, (const uint64_t *)src, n, shift, nblocks, hist.as<uint32_t>());
        CNIIC_TRY(pack_scan_counts(c, hist.as<uint32_t>(), 256 * nblocks, off.as<uint64_t>(), tot.as<uint64_t>()));
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks), dim3(kSortThreads), 0, c->stream, (const uint64_t *)src, n, shift, nblocks,
                           (const uint64_t *)off.as<uint64_t>(), dst);
        std::swap(src, dst);
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    *out_d = src;
    return CNIIC_OK;
}

// ---------------------------------------------------------------- codes and the serialised decoder of a large alphabet
// The host makes the tree (a sequential merge); with 10^5 .. 10^7 leaves what follows is the GPU's: every leaf walks to the
// root and collects its code (the side it hangs on at depth d is bit len - d), and its place in BinTrie::serialize's
// pre-order (huf.rs:305-321): a node sits 1 byte behind its parent if it is the left child, behind the parent's tag and the
// whole left subtree if it is the right one, and a subtree of m leaves takes m (2 + S) - 1 bytes (m leaves of 1 + S bytes,
// m - 1 branch tags).  Branch tags are the bytes no leaf record covers: the decoder is filled with 1s, then the leaves
// write their records.  (One host core took 30 ms for the codes and 40 ms for the decoder of 6.8 M colours.)
__global__ __launch_bounds__(256) void k_tree_parents(const uint32_t *__restrict__ left, const uint32_t *__restrict__ right, uint32_t n,
                                                      uint32_t *__restrict__ par) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i + 1 >= n) return;
    par[left[i]] = (n + i) << 1;
    par[right[i]] = ((n + i) << 1) | 1u;
}
__global__ __launch_bounds__(256) void k_tree_leaf_codes(const uint32_t *__restrict__ par, const uint32_t *__restrict__ left,
                                                         const uint32_t *__restrict__ nleaves, const uint64_t *__restrict__ counts, uint32_t n,
                                                         uint32_t root, uint32_t rec_bytes /* 2 + S */, uint8_t *__restrict__ len,
                                                         uint64_t *__restrict__ code, uint64_t *__restrict__ off,
                                                         unsigned long long *__restrict__ totals /* [0] payload bits, [1] codes longer than 64 */) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bits = 0;
    if (i < n) {
        uint32_t node = i, d = 0;
        uint64_t cd = 0, o = 0;
        while (node != root) {
            const uint32_t p = par[node], side = p & 1u, pn = p >> 1;
            if (d < 64) cd |= (uint64_t)side << d;
            d++;
            o += 1;
            if (side) {
                const uint32_t l = left[pn - n];
                o += (uint64_t)(l < n ? 1u : nleaves[l - n]) * rec_bytes - 1;
            }
            node = pn;
        }
        if (d > 64) { atomicAdd(&totals[1], 1ull); d = 0; cd = 0; }
        len[i] = (uint8_t)d;
        code[i] = cd;
        off[i] = o;
        bits = counts[i] * d;
    }
    bits = wave_reduce_sum64(bits);
    if ((threadIdx.x & 63) == 0 && bits) atomicAdd(&totals[0], bits);
}
// SER_ENUM_LEAF (0) + the symbol: Rgb<u8> = u64 length 3 + 3 bytes (ser.rs:210-214), SignedColor = three i16 LE (hilbertc.rs:561-565)
__global__ __launch_bounds__(256) void k_tree_leaf_records(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ off, uint32_t n, int rgb,
                                                           uint8_t *__restrict__ trie) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint8_t *o = trie + off[i];
    const uint32_t key = keys[i];
    // (a record starts at any byte: whole words stored at unaligned addresses -- 12 single bytes per leaf took 16 ms for 6.8 M)
    if (rgb) {
        const uint32_t w0 = 3u << 8, w1 = 0u, w2 = (key >> 16 & 255u) << 8 | (key >> 8 & 255u) << 16 | (key & 255u) << 24;
        __builtin_memcpy(o, &w0, 4);
        __builtin_memcpy(o + 4, &w1, 4);
        __builtin_memcpy(o + 8, &w2, 4);
    } else {
        uint32_t f[3];
#pragma unroll
        for (int k = 0; k < 3; k++) f[k] = (uint32_t)((int)((key >> (18 - 9 * k)) & 511) - 255) & 0xffffu;
        const uint32_t w0 = f[0] << 8 | (f[1] & 255u) << 24;
        const uint16_t h1 = (uint16_t)(f[1] >> 8 | (f[2] & 255u) << 8);
        __builtin_memcpy(o, &w0, 4);
        __builtin_memcpy(o + 4, &h1, 2);
        o[6] = (uint8_t)(f[2] >> 8);
    }
}

// left_d / right_d / nleaves_d: the tree's n - 1 branches (leaves 0 .. n - 1 = the symbols in ascending key order, branch i = node
// n + i); counts_d: the symbols' counts.  len_d / code_d / off_d: n entries out; totals_d[0] = the payload's bits, [1] = codes
// longer than 64 bits (the caller reads them after the stream)
int huff_tree_codes(Ctx *c, const uint32_t *left_d, const uint32_t *right_d, const uint32_t *nleaves_d, const uint64_t *counts_d, uint32_t n,
                    uint32_t root, int sym_kind, uint8_t *len_d, uint64_t *code_d, uint64_t *off_d, uint64_t *totals_d) {
    DevBuf par;
    CNIIC_HIP_TRY(c, par.alloc((2ull * n) * 4));
    CNIIC_HIP_TRY(c, hipMemsetAsync(totals_d, 0, 16, c->stream));
    if (n > 1) hipLaunchKernelGGL(k_tree_parents, dim3(ceil_div(n - 1, 256u)), dim3(256), 0, c->stream, left_d, right_d, n, par.as<uint32_t>());
    hipLaunchKernelGGL(k_tree_leaf_codes, dim3(ceil_div(n, 256u)), dim3(256), 0, c->stream, (const uint32_t *)par.as<uint32_t>(), left_d, nleaves_d, counts_d, n,
                       root, (uint32_t)(2 + (sym_kind == CNIIC_SYM_RGB ? 11 : 6)), len_d, code_d, off_d, reinterpret_cast<unsigned long long *>(totals_d));
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// the decoder of n leaves at trie_d: huff_tree_bytes(sym_kind, n) bytes
int huff_tree_serialize_dev(Ctx *c, const uint32_t *keys_d, const uint64_t *off_d, uint32_t n, int sym_kind, uint8_t *trie_d, uint64_t trie_bytes) {
    CNIIC_HIP_TRY(c, hipMemsetAsync(trie_d, 1, trie_bytes, c->stream));  // SER_ENUM_BRANCH huf.rs:297
    hipLaunchKernelGGL(k_tree_leaf_records, dim3(ceil_div(n, 256u)), dim3(256), 0, c->stream, keys_d, off_d, n, sym_kind == CNIIC_SYM_RGB ? 1 : 0, trie_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---------------------------------------------------------------- the tree of a large alphabet without the host's merge (round 3)
// The two-queue merge (huff_host.cpp) is sequential -- 8.8 ms for the 6.8 M colours of a photograph, most of what `hufman`
// still cost there -- but its INPUT is not 6.8 M different things: the leaves come sorted by count, and a photograph's counts
// are a few thousand values, each shared by a long run of leaves.  A run merges with itself: its elements are taken two by two,
// in order, and become a run of branches of twice the count, which queues up behind the branches made before (branches are
// made in non-decreasing order of count, so their queue stays sorted).  Only where a run has an odd element left does the next
// run's first element pair with it.  So the host merges RUNS -- leaf runs in count order, branch runs in the order they were
// made, a leaf run before a branch run of the same count -- and emits one descriptor per run ("pairs k0 ..
// k0 + m - 1 are the consecutive elements of this run") or straddling pair: tens of thousands of steps instead of millions,
// and the GPU expands the descriptors into the left / right arrays (pair k IS branch k).
// What the merge also gave the old code, the number of leaves below every branch (a leaf's place in the serialised decoder was
// summed from them along its path), is not needed: in pre-order the leaves appear in ascending order of their codes, and in
// front of the leaf of rank r lie r leaf records and as many branch tags as the leaves up to and including it are the LEFTMOST
// leaf of -- a leaf is the leftmost of one branch per trailing 0 of its code.  So: sort the leaves by code, scan the trailing
// zeros.
struct TreeDesc { uint32_t k0, kind, a, b; };   // kind 0: leaf run from sorted position a; 1: branch run from branch a; 2: one pair (refs a, b)
constexpr uint32_t kRefBranch = 0x80000000u;    // a node reference: a sorted leaf position, or kRefBranch | branch number
constexpr uint32_t kMaxLeafRuns = 1u << 16;     // more runs of equal count than this: the plain merge on the host
constexpr uint32_t kRunsMinLeaves = 1u << 16;   // (by the sweep in NOTES.md C: from 384^2 `hufman` on the runs win; below, and when the counts do not come in runs, the host's merge)

__global__ __launch_bounds__(256) void k_leaf_run_count(const uint64_t *__restrict__ sorted, uint32_t n, uint32_t *__restrict__ nruns) {
    uint32_t mine = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        mine += i == 0 || (sorted[i] >> 32) != (sorted[i - 1] >> 32);
    mine = block_reduce_sum<256>(mine);
    if (threadIdx.x == 0 && mine) atomicAdd(nruns, mine);
}
__global__ __launch_bounds__(256) void k_leaf_run_list(const uint64_t *__restrict__ sorted, uint32_t n, uint2 *__restrict__ runs, uint32_t *__restrict__ cursor) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        if (i == 0 || (sorted[i] >> 32) != (sorted[i - 1] >> 32)) runs[atomicAdd(cursor, 1u)] = make_uint2(i, (uint32_t)(sorted[i] >> 32));  // (start, count): any order
}
// pair k = branch k: its children
__global__ __launch_bounds__(256) void k_tree_expand(const TreeDesc *__restrict__ desc, uint32_t ndesc, const uint64_t *__restrict__ sorted, uint32_t n,
                                                     uint32_t *__restrict__ left, uint32_t *__restrict__ right) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k + 1 >= n) return;
    uint32_t a = 0, b = ndesc;   // last descriptor with k0 <= k
    while (b - a > 1) { const uint32_t m = a + (b - a) / 2; if (desc[m].k0 <= k) a = m; else b = m; }
    const TreeDesc d = desc[a];
    const uint32_t i = k - d.k0;
    auto node = [&](uint32_t ref) { return (ref & kRefBranch) ? n + (ref & ~kRefBranch) : (uint32_t)sorted[ref]; };
    uint32_t l, r;
    if (d.kind == 0) { l = (uint32_t)sorted[d.a + 2 * i]; r = (uint32_t)sorted[d.a + 2 * i + 1]; }
    else if (d.kind == 1) { l = n + d.a + 2 * i; r = n + d.a + 2 * i + 1; }
    else { l = node(d.a); r = node(d.b); }
    left[k] = l;
    right[k] = r;
}
// every leaf walks to the root: length, code (the deepest bit last), count x length; totals: [0] payload bits, [1] codes longer
// than 64 bits, [2] the longest code.  (Grid-stride, one set of atomics per block: per wave they are 10^5 on the same words.)
__global__ __launch_bounds__(256) void k_tree_leaf_walk(const uint32_t *__restrict__ par, const uint64_t *__restrict__ counts, uint32_t n, uint32_t root,
                                                        uint8_t *__restrict__ len, uint64_t *__restrict__ code, unsigned long long *__restrict__ totals) {
    __shared__ unsigned long long s_bits[4];
    __shared__ uint32_t s_mx[4], s_long[4];
    unsigned long long bits = 0;
    uint32_t mx = 0, toolong = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint32_t node = i, d = 0;
        uint64_t cd = 0;
        while (node != root) {
            const uint32_t p = par[node];
            if (d < 64) cd |= (uint64_t)(p & 1u) << d;
            d++;
            node = p >> 1;
        }
        if (d > 64) { toolong++; d = 0; cd = 0; }
        len[i] = (uint8_t)d;
        code[i] = cd;
        bits += counts[i] * d;
        mx = max(mx, d);
    }
    bits = wave_reduce_sum64(bits); mx = wave_reduce_max(mx); toolong = wave_reduce_sum(toolong);
    if ((threadIdx.x & 63) == 0) { s_bits[threadIdx.x >> 6] = bits; s_mx[threadIdx.x >> 6] = mx; s_long[threadIdx.x >> 6] = toolong; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long bt = s_bits[0] + s_bits[1] + s_bits[2] + s_bits[3];
        const uint32_t lg = s_long[0] + s_long[1] + s_long[2] + s_long[3];
        if (bt) atomicAdd(&totals[0], bt);
        if (lg) atomicAdd(&totals[1], (unsigned long long)lg);
        atomicMax(&totals[2], (unsigned long long)max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3])));
    }
}
// code order: key = the code left-aligned in 32 bits << 32 | leaf (codes of up to 32 bits)
__global__ __launch_bounds__(256) void k_code_keys(const uint64_t *__restrict__ code, const uint8_t *__restrict__ len, uint32_t n, uint64_t *__restrict__ keys) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = len[i];
    keys[i] = ((l ? (uint64_t)((uint32_t)code[i] << (32 - l)) : 0ull) << 32) | i;
}
// in code order: how many branches the leaf is the leftmost leaf of = the trailing zeros of its code
__global__ __launch_bounds__(256) void k_code_tz(const uint64_t *__restrict__ sorted_keys, const uint64_t *__restrict__ code, const uint8_t *__restrict__ len, uint32_t n,
                                                 uint32_t *__restrict__ z) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const uint32_t i = (uint32_t)sorted_keys[r], l = len[i], cd = (uint32_t)code[i];
    z[r] = cd ? min((uint32_t)__builtin_ctz(cd), l) : l;
}
// where the leaf's record starts in the serialised decoder: rank records of rec bytes + the branch tags in front of it
__global__ __launch_bounds__(256) void k_code_off(const uint64_t *__restrict__ sorted_keys, const uint32_t *__restrict__ z, const uint64_t *__restrict__ zex, uint32_t n,
                                                  uint32_t rec, uint64_t *__restrict__ off) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    off[(uint32_t)sorted_keys[r]] = (uint64_t)r * rec + zex[r] + z[r];
}

// the host's part: runs[] = the R runs of equal count among the n sorted leaves as (first sorted position, count), in order
static bool merge_runs(const uint2 *runs, uint32_t R, uint32_t n, std::vector<TreeDesc> &desc) {
    struct Run { uint64_t v; uint32_t first, cnt; };
    std::vector<Run> br;   // branch runs in the order they were made (their counts never decrease)
    size_t bh = 0;
    uint32_t lh = 0, made = 0;
    bool carry = false;
    uint32_t carry_ref = 0;
    uint64_t carry_v = 0;
    desc.clear();
    auto new_branches = [&](uint64_t v, uint32_t cnt) {   // branches made .. made + cnt - 1 have count v
        if (!br.empty() && br.back().v == v && br.back().first + br.back().cnt == made) br.back().cnt += cnt;
        else br.push_back({v, made, cnt});
        made += cnt;
    };
    while (made + 1 < n) {
        const bool leaf = lh < R && (bh >= br.size() || (uint64_t)runs[lh].y <= br[bh].v);   // among equally rare: a leaf before a branch
        if (!leaf && bh >= br.size()) return false;   // (cannot happen: something is left to pair)
        uint64_t v;
        uint32_t pos, rem;
        if (leaf) { v = runs[lh].y; pos = runs[lh].x; rem = (lh + 1 < R ? runs[lh + 1].x : n) - pos; lh++; }
        else { v = br[bh].v; pos = br[bh].first; rem = br[bh].cnt; bh++; }
        const uint32_t flag = leaf ? 0u : kRefBranch;
        if (carry && rem) {
            desc.push_back({made, 2u, carry_ref, flag | pos});
            new_branches(carry_v + v, 1);
            pos++; rem--; carry = false;
        }
        const uint32_t m = rem / 2;
        if (m) {
            desc.push_back({made, leaf ? 0u : 1u, pos, 0u});
            new_branches(2 * v, m);
            pos += 2 * m; rem -= 2 * m;
        }
        if (rem) { carry = true; carry_ref = flag | pos; carry_v = v; }
    }
    return true;
}

// sorted_d: the n >= 2 leaves as count << 32 | leaf, ascending (huff_sort_leaves_dev); counts_d: per leaf.  On *built: len_d / code_d /
// off_d hold every leaf's code length, code and place in the serialised decoder, *nbits_h the payload's bits.  !*built: the counts
// come in too many runs, or a code is longer than 32 bits -- the caller takes the host's merge (which handles everything).
int huff_tree_from_runs(Ctx *c, const uint64_t *sorted_d, const uint64_t *counts_d, uint32_t n, int sym_kind, uint8_t *len_d, uint64_t *code_d,
                        uint64_t *off_d, uint64_t *nbits_h, bool *built) {
    *built = false;
    // (this path has three waits for the stream and a sort by code in it, ~0.4 ms whatever n is, and its host part is linear in the
    // RUNS: `delta` at 16384^2, 54 K leaves of as many different counts: 2.06 ms against 1.68 through the host's merge.  So: from 2^16
    // leaves on, and only when the leaves outnumber the runs four to one -- an image whose colours are nearly all distinct (`hufman`
    // 512^2: 2.5 10^5 leaves, a dozen runs: 0.73 -> 0.40 ms).  CNIIC_HUF_RUNS_MIN moves the line, tests set 0 and take any runs.)
    const char *rm = test_env("CNIIC_HUF_RUNS_MIN");
    if (n < 2 || n < (rm ? (uint32_t)atoi(rm) : kRunsMinLeaves) || test_env("CNIIC_HUF_HOST_MERGE")) return CNIIC_OK;
    DevBuf small, runs_d, desc_d, tree_d, par, keys_a, keys_b, z_d, zex_d, tot_d;
    CNIIC_HIP_TRY(c, small.alloc(64));
    CNIIC_HIP_TRY(c, hipMemsetAsync(small.p, 0, 64, c->stream));
    uint32_t *nruns_d = small.as<uint32_t>(), *cursor_d = nruns_d + 1;
    unsigned long long *totals = reinterpret_cast<unsigned long long *>(small.as<uint8_t>() + 16);   // [0] bits, [1] too long, [2] longest
    const uint32_t g = std::min<uint32_t>(ceil_div(n, 256u), 2048u);
    hipLaunchKernelGGL(k_leaf_run_count, dim3(g), dim3(256), 0, c->stream, sorted_d, n, nruns_d);
    CNIIC_HIP_TRY(c, ctx_pinned_u(c));
    volatile uint64_t *pin = reinterpret_cast<volatile uint64_t *>(c->pinned_u) + 4300;   // (slots of this function's own)
    CNIIC_HIP_TRY(c, hipMemcpyAsync(const_cast<uint64_t *>(pin), nruns_d, 4, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint32_t R = (uint32_t)pin[0];
    if (R == 0 || R > kMaxLeafRuns || (!rm && R > n / 4)) return CNIIC_OK;
    CNIIC_HIP_TRY(c, runs_d.alloc((uint64_t)R * 8));
    hipLaunchKernelGGL(k_leaf_run_list, dim3(g), dim3(256), 0, c->stream, sorted_d, n, runs_d.as<uint2>(), cursor_d);
    // (pinned_huf is the caller's: used as it is when large enough, never grown here -- the caller holds pointers into it)
    std::vector<uint2> runs(R);
    const bool pinned_runs = c->pinned_huf && c->pinned_huf_bytes >= (uint64_t)R * 8;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(pinned_runs ? c->pinned_huf : (void *)runs.data(), runs_d.p, (uint64_t)R * 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (pinned_runs) memcpy(runs.data(), c->pinned_huf, (uint64_t)R * 8);
    std::sort(runs.begin(), runs.end(), [](const uint2 &x, const uint2 &y) { return x.x < y.x; });
    std::vector<TreeDesc> desc;
    if (!merge_runs(runs.data(), R, n, desc)) return CNIIC_OK;
    host_trace().mark("huf: tree by runs (host)");
    const uint64_t dbytes = desc.size() * sizeof(TreeDesc);
    CNIIC_HIP_TRY(c, desc_d.alloc(dbytes));
    const bool pinned_desc = c->pinned_huf && c->pinned_huf_bytes >= dbytes;
    if (pinned_desc) memcpy(c->pinned_huf, desc.data(), dbytes);
    CNIIC_HIP_TRY(c, hipMemcpyAsync(desc_d.p, pinned_desc ? c->pinned_huf : (const void *)desc.data(), dbytes, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, tree_d.alloc(2ull * (n - 1) * 4));
    CNIIC_HIP_TRY(c, par.alloc(2ull * n * 4));
    uint32_t *left_d = tree_d.as<uint32_t>(), *right_d = left_d + (n - 1);
    hipLaunchKernelGGL(k_tree_expand, dim3(ceil_div(n - 1, 256u)), dim3(256), 0, c->stream, (const TreeDesc *)desc_d.as<TreeDesc>(), (uint32_t)desc.size(), sorted_d, n,
                       left_d, right_d);
    hipLaunchKernelGGL(k_tree_parents, dim3(ceil_div(n - 1, 256u)), dim3(256), 0, c->stream, (const uint32_t *)left_d, (const uint32_t *)right_d, n, par.as<uint32_t>());
    hipLaunchKernelGGL(k_tree_leaf_walk, dim3(g), dim3(256), 0, c->stream, (const uint32_t *)par.as<uint32_t>(), counts_d, n, 2 * n - 2, len_d, code_d, totals);
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipMemcpyAsync(const_cast<uint64_t *>(pin), totals, 24, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint64_t nbits = pin[0], toolong = pin[1], longest = pin[2];
    if (toolong || longest > 32) return CNIIC_OK;   // (the sort below keys on 32 bits of code)
    // the leaves in code order
    CNIIC_HIP_TRY(c, keys_a.alloc((uint64_t)n * 8));
    CNIIC_HIP_TRY(c, keys_b.alloc((uint64_t)n * 8));
    CNIIC_HIP_TRY(c, z_d.alloc((uint64_t)n * 4));
    CNIIC_HIP_TRY(c, zex_d.alloc((uint64_t)n * 8));
    CNIIC_HIP_TRY(c, tot_d.alloc(8));
    hipLaunchKernelGGL(k_code_keys, dim3(ceil_div(n, 256u)), dim3(256), 0, c->stream, (const uint64_t *)code_d, (const uint8_t *)len_d, n, keys_a.as<uint64_t>());
    uint64_t *srt = nullptr;
    CNIIC_TRY(huff_sort_u64(c, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), n, 64 - (uint32_t)std::max<uint64_t>(longest, 1), &srt));
    hipLaunchKernelGGL(k_code_tz, dim3(ceil_div(n, 256u)), dim3(256), 0, c->stream, (const uint64_t *)srt, (const uint64_t *)code_d, (const uint8_t *)len_d, n, z_d.as<uint32_t>());
    CNIIC_TRY(pack_scan(c, z_d.as<uint32_t>(), n, zex_d.as<uint64_t>(), tot_d.as<uint64_t>()));
    hipLaunchKernelGGL(k_code_off, dim3(ceil_div(n, 256u)), dim3(256), 0, c->stream, (const uint64_t *)srt, (const uint32_t *)z_d.as<uint32_t>(), (const uint64_t *)zex_d.as<uint64_t>(), n,
                       (uint32_t)(1 + (sym_kind == CNIIC_SYM_RGB ? 11 : 6)), off_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    *nbits_h = nbits;
    *built = true;
    return CNIIC_OK;
}

// table_d: dense symbol table (any content; overwritten).  keys_d/len_d/code_d: the U distinct symbols
// and their codes (U < 2^26; codes longer than 26 bits escape to the per-rank tables).  src: pixels (rgb) or symbol keys (keys_d null: the
// symbols ARE ranks and table_d has U entries); packed_d: n u32 of scratch,
// may alias the symbol stream when that buffer is not needed afterwards.
int huff_pack_code32(Ctx *c, const uint32_t *syms_or_null_d, const uint8_t *rgb_or_null_d, uint64_t n, uint32_t *table_d,
                     const uint32_t *keys_d, const uint8_t *len_d, const uint64_t *code_d, uint64_t U, uint32_t *packed_d,
                     uint8_t *out_d, uint64_t bit_base, uint64_t *nbits_h) {
    *nbits_h = 0;
    if (n == 0) return CNIIC_OK;
    if (reinterpret_cast<uintptr_t>(out_d) & 3) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: output must be 4-byte aligned");
    const uint64_t nchunks64 = ceil_div(n, kPackChunk);
    if (nchunks64 > 0x7fffffffull) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: too many symbols");
    const uint32_t nchunks = (uint32_t)nchunks64;
    DevBuf cb, co, tot;
    CNIIC_HIP_TRY(c, cb.alloc((uint64_t)nchunks * 4));
    CNIIC_HIP_TRY(c, co.alloc((uint64_t)nchunks * 8));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    hipLaunchKernelGGL(k_fill_code32, dim3((uint32_t)std::min<uint64_t>(ceil_div(U, 256), 2048)), dim3(256), 0, c->stream, keys_d, len_d, code_d, U,
                       table_d);
    if (rgb_or_null_d)
        hipLaunchKernelGGL(k_pack_count32<SRC_RGB>, dim3(nchunks), dim3(kPackThreads), 0, c->stream, rgb_or_null_d, n, table_d, len_d, packed_d, cb.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_pack_count32<SRC_KEYS>, dim3(nchunks), dim3(kPackThreads), 0, c->stream, syms_or_null_d, n, table_d, len_d, packed_d, cb.as<uint32_t>());
    CNIIC_TRY(pack_scan(c, cb.as<uint32_t>(), nchunks, co.as<uint64_t>(), tot.as<uint64_t>()));
    hipLaunchKernelGGL(k_pack_write32, dim3(nchunks), dim3(kPackThreads), 0, c->stream, packed_d, n, len_d, code_d, co.as<uint64_t>(),
                       reinterpret_cast<uint32_t *>(out_d), bit_base, pack_img_cap());
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint64_t total = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&total, tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *nbits_h = total;
    return CNIIC_OK;
}

// ---- `delta` symbols (SignedColor keys, 27 bits): almost every symbol lies in the cube [-16, 15]^3 of small differences
// (the cube k_hilbert_delta counts in LDS).  Its 32768 (length, code) words fit the LDS of one persistent block per CU,
// so the per-symbol look-up -- one L2 request per lane and 1.2 ms at 16384^2 when it goes to memory, after 1.7 ms for
// turning symbols into ranks first -- is an LDS read; only the rare outlier reads the dense table.
constexpr uint32_t kHotCodes = 32 * 32 * 32;
constexpr int kHotThreads = 1024;
__device__ __forceinline__ uint32_t hot_index(uint32_t key) {  // < kHotCodes inside the cube, >= kHotCodes outside
    const uint32_t hr = (key >> 18) - (255 - 16), hg = ((key >> 9) & 511) - (255 - 16), hb = (key & 511) - (255 - 16);
    return (hr | hg | hb) < 32u ? (hr << 10) | (hg << 5) | hb : 0xffffffffu;
}

// dense[key] and, inside the cube, hot[index]: len << 26 | code (codes longer than 26 bits: escape to the per-rank tables)
__global__ void k_fill_code32_hot(const uint32_t *__restrict__ keys, const uint8_t *__restrict__ len, const uint64_t *__restrict__ code, uint64_t U,
                                  uint32_t *__restrict__ dense, uint32_t *__restrict__ hot) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride) {
        const uint32_t v = len[i] <= 26 ? ((uint32_t)len[i] << 26) | (uint32_t)code[i] : (kEscape << 26) | (uint32_t)i;
        const uint32_t k = keys[i];
        dense[k] = v;
        const uint32_t hx = hot_index(k);
        if (hx < kHotCodes) hot[hx] = v;
    }
}

__global__ __launch_bounds__(kHotThreads) void k_pack_count_hot(const uint32_t *__restrict__ syms, uint64_t n, const uint32_t *__restrict__ hot,
                                                                const uint32_t *__restrict__ dense, const uint8_t *__restrict__ len,
                                                                uint32_t *__restrict__ packed, uint32_t *__restrict__ chunk_bits, uint32_t nchunks) {
    extern __shared__ uint32_t s_hot[];  // [kHotCodes]
    __shared__ uint32_t s_part[kHotThreads / 64];
    for (uint32_t i = threadIdx.x; i < kHotCodes / 4; i += kHotThreads)
        reinterpret_cast<uint4 *>(s_hot)[i] = reinterpret_cast<const uint4 *>(hot)[i];
    __syncthreads();
    constexpr int PER = kPackChunk / kHotThreads;  // 4 symbols per thread
    static_assert(PER == 4, "one 16-byte load per thread");
    for (uint32_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const uint64_t first = (uint64_t)ch * kPackChunk + (uint64_t)threadIdx.x * PER;
        uint32_t key[PER];
        const bool full = first + PER <= n;
        if (full) { const uint4 q = *reinterpret_cast<const uint4 *>(syms + first); key[0] = q.x; key[1] = q.y; key[2] = q.z; key[3] = q.w; }
        else
            for (int i = 0; i < PER; i++) key[i] = first + i < n ? syms[first + i] : 0xffffffffu;
        uint32_t v[PER], bits = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            v[i] = 0u;
            if (full || first + i < n) {
                const uint32_t hx = hot_index(key[i]);
                v[i] = hx < kHotCodes ? s_hot[hx] : dense[key[i]];
            }
            const uint32_t L = v[i] >> 26;
            bits += L == kEscape ? (uint32_t)len[v[i] & 0x3ffffffu] : L;
        }
        if (full) *reinterpret_cast<uint4 *>(packed + first) = make_uint4(v[0], v[1], v[2], v[3]);
        else
            for (int i = 0; i < PER; i++)
                if (first + i < n) packed[first + i] = v[i];
        bits = wave_reduce_sum(bits);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = bits;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int i = 0; i < kHotThreads / 64; i++) t += s_part[i];
            chunk_bits[ch] = t;
        }
        __syncthreads();
    }
}

// huff_pack_code32 for SignedColor symbols: dense_d = the 2^27-entry table (overwritten at the U keys), packed_d may alias syms_d
int huff_pack_code32_hot(Ctx *c, const uint32_t *syms_d, uint64_t n, uint32_t *dense_d, const uint32_t *keys_d, const uint8_t *len_d,
                         const uint64_t *code_d, uint64_t U, uint32_t *packed_d, uint8_t *out_d, uint64_t bit_base, uint64_t *nbits_h) {
    *nbits_h = 0;
    if (n == 0) return CNIIC_OK;
    if ((reinterpret_cast<uintptr_t>(out_d) & 3) || (reinterpret_cast<uintptr_t>(syms_d) & 15) || (reinterpret_cast<uintptr_t>(packed_d) & 15))
        return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: misaligned buffers");
    const uint64_t nchunks64 = ceil_div(n, kPackChunk);
    if (nchunks64 > 0x7fffffffull) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: too many symbols");
    const uint32_t nchunks = (uint32_t)nchunks64;
    DevBuf cb, co, tot, hot;
    CNIIC_HIP_TRY(c, cb.alloc((uint64_t)nchunks * 4));
    CNIIC_HIP_TRY(c, co.alloc((uint64_t)nchunks * 8));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    CNIIC_HIP_TRY(c, hot.alloc((uint64_t)kHotCodes * 4));
    CNIIC_HIP_TRY(c, hipMemsetAsync(hot.p, 0, (uint64_t)kHotCodes * 4, c->stream));
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pack_count_hot), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kHotCodes * 4));
    });
    hipLaunchKernelGGL(k_fill_code32_hot, dim3((uint32_t)std::min<uint64_t>(ceil_div(U, 256), 2048)), dim3(256), 0, c->stream, keys_d, len_d, code_d, U,
                       dense_d, hot.as<uint32_t>());
    hipLaunchKernelGGL(k_pack_count_hot, dim3(std::min<uint32_t>(nchunks, 256)), dim3(kHotThreads), kHotCodes * 4, c->stream, syms_d, n,
                       (const uint32_t *)hot.as<uint32_t>(), (const uint32_t *)dense_d, len_d, packed_d, cb.as<uint32_t>(), nchunks);
    CNIIC_TRY(pack_scan(c, cb.as<uint32_t>(), nchunks, co.as<uint64_t>(), tot.as<uint64_t>()));
    hipLaunchKernelGGL(k_pack_write32, dim3(nchunks), dim3(kPackThreads), 0, c->stream, packed_d, n, len_d, code_d, co.as<uint64_t>(),
                       reinterpret_cast<uint32_t *>(out_d), bit_base, pack_img_cap());
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint64_t total = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&total, tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *nbits_h = total;
    return CNIIC_OK;
}

// Packs at bit_base of out_d: a 4-byte aligned, PRE-ZEROED buffer that is large enough (the caller
// knows the payload size from the histogram); bytes before bit_base may already hold the header.
int huff_pack_keys(Ctx *c, const uint32_t *keys_or_null_d, const uint8_t *rgb_or_null_d, uint64_t n,
                   const uint32_t *rank_table_d, const uint8_t *len_d, const uint64_t *code_d, uint8_t *out_d,
                   uint64_t bit_base, uint64_t *nbits_h) {
    if (rgb_or_null_d) return pack_impl<SRC_RGB>(c, rgb_or_null_d, n, rank_table_d, len_d, code_d, out_d, bit_base, nbits_h);
    return pack_impl<SRC_KEYS>(c, keys_or_null_d, n, rank_table_d, len_d, code_d, out_d, bit_base, nbits_h);
}

// ranks_d[i] = rank (+ 1 when one_based) of symbol i (may alias keys_or_null_d: every thread reads its symbols before it writes)
int huff_rank_stream(Ctx *c, const uint32_t *keys_or_null_d, const uint8_t *rgb_or_null_d, uint64_t n, const uint32_t *rank_table_d,
                     uint32_t *ranks_d, bool one_based) {
    if (n == 0) return CNIIC_OK;
    const uint64_t nchunks64 = ceil_div(n, kPackChunk);
    if (nchunks64 > 0x7fffffffull) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: too many symbols");
    if (rgb_or_null_d)
        hipLaunchKernelGGL(k_rank_stream<SRC_RGB>, dim3((uint32_t)nchunks64), dim3(kPackThreads), 0, c->stream, rgb_or_null_d, n, rank_table_d, ranks_d,
                           one_based ? 1u : 0u);
    else
        hipLaunchKernelGGL(k_rank_stream<SRC_KEYS>, dim3((uint32_t)nchunks64), dim3(kPackThreads), 0, c->stream, keys_or_null_d, n, rank_table_d, ranks_d,
                           one_based ? 1u : 0u);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int huff_pack_ranks(Ctx *c, const uint32_t *ranks_d, uint64_t n, const uint8_t *len_d, const uint64_t *code_d, uint8_t *out_d,
                    uint64_t bit_base, uint64_t *nbits_h) {
    return pack_impl<SRC_RANKS>(c, ranks_d, n, nullptr, len_d, code_d, out_d, bit_base, nbits_h);
}

}  // namespace cniic
