// k_points.hip -- the front and back end of ClusterColors::encode around the K-means, for large images
// (reference: count_freqs of the pixels, src/utils.rs:4-16 called at src/codec/clusterc.rs:21-24, and the
// colour -> centroid-colour remap of every pixel, clusterc.rs:31-47).
//
// What the K-means (k_kmeans_rgbw.hip) wants is the list of DISTINCT colours with their pixel counts in
// cell-major order (cell_of: 32^3 colour cells, the 4^3 cells of a 32^3-colour "super-cell" consecutive), and
// what Huffman wants afterwards is the cluster label of every pixel.  Both are a histogram / a lookup over
// 2^24 colours, far beyond LDS, and scattered 2- or 4-byte accesses to HBM-resident tables are bound by the
// number of L2 requests (one per lane), not by bytes.  So the pixels are partitioned ONCE by super-cell
// (512 buckets), staged through LDS so that every global access is a run of consecutive addresses:
//
//   k_sp_count     per 64 Ki-pixel chunk: pixels per bucket (LDS histogram)                     read 3 B/px
//   k_sp_colscan / k_sp_bstart   where each (chunk, bucket) run starts
//   k_sp_scatter   per chunk: 15-bit colour-in-bucket of every pixel, sorted by bucket in LDS and written as
//                  512 runs; the pixel's place in the chunk's sorted order goes to prank[pixel]  read 3, write 2 + 2 B/px
//   k_sp_hist      per bucket: LDS histogram of its 2^15 colours -> occupied colours per cell, occupancy bitmap,
//                  and the bucket's distinct colours with their counts, staged in cell-major order
//   (bitmap -> popcount prefix = GIdx: rank of a colour in the ascending list of all colours = the
//    reference's point order, used for init_assignment / init_centroids / the reseed index)
//   k_sp_emit      the staged colours, counts and their initial labels to their places in the K-means arrays
//   ... K-means ...
//   k_sp_partlab   per bucket: label of every partitioned pixel from an LDS table of the bucket's colours
//   k_sp_pixlab    per chunk: its 512 label runs into LDS, each pixel picks stage[prank]  -> label stream   read 1 + 2, write 1 B/px
//
// No atomics on global memory, no table of 2^24 entries, and every pixel's label is found without a random
// read.  Results are identical to the dense-table path (k_hist.hip + k_cells_write_tbl + k_pixel_labels): the
// point set, weights, initial labels and cell order are the same.
#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

constexpr uint32_t kSpBuckets = 512;           // super-cells: r[7:5] g[7:5] b[7:5]
constexpr uint32_t kSpBins = 1u << 15;         // colours of a super-cell
constexpr uint32_t kSpChunk = 1u << 16;        // pixels per chunk: a position inside a run fits 16 bits
constexpr int kSpThreads = 1024;
constexpr uint32_t kSpSliceMin = 1u << 17;     // k_sp_partlab: entries per slice of a bucket, at least
constexpr uint32_t kSpSlices = 8;              // ... and slices per bucket, at most
constexpr uint32_t kSpWaves = kSpThreads / 64;

__device__ __forceinline__ uint32_t sp_bucket(uint32_t key) {
    return (((key >> 21) & 7u) << 6) | (((key >> 13) & 7u) << 3) | ((key >> 5) & 7u);
}
// colour inside its super-cell, cell-major: cell within the super-cell (cell_of's low 6 bits) << 9 | colour within the cell
__device__ __forceinline__ uint32_t sp_bin(uint32_t key) {
    const uint32_t r = (key >> 16) & 31u, g = (key >> 8) & 31u, b = key & 31u;
    return ((r >> 3) << 13) | ((g >> 3) << 11) | ((b >> 3) << 9) | ((r & 7u) << 6) | ((g & 7u) << 3) | (b & 7u);
}
__device__ __forceinline__ uint32_t sp_key(uint32_t bucket, uint32_t bin) {
    const uint32_t r = (((bucket >> 6) & 7u) << 5) | (((bin >> 13) & 3u) << 3) | ((bin >> 6) & 7u);
    const uint32_t g = (((bucket >> 3) & 7u) << 5) | (((bin >> 11) & 3u) << 3) | ((bin >> 3) & 7u);
    const uint32_t b = ((bucket & 7u) << 5) | (((bin >> 9) & 3u) << 3) | (bin & 7u);
    return (r << 16) | (g << 8) | b;
}

// exclusive scan of 512 values held one per thread by threads 0..511 of a 1024-thread block; all threads call
__device__ __forceinline__ uint32_t scan512(uint32_t v, uint32_t *wsum) { return block_exclusive_scan<kSpThreads>(v, wsum); }

// pixels of chunk c: [c * kSpChunk, min(npx, (c + 1) * kSpChunk)); thread t takes the 16-pixel groups t, t + 1024, ...
template <typename F> __device__ __forceinline__ void sp_for_pixels(const uint8_t *__restrict__ rgb, uint64_t npx, F &&f) {
    const uint64_t p0 = (uint64_t)blockIdx.x * kSpChunk, p1 = min(npx, p0 + kSpChunk);
    const uint4 *v = reinterpret_cast<const uint4 *>(rgb);
    for (uint64_t g = p0 / 16 + threadIdx.x; g * 16 < p1; g += kSpThreads) {
        uint32_t key[16];
        if (g * 16 + 16 <= p1) {
            load16px_keys(v + 3 * g, key);
            f(g * 16, key, 16u);
        } else {  // the image's last, partial group
            const uint32_t m = (uint32_t)(p1 - g * 16);
            for (uint32_t i = 0; i < 16; i++) key[i] = i < m ? rgb_key(rgb + 3 * (g * 16 + i)) : 0u;
            f(g * 16, key, m);
        }
    }
}

__global__ __launch_bounds__(kSpThreads) void k_sp_count(const uint8_t *__restrict__ rgb, uint64_t npx, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t h[kSpBuckets];
    if (threadIdx.x < kSpBuckets) h[threadIdx.x] = 0;
    __syncthreads();
    sp_for_pixels(rgb, npx, [&](uint64_t, const uint32_t (&key)[16], uint32_t m) {
#pragma unroll
        for (uint32_t i = 0; i < 16; i++)
            if (i < m) atomicAdd(&h[sp_bucket(key[i])], 1u);
    });
    __syncthreads();
    if (threadIdx.x < kSpBuckets) cnt[(size_t)blockIdx.x * kSpBuckets + threadIdx.x] = h[threadIdx.x];
}

// one block per bucket: pre[c][b] = pixels of bucket b in the chunks before c; total[b]
__global__ __launch_bounds__(256) void k_sp_colscan(const uint32_t *__restrict__ cnt, uint32_t nchunks, uint32_t *__restrict__ pre,
                                                    uint32_t *__restrict__ total) {
    __shared__ uint32_t wsum[256 / 64];
    const uint32_t b = blockIdx.x, per = (nchunks + 255) / 256;
    const uint32_t c0 = threadIdx.x * per, c1 = min(c0 + per, nchunks);
    uint32_t s = 0;
    for (uint32_t c = c0; c < c1; c++) s += cnt[(size_t)c * kSpBuckets + b];
    uint32_t run = block_exclusive_scan<256>(s, wsum);
    for (uint32_t c = c0; c < c1; c++) {
        pre[(size_t)c * kSpBuckets + b] = run;
        run += cnt[(size_t)c * kSpBuckets + b];
    }
    if (threadIdx.x == 255) total[b] = run;  // (the last thread's range ends the column, or is empty and holds the sum)
}

// bstart: first entry of every bucket in the partition; sstart: first slot of its staged distinct colours (a bucket
// of n pixels has at most min(n, 2^15) of them)
__global__ __launch_bounds__(kSpBuckets) void k_sp_bstart(const uint32_t *__restrict__ total, uint32_t *__restrict__ bstart,
                                                          uint32_t *__restrict__ sstart) {
    __shared__ uint32_t wsum[kSpBuckets / 64];
    const uint32_t t = total[threadIdx.x];
    const uint32_t ex = block_exclusive_scan<kSpBuckets>(t, wsum);
    bstart[threadIdx.x] = ex;
    if (threadIdx.x == kSpBuckets - 1) bstart[kSpBuckets] = ex + t;
    sstart[threadIdx.x] = block_exclusive_scan<kSpBuckets>(min(t, kSpBins), wsum);
}

// The chunk's pixels sorted by bucket occupy positions [0, n) of the LDS stage; off[b] .. off[b + 1] is bucket b's run
// (off[kSpBuckets] = n).  Wave w walks the positions [w per, (w + 1) per) 64 at a time, lane <-> position, each lane
// keeping the bucket of its position (runs are short: the bucket advances about every other step).  STEPS positions
// per lane are handed over together so that their loads can be in flight at once.  A walk by runs instead costs a
// dependent round trip to memory per run, and one by rounds over all runs is quadratic when a run is long.
template <int STEPS, typename F> __device__ __forceinline__ void sp_walk_sorted(const uint32_t *off, uint32_t n, F &&f) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t per = (n + kSpThreads - 1) / kSpThreads * 64;
    const uint32_t j0 = wv * per, j1 = min(n, j0 + per);
    if (j0 >= j1) return;
    const uint32_t jj = min(j0 + lane, j1 - 1);
    uint32_t lo = 0, hi = kSpBuckets;  // off[lo] <= jj < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= jj) lo = mid; else hi = mid;
    }
    uint32_t b = lo, end = off[b + 1];
    for (uint32_t j = j0 + lane; j < j1; j += 64 * STEPS) {
        uint32_t jb[STEPS];
#pragma unroll
        for (int t = 0; t < STEPS; t++) {
            const uint32_t q = j + 64 * t;
            jb[t] = 0xffffffffu;
            if (q < j1) {
                while (q >= end) { b++; end = off[b + 1]; }
                jb[t] = b;
            }
        }
        f(j, jb);
    }
}

// LDS (dynamic): stage u16[kSpChunk] | off u32[kSpBuckets + 1] | cur u32[kSpBuckets]
__global__ __launch_bounds__(kSpThreads) void k_sp_scatter(const uint8_t *__restrict__ rgb, uint64_t npx, const uint32_t *__restrict__ cnt,
                                                           const uint32_t *__restrict__ pre, const uint32_t *__restrict__ bstart,
                                                           uint16_t *__restrict__ part, uint16_t *__restrict__ prank) {
    extern __shared__ __align__(16) uint8_t sp_lds[];
    uint16_t *stage = reinterpret_cast<uint16_t *>(sp_lds);
    uint32_t *off = reinterpret_cast<uint32_t *>(sp_lds + (size_t)kSpChunk * 2);  // [kSpBuckets + 1]
    uint32_t *cur = off + kSpBuckets + 1;
    __shared__ uint32_t wsum[kSpWaves];
    const uint32_t *mycnt = cnt + (size_t)blockIdx.x * kSpBuckets;
    const uint32_t n_b = threadIdx.x < kSpBuckets ? mycnt[threadIdx.x] : 0u;
    const uint32_t ex = scan512(n_b, wsum);
    if (threadIdx.x < kSpBuckets) { off[threadIdx.x] = ex; cur[threadIdx.x] = 0; }
    if (threadIdx.x == kSpBuckets - 1) off[kSpBuckets] = ex + n_b;
    __syncthreads();
    sp_for_pixels(rgb, npx, [&](uint64_t first, const uint32_t (&key)[16], uint32_t m) {
        uint32_t rk[16];
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            rk[i] = 0;
            if (i < m) {
                const uint32_t b = sp_bucket(key[i]);
                // its place in the chunk's bucket-sorted order (any order inside a run), remembered per pixel: the way back
                // (k_sp_pixlab) is then one LDS read per pixel -- no colour, no bucket, no offset table
                rk[i] = off[b] + atomicAdd(&cur[b], 1u);
                stage[rk[i]] = (uint16_t)sp_bin(key[i]);
            }
        }
        if (m == 16) {
            uint4 *d = reinterpret_cast<uint4 *>(prank + first);
            d[0] = make_uint4(rk[0] | (rk[1] << 16), rk[2] | (rk[3] << 16), rk[4] | (rk[5] << 16), rk[6] | (rk[7] << 16));
            d[1] = make_uint4(rk[8] | (rk[9] << 16), rk[10] | (rk[11] << 16), rk[12] | (rk[13] << 16), rk[14] | (rk[15] << 16));
        } else {
            for (uint32_t i = 0; i < m; i++) prank[first + i] = (uint16_t)rk[i];
        }
    });
    __syncthreads();
    // the chunk's 512 runs go out, each to consecutive addresses.  Wave w owns buckets w, w + 16, ...: their sizes and
    // places are fetched by 32 lanes at once (fetched per bucket they are 32 dependent round trips to memory)
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t *mypre = pre + (size_t)blockIdx.x * kSpBuckets;
    const uint32_t mb = wv + kSpWaves * (lane & 31);
    const uint32_t m_n = mycnt[mb], m_dst = bstart[mb] + mypre[mb], m_off = off[mb];
    for (uint32_t k = 0; k < kSpBuckets / kSpWaves; k++) {
        const uint32_t n = __shfl(m_n, k, 64), d0 = __shfl(m_dst, k, 64), o = __shfl(m_off, k, 64);
        for (uint32_t i = lane; i < n; i += 64) part[(size_t)d0 + i] = stage[o + i];
    }
}

// entries [s, e) of a u16 stream, four per load where the address allows (8-byte aligned), else one by one; four
// loads are in flight per thread (a crowded bucket is one block's work: with one load at a time its 220 K entries at
// C2 were 54 dependent round trips to memory and set the kernel's time)
template <typename F> __device__ __forceinline__ void sp_for_entries(const uint16_t *__restrict__ part, uint64_t s, uint64_t e, F &&f) {
    const uint64_t a = min(e, (s + 3) & ~3ull), z = a + ((e - a) & ~3ull);
    for (uint64_t i = s + threadIdx.x; i < a; i += kSpThreads) { const uint32_t v = part[i]; f(i, v, v, v, v, 1u); }
    constexpr uint64_t kStep = 4 * (uint64_t)kSpThreads;
    for (uint64_t i = a + 4 * (uint64_t)threadIdx.x; i < z; i += 4 * kStep) {
        uint2 q[4];
#pragma unroll
        for (int t = 0; t < 4; t++) q[t] = i + t * kStep < z ? *reinterpret_cast<const uint2 *>(part + i + t * kStep) : make_uint2(0u, 0u);
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (i + t * kStep < z) f(i + t * kStep, q[t].x & 0xffffu, q[t].x >> 16, q[t].y & 0xffffu, q[t].y >> 16, 4u);
    }
    for (uint64_t i = z + threadIdx.x; i < e; i += kSpThreads) { const uint32_t v = part[i]; f(i, v, v, v, v, 1u); }
}

// LDS histogram of one bucket's colours.  A flat image puts every pixel of a wave into one bin: equal
// neighbours are added together, and a wave whose 256 entries are all the same colour adds once.
__device__ __forceinline__ void sp_bucket_hist(const uint16_t *__restrict__ part, uint64_t s, uint64_t e, uint32_t *hist) {
    for (uint32_t i = threadIdx.x; i < kSpBins; i += kSpThreads) hist[i] = 0;
    __syncthreads();
    sp_for_entries(part, s, e, [&](uint64_t, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t m) {
        if (m == 1) { atomicAdd(&hist[a], 1u); return; }
        const bool same = a == b && b == c && c == d;
        const uint32_t first = __builtin_amdgcn_readfirstlane(a);
        if (__all(same && a == first) && __popcll(__ballot(1)) == 64) {
            if ((threadIdx.x & 63) == 0) atomicAdd(&hist[a], 256u);
        } else if (same) {
            atomicAdd(&hist[a], 4u);
        } else {
            atomicAdd(&hist[a], 1u); atomicAdd(&hist[b], 1u); atomicAdd(&hist[c], 1u); atomicAdd(&hist[d], 1u);
        }
    });
    __syncthreads();
}


// One block per bucket: LDS histogram of its colours -> occupied colours per cell (cell_count[bucket * 64 + cell]), the
// bucket's words of the occupancy bitmap, and the bucket's distinct colours as (bin, count) pairs in bin order = cell-major
// order, staged at sstart[bucket] (where they go in the K-means arrays is only known when every bucket has counted).
__global__ __launch_bounds__(kSpThreads) void k_sp_hist(const uint16_t *__restrict__ part, const uint32_t *__restrict__ bstart,
                                                        const uint32_t *__restrict__ sstart, uint32_t *__restrict__ cell_count,
                                                        uint32_t *__restrict__ bits32, uint16_t *__restrict__ sbin,
                                                        uint32_t *__restrict__ scnt) {
    extern __shared__ __align__(16) uint8_t sp_lds[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(sp_lds);
    __shared__ uint32_t wsum[kSpWaves];
    const uint32_t bucket = blockIdx.x;
    const uint64_t s = bstart[bucket], e = bstart[bucket + 1];
    const uint32_t r5 = threadIdx.x >> 5, g5 = threadIdx.x & 31;  // thread (r5, g5) owns one colour row of the bitmap
    const uint32_t word_at = ((((bucket >> 6) & 7u) << 5 | r5) << 11) | ((((bucket >> 3) & 7u) << 5 | g5) << 3) | (bucket & 7u);
    if (s == e) {  // no pixel here: no colours (the K-means never looks at an empty cell), but the bitmap words are ours
        if (threadIdx.x < 64) cell_count[bucket * 64 + threadIdx.x] = 0;
        bits32[word_at] = 0;
        return;
    }
    sp_bucket_hist(part, s, e, hist);
    uint32_t word = 0;
#pragma unroll
    for (uint32_t bq = 0; bq < 4; bq++) {
        const uint32_t base = ((r5 >> 3) << 13) | ((g5 >> 3) << 11) | (bq << 9) | ((r5 & 7u) << 6) | ((g5 & 7u) << 3);
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) word |= (hist[base + j] != 0 ? 1u : 0u) << (bq * 8 + j);
    }
    bits32[word_at] = word;
    // wave w owns bins [2048 w, 2048 w + 2048) = 4 cells, 64 at a time with lane <-> bin: the occupied lanes of one
    // step write consecutive positions (one or two lines per store)
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t mine = 0;
    for (uint32_t c4 = 0; c4 < 4; c4++) {
        uint32_t in_cell = 0;
        for (uint32_t st = 0; st < 8; st++) in_cell += (uint32_t)__popcll(__ballot(hist[wv * 2048 + c4 * 512 + st * 64 + lane] != 0));
        if (lane == 0) cell_count[bucket * 64 + wv * 4 + c4] = in_cell;
        mine += in_cell;
    }
    if (lane == 0) wsum[wv] = mine;
    __syncthreads();
    uint32_t pos0 = sstart[bucket];
    for (uint32_t i = 0; i < wv; i++) pos0 += wsum[i];
    for (uint32_t st = 0; st < 32; st++) {
        const uint32_t bin = wv * 2048 + st * 64 + lane, v = hist[bin];
        const unsigned long long bm = __ballot(v != 0);
        if (v) {
            const uint32_t pos = pos0 + (uint32_t)__popcll(bm & lt_mask);
            sbin[pos] = (uint16_t)bin;
            scnt[pos] = v;
        }
        pos0 += (uint32_t)__popcll(bm);
    }
}

struct SpEmit {             // where the distinct colours go: the K-means state's cell-major arrays
    const uint32_t *cell_start;
    uint32_t *ckeys, *cweight;
    void *labels;
    uint32_t K, wide;
    GIdx gx;
    const uint64_t *U_dev;      // when set: the length of the point list is here, not yet in gx.U
};

// the staged colours of every bucket -> keys, counts and initial labels at their cell-major positions: a copy, the bucket's
// place being cell_start of its first cell.  kSpEmitSplit blocks per bucket.
constexpr uint32_t kSpEmitSplit = 4;
__global__ __launch_bounds__(256) void k_sp_emit(const uint32_t *__restrict__ sstart, const uint16_t *__restrict__ sbin,
                                                 const uint32_t *__restrict__ scnt, SpEmit em) {
    const uint32_t bucket = blockIdx.x / kSpEmitSplit, part_no = blockIdx.x % kSpEmitSplit;
    const uint32_t q0 = em.cell_start[bucket * 64], n = em.cell_start[bucket * 64 + 64] - q0, s0 = sstart[bucket];
    const uint32_t U = em.U_dev ? (uint32_t)*em.U_dev : (uint32_t)em.gx.U, ppc = max(U / em.K, 1u);  // (fewer colours than clusters: the host refuses later)
    const float rcp = 1.0f / (float)ppc;
    // four entries per thread and step, each a chain of two loads (the staged colour, then its two index words): the
    // chains travel together
    constexpr uint32_t kStride = 256 * kSpEmitSplit;
    for (uint32_t i0 = part_no * 256 + threadIdx.x; i0 < n; i0 += 4 * kStride) {
        uint32_t key[4], cnt[4], rank[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t i = i0 + t * kStride;
            key[t] = i < n ? sp_key(bucket, sbin[s0 + i]) : 0u;
            cnt[t] = i < n ? scnt[s0 + i] : 0u;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) rank[t] = i0 + t * kStride < n ? gidx_rank(em.gx, key[t]) : 0u;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t i = i0 + t * kStride;
            if (i >= n) continue;
            em.ckeys[q0 + i] = key[t];
            em.cweight[q0 + i] = cnt[t];
            const uint32_t lab = init_label24(rank[t], U, em.K, ppc, rcp);  // init_assignment kmeans.rs:61-78
            if (em.wide) static_cast<uint16_t *>(em.labels)[q0 + i] = (uint16_t)lab;
            else static_cast<uint8_t *>(em.labels)[q0 + i] = (uint8_t)lab;
        }
    }
}

// bits (u64[2^18]) -> wprefix inside blocks of 1024 words + blocktot; k_gidx_finish (k_hist.hip) completes it
__global__ __launch_bounds__(256) void k_bits_prefix(const unsigned long long *__restrict__ bits, uint32_t *__restrict__ wprefix,
                                                     uint32_t *__restrict__ blocktot) {
    __shared__ uint32_t wsum[256 / 64];
    const uint32_t w0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    uint32_t cnt[4], mine = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { cnt[q] = (uint32_t)__popcll(bits[w0 + q]); mine += cnt[q]; }
    uint32_t run = block_exclusive_scan<256>(mine, wsum);
#pragma unroll
    for (int q = 0; q < 4; q++) { wprefix[w0 + q] = run; run += cnt[q]; }
    if (threadIdx.x == 255) blocktot[blockIdx.x] = run;
}

// ---- back end.  One block per bucket: LDS table bin -> final label from the bucket's points, then the label of
// every partitioned pixel, in partition order.
template <typename LabelT>
__global__ __launch_bounds__(kSpThreads) void k_sp_partlab(const uint16_t *__restrict__ part, const uint32_t *__restrict__ bstart,
                                                           const uint32_t *__restrict__ cell_start, const uint32_t *__restrict__ ckeys,
                                                           const LabelT *__restrict__ labels, LabelT *__restrict__ partlab) {
    extern __shared__ __align__(16) uint8_t sp_lds[];
    LabelT *lut = reinterpret_cast<LabelT *>(sp_lds);
    const uint32_t bucket = blockIdx.x;
    uint64_t s = bstart[bucket], e = bstart[bucket + 1];
    if (s == e) return;
    // a crowded bucket is several blocks' work (grid.y slices of at least kSpSliceMin entries; every slice builds the table: up to
    // 2^15 colours against >= 2^17 pixels)
    {
        const uint64_t len = e - s;
        const uint32_t slices = (uint32_t)min<uint64_t>(gridDim.y, max<uint64_t>(1, len / kSpSliceMin));
        if (blockIdx.y >= slices) return;
        const uint64_t per = ((len + slices - 1) / slices + 3) & ~3ull;
        const uint64_t a = s + per * blockIdx.y;
        e = min(e, a + per);
        s = a;
        if (s >= e) return;
    }
    const uint32_t q0 = cell_start[bucket * 64], q1 = cell_start[bucket * 64 + 64];
    for (uint32_t i = q0 + threadIdx.x; i < q1; i += kSpThreads) lut[sp_bin(ckeys[i])] = labels[i];
    __syncthreads();
    sp_for_entries(part, s, e, [&](uint64_t i, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t m) {
        if (m == 1) { partlab[i] = lut[a]; return; }
        if (sizeof(LabelT) == 1) {  // i is a multiple of 4
            *reinterpret_cast<uint32_t *>(partlab + i) = (uint32_t)lut[a] | ((uint32_t)lut[b] << 8) | ((uint32_t)lut[c] << 16) | ((uint32_t)lut[d] << 24);
        } else {
            *reinterpret_cast<uint2 *>(partlab + i) = make_uint2((uint32_t)lut[a] | ((uint32_t)lut[b] << 16), (uint32_t)lut[c] | ((uint32_t)lut[d] << 16));
        }
    });
}

// One block per chunk: its 512 label runs into LDS (the chunk's pixels in bucket-sorted order), then every pixel picks stage[prank].
// LDS (dynamic): stage LabelT[kSpChunk] | off u32[kSpBuckets + 1] | gsrc u32[kSpBuckets]
template <typename LabelT>
__global__ __launch_bounds__(kSpThreads) void k_sp_pixlab(const uint8_t *__restrict__ rgb, uint64_t npx, const uint32_t *__restrict__ cnt,
                                                          const uint32_t *__restrict__ pre, const uint32_t *__restrict__ bstart,
                                                          const LabelT *__restrict__ partlab, const uint16_t *__restrict__ prank,
                                                          LabelT *__restrict__ pixlab) {
    extern __shared__ __align__(16) uint8_t sp_lds[];
    LabelT *stage = reinterpret_cast<LabelT *>(sp_lds);
    uint32_t *off = reinterpret_cast<uint32_t *>(sp_lds + (size_t)kSpChunk * sizeof(LabelT));  // [kSpBuckets + 1]
    uint32_t *gsrc = off + kSpBuckets + 1;
    __shared__ uint32_t wsum[kSpWaves];
    const uint32_t *mycnt = cnt + (size_t)blockIdx.x * kSpBuckets, *mypre = pre + (size_t)blockIdx.x * kSpBuckets;
    const uint32_t n_b = threadIdx.x < kSpBuckets ? mycnt[threadIdx.x] : 0u;
    const uint32_t ex = scan512(n_b, wsum);
    if (threadIdx.x < kSpBuckets) off[threadIdx.x] = ex;
    if (threadIdx.x == kSpBuckets - 1) off[kSpBuckets] = ex + n_b;
    __syncthreads();
    // the chunk's 512 label runs come in: gsrc[b] = where bucket b's run of this chunk starts, minus its place in the stage
    if (threadIdx.x < kSpBuckets) gsrc[threadIdx.x] = bstart[threadIdx.x] + mypre[threadIdx.x] - off[threadIdx.x];
    __syncthreads();
    sp_walk_sorted<8>(off, off[kSpBuckets], [&](uint32_t j, const uint32_t (&jb)[8]) {
        LabelT got[8];
#pragma unroll
        for (int t = 0; t < 8; t++) got[t] = jb[t] != 0xffffffffu ? partlab[(size_t)(gsrc[jb[t]] + (j + 64 * t))] : (LabelT)0;
#pragma unroll
        for (int t = 0; t < 8; t++)
            if (jb[t] != 0xffffffffu) stage[j + 64 * t] = got[t];
    });
    __syncthreads();
    // every pixel picks stage[its place in the sorted order] (prank, written by k_sp_scatter): 2 + 1 bytes per pixel, the
    // image itself is not read again
    const uint64_t p0 = (uint64_t)blockIdx.x * kSpChunk, p1 = min(npx, p0 + kSpChunk);
    for (uint64_t g = p0 / 16 + threadIdx.x; g * 16 < p1; g += kSpThreads) {
        const uint64_t first = g * 16;
        const uint32_t m = (uint32_t)min<uint64_t>(16, p1 - first);
        uint32_t lab[16];
        if (m == 16) {
            const uint4 *rp = reinterpret_cast<const uint4 *>(prank + first);
            const uint4 r0 = rp[0], r1 = rp[1];
            const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) lab[i] = stage[(rw[i >> 1] >> (16 * (i & 1))) & 0xffffu];
            if (sizeof(LabelT) == 1) {
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; j++) w[j] = lab[4 * j] | (lab[4 * j + 1] << 8) | (lab[4 * j + 2] << 16) | (lab[4 * j + 3] << 24);
                *reinterpret_cast<uint4 *>(pixlab + first) = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                uint4 *d = reinterpret_cast<uint4 *>(pixlab + first);
                d[0] = make_uint4(lab[0] | (lab[1] << 16), lab[2] | (lab[3] << 16), lab[4] | (lab[5] << 16), lab[6] | (lab[7] << 16));
                d[1] = make_uint4(lab[8] | (lab[9] << 16), lab[10] | (lab[11] << 16), lab[12] | (lab[13] << 16), lab[14] | (lab[15] << 16));
            }
        } else {
            for (uint32_t i = 0; i < m; i++) pixlab[first + i] = stage[prank[first + i]];
        }
    }
}

// =========================================================================== host
static int sp_set_lds(Ctx *c) {
    static bool done = false;  // (function attributes are per process)
    if (done) return CNIIC_OK;
    const int big = 140 * 1024;
    CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_sp_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_sp_hist), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_sp_partlab<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_sp_partlab<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_sp_pixlab<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_sp_pixlab<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    done = true;
    return CNIIC_OK;
}

// pixels -> partition + occupied colours per cell + occupancy bitmap with its prefix.  Nothing waits: the number of
// distinct colours stays on the device (plan->total) for the set-up kernels that follow, and sp_wait_count fetches it
int sp_build(Ctx *c, const uint8_t *rgb_d, uint64_t npx, SpPlan *plan) {
    if (npx == 0 || (reinterpret_cast<uintptr_t>(rgb_d) & 15)) return c->fail(CNIIC_ERR_BAD_ARG, "sp_build: empty or unaligned image");
    CNIIC_TRY(sp_set_lds(c));
    plan->npx = npx;
    plan->nchunks = (uint32_t)ceil_div(npx, kSpChunk);
    const uint64_t nc = plan->nchunks;
    CNIIC_HIP_TRY(c, plan->cnt.alloc(nc * kSpBuckets * 4));
    CNIIC_HIP_TRY(c, plan->pre.alloc(nc * kSpBuckets * 4));
    CNIIC_HIP_TRY(c, plan->bstart.alloc(((uint64_t)kSpBuckets + 1) * 4));
    CNIIC_HIP_TRY(c, plan->part.alloc(npx * 2 + 16));
    CNIIC_HIP_TRY(c, plan->prank.alloc((npx + 16) * 2));
    CNIIC_HIP_TRY(c, plan->cell_count.alloc((uint64_t)kNumCells * 4));
    const uint64_t smax = std::min<uint64_t>(npx, 1ull << 24);  // distinct colours at most
    CNIIC_HIP_TRY(c, plan->sstart.alloc((uint64_t)kSpBuckets * 4));
    CNIIC_HIP_TRY(c, plan->sbin.alloc(smax * 2));
    CNIIC_HIP_TRY(c, plan->scnt.alloc(smax * 4));
    CNIIC_HIP_TRY(c, plan->bits.alloc((1ull << 18) * 8));
    CNIIC_HIP_TRY(c, plan->wprefix.alloc((1ull << 18) * 4));
    DevBuf total, blocktot;
    CNIIC_HIP_TRY(c, total.alloc((uint64_t)kSpBuckets * 4));
    CNIIC_HIP_TRY(c, blocktot.alloc(256 * 4));
    CNIIC_HIP_TRY(c, plan->total.alloc(8));
    CNIIC_HIP_TRY(c, ctx_pinned_u(c));
    if (!c->u_ev) CNIIC_HIP_TRY(c, hipEventCreateWithFlags(&c->u_ev, hipEventDisableTiming));
    hipLaunchKernelGGL(k_sp_count, dim3(plan->nchunks), dim3(kSpThreads), 0, c->stream, rgb_d, npx, plan->cnt.as<uint32_t>());
    hipLaunchKernelGGL(k_sp_colscan, dim3(kSpBuckets), dim3(256), 0, c->stream, plan->cnt.as<uint32_t>(), plan->nchunks,
                       plan->pre.as<uint32_t>(), total.as<uint32_t>());
    hipLaunchKernelGGL(k_sp_bstart, dim3(1), dim3(kSpBuckets), 0, c->stream, total.as<uint32_t>(), plan->bstart.as<uint32_t>(),
                       plan->sstart.as<uint32_t>());
    hipLaunchKernelGGL(k_sp_scatter, dim3(plan->nchunks), dim3(kSpThreads), (size_t)kSpChunk * 2 + kSpBuckets * 8 + 16, c->stream, rgb_d, npx,
                       plan->cnt.as<uint32_t>(), plan->pre.as<uint32_t>(), plan->bstart.as<uint32_t>(), plan->part.as<uint16_t>(),
                       plan->prank.as<uint16_t>());
    hipLaunchKernelGGL(k_sp_hist, dim3(kSpBuckets), dim3(kSpThreads), (size_t)kSpBins * 4, c->stream, plan->part.as<uint16_t>(),
                       plan->bstart.as<uint32_t>(), plan->sstart.as<uint32_t>(), plan->cell_count.as<uint32_t>(), plan->bits.as<uint32_t>(),
                       plan->sbin.as<uint16_t>(), plan->scnt.as<uint32_t>());
    hipLaunchKernelGGL(k_bits_prefix, dim3(256), dim3(256), 0, c->stream, plan->bits.as<unsigned long long>(), plan->wprefix.as<uint32_t>(),
                       blocktot.as<uint32_t>());
    CNIIC_TRY(gidx_finish(c, plan->wprefix.as<uint32_t>(), blocktot.as<uint32_t>(), plan->total.as<uint64_t>()));
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipMemcpyAsync(c->pinned_u, plan->total.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipEventRecord(c->u_ev, c->stream));
    return CNIIC_OK;
}

int sp_wait_count(Ctx *c, SpPlan *plan) {
    CNIIC_HIP_TRY(c, hipEventSynchronize(c->u_ev));
    plan->U = *c->pinned_u;
    return CNIIC_OK;
}

// the distinct colours, counts and initial labels into the K-means state's cell-major arrays (cell_start: its scan of
// plan->cell_count); gx: the index of the reference's point list (this image's own bitmap, or the union's)
int sp_emit(Ctx *c, const SpPlan *plan, const uint32_t *cell_start_d, uint32_t *ckeys_d, uint32_t *cweight_d, void *labels_d, bool wide,
            uint32_t K, const void *gbits_d, const uint32_t *gprefix_d, uint64_t Ug, const uint64_t *Ug_dev) {
    const GIdx gx{static_cast<const unsigned long long *>(gbits_d), gprefix_d, Ug};
    hipLaunchKernelGGL(k_sp_emit, dim3(kSpBuckets * kSpEmitSplit), dim3(256), 0, c->stream, plan->sstart.as<uint32_t>(), plan->sbin.as<uint16_t>(),
                       plan->scnt.as<uint32_t>(), SpEmit{cell_start_d, ckeys_d, cweight_d, labels_d, K, wide ? 1u : 0u, gx, Ug_dev});
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// occupancy of this image's colours as one nibble per colour (u32[2^21]), the form the ranks sum (k_hist.hip: k_occ_pack)
__global__ __launch_bounds__(256) void k_occ_from_bits(const uint32_t *__restrict__ bits32, uint32_t *__restrict__ occ) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;  // occ word i = colours 8 i .. 8 i + 7
    const uint32_t byte = (bits32[i >> 2] >> (8 * (i & 3))) & 255u;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) v |= ((byte >> j) & 1u) << (4 * j);
    occ[i] = v;
}
int sp_occupancy(Ctx *c, const SpPlan *plan, uint32_t *occ_d) {
    hipLaunchKernelGGL(k_occ_from_bits, dim3((1u << 21) / 256), dim3(256), 0, c->stream, plan->bits.as<uint32_t>(), occ_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// labels_d: the K-means' final cell-major labels -> pixlab_d: label of every pixel in image order
int sp_pixel_labels(Ctx *c, const SpPlan *plan, const uint8_t *rgb_d, const uint32_t *cell_start_d, const uint32_t *ckeys_d,
                    const void *labels_d, bool wide, void *pixlab_d) {
    DevBuf partlab;
    const uint64_t lb = wide ? 2 : 1;
    CNIIC_HIP_TRY(c, partlab.alloc(plan->npx * lb + 16));
    if (wide) {
        hipLaunchKernelGGL(k_sp_partlab<uint16_t>, dim3(kSpBuckets, plan->npx >= 2 * (uint64_t)kSpSliceMin ? kSpSlices : 1), dim3(kSpThreads), (size_t)kSpBins * 2, c->stream, plan->part.as<uint16_t>(),
                           plan->bstart.as<uint32_t>(), cell_start_d, ckeys_d, static_cast<const uint16_t *>(labels_d), partlab.as<uint16_t>());
        hipLaunchKernelGGL(k_sp_pixlab<uint16_t>, dim3(plan->nchunks), dim3(kSpThreads), (size_t)kSpChunk * 2 + kSpBuckets * 8 + 16, c->stream, rgb_d,
                           plan->npx, plan->cnt.as<uint32_t>(), plan->pre.as<uint32_t>(), plan->bstart.as<uint32_t>(), partlab.as<uint16_t>(),
                           plan->prank.as<uint16_t>(), static_cast<uint16_t *>(pixlab_d));
    } else {
        hipLaunchKernelGGL(k_sp_partlab<uint8_t>, dim3(kSpBuckets, plan->npx >= 2 * (uint64_t)kSpSliceMin ? kSpSlices : 1), dim3(kSpThreads), (size_t)kSpBins, c->stream, plan->part.as<uint16_t>(),
                           plan->bstart.as<uint32_t>(), cell_start_d, ckeys_d, static_cast<const uint8_t *>(labels_d), partlab.as<uint8_t>());
        hipLaunchKernelGGL(k_sp_pixlab<uint8_t>, dim3(plan->nchunks), dim3(kSpThreads), (size_t)kSpChunk + kSpBuckets * 8 + 16, c->stream, rgb_d,
                           plan->npx, plan->cnt.as<uint32_t>(), plan->pre.as<uint32_t>(), plan->bstart.as<uint32_t>(), partlab.as<uint8_t>(),
                           plan->prank.as<uint16_t>(), static_cast<uint8_t *>(pixlab_d));
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

}  // namespace cniic
