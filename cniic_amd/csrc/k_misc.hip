// k_misc.hip -- per-pixel passes around the K-means / Huffman kernels:
//   rgb_to_keys      Rgb<u8> -> packed symbol key (row-major pixel stream of hufc.rs:15)
//   remap_rgb        cluster-colors colour -> centroid-colour lookup (clusterc.rs:31-47)
//   voronoi_paint    VoronoiCluster::decode nearest-centroid fill (clusterc.rs:180-186)
//   mse_rgb          bench::compute_error (bench.rs:95-104)
//   synth_image      deterministic synthetic inputs; not part of the reference
// All are HBM-bound streaming passes: 16 pixels (48 B) per thread per step, 16-B accesses.
#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

static inline uint32_t grid_for(uint64_t items, uint32_t per_block = 256, uint32_t cap = 256 * 8) {
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(items, per_block), 1), cap);
}

// 16 packed keys (r<<16|g<<8|b) -> 48 interleaved bytes
__device__ __forceinline__ void store16px_keys(uint4 *p, const uint32_t key[16]) {
    uint32_t w[12];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        // little-endian 24-bit values: byte0 = r
        uint32_t v0 = __builtin_bswap32(key[4 * g + 0] << 8), v1 = __builtin_bswap32(key[4 * g + 1] << 8);
        uint32_t v2 = __builtin_bswap32(key[4 * g + 2] << 8), v3 = __builtin_bswap32(key[4 * g + 3] << 8);
        w[3 * g + 0] = v0 | (v1 << 24);
        w[3 * g + 1] = (v1 >> 8) | (v2 << 16);
        w[3 * g + 2] = (v2 >> 16) | (v3 << 8);
    }
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
    p[2] = make_uint4(w[8], w[9], w[10], w[11]);
}

__device__ __forceinline__ void store_px(uint8_t *p, uint32_t key) {
    p[0] = (uint8_t)(key >> 16); p[1] = (uint8_t)(key >> 8); p[2] = (uint8_t)key;
}

// ---------------------------------------------------------------- rgb -> keys
__global__ __launch_bounds__(256) void k_rgb_to_keys(const uint8_t *__restrict__ rgb, uint64_t npx,
                                                     uint32_t *__restrict__ keys, int aligned) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ngroups = aligned ? npx / 16 : 0;
    for (uint64_t g = tid; g < ngroups; g += stride) {
        uint32_t key[16];
        load16px_keys(reinterpret_cast<const uint4 *>(rgb) + 3 * g, key);
        uint4 *o = reinterpret_cast<uint4 *>(keys + 16 * g);
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = make_uint4(key[4 * j], key[4 * j + 1], key[4 * j + 2], key[4 * j + 3]);
    }
    for (uint64_t i = ngroups * 16 + tid; i < npx; i += stride) keys[i] = rgb_key(rgb + 3 * i);
}

int rgb_to_keys(Ctx *c, const uint8_t *rgb_d, uint64_t npx, uint32_t *keys_d) {
    if (!npx) return CNIIC_OK;
    int aligned = ((reinterpret_cast<uintptr_t>(rgb_d) | reinterpret_cast<uintptr_t>(keys_d)) & 15) == 0;
    hipLaunchKernelGGL(k_rgb_to_keys, dim3(grid_for(aligned ? npx / 16 + 1 : npx)), dim3(256), 0, c->stream, rgb_d, npx, keys_d, aligned);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---------------------------------------------------------------- remap (clusterc.rs:43-47)
__global__ __launch_bounds__(256) void k_remap_rgb(const uint8_t *__restrict__ rgb, uint64_t npx,
                                                   const uint32_t *__restrict__ rank_table,
                                                   const uint32_t *__restrict__ lut, uint8_t *__restrict__ out,
                                                   int aligned) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ngroups = aligned ? npx / 16 : 0;
    for (uint64_t g = tid; g < ngroups; g += stride) {
        uint32_t key[16];
        load16px_keys(reinterpret_cast<const uint4 *>(rgb) + 3 * g, key);
#pragma unroll
        for (int i = 0; i < 16; i++) key[i] = lut[rank_table[key[i]] - 1];  // reduced_colors.get(..).unwrap()
        store16px_keys(reinterpret_cast<uint4 *>(out) + 3 * g, key);
    }
    for (uint64_t i = ngroups * 16 + tid; i < npx; i += stride)
        store_px(out + 3 * i, lut[rank_table[rgb_key(rgb + 3 * i)] - 1]);
}

int remap_rgb(Ctx *c, const uint8_t *rgb_d, uint64_t npx, const uint32_t *rank_table_d, const uint32_t *lut_rgb_d,
              uint8_t *out_d) {
    if (!npx) return CNIIC_OK;
    int aligned = ((reinterpret_cast<uintptr_t>(rgb_d) | reinterpret_cast<uintptr_t>(out_d)) & 15) == 0;
    hipLaunchKernelGGL(k_remap_rgb, dim3(grid_for(aligned ? npx / 16 + 1 : npx)), dim3(256), 0, c->stream, rgb_d, npx,
                       rank_table_d, lut_rgb_d, out_d, aligned);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---------------------------------------------------------------- per-colour code tables
// (len, code) of the centroid colour each distinct input colour maps to (clusterc.rs:31-40)
template <typename LabelT>
__global__ __launch_bounds__(256) void k_expand_codes(const LabelT *__restrict__ labels, uint64_t U,
                                                      const uint8_t *__restrict__ clen, const uint64_t *__restrict__ ccode,
                                                      uint8_t *__restrict__ len, uint64_t *__restrict__ code) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride) {
        const uint32_t l = labels[i];
        len[i] = clen[l];
        code[i] = ccode[l];
    }
}

int expand_codes_by_label(Ctx *c, const uint8_t *labels8_d, const uint16_t *labels16_d, uint64_t U, const uint8_t *clen_d,
                          const uint64_t *ccode_d, uint8_t *len_d, uint64_t *code_d) {
    if (!U) return CNIIC_OK;
    if (labels8_d)
        hipLaunchKernelGGL(k_expand_codes<uint8_t>, dim3(grid_for(U)), dim3(256), 0, c->stream, labels8_d, U, clen_d, ccode_d, len_d, code_d);
    else
        hipLaunchKernelGGL(k_expand_codes<uint16_t>, dim3(grid_for(U)), dim3(256), 0, c->stream, labels16_d, U, clen_d, ccode_d, len_d, code_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// dense colour -> cluster label table: key2label[keys[i]] = labels[i]  (clusterc.rs:31-40)
template <typename LabelT>
__global__ __launch_bounds__(256) void k_scatter_labels(const uint32_t *__restrict__ keys, const LabelT *__restrict__ labels,
                                                        uint64_t U, LabelT *__restrict__ key2label) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride) key2label[keys[i] & 0xffffff] = labels[i];
}
int scatter_labels_by_key(Ctx *c, const uint32_t *keys_d, const void *labels_d, bool wide, uint64_t U, void *key2label_d) {
    if (!U) return CNIIC_OK;
    if (wide)
        hipLaunchKernelGGL(k_scatter_labels<uint16_t>, dim3(grid_for(U)), dim3(256), 0, c->stream, keys_d,
                           reinterpret_cast<const uint16_t *>(labels_d), U, reinterpret_cast<uint16_t *>(key2label_d));
    else
        hipLaunchKernelGGL(k_scatter_labels<uint8_t>, dim3(grid_for(U)), dim3(256), 0, c->stream, keys_d,
                           reinterpret_cast<const uint8_t *>(labels_d), U, reinterpret_cast<uint8_t *>(key2label_d));
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// pixels of ONE image per cluster, from that image's own dense colour counts:
// out[label[i]] += local_counts[keys[i]]   (or weights[i] when there is no table)
template <typename LabelT>
__global__ __launch_bounds__(256) void k_local_weights(const uint32_t *__restrict__ keys, const LabelT *__restrict__ labels,
                                                       uint64_t U, const uint32_t *__restrict__ local_counts, uint32_t K,
                                                       unsigned long long *__restrict__ out, const uint32_t *__restrict__ weights) {
    extern __shared__ unsigned long long bins[];
    for (uint32_t i = threadIdx.x; i < K; i += 256) bins[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride) {
        const uint32_t cnt = local_counts ? local_counts[keys[i] & 0xffffff] : weights[i];
        if (cnt) atomicAdd(&bins[labels[i]], (unsigned long long)cnt);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < K; i += 256)
        if (bins[i]) atomicAdd(&out[i], bins[i]);
}
int local_cluster_weights(Ctx *c, const uint32_t *keys_d, const void *labels_d, bool wide, uint64_t U,
                          const uint32_t *local_counts_d, uint32_t K, uint64_t *out_d, const uint32_t *weights_d) {
    if (!U) return CNIIC_OK;
    auto *o = reinterpret_cast<unsigned long long *>(out_d);
    if (wide)
        hipLaunchKernelGGL(k_local_weights<uint16_t>, dim3(grid_for(U, 256, 512)), dim3(256), (size_t)K * 8, c->stream, keys_d,
                           reinterpret_cast<const uint16_t *>(labels_d), U, local_counts_d, K, o, weights_d);
    else
        hipLaunchKernelGGL(k_local_weights<uint8_t>, dim3(grid_for(U, 256, 512)), dim3(256), (size_t)K * 8, c->stream, keys_d,
                           reinterpret_cast<const uint8_t *>(labels_d), U, local_counts_d, K, o, weights_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// LUT of packed centroid colours per distinct input colour, for cniic_remap_rgb
__global__ __launch_bounds__(256) void k_label_lut(const uint32_t *__restrict__ labels, uint64_t U,
                                                   const uint32_t *__restrict__ cent, uint32_t *__restrict__ lut) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride) lut[i] = cent[labels[i]];
}
__global__ __launch_bounds__(256) void k_rank_from_keys(const uint32_t *__restrict__ keys, uint64_t U,
                                                        uint32_t *__restrict__ table) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride) table[keys[i] & 0xffffff] = (uint32_t)i + 1;
}

int label_lut(Ctx *c, const uint32_t *labels_d, uint64_t U, const uint32_t *cent_d, uint32_t *lut_d) {
    if (!U) return CNIIC_OK;
    hipLaunchKernelGGL(k_label_lut, dim3(grid_for(U)), dim3(256), 0, c->stream, labels_d, U, cent_d, lut_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}
int rank_from_keys(Ctx *c, const uint32_t *keys_d, uint64_t U, uint32_t *table_d) {
    if (!U) return CNIIC_OK;
    hipLaunchKernelGGL(k_rank_from_keys, dim3(grid_for(U)), dim3(256), 0, c->stream, keys_d, U, table_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---------------------------------------------------------------- voronoi decode (clusterc.rs:180-186)
// key = (cx-x)^2 + (cy-y)^2 in wrapping u32 arithmetic, FIRST minimum wins (Iterator::min_by_key).
// Centroid index k is wave-uniform -> scalar loads; 4 pixels per thread.
__global__ __launch_bounds__(256) void k_voronoi_paint(const cniic_colorpos *__restrict__ cent, uint32_t K,
                                                       uint32_t w, uint32_t h, uint8_t *__restrict__ out) {
    const uint64_t npx = (uint64_t)w * h;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride) {
        const uint32_t x = (uint32_t)(i % w), y = (uint32_t)(i / w);
        uint32_t best = 0xffffffffu, bk = 0;
        for (uint32_t k = 0; k < K; k++) {
            uint32_t dx = cent[k].x - x, dy = cent[k].y - y;
            uint32_t d = dx * dx + dy * dy;
            if (d < best || k == 0) { best = d; bk = k; }
        }
        const uint8_t *col = cent[bk].rgb;
        out[3 * i] = col[0]; out[3 * i + 1] = col[1]; out[3 * i + 2] = col[2];
    }
}

// The same with the centroids pruned per 64x32-pixel tile (one 256-thread block): pivot = centroid nearest the tile centre;
// centroid k is kept iff max over the tile of d(p, pivot) - d(p, c_k) >= 0 (linear in p: taken at the corners, as in
// k_kmeans_xyrgb.hip), i.e. unless the pivot is STRICTLY nearer everywhere -- so every minimum and every tie survives,
// in ascending id, and the first minimum among the kept ones is the first minimum among all.  Needs true (not
// wrapping) arithmetic: coordinates below 2^14 (the host checks; any other stream takes the brute-force kernel).
constexpr int kVTW = 64, kVTH = 32, kVCap = 256;  // (K <= 4096: at most 16 consecutive ids per thread, a bit each in `keep`)
__global__ __launch_bounds__(256) void k_voronoi_paint_tiles(const cniic_colorpos *__restrict__ cent, uint32_t K, uint32_t w, uint32_t h,
                                                             uint32_t tiles_x, uint8_t *__restrict__ out) {
    __shared__ int2 s_c[kVCap];
    __shared__ uint16_t s_k[kVCap];
    __shared__ unsigned long long s_key;
    __shared__ uint32_t wsum[256 / 64], s_n;
    const uint32_t tx0 = (blockIdx.x % tiles_x) * kVTW, ty0 = (blockIdx.x / tiles_x) * kVTH;
    const int32_t x0 = (int32_t)tx0, x1 = (int32_t)min(w, tx0 + kVTW) - 1, y0 = (int32_t)ty0, y1 = (int32_t)min(h, ty0 + kVTH) - 1;
    const int32_t cx2 = x0 + x1, cy2 = y0 + y1;  // twice the tile centre
    const uint32_t per = (K + 255) / 256, k_lo = threadIdx.x * per;
    if (threadIdx.x == 0) s_key = ~0ull;
    __syncthreads();
    unsigned long long key = ~0ull;
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t k = k_lo + i;
        if (k >= K) break;
        const int32_t dx = 2 * (int32_t)cent[k].x - cx2, dy = 2 * (int32_t)cent[k].y - cy2;
        key = min(key, ((unsigned long long)(uint32_t)(dx * dx + dy * dy) << 12) | k);  // (< 2^31: coordinates < 2^14)
    }
    key = wave_reduce_min64(key);
    if ((threadIdx.x & 63) == 0) atomicMin(&s_key, key);
    __syncthreads();
    const uint32_t pk = (uint32_t)(s_key & 4095ull);
    const int32_t px = (int32_t)cent[pk].x, py = (int32_t)cent[pk].y;
    const int32_t ax0 = px - 2 * x0, ax1 = px - 2 * x1, ay0 = py - 2 * y0, ay1 = py - 2 * y1;
    uint32_t keep = 0, cnt = 0;
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t k = k_lo + i;
        if (k >= K) break;
        const int32_t kx = (int32_t)cent[k].x, ky = (int32_t)cent[k].y, dx = px - kx, dy = py - ky;
        const int32_t f = max(dx * (kx + ax0), dx * (kx + ax1)) + max(dy * (ky + ay0), dy * (ky + ay1));  // |terms| < 2^30
        if (f >= 0) { keep |= 1u << i; cnt++; }
    }
    uint32_t off = block_exclusive_scan<256>(cnt, wsum);
    if (threadIdx.x == 255) s_n = off + cnt;
    for (uint32_t i = 0; i < per; i++)
        if ((keep >> i) & 1u) {
            if (off < kVCap) { s_c[off] = make_int2((int32_t)cent[k_lo + i].x, (int32_t)cent[k_lo + i].y); s_k[off] = (uint16_t)(k_lo + i); }
            off++;
        }
    __syncthreads();
    const uint32_t n = s_n;
    const uint32_t x = tx0 + (threadIdx.x & 63);
    for (uint32_t j = 0; j < kVTH / 4; j++) {
        const uint32_t y = ty0 + (threadIdx.x >> 6) + 4 * j;
        if (x >= w || y >= h) continue;
        uint32_t best = 0xffffffffu, bk = 0;
        if (n <= kVCap) {
            uint32_t bq = 0;
            for (uint32_t q = 0; q < n; q++) {
                const int2 cc = s_c[q];  // LDS broadcast
                const int32_t dx = cc.x - (int32_t)x, dy = cc.y - (int32_t)y;
                const uint32_t d = (uint32_t)(dx * dx + dy * dy);
                if (d < best) { best = d; bq = q; }
            }
            bk = s_k[bq];
        } else {
            for (uint32_t k = 0; k < K; k++) {
                const uint32_t dx = cent[k].x - x, dy = cent[k].y - y, d = dx * dx + dy * dy;
                if (d < best || k == 0) { best = d; bk = k; }
            }
        }
        const uint8_t *col = cent[bk].rgb;
        uint8_t *o = out + 3 * ((uint64_t)y * w + x);
        o[0] = col[0]; o[1] = col[1]; o[2] = col[2];
    }
}

// small_coords: every centroid coordinate and both image sides are below 2^14 (true arithmetic = the reference's wrapping one)
int voronoi_paint(Ctx *c, const cniic_colorpos *cent_d, uint32_t K, uint32_t w, uint32_t h, uint8_t *out_d, bool small_coords) {
    uint64_t npx = (uint64_t)w * h;
    if (!npx) return CNIIC_OK;
    if (small_coords && K >= 1 && K <= 4096) {
        const uint32_t tiles_x = (w + kVTW - 1) / kVTW, tiles_y = (h + kVTH - 1) / kVTH;
        hipLaunchKernelGGL(k_voronoi_paint_tiles, dim3(tiles_x * tiles_y), dim3(256), 0, c->stream, cent_d, K, w, h, tiles_x, out_d);
    } else {
        hipLaunchKernelGGL(k_voronoi_paint, dim3(grid_for(npx, 256, 256 * 16)), dim3(256), 0, c->stream, cent_d, K, w, h, out_d);
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---------------------------------------------------------------- MSE (bench.rs:95-104)
__global__ __launch_bounds__(256) void k_sqerr(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                               uint64_t nbytes, unsigned long long *__restrict__ total) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += stride) {
        int d = (int)a[i] - (int)b[i];
        s += (uint32_t)(d * d);
    }
    s = wave_reduce_sum64(s);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, (unsigned long long)s);
}

int mse_rgb(Ctx *c, const uint8_t *a_d, const uint8_t *b_d, uint64_t npx, double *mse_h) {
    if (!npx) { *mse_h = 0.0; return CNIIC_OK; }
    DevBuf tot;
    CNIIC_HIP_TRY(c, tot.alloc(8));
    CNIIC_HIP_TRY(c, hipMemsetAsync(tot.p, 0, 8, c->stream));
    hipLaunchKernelGGL(k_sqerr, dim3(grid_for(npx * 3, 256, 256 * 8)), dim3(256), 0, c->stream, a_d, b_d, npx * 3,
                       tot.as<unsigned long long>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    unsigned long long t = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&t, tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *mse_h = (double)t / (double)npx;  // exact integer sum; the reference sums sqrt(n)^2 in f64
    return CNIIC_OK;
}

// ---------------------------------------------------------------- synthetic images
// "U": byte k of the splitmix64 stream seeded with `seed` (LSB-first bytes of successive outputs).
// "P": per channel, integer bilinear interpolation of a hashed 64-px lattice plus +-8 noise.
__device__ __forceinline__ uint32_t synth_lattice(uint64_t seed, uint32_t i, uint32_t j, uint32_t ch) {
    uint64_t id = ((uint64_t)j << 32) | ((uint64_t)i << 2) | ch;
    return (uint32_t)(splitmix_mix(seed + 0x9E3779B97F4A7C15ULL * (id + 1)) & 255);
}

__global__ __launch_bounds__(256) void k_synth(int kind, uint64_t seed, uint32_t w, uint32_t h, uint8_t *__restrict__ out) {
    const uint64_t npx = (uint64_t)w * h;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride) {
        if (kind == CNIIC_SYNTH_UNIFORM) {
            for (uint32_t ch = 0; ch < 3; ch++) {
                uint64_t k = 3 * i + ch;
                uint64_t word = splitmix_mix(seed + 0x9E3779B97F4A7C15ULL * (k / 8 + 1));
                out[k] = (uint8_t)(word >> (8 * (k % 8)));
            }
        } else {
            const uint32_t x = (uint32_t)(i % w), y = (uint32_t)(i / w);
            const uint32_t cx = x >> 6, fx = x & 63, cy = y >> 6, fy = y & 63;
            const uint64_t seed2 = seed ^ 0xD1B54A32D192ED03ULL;
            for (uint32_t ch = 0; ch < 3; ch++) {
                uint32_t a = synth_lattice(seed, cx, cy, ch), b = synth_lattice(seed, cx + 1, cy, ch);
                uint32_t cc = synth_lattice(seed, cx, cy + 1, ch), d = synth_lattice(seed, cx + 1, cy + 1, ch);
                int v = (int)(((a * (64 - fx) + b * fx) * (64 - fy) + (cc * (64 - fx) + d * fx) * fy) >> 12);
                int noise = (int)(splitmix_mix(seed2 + 0x9E3779B97F4A7C15ULL * (i * 4 + ch + 1)) & 15) - 8;
                v += noise;
                out[3 * i + ch] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
}

int synth_image(Ctx *c, int kind, uint64_t seed, uint32_t w, uint32_t h, uint8_t *out_d) {
    uint64_t npx = (uint64_t)w * h;
    if (!npx) return CNIIC_OK;
    if (kind != CNIIC_SYNTH_UNIFORM && kind != CNIIC_SYNTH_PHOTO) return c->fail(CNIIC_ERR_BAD_ARG, "synth: unknown kind %d", kind);
    hipLaunchKernelGGL(k_synth, dim3(grid_for(npx, 256, 256 * 16)), dim3(256), 0, c->stream, kind, seed, w, h, out_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

}  // namespace cniic
