// k_kmeans_xyrgb.hip -- kmeans::cluster::<ColorPos> on gfx950: 5-D (x, y, r, g, b) K-means over every
// pixel of an image (reference: src/codec/clusterc.rs:148-153, 200-248; src/kmeans.rs:21-143,330-416).
//
// Layout: the RGB8 image stays as given (3 B/px, x and y are implicit in the pixel index) plus one
// u16 label per pixel: 3 + 2 read, 2 written = 7 B/px/iteration.
//
// Exactness: assign is exact Lloyd under the reference's rules (stay unless STRICTLY closer,
// kmeans.rs:375; lowest id among equidistant minima) on integer squared distances.
//
// Pruning (replaces the reference's per-cluster neighbour lists, kmeans.rs:150-323, which are
// sequential and heuristic once truncated): the image is cut into 64x16-pixel tiles.  For each tile
// and iteration the block computes the tile's 5-D bounding box (pixel extents + min/max of each
// colour channel), then for every centroid a lower bound lb_k and an upper bound ub_k of the squared
// distance to ANY point of the box.  With T = min_k ub_k, a centroid with lb_k > T cannot be the
// nearest (not even tied) for any pixel of the tile, so only {k : lb_k <= T} are evaluated per
// pixel.  This is the triangle-inequality idea of kmeans.rs:355-370 applied to a box of points
// instead of one point at a time; it never changes the result.
//
// Sums: per-block LDS accumulators (u32, flushed before they can overflow) -> u64 global atomics.
#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

constexpr int kTW = 64, kTH = 16;          // tile: 64 x 16 pixels, one wave per 64-px row segment
constexpr int kXThreads = 256;
constexpr int kXPPT = (kTW * kTH) / kXThreads;  // 4 pixels per thread
constexpr int kMaxCand = 1024;             // LDS candidate list capacity; beyond it the tile is brute-forced
constexpr uint32_t kXMaxK = 4096;

struct XyCent {  // device centroid table, structure of arrays
    int32_t  *cx, *cy;   // [K]
    uint32_t *crgb;      // [K] packed r<<16|g<<8|b
    int32_t  *c2;        // [K] cx^2 + cy^2 + |rgb|^2
};

struct KmXyState {
    Ctx *c = nullptr;
    const uint8_t *rgb = nullptr;
    uint32_t w = 0, h = 0, K = 0, nblocks = 1, tiles_x = 0, tiles_y = 0, flush_every = 1;
    uint64_t N = 0, seed = 0, max_iters = 0;
    bool brute = false;
    DevBuf labels, cx, cy, crgb, c2, partials, running, dstate, members_last, tile_box, tile_T, tile_mask, moved_list;
    bool no_skip = false;
    XyCent cent() const { return XyCent{cx.as<int32_t>(), cy.as<int32_t>(), crgb.as<uint32_t>(), c2.as<int32_t>()}; }
};

__device__ __forceinline__ uint32_t xdot4(uint32_t a, uint32_t b) { return __builtin_amdgcn_udot4(a, b, 0u, false); }

__global__ void k_xy_init(const uint8_t *__restrict__ rgb, uint32_t w, uint64_t N, uint32_t K,
                          uint16_t *__restrict__ labels, XyCent ct) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = tid; i < N; i += stride) labels[i] = (uint16_t)init_label(i, N, K);  // kmeans.rs:61-78
    if (tid < K) {
        uint32_t k = (uint32_t)tid;
        uint64_t ppc = N / K;
        uint64_t first = (k < K - 1) ? N - ((uint64_t)k + 1) * ppc : 0;  // init_centroids kmeans.rs:101-108
        int32_t x = (int32_t)(first % w), y = (int32_t)(first / w);
        uint32_t col = rgb_key(rgb + 3 * first);
        ct.cx[k] = x; ct.cy[k] = y; ct.crgb[k] = col;
        ct.c2[k] = x * x + y * y + (int32_t)xdot4(col, col);
    }
}

// distance bounds from centroid coordinate c to the interval [a, b]
__device__ __forceinline__ void bound1(int32_t c, int32_t a, int32_t b, uint32_t &lb, uint32_t &ub) {
    int32_t da = c - a, db = c - b;
    int32_t lo = c < a ? -da : (c > b ? db : 0);
    int32_t hi = max(abs(da), abs(db));
    lb += (uint32_t)(lo * lo);
    ub += (uint32_t)(hi * hi);
}

struct TileBox { int32_t x0, x1, y0, y1, r0, r1, g0, g1, b0, b1; };

__device__ __forceinline__ void cent_bounds(const XyCent &ct, uint32_t k, const TileBox &bx, uint32_t &lb, uint32_t &ub) {
    lb = 0; ub = 0;
    uint32_t col = ct.crgb[k];
    bound1(ct.cx[k], bx.x0, bx.x1, lb, ub);
    bound1(ct.cy[k], bx.y0, bx.y1, lb, ub);
    bound1((int32_t)((col >> 16) & 255), bx.r0, bx.r1, lb, ub);
    bound1((int32_t)((col >> 8) & 255), bx.g0, bx.g1, lb, ub);
    bound1((int32_t)(col & 255), bx.b0, bx.b1, lb, ub);
}

// partials layout (u64 words): [5k+d] sums of x,y,r,g,b ; [5K+k] member count (also wsum) ;
// [6K] moved ; [6K+1] pair evaluations.  At iteration 0 the partials are the full sums of the new
// assignment; afterwards they are SIGNED deltas of the pixels that moved, added to running sums.
constexpr uint32_t kXMaxMovedSkip = 512;   // tile-skip schedule when at most this many centroids moved
constexpr int kXMaxR = kXMaxK / kXThreads; // centroids per thread in the candidate pass (16)

struct TileState {              // per tile, carried between iterations
    uint2 *box;                 // colour extents: x = r0|r1<<8|g0<<16|g1<<24, y = b0|b1<<8   (static)
    uint32_t *T;                // min_k ub_k of the last candidate build
    unsigned long long *mask;   // [ntiles][K/64 rounded up] candidate bitmask of the last build
    const uint32_t *moved;      // [0] = number of centroids changed by the last update, then their ids
    uint32_t max_moved;         // skip schedule when moved[0] <= max_moved (0 disables it)
};

__global__ __launch_bounds__(kXThreads) void k_xy_assign(const uint8_t *__restrict__ rgb, uint32_t w, uint32_t h,
                                                         uint32_t tiles_x, uint32_t ntiles, uint32_t K, XyCent ct,
                                                         uint16_t *__restrict__ labels,
                                                         unsigned long long *__restrict__ partials,
                                                         const KmDevState *__restrict__ st, uint32_t flush_every,
                                                         int brute, TileState ts) {
    extern __shared__ __align__(16) uint32_t acc[];  // [K][6] per-block partial sums (x,y,r,g,b,count), wrap-around signed
    __shared__ int4 cand[kMaxCand];                   // (cx, cy, crgb, c2) of the tile's candidates
    __shared__ uint16_t cand_k[kMaxCand];
    __shared__ unsigned long long s_mask[kXMaxK / 64];
    __shared__ int32_t red[8];
    __shared__ uint32_t s_minub, s_ncand;
    __shared__ uint32_t wsum[kXThreads / 64];
    if (st->done) return;
    for (uint32_t i = threadIdx.x; i < 6 * K; i += kXThreads) acc[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool first = st->iter == 0;
    const uint32_t nS = ts.moved[0];
    const bool skip_mode = !first && !brute && nS <= ts.max_moved;
    const uint32_t MW = (K + 63) >> 6;
    uint32_t moved = 0;
    unsigned long long evals = 0;
    uint32_t since_flush = 0;
    const uint32_t R = (K + kXThreads - 1) / kXThreads;  // centroids per thread in the candidate pass (<= kXMaxR)
    const uint32_t k0 = threadIdx.x * R, k1 = min(k0 + R, K);

    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t tx0 = (tile % tiles_x) * kTW, ty0 = (tile / tiles_x) * kTH;
        const uint32_t tw = min((uint32_t)kTW, w - tx0), th = min((uint32_t)kTH, h - ty0);
        TileBox bx{(int32_t)tx0, (int32_t)(tx0 + tw - 1), (int32_t)ty0, (int32_t)(ty0 + th - 1), 0, 0, 0, 0, 0, 0};
        if (!first) {
            const uint2 pb = ts.box[tile];
            bx.r0 = pb.x & 255; bx.r1 = (pb.x >> 8) & 255; bx.g0 = (pb.x >> 16) & 255; bx.g1 = pb.x >> 24;
            bx.b0 = pb.y & 255; bx.b1 = (pb.y >> 8) & 255;
        }
        // ---- skip test: did anything that matters to this tile change?
        if (skip_mode) {
            const uint32_t Tp = ts.T[tile];
            int dirty = 0;
            for (uint32_t j = threadIdx.x; j < nS; j += kXThreads) {
                const uint32_t k = ts.moved[1 + j];
                uint32_t lb, ub;
                cent_bounds(ct, k, bx, lb, ub);
                dirty |= lb <= Tp || ((ts.mask[(size_t)tile * MW + (k >> 6)] >> (k & 63)) & 1ull);
            }
            if (!__syncthreads_or(dirty)) continue;  // same T, same candidates, same centroid values: every pixel repeats its decision
        }
        // ---- load this thread's pixels: pixel j of thread t is (t & 63, (t >> 6) + 4 j)
        uint32_t px[kXPPT];
        bool valid[kXPPT];
        int32_t r0 = 255, r1 = 0, g0 = 255, g1 = 0, b0 = 255, b1 = 0;
#pragma unroll
        for (int j = 0; j < kXPPT; j++) {
            const uint32_t lx = threadIdx.x & 63, ly = (threadIdx.x >> 6) + 4 * j;
            valid[j] = lx < tw && ly < th;
            px[j] = 0;
            if (valid[j]) {
                px[j] = rgb_key(rgb + 3 * ((uint64_t)(ty0 + ly) * w + tx0 + lx));
                int32_t r = (px[j] >> 16) & 255, g = (px[j] >> 8) & 255, b = px[j] & 255;
                r0 = min(r0, r); r1 = max(r1, r); g0 = min(g0, g); g1 = max(g1, g); b0 = min(b0, b); b1 = max(b1, b);
            }
        }
        if (threadIdx.x == 0) {
            red[0] = 255; red[1] = 0; red[2] = 255; red[3] = 0; red[4] = 255; red[5] = 0;
            s_minub = 0xffffffffu; s_ncand = 0;
        }
        for (uint32_t i = threadIdx.x; i < MW; i += kXThreads) s_mask[i] = 0ull;
        __syncthreads();
        if (first) {  // tile bounding box (block reduction of the colour extents); static, kept for later iterations
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                r0 = min(r0, __shfl_down(r0, off, 64)); r1 = max(r1, __shfl_down(r1, off, 64));
                g0 = min(g0, __shfl_down(g0, off, 64)); g1 = max(g1, __shfl_down(g1, off, 64));
                b0 = min(b0, __shfl_down(b0, off, 64)); b1 = max(b1, __shfl_down(b1, off, 64));
            }
            if (lane == 0) {
                atomicMin(&red[0], r0); atomicMax(&red[1], r1); atomicMin(&red[2], g0);
                atomicMax(&red[3], g1); atomicMin(&red[4], b0); atomicMax(&red[5], b1);
            }
            __syncthreads();
            bx.r0 = red[0]; bx.r1 = red[1]; bx.g0 = red[2]; bx.g1 = red[3]; bx.b0 = red[4]; bx.b1 = red[5];
            if (threadIdx.x == 0)
                ts.box[tile] = make_uint2((uint32_t)bx.r0 | ((uint32_t)bx.r1 << 8) | ((uint32_t)bx.g0 << 16) | ((uint32_t)bx.g1 << 24),
                                          (uint32_t)bx.b0 | ((uint32_t)bx.b1 << 8));
        }
        // ---- candidate set: thread t owns centroids [t R, (t+1) R) so the list comes out in ascending k
        uint32_t ncand = K;
        bool use_list = !brute;
        if (use_list) {
            uint32_t lbv[kXMaxR];
            uint32_t mub = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < kXMaxR; i++) {
                lbv[i] = 0xffffffffu;
                const uint32_t k = k0 + i;
                if ((uint32_t)i < R && k < k1) {
                    uint32_t ub;
                    cent_bounds(ct, k, bx, lbv[i], ub);
                    mub = min(mub, ub);
                }
            }
            mub = wave_reduce_min(mub);
            if (lane == 0) atomicMin(&s_minub, mub);
            __syncthreads();
            const uint32_t T = s_minub;
            uint32_t mine = 0;
#pragma unroll
            for (int i = 0; i < kXMaxR; i++) mine += lbv[i] <= T;
            uint32_t off = block_exclusive_scan<kXThreads>(mine, wsum);
            if (threadIdx.x == kXThreads - 1) s_ncand = off + mine;
            const bool fits = off + mine <= kMaxCand;
#pragma unroll
            for (int i = 0; i < kXMaxR; i++) {
                if (lbv[i] <= T) {
                    const uint32_t k = k0 + i;
                    atomicOr(&s_mask[k >> 6], 1ull << (k & 63));
                    if (fits) {
                        cand[off] = make_int4(ct.cx[k], ct.cy[k], (int32_t)ct.crgb[k], ct.c2[k]);
                        cand_k[off] = (uint16_t)k;
                        off++;
                    }
                }
            }
            __syncthreads();
            ncand = s_ncand;
            use_list = ncand <= kMaxCand;
            if (!use_list) ncand = K;
            if (threadIdx.x == 0) ts.T[tile] = T;
            for (uint32_t i = threadIdx.x; i < MW; i += kXThreads) ts.mask[(size_t)tile * MW + i] = s_mask[i];
        }
        // ---- assign
#pragma unroll
        for (int j = 0; j < kXPPT; j++) {
            if (!valid[j]) continue;
            const uint32_t lx = threadIdx.x & 63, ly = (threadIdx.x >> 6) + 4 * j;
            const int32_t x = (int32_t)(tx0 + lx), y = (int32_t)(ty0 + ly);
            const uint64_t idx = (uint64_t)y * w + x;
            int32_t best = INT32_MIN;
            uint32_t bk = 0;
            if (use_list) {
                uint32_t bpos = 0;
                for (uint32_t q = 0; q < ncand; q++) {
                    int4 cc = cand[q];  // LDS broadcast
                    int32_t dot = x * cc.x + y * cc.y + (int32_t)xdot4(px[j], (uint32_t)cc.z);
                    int32_t g = 2 * dot - cc.w;
                    if (g > best) { best = g; bpos = q; }  // ascending k: first maximum = lowest id
                }
                bk = cand_k[bpos];
            } else {
                for (uint32_t k = 0; k < K; k++) {  // k wave-uniform: scalar loads
                    int32_t dot = x * ct.cx[k] + y * ct.cy[k] + (int32_t)xdot4(px[j], ct.crgb[k]);
                    int32_t g = 2 * dot - ct.c2[k];
                    if (g > best) { best = g; bk = k; }
                }
            }
            const uint32_t cur = labels[idx];
            const int32_t gcur = 2 * (x * ct.cx[cur] + y * ct.cy[cur] + (int32_t)xdot4(px[j], ct.crgb[cur])) - ct.c2[cur];
            const bool mv = best > gcur;  // strictly closer (kmeans.rs:375)
            const uint32_t nl = mv ? bk : cur;
            if (mv) { labels[idx] = (uint16_t)nl; moved++; }
            if (mv || first) {  // vector_add clusterc.rs:221-228, as +new / -old
                uint32_t *a = acc + 6 * nl;
                atomicAdd(a + 0, (uint32_t)x);
                atomicAdd(a + 1, (uint32_t)y);
                atomicAdd(a + 2, (px[j] >> 16) & 255);
                atomicAdd(a + 3, (px[j] >> 8) & 255);
                atomicAdd(a + 4, px[j] & 255);
                atomicAdd(a + 5, 1u);
                if (!first) {
                    uint32_t *o = acc + 6 * cur;
                    atomicAdd(o + 0, 0u - (uint32_t)x);
                    atomicAdd(o + 1, 0u - (uint32_t)y);
                    atomicAdd(o + 2, 0u - ((px[j] >> 16) & 255));
                    atomicAdd(o + 3, 0u - ((px[j] >> 8) & 255));
                    atomicAdd(o + 4, 0u - (px[j] & 255));
                    atomicAdd(o + 5, 0u - 1u);
                }
            }
            evals += ncand + 1;
        }
        // ---- flush the LDS partials before a signed 32-bit lane can overflow
        if (++since_flush >= flush_every) {
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < 6 * K; i += kXThreads) {
                const uint32_t v = acc[i];
                if (v) {
                    const uint32_t k = i / 6, d = i % 6;
                    atomicAdd(&partials[d < 5 ? 5 * (size_t)k + d : 5 * (size_t)K + k], (unsigned long long)(long long)(int32_t)v);
                    acc[i] = 0;
                }
            }
            since_flush = 0;
        }
        __syncthreads();
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 6 * K; i += kXThreads) {
        const uint32_t v = acc[i];
        if (v) {
            const uint32_t k = i / 6, d = i % 6;
            atomicAdd(&partials[d < 5 ? 5 * (size_t)k + d : 5 * (size_t)K + k], (unsigned long long)(long long)(int32_t)v);
        }
    }
    moved = block_reduce_sum<kXThreads>(moved);
    evals = wave_reduce_sum64(evals);
    if (threadIdx.x == 0 && moved) atomicAdd(&partials[6 * (size_t)K], (unsigned long long)moved);
    if (lane == 0 && evals) atomicAdd(&partials[6 * (size_t)K + 1], evals);
}

// Point::mean for ColorPos (clusterc.rs:215-247) + empty-cluster reseed (kmeans.rs:110-137)
__global__ __launch_bounds__(256) void k_xy_update(unsigned long long *__restrict__ partials,
                                                   unsigned long long *__restrict__ running,
                                                   const uint8_t *__restrict__ rgb, uint32_t w, uint64_t N,
                                                   uint32_t K, uint64_t seed, uint64_t max_iters, XyCent ct,
                                                   uint64_t *__restrict__ members_out, uint32_t *__restrict__ moved_list,
                                                   KmDevState *__restrict__ st) {
    if (st->done) return;
    __shared__ uint32_t s_reseed, s_active, s_nmoved;
    if (threadIdx.x == 0) { s_reseed = 0; s_active = 0; s_nmoved = 0; }
    for (uint32_t i = threadIdx.x; i < 6 * K; i += blockDim.x) running[i] += partials[i];
    __syncthreads();
    const uint64_t iter = st->iter;
    for (uint32_t k = threadIdx.x; k < K; k += blockDim.x) {
        const unsigned long long m = running[5 * (size_t)K + k];
        members_out[k] = m;
        int32_t x, y;
        uint32_t col;
        if (m == 0) {
            uint64_t idx = reseed_index(seed, iter, k, N);  // fake_clone of the stolen pixel
            x = (int32_t)(idx % w); y = (int32_t)(idx / w);
            col = rgb_key(rgb + 3 * idx);
            atomicAdd(&s_reseed, 1u);
        } else {
            x = (int32_t)(uint32_t)(running[5 * (size_t)k + 0] / m);
            y = (int32_t)(uint32_t)(running[5 * (size_t)k + 1] / m);
            uint32_t r = (uint32_t)(running[5 * (size_t)k + 2] / m) & 255;
            uint32_t g = (uint32_t)(running[5 * (size_t)k + 3] / m) & 255;
            uint32_t b = (uint32_t)(running[5 * (size_t)k + 4] / m) & 255;
            col = (r << 16) | (g << 8) | b;
            atomicAdd(&s_active, 1u);
        }
        if (ct.cx[k] != x || ct.cy[k] != y || ct.crgb[k] != col) moved_list[1 + atomicAdd(&s_nmoved, 1u)] = k;
        ct.cx[k] = x; ct.cy[k] = y; ct.crgb[k] = col;
        ct.c2[k] = x * x + y * y + (int32_t)xdot4(col, col);
    }
    __syncthreads();
    const unsigned long long changed = partials[6 * (size_t)K];
    const unsigned long long evals = partials[6 * (size_t)K + 1];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 6 * K + 2; i += blockDim.x) partials[i] = 0;
    if (threadIdx.x == 0) {
        moved_list[0] = s_nmoved;
        st->changed_ring[iter % kHistRing] = changed;
        st->moved_last = changed;
        st->reseeds += s_reseed;
        st->active = s_active;
        st->pair_evals += evals;
        st->iter = iter + 1;
        if (changed == 0 || (max_iters && iter + 1 >= max_iters)) st->done = 1;
    }
}

__global__ void k_xy_widen(const uint16_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
__global__ void k_xy_narrow(const uint32_t *__restrict__ in, uint16_t *__restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint16_t)in[i];
}

// =========================================================================== host
static int xy_create(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K, const cniic_kmeans_opts *opts,
                     KmXyState &s) {
    const uint64_t N = (uint64_t)w * h;
    if (K == 0 || N == 0) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_xyrgb: empty problem");
    if (N / K == 0) return c->fail(CNIIC_ERR_TOO_FEW_POINTS, "kmeans: %llu points for %u clusters (src/kmeans.rs:68)",
                                   (unsigned long long)N, K);
    if (K > kXMaxK) return c->fail(CNIIC_ERR_UNSUPPORTED, "kmeans_xyrgb: K=%u > %u not supported", K, kXMaxK);
    if (w > 16384 || h > 16384) return c->fail(CNIIC_ERR_UNSUPPORTED, "kmeans_xyrgb: image side > 16384 not supported");
    s.c = c; s.rgb = rgb_d; s.w = w; s.h = h; s.K = K; s.N = N;
    s.seed = (opts && opts->seed) ? opts->seed : kDefaultSeed;
    s.max_iters = opts ? opts->max_iters : 0;
    s.brute = opts && (opts->flags & CNIIC_KM_BRUTE_FORCE);
    s.tiles_x = (uint32_t)ceil_div(w, kTW);
    s.tiles_y = (uint32_t)ceil_div(h, kTH);
    const uint32_t ntiles = s.tiles_x * s.tiles_y;
    s.nblocks = std::min<uint32_t>(ntiles, 512);
    // signed 32-bit LDS partials: (pixels between flushes) * max coordinate < 2^31
    s.flush_every = std::max<uint32_t>(1, (uint32_t)((1ull << 31) / ((uint64_t)std::max(w, h) * kTW * kTH)) - 1);
    CNIIC_HIP_TRY(c, s.labels.alloc(N * 2));
    CNIIC_HIP_TRY(c, s.cx.alloc((uint64_t)K * 4));
    CNIIC_HIP_TRY(c, s.cy.alloc((uint64_t)K * 4));
    CNIIC_HIP_TRY(c, s.crgb.alloc((uint64_t)K * 4));
    CNIIC_HIP_TRY(c, s.c2.alloc((uint64_t)K * 4));
    CNIIC_HIP_TRY(c, s.partials.alloc((6 * (uint64_t)K + 2) * 8));
    CNIIC_HIP_TRY(c, s.members_last.alloc((uint64_t)K * 8));
    CNIIC_HIP_TRY(c, s.dstate.alloc(sizeof(KmDevState)));
    CNIIC_HIP_TRY(c, s.running.alloc((6 * (uint64_t)K + 2) * 8));
    CNIIC_HIP_TRY(c, s.tile_box.alloc((uint64_t)ntiles * 8));
    CNIIC_HIP_TRY(c, s.tile_T.alloc((uint64_t)ntiles * 4));
    CNIIC_HIP_TRY(c, s.tile_mask.alloc((uint64_t)ntiles * ((K + 63) / 64) * 8));
    CNIIC_HIP_TRY(c, s.moved_list.alloc(((uint64_t)K + 1) * 4));
    s.no_skip = opts && (opts->flags & CNIIC_KM_NO_SKIP);
    CNIIC_HIP_TRY(c, hipMemsetAsync(s.partials.p, 0, (6 * (uint64_t)K + 2) * 8, c->stream));
    CNIIC_HIP_TRY(c, hipMemsetAsync(s.running.p, 0, (6 * (uint64_t)K + 2) * 8, c->stream));
    CNIIC_HIP_TRY(c, hipMemsetAsync(s.dstate.p, 0, sizeof(KmDevState), c->stream));
    const uint32_t all = K;  // before the first update every centroid counts as moved
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s.moved_list.p, &all, 4, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

static int xy_assign(KmXyState &s) {
    Ctx *c = s.c;
    hipLaunchKernelGGL(k_xy_assign, dim3(s.nblocks), dim3(kXThreads), (size_t)s.K * 6 * 4, c->stream, s.rgb, s.w, s.h,
                       s.tiles_x, s.tiles_x * s.tiles_y, s.K, s.cent(), s.labels.as<uint16_t>(),
                       s.partials.as<unsigned long long>(), s.dstate.as<KmDevState>(), s.flush_every, s.brute ? 1 : 0,
                       TileState{s.tile_box.as<uint2>(), s.tile_T.as<uint32_t>(), s.tile_mask.as<unsigned long long>(),
                                 s.moved_list.as<uint32_t>(), s.no_skip ? 0u : kXMaxMovedSkip});
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

static int xy_update(KmXyState &s) {
    Ctx *c = s.c;
    hipLaunchKernelGGL(k_xy_update, dim3(1), dim3(256), 0, c->stream, s.partials.as<unsigned long long>(),
                       s.running.as<unsigned long long>(), s.rgb, s.w, s.N, s.K, s.seed, s.max_iters, s.cent(),
                       s.members_last.as<uint64_t>(), s.moved_list.as<uint32_t>(), s.dstate.as<KmDevState>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

static uint32_t xy_grid(uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, 256), 1), 4096); }

int km_xyrgb_run(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K, const cniic_kmeans_opts *opts,
                 cniic_colorpos *centroids_h, uint32_t *labels_d_u32, uint64_t *members_h, cniic_kmeans_stats *stats) {
    KmXyState s;
    CNIIC_TRY(xy_create(c, rgb_d, w, h, K, opts, s));
    hipLaunchKernelGGL(k_xy_init, dim3(xy_grid(std::max<uint64_t>(s.N, K))), dim3(256), 0, c->stream, rgb_d, w, s.N, K,
                       s.labels.as<uint16_t>(), s.cent());
    CNIIC_HIP_TRY(c, hipGetLastError());
    KmDevState hst;
    ScopedKernelTimer timer(c, "kmeans_xyrgb_iter");
    for (;;) {
        for (int b = 0; b < 8; b++) {
            CNIIC_TRY(xy_assign(s));
            CNIIC_TRY(xy_update(s));
        }
        CNIIC_HIP_TRY(c, hipMemcpyAsync(&hst, s.dstate.p, sizeof hst, hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (hst.done) break;
    }
    timer.stop(hst.iter);
    std::vector<int32_t> cx(K), cy(K);
    std::vector<uint32_t> col(K);
    CNIIC_HIP_TRY(c, hipMemcpy(cx.data(), s.cx.p, (size_t)K * 4, hipMemcpyDeviceToHost));
    CNIIC_HIP_TRY(c, hipMemcpy(cy.data(), s.cy.p, (size_t)K * 4, hipMemcpyDeviceToHost));
    CNIIC_HIP_TRY(c, hipMemcpy(col.data(), s.crgb.p, (size_t)K * 4, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < K; k++) {
        centroids_h[k].x = (uint32_t)cx[k]; centroids_h[k].y = (uint32_t)cy[k];
        centroids_h[k].rgb[0] = (uint8_t)(col[k] >> 16); centroids_h[k].rgb[1] = (uint8_t)(col[k] >> 8);
        centroids_h[k].rgb[2] = (uint8_t)col[k]; centroids_h[k].pad = 0;
    }
    if (members_h) CNIIC_HIP_TRY(c, hipMemcpy(members_h, s.members_last.p, (size_t)K * 8, hipMemcpyDeviceToHost));
    if (labels_d_u32) {
        hipLaunchKernelGGL(k_xy_widen, dim3(xy_grid(s.N)), dim3(256), 0, c->stream, s.labels.as<uint16_t>(), labels_d_u32, s.N);
        CNIIC_HIP_TRY(c, hipGetLastError());
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (stats) {
        stats->iterations = hst.iter;
        stats->moved_last = hst.moved_last;
        stats->empty_reseeds = hst.reseeds;
        stats->active = hst.active;
        stats->pair_evals = hst.pair_evals;
    }
    return CNIIC_OK;
}

int km_xyrgb_step(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K, const cniic_colorpos *centroids_h,
                  uint32_t *labels_d_u32, uint64_t *sums_h, uint64_t *wsum_h, uint64_t *members_h, uint64_t *changed_h,
                  const cniic_kmeans_opts *opts) {
    KmXyState s;
    CNIIC_TRY(xy_create(c, rgb_d, w, h, K, opts, s));
    std::vector<int32_t> cx(K), cy(K), c2(K);
    std::vector<uint32_t> col(K);
    for (uint32_t k = 0; k < K; k++) {
        cx[k] = (int32_t)centroids_h[k].x; cy[k] = (int32_t)centroids_h[k].y;
        const uint8_t *q = centroids_h[k].rgb;
        col[k] = ((uint32_t)q[0] << 16) | ((uint32_t)q[1] << 8) | q[2];
        c2[k] = cx[k] * cx[k] + cy[k] * cy[k] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    }
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s.cx.p, cx.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s.cy.p, cy.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s.crgb.p, col.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s.c2.p, c2.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_xy_narrow, dim3(xy_grid(s.N)), dim3(256), 0, c->stream, labels_d_u32, s.labels.as<uint16_t>(), s.N);
    CNIIC_TRY(xy_assign(s));
    hipLaunchKernelGGL(k_xy_widen, dim3(xy_grid(s.N)), dim3(256), 0, c->stream, s.labels.as<uint16_t>(), labels_d_u32, s.N);
    CNIIC_HIP_TRY(c, hipGetLastError());
    std::vector<uint64_t> p(6 * (size_t)K + 2);
    CNIIC_HIP_TRY(c, hipMemcpyAsync(p.data(), s.partials.p, p.size() * 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (sums_h) memcpy(sums_h, p.data(), 5 * (size_t)K * 8);
    if (wsum_h) memcpy(wsum_h, p.data() + 5 * (size_t)K, (size_t)K * 8);
    if (members_h) memcpy(members_h, p.data() + 5 * (size_t)K, (size_t)K * 8);
    if (changed_h) *changed_h = p[6 * (size_t)K];
    return CNIIC_OK;
}

}  // namespace cniic
