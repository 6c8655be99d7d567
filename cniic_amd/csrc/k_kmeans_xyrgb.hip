// k_kmeans_xyrgb.hip -- kmeans::cluster::<ColorPos> on gfx950: 5-D (x, y, r, g, b) K-means over every
// pixel of an image (reference: src/codec/clusterc.rs:148-153, 200-248; src/kmeans.rs:21-143,330-416).
//
// Layout: the RGB8 image stays as given (3 B/px, x and y are implicit in the pixel index) plus one
// u16 label per pixel: 3 + 2 read, 2 written = 7 B/px/iteration.  The centroid table is
// an array of int4 (cx, cy, r<<16|g<<8|b, -|c|^2): one 16-byte load per centroid.
//
// Exactness: assign is exact Lloyd under the reference's rules (stay unless STRICTLY closer,
// kmeans.rs:375; lowest id among equidistant minima) on integer squared distances.
//
// Pruning (replaces the reference's per-cluster neighbour lists, kmeans.rs:150-323, which are
// sequential and heuristic once truncated).  The image is cut into 64x16-pixel tiles (one wave each)
// grouped into 4x4 super-tiles (one 16-wave block each).  For a box B of pixels -- pixel extents and
// min/max of each colour channel -- and a pivot centroid c*, the difference
//     d(p, c*) - d(p, c_k) = sum_dim (c*_d - k_d) (c*_d + k_d - 2 p_d)
// is linear in p, so its maximum over B is a sum of per-dimension maxima taken at the box faces.  If
// that maximum is negative, c* is strictly closer than c_k for every pixel of B and c_k cannot win (or
// tie) anywhere in B: it is dropped.  The pivot is the centroid nearest the box centre.  The test runs
// twice: over the super-tile's box against all K centroids (the block builds the list S), then per tile
// over S only (the wave builds its candidate strip).  This is the triangle-inequality idea of
// kmeans.rs:355-370 applied to a box of points and one pivot; it never changes the result.
//
// Skip schedule: a tile keeps its pivot and candidate bitmask between iterations.  When few centroids
// moved, a tile whose candidates did not move and for which every moved centroid is still dominated by
// the (unmoved) pivot repeats all its decisions, so it is not processed at all.  (After a tile has been
// processed every pixel's label is one of its candidates, so a moved "current" centroid always shows.)
//
// Sums: running u64 sums fed by signed deltas of the pixels that moved (full sums at iteration 0),
// collected in per-block LDS accumulators (u32, wrap-around signed) -> u64 global atomics.
//
// Arithmetic: coordinates are < 2^14, so every product uses the full-rate 24-bit multiplier
// (v_mul_i32_i24 / v_mad_i32_i24); the 32-bit v_mul_lo_u32 is quarter rate on CDNA.
#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

constexpr int kTW = 64, kTH = 16;          // tile: 64 x 16 pixels = one wave, 16 pixels (a column) per lane
#ifndef CNIIC_XY_BUFLOADS
#define CNIIC_XY_BUFLOADS 1
#endif
#ifndef CNIIC_XY_STY
#define CNIIC_XY_STY 4                     // (measuring builds: 2 = super-tiles of 4 x 2 tiles, 8-wave blocks, two per CU while K <= 1024 -- NOTES D)
#endif
constexpr int kSTX = 4, kSTY = CNIIC_XY_STY;  // super-tile: 4 x 4 tiles = one block
constexpr int kXWaves = kSTX * kSTY;
constexpr int kXThreads = 64 * kXWaves;    // 1024
constexpr int kXRows = 4;                  // rows of a tile evaluated together (4 groups per tile)
constexpr int kXUS = kXWaves / 4;          // dirty tiles in work at a time: wave v takes row group v & 3 of dirty tiles number v >> 2, (v >> 2) + kXUS, ...
constexpr uint32_t kXMaxK = kSTY == 4 ? 4096 : 2048;
constexpr int kXMaxR = kXMaxK / kXThreads; // centroids per thread in the super-tile pass (4)
constexpr uint32_t kSCap = kSTY == 4 ? 1024 : 512;  // super-tile list capacity; beyond it the super-tile is brute-forced
constexpr uint32_t kXMaxMovedSkip = 512;   // skip schedule when at most this many centroids moved
constexpr uint64_t kSuperPx = (uint64_t)kSTX * kTW * kSTY * kTH;

struct KmXyState {
    Ctx *c = nullptr;
    const uint8_t *rgb = nullptr;
    uint32_t w = 0, h = 0, K = 0, nblocks = 1, tiles_x = 0, tiles_y = 0, super_x = 0, super_y = 0, wcap = 0;
    size_t lds = 0;
    uint64_t N = 0, seed = 0, max_iters = 0;
    bool brute = false, no_skip = false;
    DevBuf labels, cent, partials, running, dstate, members_last, tile_box, super_box, tile_piv, tile_mask, moved_list;
    DevBuf sup_piv, sup_mask;              // super-tile filter of the skip schedule
    DevBuf f_partials, f_running, f_cent;  // the loop with the update folded into the assign launches: 3 / 2 / 2 buffers (XyFused)
    uint32_t sup_cap = 1;                  // super-tiles a block may take in one launch (its u32 accumulators)
    uint32_t dyn = 16;                     // ... which draws its super-tiles from a counter while at least this many centroids move (CNIIC_XY_DYN; 0: a block takes every gridDim-th)
    bool use_tab = false, fused = false;
    uint32_t launch_no = 0;
};

__device__ __forceinline__ uint32_t xdot4(uint32_t a, uint32_t b) { return __builtin_amdgcn_udot4(a, b, 0u, false); }
__device__ __forceinline__ int32_t xmad24(int32_t a, int32_t b, int32_t c) { return __mul24(a, b) + c; }
__device__ __forceinline__ int4 make_cent(int32_t x, int32_t y, uint32_t col) {
    return make_int4(x, y, (int32_t)col, -xmad24(x, x, xmad24(y, y, (int32_t)xdot4(col, col))));
}

__global__ void k_xy_init(const uint8_t *__restrict__ rgb, uint32_t w, uint64_t N, uint32_t K,
                          uint16_t *__restrict__ labels, int4 *__restrict__ cent) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = tid; i < N; i += stride) labels[i] = (uint16_t)init_label(i, N, K);  // kmeans.rs:61-78
    if (tid < K) {
        uint32_t k = (uint32_t)tid;
        uint64_t ppc = N / K;
        uint64_t first = (k < K - 1) ? N - ((uint64_t)k + 1) * ppc : 0;  // init_centroids kmeans.rs:101-108
        cent[k] = make_cent((int32_t)(first % w), (int32_t)(first / w), rgb_key(rgb + 3 * first));
    }
}

// static colour extents of every tile: x = r0 | r1<<8 | g0<<16 | g1<<24, y = b0 | b1<<8.  One wave per tile.
__global__ __launch_bounds__(256) void k_xy_boxes(const uint8_t *__restrict__ rgb, uint32_t w, uint32_t h, uint32_t tiles_x,
                                                  uint32_t ntiles, uint2 *__restrict__ box) {
    const uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t x = (tile % tiles_x) * kTW + lane, y0 = (tile / tiles_x) * kTH;
    uint32_t lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0};
    if (x < w) {
        for (uint32_t y = y0; y < min(y0 + kTH, h); y++) {
            const uint32_t p = rgb_key(rgb + 3 * ((uint64_t)y * w + x));
#pragma unroll
            for (int s = 0; s < 3; s++) {
                const uint32_t v = (p >> (16 - 8 * s)) & 255;
                lo[s] = min(lo[s], v); hi[s] = max(hi[s], v);
            }
        }
    }
    uint32_t m[6];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        m[2 * s] = wave_reduce_min(lo[s]);
        m[2 * s + 1] = wave_reduce_max(hi[s]);
    }
    if (lane == 0) box[tile] = make_uint2(m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24), m[4] | (m[5] << 8));
}

// the same for every super-tile: union of its tiles' extents (also static)
__global__ void k_xy_super_boxes(const uint2 *__restrict__ box, uint32_t tiles_x, uint32_t tiles_y, uint32_t super_x, uint32_t nsuper,
                                 uint2 *__restrict__ sbox) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsuper) return;
    const uint32_t stx = (s % super_x) * kSTX, sty = (s / super_x) * kSTY;
    uint32_t lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0};
    for (uint32_t t = 0; t < (uint32_t)kXWaves; t++) {
        const uint32_t ux = stx + (t % kSTX), uy = sty + (t / kSTX);
        if (ux >= tiles_x || uy >= tiles_y) continue;
        const uint2 b = box[uy * tiles_x + ux];
        lo[0] = min(lo[0], b.x & 255); hi[0] = max(hi[0], (b.x >> 8) & 255);
        lo[1] = min(lo[1], (b.x >> 16) & 255); hi[1] = max(hi[1], b.x >> 24);
        lo[2] = min(lo[2], b.y & 255); hi[2] = max(hi[2], (b.y >> 8) & 255);
    }
    sbox[s] = make_uint2(lo[0] | (hi[0] << 8) | (lo[1] << 16) | (hi[1] << 24), lo[2] | (hi[2] << 8));
}

// floor(sum / m) for a centroid's coordinate: sum < 2^42 (a coordinate below 2^14 times at most 2^28 members), m < 2^32, quotient < 2^14.
// A SINGLE-precision estimate is within one of it (relative error 3 * 2^-24 on a value below 2^14) and is put right with one 32 x 32 -> 64
// product; two steps either way are allowed for.  (Until round 4: a double quotient and 64 x 64 products, five per changed centroid and
// 2048 centroids per block and launch -- the folded-in update's 2.5 us.)
__device__ __forceinline__ uint32_t xy_div_floor(unsigned long long sum, uint32_t m, float rm) {
    uint32_t e = (uint32_t)((float)sum * rm);
    unsigned long long em = (unsigned long long)e * m;
    if (em > sum) { e--; em -= m; if (em > sum) e--; }
    else if (em + m <= sum) { e++; em += m; if (em + m <= sum) e++; }
    return e;
}

struct Box5 { int32_t lo[5], hi[5]; };  // x, y, r, g, b extents

__device__ __forceinline__ void box_colours(Box5 &b, uint2 pb) {
    b.lo[2] = pb.x & 255; b.hi[2] = (pb.x >> 8) & 255; b.lo[3] = (pb.x >> 16) & 255; b.hi[3] = pb.x >> 24;
    b.lo[4] = pb.y & 255; b.hi[4] = (pb.y >> 8) & 255;
}

// squared distance from the box centre to a centroid
__device__ __forceinline__ uint32_t centre_dist(const Box5 &b, int4 c) {
    int32_t d = 0;
    const int32_t v[5] = {c.x, c.y, (c.z >> 16) & 255, (c.z >> 8) & 255, c.z & 255};
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int32_t e = v[i] - ((b.lo[i] + b.hi[i]) >> 1);
        d = xmad24(e, e, d);
    }
    return (uint32_t)d;
}

// the pivot against one box: a[2 i] = c*_i - 2 lo_i, a[2 i + 1] = c*_i - 2 hi_i
struct Dominance {
    int32_t p[5], a[10];
    __device__ __forceinline__ void set(const Box5 &b, int4 pv) {
        p[0] = pv.x; p[1] = pv.y; p[2] = (pv.z >> 16) & 255; p[3] = (pv.z >> 8) & 255; p[4] = pv.z & 255;
#pragma unroll
        for (int i = 0; i < 5; i++) { a[2 * i] = p[i] - 2 * b.lo[i]; a[2 * i + 1] = p[i] - 2 * b.hi[i]; }
    }
    // max over the box of d(p, pivot) - d(p, c): c can be nearest (or tie) somewhere in the box only if >= 0.
    // |c* - k| < 2^14 and |c* + k - 2p| < 2^15: 24-bit products, and the five terms sum below 2^31.
    __device__ __forceinline__ int32_t worst(int4 c) const {
        const int32_t v[5] = {c.x, c.y, (c.z >> 16) & 255, (c.z >> 8) & 255, c.z & 255};
        int32_t f = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int32_t d = p[i] - v[i];
            f += max(__mul24(d, v[i] + a[2 * i]), __mul24(d, v[i] + a[2 * i + 1]));
        }
        return f;
    }
};

// position of the i-th set bit of m (i < popcount(m)); wave-uniform arguments: scalar code
__device__ __forceinline__ uint32_t nth_set_bit(uint32_t m, uint32_t i) {
    for (uint32_t t = 0; t < i; t++) m &= m - 1;
    return (uint32_t)__ffs((int)m) - 1u;
}
// the same for a 64-bit word held by one lane: six halvings
__device__ __forceinline__ uint32_t select64(unsigned long long m, uint32_t r) {
    uint32_t pos = 0;
#pragma unroll
    for (int wd = 32; wd >= 1; wd >>= 1) {
        const uint32_t c = (uint32_t)__popcll(m & ((1ull << wd) - 1ull));
        if (r >= c) { r -= c; m >>= wd; pos += wd; }
    }
    return pos;
}
constexpr uint32_t kXGroupLaunches = 3;  // launches 1 .. this book a row's movers label by label (wave reductions) when at least kXGroupMin of the row move
constexpr uint32_t kXGroupMin = 8;
constexpr uint32_t kXHeavyWords = 64;   // super-tiles whose order of issue follows the launch before: 64 x 64 (a 8192 x 8192 image); more: index order

// partials layout (u64 words): [5k+d] sums of x,y,r,g,b ; [5K+k] member count (also wsum) -- the step ABI and the loop with a separate
// update kernel; the loop with the folded-in update keeps [d K + k] (d = 0..4 the sums, 5 the members) so that a wave reads consecutive words;
// [6K] moved ; [6K+1] pair evaluations.  At iteration 0 the partials are the full sums of the new
// assignment; afterwards they are SIGNED deltas of the pixels that moved, added to running sums.
struct TileState {              // per tile, carried between iterations
    const uint2 *box;           // colour extents (static, k_xy_boxes)
    const uint2 *sbox;          // colour extents of every super-tile (static)
    int4 *piv;                  // pivot of the last candidate build: (cx, cy, colour, id)
    unsigned long long *mask;   // [ntiles][K/64 rounded up] candidate bitmask of the last build
    const uint32_t *moved;      // [0] = number of centroids changed by the last update, then their ids
    uint32_t max_moved;         // skip schedule when moved[0] <= max_moved (0 disables it)
    uint32_t *spiv;             // per super-tile: id of the pivot of its last list build (0xffffffff: none yet)
    unsigned long long *smask;  // per super-tile: union of its tiles' candidate masks
    uint32_t dyn;               // the loop with the folded-in update: while at least this many centroids move (0: never), super-tiles beyond a block's first are
                                // drawn from a counter (word 6 K + 2 of the launch's sums)
    uint32_t sup_cap;           // ... at most this many per block: its u32 accumulators hold what that many super-tiles can add (xy_create: per_block_max)
    uint32_t tl_launch;         // measuring builds (-DCNIIC_XY_PHASES): 1 + the launch whose blocks write their timeline (CNIIC_XY_TL_LAUNCH)
};

// -DCNIIC_XY_PHASES: wave-clock totals per phase of k_xy_assign (a measuring build, never the shipped one)
#ifdef CNIIC_XY_PHASES
__device__ unsigned long long g_xy_tl[256][8];   // one launch (CNIIC_XY_TL_LAUNCH): per block the 100 MHz clock at entry, set-up loads out, prologue done, loop done, flush done; [5] super-tiles taken, [6] dirty ones
__device__ unsigned long long g_xy_phase[12];
#define XY_PHASE(i) do { const long long now_ = clock64(); ph_[i] += (unsigned long long)(now_ - t_ph); t_ph = now_; } while (0)
#define XY_COUNT(i, v) do { ph_[i] += (unsigned long long)(v); } while (0)
#else
#define XY_PHASE(i) do {} while (0)
#define XY_COUNT(i, v) do {} while (0)
#endif

__host__ __device__ constexpr uint32_t xy_acc_words(uint32_t K) { return (6 * K + 3) & ~3u; }  // keeps the int4 arrays aligned

// Centroid update folded into the next assign launch (round 3; the colour kernel has had it since round 1): launch j first finishes
// iteration j - 1 -- every block turns (running sums + the deltas of launch j - 1) into the K centroids it loads into LDS anyway,
// two per thread, and block 0 also writes the global state -- and then assigns.  One dependent kernel less per iteration (8.5 us +
// the gap in front of it, x 188 launches of the configs[2] run).  The deltas are triple-buffered (launch j adds into buffer j % 3,
// reads (j - 1) % 3, block 0 clears (j + 1) % 3); running sums and the centroids to compare with ping-pong.  Only with the table
// in LDS (K <= 2048): without it other blocks would read centroids from memory while block 0 writes them.
struct XyFused {
    uint32_t on, launch_no;
    uint64_t seed, max_iters, N;
    const unsigned long long *partials_prev;   // deltas of launch j - 1
    unsigned long long *partials_clear;        // the buffer launch j + 1 adds into
    const unsigned long long *running_prev;
    unsigned long long *running_new;
    const int4 *cent_prev;                     // centroids launch j - 1 assigned with
    int4 *cent_new;                            // block 0: this launch's, for launch j + 1 to start from
    int4 *cent_g;                              // ... and the result copy
    uint64_t *members_out;
    KmDevState *st_rw;
};

// LDS (dynamic): acc[K][6] u32 | S_c[kSCap] int4 | W_c[16][wcap] int4 | M_c[512] int4 | W_mask[16][MW] u64 |
//                S_k[kSCap] u16 | W_k[16][wcap] u16 | (use_tab) tab[K] int4: the centroid table itself
// GROUP: the instance for the launches right after the first (the loop with the folded-in update knows which launch it enqueues):
// the same body plus the label-by-label booking of a row's movers -- whose mere presence costs the other 180 launches 6 us each.
template <bool GROUP>
__global__ __launch_bounds__(kXThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_xy_assign(const uint8_t *__restrict__ rgb, uint32_t w, uint32_t h,
                                                         uint32_t tiles_x, uint32_t tiles_y, uint32_t super_x, uint32_t nsuper,
                                                         uint32_t K, const int4 *__restrict__ cent,
                                                         uint16_t *__restrict__ labels,
                                                         unsigned long long *__restrict__ partials,
                                                         const KmDevState *__restrict__ st, uint32_t wcap, int use_tab,
                                                         int brute, TileState ts, XyFused fz) {
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t MW = (K + 63) >> 6;
    uint32_t *acc = lds;
    int4 *S_c = reinterpret_cast<int4 *>(acc + xy_acc_words(K));
    int4 *W_c = S_c + kSCap;
    int4 *M_c = W_c + (size_t)kXWaves * wcap;
    unsigned long long *W_mask = reinterpret_cast<unsigned long long *>(M_c + kXMaxMovedSkip);
    uint16_t *S_k = reinterpret_cast<uint16_t *>(W_mask + (size_t)kXWaves * MW);
    uint16_t *W_k = S_k + kSCap;
    // 16-byte aligned: the u16 arrays before it hold 16 * (64 + wcap) * 2 bytes
    int4 *tab = reinterpret_cast<int4 *>(W_k + (size_t)kXWaves * wcap);
    __shared__ unsigned long long s_key, s_evals;
    __shared__ uint32_t s_n;
    __shared__ uint32_t wsum[kXWaves], s_dirty[2][kXWaves], s_ncand[kXWaves];
    __shared__ uint16_t s_rel[3][kXMaxMovedSkip];   // skip schedule: the moved centroids that matter to the super-tile (indices into M_c)
    __shared__ uint32_t s_nrel[3];
    __shared__ unsigned long long s_smask[64];      // union of the super-tile's tiles' candidate masks, being collected
    const uint32_t done = st->done;  // acted on once the set-up loads are out

    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the order of issue of the super-tiles (see the loop below): the bitmap of the launch before, its complement, and how many bits lie
    // before each word of either; this block's first draw
    __shared__ unsigned long long s_cw[3][64];   // class 0: at least half of the tiles were dirty in the launch before, 1: some were, 2: none
    __shared__ uint32_t s_hpre[3][65];
    unsigned long long drawn = 0;
    if (fz.on && ts.dyn) {
        if (threadIdx.x == 0) drawn = atomicAdd(&partials[6 * (size_t)K + 2], 1ull);
        const uint32_t nw = (nsuper + 63) >> 6;
        if (wv == 0 && fz.launch_no && nw <= kXHeavyWords) {
            const unsigned long long valid = lane + 1 < nw ? ~0ull : lane + 1 == nw ? ((nsuper & 63) ? (1ull << (nsuper & 63)) - 1ull : ~0ull) : 0ull;
            const unsigned long long any = lane < nw ? fz.partials_prev[6 * (size_t)K + 4 + lane] & valid : 0ull;
            const unsigned long long many = lane < nw ? fz.partials_prev[6 * (size_t)K + 4 + kXHeavyWords + lane] & any : 0ull;
            const unsigned long long cw[3] = {many, any & ~many, ~any & valid};
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const uint32_t pc = (uint32_t)__popcll(cw[q]), inc = wave_inclusive_scan(pc);
                s_cw[q][lane] = cw[q];
                s_hpre[q][lane] = inc - pc;
                if (lane == 63) s_hpre[q][64] = inc;
            }
        }
    }
#ifdef CNIIC_XY_PHASES
    const bool tl_on_ = fz.on && ts.tl_launch == fz.launch_no + 1 && blockIdx.x < 256 && threadIdx.x == 0;
    unsigned long long tl_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tl_on_) tl_[0] = wall_clock64();
    long long t_ph = clock64();
    unsigned long long ph_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __shared__ unsigned long long s_ph[12];
    if (threadIdx.x < 12) s_ph[threadIdx.x] = 0;
#endif
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const bool first = fz.on ? fz.launch_no == 0 : st->iter == 0;
    uint32_t nS = fz.on ? K : ts.moved[0];
    const bool upd = fz.on && !first;
    __shared__ uint32_t s_nmoved, s_reseed, s_active;
    for (uint32_t i = threadIdx.x; i < 6 * K; i += kXThreads) acc[i] = 0;
    if (threadIdx.x == 0) { s_key = ~0ull; s_evals = 0; s_nmoved = 0; s_reseed = 0; s_active = 0; s_nrel[0] = s_nrel[1] = s_nrel[2] = 0; }
    // this thread's slice of the centroid table, for the whole launch: [t R, (t+1) R) so lists come out ascending
    const uint32_t R = (K + kXThreads - 1) / kXThreads;
    int4 mine[kXMaxR];
    int4 *const my_c = W_c + (size_t)wv * wcap;
    uint16_t *const my_k = W_k + (size_t)wv * wcap;
    unsigned long long *const my_mask = W_mask + (size_t)wv * MW;
    uint32_t moved = 0;
    unsigned long long evals = 0;
    if (!upd) {
        if (!fz.on && !first && !brute && nS <= ts.max_moved)
            for (uint32_t j = threadIdx.x; j < nS; j += kXThreads) {
                const uint32_t k = ts.moved[1 + j];
                int4 c = cent[k];
                c.w = (int32_t)k;
                M_c[j] = c;
            }
        if (use_tab)
            for (uint32_t k = threadIdx.x; k < K; k += kXThreads) tab[k] = cent[k];
#pragma unroll
        for (int i = 0; i < kXMaxR; i++) {
            const uint32_t k = threadIdx.x * R + i;
            mine[i] = ((uint32_t)i < R && k < K) ? cent[k] : make_int4(0, 0, 0, 0);
        }
        if (done) return;
        if (fz.on && blockIdx.x == 0)
            for (uint32_t i = threadIdx.x; i < 6 * K + 4 + 2 * kXHeavyWords; i += kXThreads) fz.partials_clear[i] = 0ull;
        __syncthreads();
    } else {
        // ---- finish iteration j - 1: Point::mean for ColorPos (clusterc.rs:215-247) + empty-cluster reseed (kmeans.rs:110-137),
        // two clusters per thread, everything requested before anything is looked at
        const uint32_t j = fz.launch_no;
        const unsigned long long changed = fz.partials_prev[6 * (size_t)K], pev = fz.partials_prev[6 * (size_t)K + 1];
        unsigned long long r[kXMaxR][6];   // running sums + the deltas (kept apart they were 96 registers and spilled)
        bool anyd[kXMaxR];
        int4 oc[kXMaxR];
#pragma unroll
        for (int i = 0; i < kXMaxR; i++) {
            const uint32_t k = threadIdx.x * R + i;
            oc[i] = make_int4(0, 0, 0, 0);
            anyd[i] = false;
#pragma unroll
            for (int q = 0; q < 6; q++) r[i][q] = 0;
            if ((uint32_t)i < R && k < K) {
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const size_t at = (size_t)q * K + k;   // (the fused loop's buffers are [6][K]: a wave's loads of one sum are consecutive words -- [K][5] | [K] cost 40 cache lines an instruction)
                    const unsigned long long dd = fz.partials_prev[at];
                    anyd[i] |= dd != 0;
                    r[i][q] = fz.running_prev[at] + dd;
                }
                oc[i] = fz.cent_prev[k];
            }
        }
        if (done) return;   // a launch past convergence
#ifdef CNIIC_XY_PHASES
        if (tl_on_) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tl_[1] = wall_clock64(); }
#endif
        __syncthreads();    // (the accumulators and counters above are in place)
#pragma unroll
        for (int i = 0; i < kXMaxR; i++) {
            const uint32_t k = threadIdx.x * R + i;
            mine[i] = make_int4(0, 0, 0, 0);
            if ((uint32_t)i < R && k < K) {
                const bool any = anyd[i];
                const unsigned long long m = r[i][5];
                int4 nc = oc[i];
                if (m == 0) {  // reseeded every iteration it stays empty (the index depends on the iteration)
                    const uint64_t idx = reseed_index(fz.seed, (uint64_t)j - 1, k, fz.N);  // fake_clone of the stolen pixel
                    nc = make_cent((int32_t)(idx % w), (int32_t)(idx / w), rgb_key(rgb + 3 * idx));
                    atomicAdd(&s_reseed, 1u);
                } else {
                    atomicAdd(&s_active, 1u);
                    if (any) {  // floor(sum / m): see xy_div_floor
                        uint32_t q5[5];
                        const uint32_t m32 = (uint32_t)m;
                        const float rm = 1.0f / (float)m32;
#pragma unroll
                        for (int q = 0; q < 5; q++) q5[q] = xy_div_floor(r[i][q], m32, rm);
                        nc = make_cent((int32_t)q5[0], (int32_t)q5[1], ((q5[2] & 255) << 16) | ((q5[3] & 255) << 8) | (q5[4] & 255));
                    }
                }
                mine[i] = nc;
                tab[k] = nc;
                if (oc[i].x != nc.x || oc[i].y != nc.y || oc[i].z != nc.z) {
                    const uint32_t pos = atomicAdd(&s_nmoved, 1u);
                    if (pos < kXMaxMovedSkip) { int4 mc = nc; mc.w = (int32_t)k; M_c[pos] = mc; }
                }
                if (blockIdx.x == 0) {
#pragma unroll
                    for (int q = 0; q < 6; q++) fz.running_new[(size_t)q * K + k] = r[i][q];
                    fz.cent_new[k] = nc;
                    fz.cent_g[k] = nc;
                    fz.members_out[k] = m;
                }
            }
        }
        if (blockIdx.x == 0)
            for (uint32_t i = threadIdx.x; i < 6 * K + 4 + 2 * kXHeavyWords; i += kXThreads) fz.partials_clear[i] = 0ull;
        __syncthreads();
        const bool fin = changed == 0 || (fz.max_iters && j >= fz.max_iters);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            KmDevState *sw = fz.st_rw;
            sw->changed_ring[(j - 1) % kHistRing] = changed;
            sw->moved_last = changed;
            sw->reseeds += s_reseed;
            sw->active = s_active;
            sw->pair_evals += pev;
            sw->iter = j;
            if (fin) sw->done = 1;
        }
        if (fin) return;  // converged (or the iteration cap): nothing to assign
        nS = s_nmoved;
    }
    const bool skip_mode = !first && !brute && nS <= ts.max_moved;
    XY_PHASE(0);
#ifdef CNIIC_XY_PHASES
    if (tl_on_) tl_[2] = wall_clock64();
#endif

    const uint64_t npix = (uint64_t)w * h;
#if CNIIC_XY_BUFLOADS
    const __amdgpu_buffer_rsrc_t rs_rgb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(rgb), 0, (int)(3u * (uint32_t)npix), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lab = __builtin_amdgcn_make_buffer_rsrc(labels, 0, (int)(2u * (uint32_t)npix), 0x00020000);
#endif
    uint32_t par = 0, sit = 0;
    // Which super-tiles a block takes.  Statically: every gridDim-th.  The loop with the folded-in update, while centroids still move
    // (dyn): whatever the launch's counter hands out next -- the dirty tiles lie in patches of the image, and a block that met four
    // busy super-tiles kept the launch waiting: the waves were alive for 0.63-0.68 of a launch's duration (SQ_WAVE_CYCLES).  The
    // draws are handed out in the order "at least half of its tiles dirty in the launch before, some dirty, none" (two bits per
    // super-tile, set by whoever worked on it): what is still out when the blocks run dry is then the super-tiles that take a microsecond.  The draw for
    // the NEXT super-tile is asked for before this one is worked on (a round trip to the L2).
    __shared__ uint32_t s_sup[2];
    const bool dyn = fz.on && ts.dyn && (!skip_mode || nS >= ts.dyn);   // (late in a run most super-tiles are passed over in a microsecond: nothing to balance)
    const uint32_t hNW = (nsuper + 63) >> 6;
    const bool ordered = dyn && !first && hNW <= kXHeavyWords;
    unsigned long long *const heavy_cur = fz.on && hNW <= kXHeavyWords ? partials + 6 * (size_t)K + 4 : nullptr;
    auto resolve = [&]() -> uint32_t {   // the draw thread 0 holds -> a super-tile, for every thread
        if (wv == 0) {
            const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)drawn);
            uint32_t v = d;
            if (ordered) {
                const uint32_t n0 = s_hpre[0][64], n1 = s_hpre[1][64];
                const uint32_t cls = d < n0 ? 0u : d < n0 + n1 ? 1u : 2u;   // (wave-uniform)
                const uint32_t r = d - (cls == 0 ? 0u : cls == 1 ? n0 : n0 + n1);
                const uint32_t e0 = s_hpre[cls][lane], e1 = s_hpre[cls][lane + 1];
                const unsigned long long bm = __ballot(r >= e0 && r < e1);
                v = 0xffffffffu;
                if (bm) {
                    const int src = __builtin_ctzll(bm);
                    const unsigned long long word = s_cw[cls][lane];
                    v = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)lane * 64u + select64(word, r - e0)), src);
                }
            }
            if (lane == 0) s_sup[sit & 1] = v;
        }
        __syncthreads();
        return s_sup[sit & 1];
    };
    uint32_t sup = blockIdx.x;
    if (dyn) { sit = 1; sup = resolve(); sit = 0; }
    // (a block that has taken sup_cap super-tiles stops: the grid is sized so that every block taking that many covers the image, and
    // the blocks that start later -- a large image has more blocks than fit the machine -- draw what it leaves)
    for (; sup < nsuper && (!dyn || sit < ts.sup_cap); sup = dyn ? resolve() : sup + gridDim.x, par ^= 1, sit++) {
        if (dyn && threadIdx.x == 0 && sit + 1 < ts.sup_cap) drawn = atomicAdd(&partials[6 * (size_t)K + 2], 1ull);
        const uint32_t stx = (sup % super_x) * kSTX, sty = (sup / super_x) * kSTY;
        const uint32_t tix = stx + (wv & (kSTX - 1)), tiy = sty + wv / kSTX;
        const bool has_tile = tix < tiles_x && tiy < tiles_y;
        const uint32_t tile = has_tile ? tiy * tiles_x + tix : 0;
        const uint32_t tx0 = tix * kTW, ty0 = tiy * kTH;
        const uint32_t tw = has_tile ? min((uint32_t)kTW, w - tx0) : 0, th = has_tile ? min((uint32_t)kTH, h - ty0) : 0;
        Box5 tb;
        tb.lo[0] = (int32_t)tx0; tb.hi[0] = (int32_t)(tx0 + tw) - 1; tb.lo[1] = (int32_t)ty0; tb.hi[1] = (int32_t)(ty0 + th) - 1;
        box_colours(tb, has_tile ? ts.box[tile] : make_uint2(0, 0));

        // ---- skip schedule, first the whole super-tile (round 3): of the centroids that moved, which matter here at all?  One per
        // thread: a moved centroid that no tile of the super-tile has as a candidate (the union of their masks) and that the
        // super-tile's pivot -- any centroid will do for the argument; the pivot of its last list build is a good one -- strictly
        // dominates over the super-tile's whole box cannot be nearest or tied for any pixel of it, before or after its move: no
        // tile's decisions depend on it.  The tiles then test what is left, typically a handful of the up to 512 (every tile
        // against every moved centroid was a quarter of the kernel's instructions in mid-run), and a super-tile nothing
        // matters to is done without reading a tile record.
        const uint32_t it3 = sit % 3;
        uint32_t nrel = nS;
        const uint32_t prev_piv = (first || brute) ? 0xffffffffu : ts.spiv[sup];   // the pivot of the super-tile's last list build
        if (skip_mode) {
            if (threadIdx.x == 0) s_nrel[(sit + 1) % 3] = 0;   // (the counter of the NEXT super-tile; its last readers are two barriers back)
            const uint32_t pid = prev_piv;
            if (threadIdx.x < nS) {
                const int4 m = M_c[threadIdx.x];
                bool rel = true;
                if (pid < K) {
                    Box5 sb;
                    sb.lo[0] = (int32_t)(stx * kTW); sb.hi[0] = (int32_t)min(w, (stx + kSTX) * kTW) - 1;
                    sb.lo[1] = (int32_t)(sty * kTH); sb.hi[1] = (int32_t)min(h, (sty + kSTY) * kTH) - 1;
                    box_colours(sb, ts.sbox[sup]);
                    Dominance ds;
                    ds.set(sb, use_tab ? tab[pid] : cent[pid]);
                    const unsigned long long uw = ts.smask[(size_t)sup * MW + ((uint32_t)m.w >> 6)];
                    rel = ((uw >> (m.w & 63)) & 1ull) || ds.worst(m) >= 0;
                }
                if (rel) s_rel[it3][atomicAdd(&s_nrel[it3], 1u)] = (uint16_t)threadIdx.x;
            }
            __syncthreads();
            nrel = s_nrel[it3];
            if (nrel == 0) continue;   // nothing that moved matters to any tile here (block-uniform)
        }
        if (wv == 0 && lane < MW) s_smask[lane] = 0ull;   // (collected below, between the two barriers of a super-tile that has dirty tiles)
        // ---- skip test (per wave): did anything that matters to this tile change?
        bool dirty = has_tile;
        unsigned long long mword = 0ull;
        if (skip_mode && has_tile) {
            Dominance dm;
            dm.set(tb, ts.piv[tile]);
            mword = lane < MW ? ts.mask[(size_t)tile * MW + lane] : 0ull;
            bool d = false;
            for (uint32_t j0 = 0; j0 < nrel; j0 += 64) {  // wave-uniform trip count: the shuffle needs every lane
                const uint32_t j = j0 + lane;
                const int4 m = M_c[s_rel[it3][min(j, nrel - 1)]];
                const unsigned long long wd = __shfl(mword, (m.w >> 6) & 63, 64);
                d |= j < nrel && (((wd >> (m.w & 63)) & 1ull) || dm.worst(m) >= 0);
            }
            dirty = __ballot(d) != 0ull;
        }
        // ---- the dirty tiles of the super-tile are evaluated by ALL 16 waves: the work unit is (tile, group of
        // kXRows rows); wave v takes row group v & 3 of dirty tiles number (v >> 2), (v >> 2) + 4, ... so the wave of a
        // clean tile works for its neighbours instead of waiting for them.
        // (s_dirty alternates between two copies: with no dirty tile there is no second barrier before the next write)
        if (lane == 0) s_dirty[par][wv] = dirty ? 1u : 0u;
        XY_PHASE(1);
        __syncthreads();  // (also frees S and the strips of the previous super-tile)
        XY_PHASE(2);
        const uint32_t dm16 = (uint32_t)__ballot(s_dirty[par][lane & (kXWaves - 1)] != 0u) & ((1u << kXWaves) - 1u);  // wave-uniform
        const uint32_t nd = (uint32_t)__popc(dm16);
        if (nd == 0) continue;
        if (heavy_cur && threadIdx.x == 0) {   // busy: among the first to be handed out next time, the busiest before the others
            atomicOr(&heavy_cur[sup >> 6], 1ull << (sup & 63));
            if (nd >= (uint32_t)kXWaves / 2) atomicOr(&heavy_cur[kXHeavyWords + (sup >> 6)], 1ull << (sup & 63));
        }
        const uint32_t g4 = (wv & 3) * kXRows;  // this wave's rows within a tile
        uint32_t px[2][kXRows], cur[2][kXRows];
        // pixel (lane, row j) of the unit: x = first column of the tile + lane, y = first row of the tile + g4 + j
        auto load_unit = [&](uint32_t slot, uint32_t (&p)[kXRows], uint32_t (&c)[kXRows]) {
            const uint32_t ux = (stx + (slot & (kSTX - 1))) * kTW + lane, uy0 = (sty + slot / kSTX) * kTH + g4;
#if CNIIC_XY_BUFLOADS
            // (round 4) buffer loads: a pixel outside the image gets an offset outside the buffer and reads as 0 -- no exec mask per row -- and
            // a pixel's 32-bit index is its offset (N <= 2^28).  The image's last pixel, whose dword would end a byte behind the buffer, reads
            // the dword a byte earlier and shifts.
            const uint32_t i0 = uy0 * w + ux;
#pragma unroll
            for (int j = 0; j < kXRows; j++) {
                const uint32_t idx = i0 + (uint32_t)j * w;
                const bool in = ux < w && uy0 + j < h, last = idx == (uint32_t)npix - 1u;
                const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_rgb, (int)(in ? 3u * idx - (last ? 1u : 0u) : 0xffffffffu), 0, 0);
                p[j] = key_from_le24(last ? v >> 8 : v);
                c[j] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_lab, (int)(in ? 2u * idx : 0xffffffffu), 0, 0);
            }
#else
#pragma unroll
            for (int j = 0; j < kXRows; j++) {
                p[j] = 0; c[j] = 0;
                if (ux < w && uy0 + j < h) {
                    const uint64_t idx = (uint64_t)(uy0 + j) * w + ux;
                    p[j] = rgb_key_at(rgb, idx, npix);
                    c[j] = labels[idx];
                }
            }
#endif
        };
        // the first unit's pixels and labels are requested now and arrive while the block builds S
        if ((wv >> 2) < nd) load_unit(nth_set_bit(dm16, wv >> 2), px[0], cur[0]);

        // ---- super-tile list S (whole block): pivot = centroid nearest the super-tile's box centre
        bool s_over = brute != 0;
        uint32_t nSl = 0;
        if (!brute) {
            Box5 sb;
            sb.lo[0] = (int32_t)(stx * kTW); sb.hi[0] = (int32_t)min(w, (stx + kSTX) * kTW) - 1;
            sb.lo[1] = (int32_t)(sty * kTH); sb.hi[1] = (int32_t)min(h, (sty + kSTY) * kTH) - 1;
            box_colours(sb, ts.sbox[sup]);
            // The pivot: the centroid nearest the box centre -- searched for every eighth launch; in between the pivot of the last build
            // serves (ANY centroid gives a correct list; one that was nearest a few launches ago is still near: centroids creep), which
            // saves the search, its reduction and a block barrier per super-tile (round 4).
            uint32_t pk = prev_piv;
            bool reuse = prev_piv < K && fz.on && (fz.launch_no & 7u) != 0u;   // (block-uniform)
            if (reuse) {   // ... unless it has left the box's neighbourhood (a reseeded centroid lands anywhere, and a far pivot dominates nothing)
                uint32_t r2 = 0;
#pragma unroll
                for (int i = 0; i < 5; i++) { const int32_t hd = (sb.hi[i] - sb.lo[i] + 1) >> 1; r2 += (uint32_t)(hd * hd); }
                reuse = centre_dist(sb, tab[prev_piv]) <= 2u * r2;
            }
            if (!reuse) {
                unsigned long long key = ~0ull;
#pragma unroll
                for (int i = 0; i < kXMaxR; i++) {
                    const uint32_t k = threadIdx.x * R + i;
                    if ((uint32_t)i < R && k < K) key = min(key, ((unsigned long long)centre_dist(sb, mine[i]) << 12) | k);
                }
                key = wave_reduce_min64(key);
                if (lane == 0) atomicMin(&s_key, key);
                __syncthreads();
                pk = (uint32_t)(s_key & 4095ull);
                if (threadIdx.x == 0) ts.spiv[sup] = pk;
            }
            Dominance dm;
            dm.set(sb, use_tab ? tab[pk] : cent[pk]);
            bool keep[kXMaxR];
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < kXMaxR; i++) {
                const uint32_t k = threadIdx.x * R + i;
                keep[i] = (uint32_t)i < R && k < K && dm.worst(mine[i]) >= 0;
                cnt += keep[i];
            }
            uint32_t off = block_exclusive_scan<kXThreads>(cnt, wsum);  // (two barriers: every wave has read s_key)
            if (threadIdx.x == kXThreads - 1) { s_n = off + cnt; s_key = ~0ull; }
#pragma unroll
            for (int i = 0; i < kXMaxR; i++)
                if (keep[i]) {
                    if (off < kSCap) { S_c[off] = mine[i]; S_k[off] = (uint16_t)(threadIdx.x * R + i); }
                    off++;
                }
            __syncthreads();
            nSl = s_n;
            s_over = nSl > kSCap;
        }
        XY_PHASE(3);
        // ---- candidate strip of this wave's own tile, if dirty: pivot = member of S nearest the tile's box centre,
        // then the dominance test over S
        if (dirty) XY_COUNT(7, 1);
        if (dirty && !s_over) {
            uint32_t bd = 0xffffffffu, be = 0;
            for (uint32_t e = lane; e < nSl; e += 64) {
                const uint32_t d = centre_dist(tb, S_c[e]);
                if (d < bd) { bd = d; be = e; }
            }
            const uint32_t dmin = wave_reduce_min(bd);
            const uint32_t pe = wave_reduce_min(bd == dmin ? be : 0xffffffffu);
            int4 pv = S_c[pe];
            Dominance dm;
            dm.set(tb, pv);
            for (uint32_t i = lane; i < MW; i += 64) my_mask[i] = 0ull;
            __builtin_amdgcn_wave_barrier();
            uint32_t n = 0;
            for (uint32_t e0 = 0; e0 < nSl; e0 += 64) {
                const uint32_t e = e0 + lane;
                int4 c = make_int4(0, 0, 0, 0);
                bool kp = false;
                if (e < nSl) { c = S_c[e]; kp = dm.worst(c) >= 0; }
                const unsigned long long bm = __ballot(kp);
                if (kp) {
                    const uint32_t pos = n + (uint32_t)__popcll(bm & lt_mask);
                    const uint32_t k = S_k[e];
                    if (pos < wcap) { my_c[pos] = c; my_k[pos] = (uint16_t)k; }
                    atomicOr(&my_mask[k >> 6], 1ull << (k & 63));
                }
                n += (uint32_t)__popcll(bm);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) s_ncand[wv] = n <= wcap ? n : (0x80000000u | nSl);  // (beyond the strip: S, a superset in the same order)
            pv.w = (int32_t)S_k[pe];
            if (lane == 0) ts.piv[tile] = pv;
            for (uint32_t i = lane; i < MW; i += 64) ts.mask[(size_t)tile * MW + i] = my_mask[i];
        } else if (dirty) {
            if (lane == 0) ts.piv[tile] = use_tab ? tab[0] : cent[0];   // every centroid is a candidate:
            for (uint32_t i = lane; i < MW; i += 64) ts.mask[(size_t)tile * MW + i] = ~0ull;  // any move makes the tile dirty
        }

        // the union of the tiles' masks, for the super-tile filter of later launches: a dirty tile's new mask, a clean tile's old one
        if (has_tile && lane < MW) {
            const unsigned long long v = !dirty ? mword : s_over ? ~0ull : my_mask[lane];
            if (v) atomicOr(&s_smask[lane], v);
        }
        XY_PHASE(4);
        __syncthreads();  // every dirty tile's strip is in LDS
        XY_PHASE(2);
        if (wv == 0 && lane < MW) ts.smask[(size_t)sup * MW + lane] = s_smask[lane];

        // ---- assign, one unit at a time; the next unit's pixels and labels are in flight meanwhile
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t ui = (wv >> 2) + kXUS * r;
            if (ui >= nd) break;  // wave-uniform
            const uint32_t (&p)[kXRows] = px[r & 1];
            const uint32_t (&c)[kXRows] = cur[r & 1];
            const uint32_t slot = nth_set_bit(dm16, ui);
#ifdef CNIIC_XY_PHASES
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            XY_PHASE(8);
#endif
            if (ui + kXUS < nd) load_unit(nth_set_bit(dm16, ui + kXUS), px[(r + 1) & 1], cur[(r + 1) & 1]);
            const uint32_t ux0 = (stx + (slot & (kSTX - 1))) * kTW, uy0 = (sty + slot / kSTX) * kTH + g4;
            if (uy0 >= h) continue;  // wave-uniform: the tile ends above this row group
            const uint32_t nrows = min((uint32_t)kXRows, h - uy0), ucols = min((uint32_t)kTW, w - ux0);
            const int32_t x = (int32_t)(ux0 + lane);
            const bool okx = lane < ucols;
            const uint32_t enc = s_ncand[slot];
            const uint32_t ncand = s_over ? K : (enc & 0x7fffffffu);
            const int4 *list_c = (enc >> 31) ? S_c : W_c + (size_t)slot * wcap;
            const uint16_t *list_k = (enc >> 31) ? S_k : W_k + (size_t)slot * wcap;
            int4 own[kXRows];  // the centroid each pixel belongs to now
#pragma unroll
            for (int j = 0; j < kXRows; j++) own[j] = use_tab ? tab[c[j]] : cent[c[j]];
            int32_t best[kXRows];
            uint32_t bpos[kXRows];
#pragma unroll
            for (int j = 0; j < kXRows; j++) { best[j] = INT32_MIN; bpos[j] = 0; }
            // maximise 2 p.c - |c|^2; ascending k and a strict compare: first maximum = lowest id
            if (!s_over) {
                const int32_t x2 = 2 * x, y20 = 2 * (int32_t)uy0;
                const uint32_t ncu = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncand);  // (wave-uniform, which the compiler cannot see: a scalar loop counter instead of an exec-mask loop)
#pragma unroll 2   // (two candidates' reads together; costs two spilled registers and is still 0.5 % faster than one at a time)
                for (uint32_t q = 0; q < ncu; q++) {
                    const int4 cc = list_c[q];  // LDS broadcast
                    // 2 (x cx + y cy) - |c|^2 for the unit's first row; each further row adds 2 cy
                    int32_t t = xmad24(x2, cc.x, xmad24(y20, cc.y, cc.w));
                    const int32_t cy2 = cc.y << 1;
#pragma unroll
                    for (int j = 0; j < kXRows; j++) {
                        const int32_t gq = (int32_t)((xdot4(p[j], (uint32_t)cc.z) << 1) + (uint32_t)t);
                        if (gq > best[j]) { best[j] = gq; bpos[j] = q; }
                        t += cy2;
                    }
                }
            } else {
                for (uint32_t q = 0; q < K; q++) {  // q wave-uniform: scalar loads (an LDS broadcast with the table there)
                    const int4 cc = use_tab ? tab[q] : cent[q];
                    const int32_t ax = xmad24(2 * x, cc.x, cc.w);
#pragma unroll
                    for (int j = 0; j < kXRows; j++) {
                        const int32_t y2 = 2 * (int32_t)(uy0 + j);
                        const int32_t gq = (int32_t)(xdot4(p[j], (uint32_t)cc.z) << 1) + xmad24(y2, cc.y, ax);
                        if (gq > best[j]) { best[j] = gq; bpos[j] = q; }
                    }
                }
            }
#ifdef CNIIC_XY_PHASES
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            XY_PHASE(9);
#endif
#pragma unroll
            for (int j = 0; j < kXRows; j++) {
                const bool ok = okx && (uint32_t)j < nrows;
                const int32_t y = (int32_t)(uy0 + j);
                const int32_t gcur = (int32_t)(xdot4(p[j], (uint32_t)own[j].z) << 1) +
                                     xmad24(2 * y, own[j].y, xmad24(2 * x, own[j].x, own[j].w));
                const bool mv = ok && best[j] > gcur;  // strictly closer (kmeans.rs:375)
                uint32_t nl = c[j];
                if (mv) {
                    nl = s_over ? bpos[j] : (uint32_t)list_k[bpos[j]];
                    labels[(uint64_t)y * w + (uint32_t)x] = (uint16_t)nl;
                    moved++;
                }
                const uint32_t r8 = (p[j] >> 16) & 255, gg = (p[j] >> 8) & 255, b = p[j] & 255;
                // A row of 64 pixels holds few distinct labels: one wave reduction per label instead of 64 colliding LDS atomics
                // (sgn: 0 adds the pixels, ~0 subtracts them -- the sums are u32 that wrap).
                auto book_row = [&](bool sel, uint32_t lab, uint32_t sgn) {
                    unsigned long long todo = __ballot(sel);
                    while (todo) {
                        const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)lab, __ffsll((long long)todo) - 1);
                        const bool in = sel && lab == lk;
                        const unsigned long long grp = __ballot(in);
                        const uint32_t cn = (uint32_t)__popcll(grp);
                        const uint32_t sx = wave_reduce_sum(in ? (uint32_t)x : 0u), sr = wave_reduce_sum(in ? r8 : 0u);
                        const uint32_t sg = wave_reduce_sum(in ? gg : 0u), sbb = wave_reduce_sum(in ? b : 0u);
                        if (lane == 0) {
                            uint32_t *a = acc + 6 * lk;
                            atomicAdd(a + 0, (sx ^ sgn) - sgn); atomicAdd(a + 1, (((uint32_t)y * cn) ^ sgn) - sgn); atomicAdd(a + 2, (sr ^ sgn) - sgn);
                            atomicAdd(a + 3, (sg ^ sgn) - sgn); atomicAdd(a + 4, (sbb ^ sgn) - sgn); atomicAdd(a + 5, (cn ^ sgn) - sgn);
                        }
                        todo &= ~grp;
                    }
                };
                if (first) {
                    book_row(ok, nl, 0u);  // vector_add clusterc.rs:221-228 for every pixel
                } else if (GROUP && __popcll(__ballot(mv)) >= (int)kXGroupMin) {
                    // the launches after the first, where most pixels change hands again (16.1 M, 6.6 M, 2.5 M of 16.8 M in launches
                    // 1-3 at configs[2]): twelve atomics per mover, 64 movers a row on the same few words, kept every CU's LDS busy
                    // for 0.36 ms in launch 1
                    book_row(mv, nl, 0u);
                    book_row(mv, c[j], ~0u);
                } else if (mv) {  // +pixel to its new cluster, -pixel from its old one
                    uint32_t *a = acc + 6 * nl, *o = acc + 6 * c[j];
                    atomicAdd(a + 0, (uint32_t)x); atomicAdd(a + 1, (uint32_t)y); atomicAdd(a + 2, r8);
                    atomicAdd(a + 3, gg); atomicAdd(a + 4, b); atomicAdd(a + 5, 1u);
                    atomicAdd(o + 0, 0u - (uint32_t)x); atomicAdd(o + 1, 0u - (uint32_t)y); atomicAdd(o + 2, 0u - r8);
                    atomicAdd(o + 3, 0u - gg); atomicAdd(o + 4, 0u - b); atomicAdd(o + 5, 0u - 1u);
                }
            }
            if (lane == 0) evals += (unsigned long long)(ncand + 1) * ucols * nrows;
        }
        XY_PHASE(5);
    }
#ifdef CNIIC_XY_PHASES
    if (tl_on_) { tl_[3] = wall_clock64(); tl_[5] = sit; }
#endif
    __syncthreads();
    XY_PHASE(2);
    // the host sizes the grid so that one block's pixels * max coordinate stays below 2^31: one flush at the end
    for (uint32_t i = threadIdx.x; i < 6 * K; i += kXThreads) {
        const uint32_t v = acc[i];
        if (v) {
            const uint32_t k = i / 6, d = i % 6;
            atomicAdd(&partials[fz.on ? (size_t)d * K + k : (d < 5 ? 5 * (size_t)k + d : 5 * (size_t)K + k)], (unsigned long long)(long long)(int32_t)v);   // (fused loop: [6][K])
        }
    }
    moved = block_reduce_sum<kXThreads>(moved);
    // ONE atomic per block and counter: atomics on one address are served ~26 ns apart, and an addition per wave (4096 a
    // launch) kept every launch busy for ~0.1 ms whatever else it did
    if (lane == 0 && evals) atomicAdd(&s_evals, evals);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (moved) atomicAdd(&partials[6 * (size_t)K], (unsigned long long)moved);
        if (s_evals) atomicAdd(&partials[6 * (size_t)K + 1], s_evals);
    }
    XY_PHASE(6);
#ifdef CNIIC_XY_PHASES
    if (tl_on_) { tl_[4] = wall_clock64(); for (int i = 0; i < 8; i++) g_xy_tl[blockIdx.x][i] = tl_[i]; }
    if (lane == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&s_ph[i], ph_[i]);
    __syncthreads();
    if (threadIdx.x < 12) atomicAdd(&g_xy_phase[threadIdx.x], s_ph[threadIdx.x]);
#endif
}

// Point::mean for ColorPos (clusterc.rs:215-247) + empty-cluster reseed (kmeans.rs:110-137).
// A cluster whose partials are all zero keeps members and sums, hence its centroid: no division.
__global__ __launch_bounds__(1024) void k_xy_update(unsigned long long *__restrict__ partials,
                                                    unsigned long long *__restrict__ running,
                                                    const uint8_t *__restrict__ rgb, uint32_t w, uint64_t N,
                                                    uint32_t K, uint64_t seed, uint64_t max_iters, int4 *__restrict__ cent,
                                                    uint64_t *__restrict__ members_out, uint32_t *__restrict__ moved_list,
                                                    KmDevState *__restrict__ st) {
    // everything a cluster needs is requested before anything is looked at: one memory round trip per launch
    const uint32_t done = st->done;
    const uint64_t iter = st->iter;
    const unsigned long long changed = partials[6 * (size_t)K];
    const unsigned long long evals = partials[6 * (size_t)K + 1];
    __shared__ uint32_t s_reseed, s_active, s_nmoved;
    if (threadIdx.x == 0) { s_reseed = 0; s_active = 0; s_nmoved = 0; }
    __syncthreads();
    if (done) return;
    for (uint32_t k = threadIdx.x; k < K; k += blockDim.x) {
        unsigned long long d[6], r[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const size_t at = i < 5 ? 5 * (size_t)k + i : 5 * (size_t)K + k;
            d[i] = partials[at];
            r[i] = running[at];
        }
        const int4 oc = cent[k];
        bool any = false;
#pragma unroll
        for (int i = 0; i < 6; i++) any |= d[i] != 0;
        if (any) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const size_t at = i < 5 ? 5 * (size_t)k + i : 5 * (size_t)K + k;
                r[i] += d[i];
                running[at] = r[i];
                partials[at] = 0;
            }
        }
        const unsigned long long m = r[5];
        members_out[k] = m;
        if (m == 0) {  // reseeded every iteration it stays empty (the index depends on iter)
            const uint64_t idx = reseed_index(seed, iter, k, N);  // fake_clone of the stolen pixel
            const int4 nc = make_cent((int32_t)(idx % w), (int32_t)(idx / w), rgb_key(rgb + 3 * idx));
            if (oc.x != nc.x || oc.y != nc.y || oc.z != nc.z) moved_list[1 + atomicAdd(&s_nmoved, 1u)] = k;
            cent[k] = nc;
            atomicAdd(&s_reseed, 1u);
        } else {
            atomicAdd(&s_active, 1u);
            if (any) {
                // floor(sum / m) with sum < 2^42 and m < 2^28: a double quotient is within one of it
                uint32_t q[5];
                const uint32_t m32 = (uint32_t)m;
                const float rm = 1.0f / (float)m32;
#pragma unroll
                for (int i = 0; i < 5; i++) q[i] = xy_div_floor(r[i], m32, rm);
                const int4 nc = make_cent((int32_t)q[0], (int32_t)q[1], ((q[2] & 255) << 16) | ((q[3] & 255) << 8) | (q[4] & 255));
                if (oc.x != nc.x || oc.y != nc.y || oc.z != nc.z) moved_list[1 + atomicAdd(&s_nmoved, 1u)] = k;
                cent[k] = nc;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[6 * (size_t)K] = 0;
        partials[6 * (size_t)K + 1] = 0;
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
    s.super_x = (uint32_t)ceil_div(s.tiles_x, kSTX);
    s.super_y = (uint32_t)ceil_div(s.tiles_y, kSTY);
    const uint32_t ntiles = s.tiles_x * s.tiles_y, nsuper = s.super_x * s.super_y;
    // One 16-wave block per CU (the LDS accumulators allow no more).  The accumulators are signed 32-bit and
    // flushed once, so (pixels one block sees) * max coordinate must stay below 2^31; a larger image gets
    // more blocks than CUs and the surplus queues behind the resident ones.
    const uint64_t per_block_max = std::max<uint64_t>(((1ull << 31) - 1) / ((uint64_t)std::max(w, h) * kSuperPx), 1);
    s.nblocks = (uint32_t)std::max<uint64_t>(std::min<uint32_t>(nsuper, 256 * 16 / kXWaves), ceil_div(nsuper, per_block_max));
    s.sup_cap = (uint32_t)std::min<uint64_t>(per_block_max, 0x7fffffffull);
    const uint32_t MW = (K + 63) / 64;
    // LDS budget (153 KiB dynamic): accumulators, S, the moved list, the masks, then the centroid table if it fits next
    // to at least 64 candidates per wave, and the per-wave strips take what is left (up to 256 entries each)
    const size_t lds_max = (kSTY == 4 ? 153 : 73) * 1024;  // (160 KiB less the kernel's static arrays: 6.2 KiB with the super-tile filter's lists and the order of issue)
    size_t fixed = (size_t)xy_acc_words(K) * 4 + (size_t)kSCap * 18 + (size_t)kXMaxMovedSkip * 16 + (size_t)kXWaves * MW * 8;
    s.use_tab = fixed + (size_t)K * 16 + (size_t)kXWaves * 64 * 18 <= lds_max;
    if (s.use_tab) fixed += (size_t)K * 16;
    s.wcap = (uint32_t)std::min<size_t>(256, (lds_max - fixed) / ((size_t)kXWaves * 18) / 32 * 32);
    s.lds = fixed + (size_t)kXWaves * s.wcap * 18;
    CNIIC_HIP_TRY(c, s.labels.alloc(N * 2));
    CNIIC_HIP_TRY(c, s.cent.alloc((uint64_t)K * 16));
    CNIIC_HIP_TRY(c, s.partials.alloc((6 * (uint64_t)K + 2) * 8));
    CNIIC_HIP_TRY(c, s.members_last.alloc((uint64_t)K * 8));
    CNIIC_HIP_TRY(c, s.dstate.alloc(sizeof(KmDevState)));
    CNIIC_HIP_TRY(c, s.running.alloc((6 * (uint64_t)K + 2) * 8));
    CNIIC_HIP_TRY(c, s.tile_box.alloc((uint64_t)ntiles * 8));
    CNIIC_HIP_TRY(c, s.super_box.alloc((uint64_t)nsuper * 8));
    CNIIC_HIP_TRY(c, s.tile_piv.alloc((uint64_t)ntiles * 16));
    CNIIC_HIP_TRY(c, s.tile_mask.alloc((uint64_t)ntiles * MW * 8));
    CNIIC_HIP_TRY(c, s.moved_list.alloc(((uint64_t)K + 1) * 4));
    CNIIC_HIP_TRY(c, s.sup_piv.alloc((uint64_t)nsuper * 4));
    CNIIC_HIP_TRY(c, s.sup_mask.alloc((uint64_t)nsuper * MW * 8));
    CNIIC_HIP_TRY(c, hipMemsetAsync(s.sup_piv.p, 0xff, (uint64_t)nsuper * 4, c->stream));
    CNIIC_HIP_TRY(c, hipMemsetAsync(s.sup_mask.p, 0xff, (uint64_t)nsuper * MW * 8, c->stream));
    s.no_skip = opts && (opts->flags & CNIIC_KM_NO_SKIP);
    s.fused = s.use_tab && !s.brute && !(test_env("CNIIC_XY_UNFUSED") && atoi(test_env("CNIIC_XY_UNFUSED")));
    if (const char *e = test_env("CNIIC_XY_DYN")) s.dyn = (uint32_t)atoi(e);
    if (s.fused) {
        const uint64_t Wb = (6 * (uint64_t)K + 4 + 2 * kXHeavyWords) * 8;   // (+ the launch's super-tile counter, a word of padding, the two bitmaps of its busy super-tiles)
        CNIIC_HIP_TRY(c, s.f_partials.alloc(3 * Wb));
        CNIIC_HIP_TRY(c, s.f_running.alloc(2 * Wb));
        CNIIC_HIP_TRY(c, s.f_cent.alloc(2 * (uint64_t)K * 16));
        CNIIC_HIP_TRY(c, hipMemsetAsync(s.f_partials.p, 0, 3 * Wb, c->stream));
        CNIIC_HIP_TRY(c, hipMemsetAsync(s.f_running.p, 0, 2 * Wb, c->stream));
    }
    CNIIC_HIP_TRY(c, hipMemsetAsync(s.partials.p, 0, (6 * (uint64_t)K + 2) * 8, c->stream));
    CNIIC_HIP_TRY(c, hipMemsetAsync(s.running.p, 0, (6 * (uint64_t)K + 2) * 8, c->stream));
    CNIIC_HIP_TRY(c, hipMemsetAsync(s.dstate.p, 0, sizeof(KmDevState), c->stream));
    const uint32_t all = K;  // before the first update every centroid counts as moved
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s.moved_list.p, &all, 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_xy_boxes, dim3((uint32_t)ceil_div(ntiles, 4)), dim3(256), 0, c->stream, rgb_d, w, h, s.tiles_x, ntiles,
                       s.tile_box.as<uint2>());
    hipLaunchKernelGGL(k_xy_super_boxes, dim3((uint32_t)ceil_div(nsuper, 256)), dim3(256), 0, c->stream, s.tile_box.as<uint2>(),
                       s.tiles_x, s.tiles_y, s.super_x, nsuper, s.super_box.as<uint2>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    // the assign kernel carves up to 153 KiB of the CU's 160 KiB LDS
    CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_xy_assign<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         153 * 1024));
    CNIIC_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_xy_assign<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         153 * 1024));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

static int xy_assign(KmXyState &s, bool fused = false) {
    Ctx *c = s.c;
    XyFused fz{};
    unsigned long long *part = s.partials.as<unsigned long long>();
    if (fused) {
        const uint64_t W = 6 * (uint64_t)s.K + 4 + 2 * kXHeavyWords;
        const uint32_t j = s.launch_no++;
        auto *P = s.f_partials.as<unsigned long long>();
        auto *Rn = s.f_running.as<unsigned long long>();
        int4 *Cn = s.f_cent.as<int4>();
        fz.on = 1; fz.launch_no = j; fz.seed = s.seed; fz.max_iters = s.max_iters; fz.N = s.N;
        fz.partials_prev = P + ((j + 2) % 3) * W;
        fz.partials_clear = P + ((j + 1) % 3) * W;
        fz.running_prev = Rn + ((j + 1) % 2) * W;
        fz.running_new = Rn + (j % 2) * W;
        fz.cent_prev = Cn + ((j + 1) % 2) * (size_t)s.K;
        fz.cent_new = Cn + (j % 2) * (size_t)s.K;
        fz.cent_g = s.cent.as<int4>();
        fz.members_out = s.members_last.as<uint64_t>();
        fz.st_rw = s.dstate.as<KmDevState>();
        part = P + (j % 3) * W;
    }
    // (launch 0 of the fused loop reads the initial centroids from s.cent; every later one computes its table from the sums)
    const bool group = fused && fz.launch_no >= 1 && fz.launch_no <= kXGroupLaunches;
    hipLaunchKernelGGL(group ? k_xy_assign<true> : k_xy_assign<false>, dim3(s.nblocks), dim3(kXThreads), s.lds, c->stream, s.rgb, s.w, s.h, s.tiles_x, s.tiles_y,
                       s.super_x, s.super_x * s.super_y, s.K, s.cent.as<int4>(), s.labels.as<uint16_t>(),
                       part, s.dstate.as<KmDevState>(), s.wcap, s.use_tab ? 1 : 0, s.brute ? 1 : 0,
                       TileState{s.tile_box.as<uint2>(), s.super_box.as<uint2>(), s.tile_piv.as<int4>(), s.tile_mask.as<unsigned long long>(),
                                 s.moved_list.as<uint32_t>(), s.no_skip ? 0u : kXMaxMovedSkip, s.sup_piv.as<uint32_t>(),
                                 s.sup_mask.as<unsigned long long>(), s.dyn, s.sup_cap,
                                 test_env("CNIIC_XY_TL_LAUNCH") ? (uint32_t)atoi(test_env("CNIIC_XY_TL_LAUNCH")) + 1u : 0u}, fz);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

static int xy_update(KmXyState &s) {
    Ctx *c = s.c;
    hipLaunchKernelGGL(k_xy_update, dim3(1), dim3(1024), 0, c->stream, s.partials.as<unsigned long long>(),
                       s.running.as<unsigned long long>(), s.rgb, s.w, s.N, s.K, s.seed, s.max_iters, s.cent.as<int4>(),
                       s.members_last.as<uint64_t>(), s.moved_list.as<uint32_t>(), s.dstate.as<KmDevState>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

static uint32_t xy_grid(uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, 256), 1), 4096); }

int km_xyrgb_run(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K, const cniic_kmeans_opts *opts,
                 cniic_colorpos *centroids_h, uint32_t *labels_d_u32, uint64_t *members_h, cniic_kmeans_stats *stats) {
    // beyond what the tiled kernel is laid out for (table and sums in LDS, 24-bit coordinate products): the exact, slow route
    if (K > kXMaxK || w > 16384 || h > 16384) return km_xyrgb_run_wide(c, rgb_d, w, h, K, opts, centroids_h, labels_d_u32, members_h, stats);
    KmXyState s;
    CNIIC_TRY(xy_create(c, rgb_d, w, h, K, opts, s));
    hipLaunchKernelGGL(k_xy_init, dim3(xy_grid(std::max<uint64_t>(s.N, K))), dim3(256), 0, c->stream, rgb_d, w, s.N, K,
                       s.labels.as<uint16_t>(), s.cent.as<int4>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    KmDevState hst;
    LaggedPoll poll(c, s.dstate.p);
    CNIIC_TRY(poll.prepare());
    ScopedKernelTimer timer(c, "kmeans_xyrgb_iter", opts && (opts->flags & CNIIC_KM_PROFILE));  // (stop() synchronises)
    if (s.fused)  // the centroids launch 0 assigns with are what launch 1 compares its new ones with (cent_prev of launch 1 = buffer 0)
        CNIIC_HIP_TRY(c, hipMemcpyAsync(s.f_cent.p, s.cent.p, (size_t)K * 16, hipMemcpyDeviceToDevice, c->stream));
    for (;;) {
        for (int b = 0; b < 4; b++) {
            if (s.fused) { CNIIC_TRY(xy_assign(s, true)); continue; }
            CNIIC_TRY(xy_assign(s));
            CNIIC_TRY(xy_update(s));
        }
        bool have = false;
        CNIIC_TRY(poll.after_batch(&hst, &have));
        if (have && hst.done) break;
    }
    timer.stop(hst.iter);
#ifdef CNIIC_XY_PHASES
    {
        unsigned long long ph[12], zero[12] = {0};
        CNIIC_HIP_TRY(c, hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_xy_phase), sizeof ph));
        CNIIC_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_xy_phase), zero, sizeof zero));
        if (test_env("CNIIC_XY_TL_LAUNCH")) {
            static unsigned long long tl[256][8];
            CNIIC_HIP_TRY(c, hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_xy_tl), sizeof tl));
            unsigned long long t0 = ~0ull;
            for (uint32_t b = 0; b < s.nblocks && b < 256; b++) if (tl[b][0]) t0 = std::min(t0, tl[b][0]);
            const char *names[5] = {"entry", "set-up loads back", "prologue done", "super-tile loop done", "flush done"};
            for (int q = 0; q < 5; q++) {
                std::vector<double> v;
                for (uint32_t b = 0; b < s.nblocks && b < 256; b++) if (tl[b][q]) v.push_back((double)(tl[b][q] - t0) / 100.0);
                std::sort(v.begin(), v.end());
                if (!v.empty()) fprintf(stderr, "xy timeline launch %s: %-22s min %7.2f p50 %7.2f p90 %7.2f max %7.2f us (%zu blocks)\n", test_env("CNIIC_XY_TL_LAUNCH"), names[q], v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back(), v.size());
            }
        }
        fprintf(stderr, "xy phases (wave clocks): prologue %llu skiptest %llu barrier %llu S %llu tile %llu eval %llu epilogue %llu | dirty tiles %llu iters %llu | eval: loadwait %llu candloop %llu\n",
                ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], (unsigned long long)hst.iter, ph[8], ph[9]);
    }
#endif
    std::vector<int4> cent(K);
    CNIIC_HIP_TRY(c, hipMemcpy(cent.data(), s.cent.p, (size_t)K * 16, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t col = (uint32_t)cent[k].z;
        centroids_h[k].x = (uint32_t)cent[k].x; centroids_h[k].y = (uint32_t)cent[k].y;
        centroids_h[k].rgb[0] = (uint8_t)(col >> 16); centroids_h[k].rgb[1] = (uint8_t)(col >> 8);
        centroids_h[k].rgb[2] = (uint8_t)col; centroids_h[k].pad = 0;
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
    std::vector<int4> cent(K);
    for (uint32_t k = 0; k < K; k++) {
        const int32_t cx = (int32_t)centroids_h[k].x, cy = (int32_t)centroids_h[k].y;
        const uint8_t *q = centroids_h[k].rgb;
        cent[k] = make_int4(cx, cy, (int32_t)(((uint32_t)q[0] << 16) | ((uint32_t)q[1] << 8) | q[2]),
                            -(cx * cx + cy * cy + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]));
    }
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s.cent.p, cent.data(), (size_t)K * 16, hipMemcpyHostToDevice, c->stream));
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
