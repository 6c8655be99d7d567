// k_hilbert.hip -- Hilbert-order traversal + neighbour delta on gfx950
// (reference: src/hilbert.rs:34-43 iter/linearize, src/codec/hilbertc.rs:445-477 DiffStream,
//  hilbertc.rs:417-431 Delta::decode scatter).
//
// The reference walks an iterator (zhang_hilbert::ArbHilbertScan32) one cell at a time.  Here the
// scan is a pure function d -> (x, y), so the index map, the pixel gather, the delta and the
// histogram are all data-parallel.  The scan definition is the one frozen in DESIGN.md ("Hilbert
// scan: parity unpinned"): the generalised Hilbert curve, identical to the classic Hilbert curve
// on 2^n squares.
//   * 2^n x 2^n images (all BASELINE configs): table-driven state machine, 4 levels (8 bits of d)
//     per LDS look-up: the four orientations of a Hilbert sub-curve form the Klein group
//     {id, transpose, anti-transpose, rot180}, so a state is 2 bits and composition is XOR.
//   * any other rectangle: each thread descends the recursion of the generalised curve.
//
// Fused delta kernel: the SignedColor histogram (src/utils.rs:4-16 via src/huf.rs:30) is kept in
// LDS-private bins for the hot cube of deltas in [-16,15]^3 (u32[32768] = 128 KiB per 1024-thread
// block), flushed once per block with global atomics; symbols outside the cube go straight to the
// dense 2^27-bin table.  Without this the kernel runs at the global-atomic rate (~27 G/s).
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "device_utils.hpp"
#include "hilbert_scan.hpp"

namespace cniic {

// ---------------------------------------------------------------- 2^n squares: state machine
// One level: quadrant q of a sub-curve in state s sits at block (x,y) and continues in state s'.
// Base state 0 (enter at (0,0), leave at (n-1,0)): q0 -> (0,0) transposed, q1 -> (0,1), q2 -> (1,1),
// q3 -> (1,0) anti-transposed.  States: 0 id, 1 transpose, 2 anti-transpose, 3 rot180.
static void lut1_entry(int s, int q, int &x, int &y, int &ns) {
    static const int bx[4] = {0, 0, 1, 1}, by[4] = {0, 1, 1, 0}, bs[4] = {1, 0, 0, 2};
    int X = bx[q], Y = by[q];
    switch (s) {
    case 1: { int t = X; X = Y; Y = t; break; }               // transpose
    case 2: { int t = X; X = 1 - Y; Y = 1 - t; break; }       // anti-transpose
    case 3: X = 1 - X; Y = 1 - Y; break;                       // rot180
    }
    x = X; y = Y; ns = s ^ bs[q];
}

static void host_classic_d2xy(uint32_t n, uint64_t d, uint32_t &x, uint32_t &y) {  // textbook form, for the self-check
    uint64_t t = d;
    x = y = 0;
    for (uint32_t s = 1; s < n; s *= 2) {
        uint32_t rx = 1 & (uint32_t)(t / 2), ry = 1 & (uint32_t)(t ^ rx);
        if (ry == 0) {
            if (rx == 1) { x = s - 1 - x; y = s - 1 - y; }
            uint32_t tmp = x; x = y; y = tmp;
        }
        x += s * rx; y += s * ry;
        t /= 4;
    }
}

static bool build_hilbert_lut(HilbertLut &L) {
    for (int s = 0; s < 4; s++)
        for (int q = 0; q < 4; q++) {
            int x, y, ns;
            lut1_entry(s, q, x, y, ns);
            L.l1[s * 4 + q] = (uint8_t)(x | (y << 1) | (ns << 2));
        }
    for (int s = 0; s < 4; s++)
        for (int b = 0; b < 256; b++) {
            int st = s, X = 0, Y = 0;
            for (int lv = 3; lv >= 0; lv--) {
                int x, y, ns;
                lut1_entry(st, (b >> (2 * lv)) & 3, x, y, ns);
                X = (X << 1) | x; Y = (Y << 1) | y; st = ns;
            }
            L.l4[s * 256 + b] = (uint16_t)(X | (Y << 4) | (st << 8));
        }
    // self-check against the textbook algorithm on every order up to 9 (covers l4, l1 and their mix)
    for (uint32_t order = 1; order <= 9; order++) {
        const uint32_t n = 1u << order;
        for (uint64_t d = 0; d < (uint64_t)n * n; d += (order > 6 ? 37 : 1)) {
            uint32_t st = 0, x = 0, y = 0, rem = order;
            while (rem >= 4) { uint16_t e = L.l4[st * 256 + ((d >> (2 * (rem - 4))) & 255)]; x = (x << 4) | (e & 15); y = (y << 4) | ((e >> 4) & 15); st = e >> 8; rem -= 4; }
            while (rem >= 1) { uint8_t e = L.l1[st * 4 + ((d >> (2 * (rem - 1))) & 3)]; x = (x << 1) | (e & 1); y = (y << 1) | ((e >> 1) & 1); st = e >> 2; rem -= 1; }
            uint32_t cx, cy;
            host_classic_d2xy(n, d, cx, cy);
            if (cx != x || cy != y) return false;
        }
    }
    return true;
}

// hilbert::iter (hilbert.rs:40-43)
__global__ __launch_bounds__(256) void k_hilbert_xy(uint32_t w, uint32_t h, uint32_t order, const HilbertLut *__restrict__ lut,
                                                    uint32_t *__restrict__ xy) {
    __shared__ uint16_t s_l4[1024];
    __shared__ uint8_t s_l1[16];
    const Scan sc = load_scan(w, h, order, lut, s_l4, s_l1);
    const uint64_t n = (uint64_t)w * h;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += stride) {
        uint32_t x, y;
        sc.xy(d, x, y);
        reinterpret_cast<uint2 *>(xy)[d] = make_uint2(x, y);
    }
}

// hilbert::linearize (hilbert.rs:10-12, 34-38): out[d] = pixel(scan(d)); SCATTER = inverse
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_hilbert_move(const uint8_t *__restrict__ src, uint32_t w, uint32_t h, uint32_t order,
                                                      const HilbertLut *__restrict__ lut, uint8_t *__restrict__ dst) {
    __shared__ uint16_t s_l4[1024];
    __shared__ uint8_t s_l1[16];
    const Scan sc = load_scan(w, h, order, lut, s_l4, s_l1);
    const uint64_t n = (uint64_t)w * h;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += stride) {
        uint32_t x, y;
        sc.xy(d, x, y);
        const uint64_t p = (uint64_t)y * w + x;
        const uint32_t v = px_le24(src, SCATTER ? d : p, n);
        uint8_t *o = dst + 3 * (SCATTER ? p : d);
        o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16);
    }
}

// The same along the leaves of a large rectangle (ScanLeavesDev), four consecutive positions per thread: the scan side moves as 12
// contiguous bytes (one load or store instead of twelve byte accesses a wave instruction each), the four table entries are
// neighbours, and the leaf is looked up once unless the four straddle its end.  Scan-side buffer 4-byte aligned.
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_hilbert_move_leaves(const uint8_t *__restrict__ src, uint32_t w, uint32_t h, const ScanLeavesDev *__restrict__ hdr,
                                                             uint8_t *__restrict__ dst) {
    const ScanLeavesDev L = *hdr;
    const uint64_t n = (uint64_t)w * h, nquad = n >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride) {
        const uint32_t d = (uint32_t)(q << 2);
        uint32_t k = L.idx[d >> L.shift];
        while (L.d0[k + 1] <= d) k++;
        int4 r = L.rec[k];
        uint32_t first = L.d0[k], end = L.d0[k + 1];
        uint32_t v[4];
        if (SCATTER) {
            const uint32_t *s3 = reinterpret_cast<const uint32_t *>(src + 3ull * d);
            const uint32_t a = s3[0], b = s3[1], c3 = s3[2];
            v[0] = a; v[1] = (a >> 24) | (b << 8); v[2] = (b >> 16) | (c3 << 16); v[3] = c3 >> 8;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t dj = d + j;
            if (dj >= end) {   // the next leaf (leaves are never empty)
                k++;
                r = L.rec[k]; first = end; end = L.d0[k + 1];
            }
            const uint32_t e = L.lut[(uint32_t)r.z + (dj - first)];
            const uint32_t x = (uint32_t)(r.x + (int32_t)(int16_t)(e & 0xffffu)), y = (uint32_t)(r.y + (int32_t)(int16_t)(e >> 16));
            const uint64_t p = (uint64_t)y * w + x;
            if (SCATTER) {
                store_px3(dst + 3 * p, v[j]);
            } else {
                v[j] = px_le24(src, p, n) & 0xffffffu;
            }
        }
        if (!SCATTER) {
            uint32_t *o3 = reinterpret_cast<uint32_t *>(dst + 3ull * d);
            o3[0] = v[0] | (v[1] << 24); o3[1] = (v[1] >> 8) | (v[2] << 16); o3[2] = (v[2] >> 16) | (v[3] << 8);
        }
    }
    // the last n mod 4 positions, one by one
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const Scan sc{w, h, 0u, nullptr, nullptr, nullptr, L};
        const uint64_t d = (nquad << 2) + threadIdx.x;
        uint32_t x, y;
        sc.xy(d, x, y);
        const uint64_t p = (uint64_t)y * w + x;
        const uint32_t vv = px_le24(src, SCATTER ? d : p, n);
        uint8_t *o = dst + 3 * (SCATTER ? p : d);
        o[0] = (uint8_t)vv; o[1] = (uint8_t)(vv >> 8); o[2] = (uint8_t)(vv >> 16);
    }
}

// The same on 2^n squares of side >= 64 with 16-byte aligned buffers, by 64 x 64 tiles = 4096 consecutive scan positions
// (k_delta_gather_p2's scheme): the image side of a tile is read or written as rows of 48-byte pieces, the scan side as
// 48-byte pieces of 16 positions, and the permutation happens in LDS -- one word per pixel, stored 8 x 8 block by block
// (a block is 64 consecutive positions) with the rows of a block XOR-ed so that neither side's accesses pile up on a bank.
// UNDIFF (with SCATTER): the scan side is not colours but the decoded `delta` symbols (packed SignedColor keys, 4 bytes each),
// turned into colours on the way -- FromDiff (hilbertc.rs:482-509) as a prefix sum: a tile is 4096 consecutive positions = one
// chunk of k_ud_sums, whose exclusive channel sums chunk_off holds; a colour outside 0..255 is reported through *bad.  Saves
// writing and re-reading the linearised image (6 of 17 bytes per pixel) and a launch.
template <bool SCATTER, bool UNDIFF = false>
__global__ __launch_bounds__(256) void k_hilbert_move_p2(const uint8_t *__restrict__ src, uint32_t order, const HilbertLut *__restrict__ lut,
                                                         uint8_t *__restrict__ dst, const int32_t *__restrict__ chunk_off = nullptr,
                                                         uint32_t *__restrict__ bad = nullptr) {
    static_assert(!UNDIFF || SCATTER, "undiff feeds the scatter");
    __shared__ __align__(16) uint32_t s_tile[64 * 64];
    __shared__ int32_t s_wsum[3][4];
    __shared__ uint16_t s_l4[1024];
    __shared__ uint8_t s_l1[16];
    __shared__ uint8_t s_l3[4 * 64];  // three levels from state s for six bits q: x:3 | y:3 << 3 | end state << 6
    const uint32_t w = 1u << order;
    load_scan(w, w, order, lut, s_l4, s_l1);
    {
        uint32_t st = threadIdx.x >> 6, x = 0, y = 0;
        for (int lv = 2; lv >= 0; lv--) {
            const uint32_t e = s_l1[st * 4 + ((threadIdx.x >> (2 * lv)) & 3)];
            x = (x << 1) | (e & 1); y = (y << 1) | ((e >> 1) & 1); st = e >> 2;
        }
        s_l3[threadIdx.x] = (uint8_t)(x | (y << 3) | (st << 6));
    }
    __syncthreads();
    const uint32_t ntiles = (uint32_t)(((uint64_t)w * w) >> 12);
    const uint32_t row = threadIdx.x >> 2, seg = threadIdx.x & 3;
    auto blk_base = [](uint32_t e) { return (e & 63u) << 6; };                                        // from x:3 | y:3 << 3
    auto blk_xor = [](uint32_t e) { return (((e >> 1) & 3u) | (((e >> 3) & 1u) << 2)) << 3; };
    // 16 pixels <-> 12 words (r, g, b bytes in a row; a pixel word is r | g << 8 | b << 16, bits 24..31 anything on the way in)
    auto unpack4 = [](uint32_t a, uint32_t b, uint32_t c) { return make_uint4(a, (a >> 24) | (b << 8), (b >> 16) | (c << 16), c >> 8); };
    auto pack4 = [](uint4 p, uint32_t &a, uint32_t &b, uint32_t &c) {
        a = (p.x & 0xffffffu) | (p.y << 24);
        b = ((p.y >> 8) & 0xffffu) | (p.z << 16);
        c = ((p.z >> 16) & 0xffu) | (p.w << 8);
    };
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t st = 0, tx = 0, ty = 0, rem = order - 6;  // where the tile lies, in which orientation the curve enters it
        while (rem >= 4) {
            const uint32_t e = s_l4[st * 256 + ((tile >> (2 * (rem - 4))) & 255)];
            tx = (tx << 4) | (e & 15); ty = (ty << 4) | ((e >> 4) & 15); st = e >> 8; rem -= 4;
        }
        while (rem >= 1) {
            const uint32_t e = s_l1[st * 4 + ((tile >> (2 * (rem - 1))) & 3)];
            tx = (tx << 1) | (e & 1); ty = (ty << 1) | ((e >> 1) & 1); st = e >> 2; rem -= 1;
        }
        // the image side: row `row` of the tile, pixels 16 seg ..; the scan side: positions 16 t .. of the tile
        uint8_t *const img_piece = const_cast<uint8_t *>(SCATTER ? dst : src) + ((uint64_t)((ty << 6) + row) * w + (tx << 6) + seg * 16) * 3;
        uint8_t *const lin_piece = const_cast<uint8_t *>(SCATTER ? src : dst) + ((uint64_t)tile * 4096 + threadIdx.x * 16) * (UNDIFF ? 4 : 3);
        const uint4 *in = reinterpret_cast<const uint4 *>(SCATTER ? lin_piece : img_piece);
        const uint4 q0 = in[0], q1 = in[1], q2 = in[2];
        const uint32_t q[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
        uint32_t kq[4] = {0, 0, 0, 0};
        int32_t run[3] = {0, 0, 0};
        if (UNDIFF) {  // 16 symbols: three words more; this thread's channel sums, scanned over the block (the tile) below
            const uint4 q3 = in[3];
            kq[0] = q3.x; kq[1] = q3.y; kq[2] = q3.z; kq[3] = q3.w;
            int32_t sm[3] = {0, 0, 0};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t key = i < 12 ? q[i] : kq[i - 12];
                sm[0] += (int32_t)((key >> 18) & 511) - 255; sm[1] += (int32_t)((key >> 9) & 511) - 255; sm[2] += (int32_t)(key & 511) - 255;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const int32_t inc = (int32_t)wave_inclusive_scan((uint32_t)sm[ch]);
                if ((threadIdx.x & 63) == 63) s_wsum[ch][threadIdx.x >> 6] = inc;
                run[ch] = inc - sm[ch];
            }
        }
        // this thread's 16 positions lie in one block: block number threadIdx.x / 4 of the tile's 64, places 16 (t & 3) ..
        const uint32_t eb = s_l3[st * 64 + (threadIdx.x >> 2)];
        const uint32_t yhi = row >> 3, y3 = row & 7;
        __syncthreads();  // the tile before has left LDS
        if (!SCATTER) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t e = (seg * 2 + (g >> 1)) | (yhi << 3);
                *reinterpret_cast<uint4 *>(&s_tile[blk_base(e) + (((y3 << 3) ^ blk_xor(e)) | ((g & 1) << 2))]) = unpack4(q[3 * g], q[3 * g + 1], q[3 * g + 2]);
            }
        } else {
            if (UNDIFF) {
                bool oob = false;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    for (uint32_t i = 0; i < (threadIdx.x >> 6); i++) run[ch] += s_wsum[ch][i];
                    run[ch] += chunk_off[3 * (size_t)tile + ch];
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t key = 4 * g + i < 12 ? q[4 * g + i] : kq[4 * g + i - 12];
                        run[0] += (int32_t)((key >> 18) & 511) - 255; run[1] += (int32_t)((key >> 9) & 511) - 255; run[2] += (int32_t)(key & 511) - 255;
                        oob |= ((uint32_t)run[0] | (uint32_t)run[1] | (uint32_t)run[2]) > 255u;   // (a negative value has its top bits set)
                        const uint32_t place = s_l3[(eb >> 6) * 64 + (threadIdx.x & 3) * 16 + 4 * g + i] & 63u;
                        s_tile[blk_base(eb) + (place ^ blk_xor(eb))] = ((uint32_t)run[0] & 255u) | (((uint32_t)run[1] & 255u) << 8) | (((uint32_t)run[2] & 255u) << 16);
                    }
                }
                if (oob) *bad = 1u;
            } else {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint4 p4 = unpack4(q[3 * g], q[3 * g + 1], q[3 * g + 2]);
                const uint32_t p[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t place = s_l3[(eb >> 6) * 64 + (threadIdx.x & 3) * 16 + 4 * g + i] & 63u;
                    s_tile[blk_base(eb) + (place ^ blk_xor(eb))] = p[i];
                }
            }
            }
        }
        __syncthreads();
        uint32_t o[12];
        if (!SCATTER) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                uint32_t p[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t place = s_l3[(eb >> 6) * 64 + (threadIdx.x & 3) * 16 + 4 * g + i] & 63u;
                    p[i] = s_tile[blk_base(eb) + (place ^ blk_xor(eb))];
                }
                pack4(make_uint4(p[0], p[1], p[2], p[3]), o[3 * g], o[3 * g + 1], o[3 * g + 2]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t e = (seg * 2 + (g >> 1)) | (yhi << 3);
                const uint4 p4 = *reinterpret_cast<const uint4 *>(&s_tile[blk_base(e) + (((y3 << 3) ^ blk_xor(e)) | ((g & 1) << 2))]);
                pack4(p4, o[3 * g], o[3 * g + 1], o[3 * g + 2]);
            }
        }
        uint4 *out = reinterpret_cast<uint4 *>(SCATTER ? img_piece : lin_piece);
        out[0] = make_uint4(o[0], o[1], o[2], o[3]);
        out[1] = make_uint4(o[4], o[5], o[6], o[7]);
        out[2] = make_uint4(o[8], o[9], o[10], o[11]);
    }
}

// DiffStream (hilbertc.rs:449-477) over the Hilbert-ordered pixels, START = [0;3] (hilbertc.rs:445).
// Each thread owns 4 consecutive scan positions (one extra look-up for the predecessor of the
// first).  The packed SignedColor key goes to syms (16-B store per thread) and, when HIST, into the
// histogram: LDS bins for the cube [-16,15]^3, the dense table for everything else.
constexpr int kDeltaRun = 4;
constexpr int kDeltaThreads = 1024;
constexpr uint32_t kHotBins = 32 * 32 * 32;

template <bool HIST>
__global__ __launch_bounds__(kDeltaThreads) void k_hilbert_delta(const uint8_t *__restrict__ rgb, uint32_t w, uint32_t h, uint32_t order,
                                                                 const HilbertLut *__restrict__ lut, uint32_t *__restrict__ syms,
                                                                 uint32_t *__restrict__ table) {
    extern __shared__ uint32_t s_bins[];  // HIST: u32[kHotBins]
    __shared__ uint16_t s_l4[1024];
    __shared__ uint8_t s_l1[16];
    __shared__ uint32_t s_ck[64], s_cc[64];   // the block's cache of keys outside the cube (cold_count)
    if (HIST) {
        for (uint32_t i = threadIdx.x; i < kHotBins; i += kDeltaThreads) s_bins[i] = 0;
        if (threadIdx.x < 64) { s_ck[threadIdx.x] = 0xffffffffu; s_cc[threadIdx.x] = 0; }
    }
    const Scan sc = load_scan(w, h, order, lut, s_l4, s_l1);
    if (HIST && !order) __syncthreads();
    const uint64_t n = (uint64_t)w * h;
    const uint64_t nruns = (n + kDeltaRun - 1) / kDeltaRun;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t run = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; run < nruns; run += stride) {
        const uint64_t d0 = run * kDeltaRun;
        // the run's pixels: one (unaligned) dword each -- three byte loads per pixel were 15 load instructions per thread
        uint32_t px[kDeltaRun];
        ScanCursor cu;
#pragma unroll
        for (int i = 0; i < kDeltaRun; i++) {
            px[i] = 0;
            if (d0 + i < n) {
                uint32_t x, y;
                sc.xy_seq(cu, d0 + i, x, y);
                px[i] = px_le24(rgb, (uint64_t)y * w + x, n);
            }
        }
        // the pixel before the run is the neighbouring lane's last one (runs are consecutive across the lanes of a wave)
        uint32_t prev = wave_prev_lane(px[kDeltaRun - 1], 0u);
        if ((threadIdx.x & 63) == 0) {
            prev = 0;  // START = (0, 0, 0) hilbertc.rs:445
            if (d0 > 0) {
                uint32_t x, y;
                sc.xy(d0 - 1, x, y);
                prev = px_le24(rgb, (uint64_t)y * w + x, n);
            }
        }
        int32_t pr = prev & 255, pg = (prev >> 8) & 255, pb = (prev >> 16) & 255;
        uint32_t key[kDeltaRun];
#pragma unroll
        for (int i = 0; i < kDeltaRun; i++) {
            key[i] = 0;
            if (d0 + i < n) {
                const int32_t r = px[i] & 255, g = (px[i] >> 8) & 255, b = (px[i] >> 16) & 255;
                const int32_t dr = r - pr, dg = g - pg, db = b - pb;
                key[i] = ((uint32_t)(dr + 255) << 18) | ((uint32_t)(dg + 255) << 9) | (uint32_t)(db + 255);
                pr = r; pg = g; pb = b;
                if (HIST) {
                    const uint32_t hr = (uint32_t)(dr + 16), hg = (uint32_t)(dg + 16), hb = (uint32_t)(db + 16);
                    if ((hr | hg | hb) < 32u) atomicAdd(&s_bins[(hr << 10) | (hg << 5) | hb], 1u);
                    else cold_count(table, s_ck, s_cc, key[i]);
                }
            }
        }
        if (syms) {
            if (d0 + kDeltaRun <= n && (reinterpret_cast<uintptr_t>(syms) & 15) == 0)
                reinterpret_cast<uint4 *>(syms)[run] = make_uint4(key[0], key[1], key[2], key[3]);
            else
                for (int i = 0; i < kDeltaRun && d0 + i < n; i++) syms[d0 + i] = key[i];
        }
    }
    if (HIST) {
        __syncthreads();
        cold_flush(table, s_ck, s_cc);
        for (uint32_t i = threadIdx.x; i < kHotBins; i += kDeltaThreads) {
            const uint32_t cnt = s_bins[i];
            if (cnt) {
                const uint32_t dr = (i >> 10) + 255 - 16, dg = ((i >> 5) & 31) + 255 - 16, db = (i & 31) + 255 - 16;
                atomicAdd(&table[(dr << 18) | (dg << 9) | db], cnt);
            }
        }
    }
}

// The same on 2^n squares of side >= 8, one scan position per LANE: the 64 positions of an aligned group share every
// level of the curve but the last three, so the wave walks the shared levels once (the same table entries in every lane)
// and a lane finishes with ONE look-up of its own -- its 6 low bits in the state the group ends in -- instead of five
// dependent ones per pixel; a wave owns a contiguous run of groups, the predecessor of a group's first pixel is the
// previous group's last one (carried in a register), and the symbol store of a group is one coalesced 256-byte row.
template <bool HIST>
__global__ __launch_bounds__(kDeltaThreads) void k_hilbert_delta_p2(const uint8_t *__restrict__ rgb, uint32_t order, const HilbertLut *__restrict__ lut,
                                                                    uint32_t *__restrict__ syms, uint32_t *__restrict__ table) {
    extern __shared__ uint32_t s_bins[];  // HIST: u32[kHotBins]
    __shared__ uint16_t s_l4[1024];
    __shared__ uint8_t s_l1[16];
    __shared__ uint8_t s_l3[4 * 64];  // three levels from state s for the 6 low bits q: x:3 | y:3 << 3
    __shared__ uint32_t s_ck[64], s_cc[64];   // the block's cache of keys outside the cube (cold_count)
    if (HIST) {
        for (uint32_t i = threadIdx.x; i < kHotBins; i += kDeltaThreads) s_bins[i] = 0;
        if (threadIdx.x < 64) { s_ck[threadIdx.x] = 0xffffffffu; s_cc[threadIdx.x] = 0; }
    }
    const uint32_t w = 1u << order;
    const Scan sc = load_scan(w, w, order, lut, s_l4, s_l1);
    if (threadIdx.x < 256) {
        uint32_t st = threadIdx.x >> 6, x = 0, y = 0;
        for (int lv = 2; lv >= 0; lv--) {
            const uint32_t e = s_l1[st * 4 + ((threadIdx.x >> (2 * lv)) & 3)];
            x = (x << 1) | (e & 1); y = (y << 1) | ((e >> 1) & 1); st = e >> 2;
        }
        s_l3[threadIdx.x] = (uint8_t)(x | (y << 3));
    }
    __syncthreads();
    const uint64_t n = (uint64_t)w * w;
    const uint32_t ngroups = (uint32_t)(n >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t nwaves = gridDim.x * (kDeltaThreads / 64), gw = blockIdx.x * (kDeltaThreads / 64) + (threadIdx.x >> 6);
    const uint32_t per = (ngroups + nwaves - 1) / nwaves;
    const uint32_t g0 = gw * per, g1 = g0 + per < ngroups ? g0 + per : ngroups;
    uint32_t carried = 0;  // START = (0, 0, 0) hilbertc.rs:445
    if (g0 < g1 && g0 > 0) {
        uint32_t x, y;
        sc.xy((uint64_t)g0 * 64 - 1, x, y);
        carried = px_le24(rgb, (uint64_t)y * w + x, n);
    }
    const uint32_t top = order - 3;  // levels the 64 positions of a group share
    constexpr int kBatch = 4;        // groups whose pixel loads are in flight together (one block per CU: the waves must hide the latency themselves)
    for (uint32_t gb = g0; gb < g1; gb += kBatch) {
        uint32_t px[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; j++) {
            const uint32_t g = gb + j;
            px[j] = 0;
            if (g < g1) {
                uint32_t st = 0, x = 0, y = 0, rem = top;
                while (rem >= 4) {
                    const uint32_t e = s_l4[st * 256 + ((g >> (2 * (rem - 4))) & 255)];
                    x = (x << 4) | (e & 15); y = (y << 4) | ((e >> 4) & 15); st = e >> 8; rem -= 4;
                }
                while (rem >= 1) {
                    const uint32_t e = s_l1[st * 4 + ((g >> (2 * (rem - 1))) & 3)];
                    x = (x << 1) | (e & 1); y = (y << 1) | ((e >> 1) & 1); st = e >> 2; rem -= 1;
                }
                const uint32_t e3 = s_l3[st * 64 + lane];
                const uint32_t X = (x << 3) | (e3 & 7), Y = (y << 3) | (e3 >> 3);
                px[j] = px_le24(rgb, (uint64_t)Y * w + X, n);
            }
        }
#pragma unroll
        for (int j = 0; j < kBatch; j++) {
            const uint32_t g = gb + j;
            if (g >= g1) break;
            const uint32_t prev = wave_prev_lane(px[j], carried);
            carried = (uint32_t)__builtin_amdgcn_readlane((int)px[j], 63);
            const int32_t dr = (int32_t)(px[j] & 255) - (int32_t)(prev & 255), dg = (int32_t)((px[j] >> 8) & 255) - (int32_t)((prev >> 8) & 255),
                          db = (int32_t)((px[j] >> 16) & 255) - (int32_t)((prev >> 16) & 255);
            const uint32_t key = ((uint32_t)(dr + 255) << 18) | ((uint32_t)(dg + 255) << 9) | (uint32_t)(db + 255);
            if (HIST) {
                const uint32_t hr = (uint32_t)(dr + 16), hg = (uint32_t)(dg + 16), hb = (uint32_t)(db + 16);
                if ((hr | hg | hb) < 32u) atomicAdd(&s_bins[(hr << 10) | (hg << 5) | hb], 1u);
                else cold_count(table, s_ck, s_cc, key);
            }
            if (syms) syms[(uint64_t)g * 64 + lane] = key;
        }
    }
    if (HIST) {
        __syncthreads();
        cold_flush(table, s_ck, s_cc);
        for (uint32_t i = threadIdx.x; i < kHotBins; i += kDeltaThreads) {
            const uint32_t cnt = s_bins[i];
            if (cnt) {
                const uint32_t dr = (i >> 10) + 255 - 16, dg = ((i >> 5) & 31) + 255 - 16, db = (i & 31) + 255 - 16;
                atomicAdd(&table[(dr << 18) | (dg << 9) | db], cnt);
            }
        }
    }
}

// ---------------------------------------------------------------- host
static inline uint32_t hgrid(uint64_t items) {
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(items, 256), 1), 256 * 16);
}

static int check_dims(Ctx *c, uint32_t w, uint32_t h) {
    if (w >= (1u << 30) || h >= (1u << 30) || (uint64_t)w * h >= (1ull << 32))
        return c->fail(CNIIC_ERR_BAD_ARG, "hilbert: image %ux%u too large", w, h);
    return CNIIC_OK;
}

// order of a 2^n square (n >= 1), else 0
uint32_t pow2_order(uint32_t w, uint32_t h) {
    if (w != h || w < 2 || (w & (w - 1))) return 0;
    uint32_t o = 0;
    while ((1u << o) < w) o++;
    return o;
}

int hilbert_lut(Ctx *c, const HilbertLut **lut_d) {
    if (!c->hilbert_lut.p) {
        static HilbertLut host_lut;
        static bool ok = false;
        static std::once_flag once;
        std::call_once(once, [] { ok = build_hilbert_lut(host_lut); });
        if (!ok) return c->fail(CNIIC_ERR_HIP, "hilbert: look-up tables failed their self-check");
        DevPool *saved = current_pool();
        current_pool() = nullptr;  // lives as long as the context, not recycled
        hipError_t e = c->hilbert_lut.alloc(sizeof(HilbertLut));
        current_pool() = saved;
        if (e != hipSuccess) return c->fail(CNIIC_ERR_HIP, "hilbert: hipMalloc failed");
        CNIIC_HIP_TRY(c, hipMemcpyAsync(c->hilbert_lut.p, &host_lut, sizeof(HilbertLut), hipMemcpyHostToDevice, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    *lut_d = c->hilbert_lut.as<HilbertLut>();
    return CNIIC_OK;
}

// 2^n squares from 64 x 64 with 16-byte aligned buffers go by tiles (CNIIC_HILBERT_MOVE=any: the per-position kernel; tests)
static bool move_by_tiles(uint32_t order, const void *a, const void *b) {
    const char *e = test_env("CNIIC_HILBERT_MOVE");
    return order >= 6 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 && !(e && e[0] == 'a');
}

// ---------------------------------------------------------------- the leaves of an image size (hilbert_scan.hpp, ScanLeavesDev)
namespace {
struct LeafClass { int32_t ax, ay, bx, by; uint32_t base, area; };
__global__ void k_scan_leaf_lut(const LeafClass *__restrict__ cls, uint32_t ncls, uint32_t total, uint32_t *__restrict__ lut) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint32_t lo = 0, hi = ncls - 1;   // the class whose entries hold e
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (cls[mid].base <= e) lo = mid; else hi = mid - 1; }
    const LeafClass c = cls[lo];
    int32_t x = 0, y = 0;
    gilbert_walk(c.ax, c.ay, c.bx, c.by, (int64_t)(e - c.base), x, y);
    lut[e] = ((uint32_t)x & 0xffffu) | ((uint32_t)y << 16);
}
// the tables against the recursion itself: every leaf's first and last position and every 997th position of the scan
__global__ void k_scan_leaf_check(ScanLeavesDev L, uint32_t w, uint32_t h, uint64_t n, uint32_t *__restrict__ bad) {
    const Scan sc{w, h, 0u, nullptr, nullptr, nullptr, L};
    const uint64_t nsamp = n / 997 + 1, total = nsamp + 2ull * L.nleaf;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        uint64_t d;
        if (t < nsamp) d = t * 997;
        else { const uint64_t k = (t - nsamp) >> 1; d = ((t - nsamp) & 1) ? (uint64_t)L.d0[k + 1] - 1 : L.d0[k]; }
        if (d >= n) continue;
        uint32_t x, y, ex, ey;
        sc.xy(d, x, y);
        gilbert_d2xy(w, h, d, ex, ey);
        if (x != ex || y != ey) *bad = 1u;
    }
}
struct ScanLeavesEntry {
    uint32_t w = 0, h = 0;
    uint64_t stamp = 0;
    DevBuf idx, d0, rec, lut, hdr;
};
struct ScanLeavesCache {
    std::vector<std::unique_ptr<ScanLeavesEntry>> e;
    uint64_t clock = 0;
};
constexpr size_t kScanLeavesKept = 4;          // image sizes whose leaves a context keeps
constexpr uint64_t kScanLeavesMinPx = 1ull << 20;   // smaller images: the recursion per position (a kernel of some microseconds either way)
}  // namespace

void scan_leaves_drop(Ctx *c) { c->scan_leaves.reset(); }

// the leaves of a w x h image, built on first use of the size (host: the upper levels of the recursion; a kernel: the class tables;
// another: the self-check) and kept by the context
static int scan_leaves_get(Ctx *c, uint32_t w, uint32_t h, const ScanLeavesDev **hdr_d) {
    if (!c->scan_leaves) c->scan_leaves = std::make_shared<ScanLeavesCache>();
    ScanLeavesCache *cache = static_cast<ScanLeavesCache *>(c->scan_leaves.get());
    for (auto &en : cache->e)
        if (en->w == w && en->h == h) { en->stamp = ++cache->clock; *hdr_d = en->hdr.as<ScanLeavesDev>(); return CNIIC_OK; }
    const uint64_t n = (uint64_t)w * h;
    uint64_t max_area = 4096;
    while (max_area < n / 8192) max_area <<= 1;   // <= ~3 x 10^4 leaves whatever the image
    if (const char *e = test_env("CNIIC_SCAN_LEAF_AREA")) max_area = std::max<uint64_t>(2, strtoull(e, nullptr, 10));   // (tests: deep trees on small images)
    max_area = std::min<uint64_t>(max_area, 1u << 14);   // (a leaf's offsets are int16)
    uint32_t shift = 0;
    while ((1ull << (shift + 1)) <= max_area) shift++;
    struct Node { int32_t x, y, ax, ay, bx, by; };
    std::vector<Node> stack;
    std::vector<uint32_t> d0;
    std::vector<int4> rec;
    std::vector<LeafClass> cls;
    std::map<std::array<int32_t, 4>, uint32_t> cls_of;
    uint64_t at = 0, total = 0;
    auto leaf = [&](int32_t x, int32_t y, int32_t ax, int32_t ay, int32_t bx, int32_t by, uint32_t area) {
        const std::array<int32_t, 4> key{ax, ay, bx, by};
        auto it = cls_of.find(key);
        uint32_t ci;
        if (it == cls_of.end()) {
            ci = (uint32_t)cls.size();
            cls_of.emplace(key, ci);
            cls.push_back(LeafClass{ax, ay, bx, by, (uint32_t)total, area});
            total += area;
        } else ci = it->second;
        d0.push_back((uint32_t)at);
        rec.push_back(make_int4(x, y, (int32_t)cls[ci].base, 0));
        at += area;
    };
    if (w >= h) stack.push_back(Node{0, 0, (int32_t)w, 0, 0, (int32_t)h});
    else stack.push_back(Node{0, 0, 0, (int32_t)h, (int32_t)w, 0});
    while (!stack.empty()) {   // gilbert_walk's own case distinctions, every branch taken
        const Node q = stack.back();
        stack.pop_back();
        const int32_t W = gl_abs(q.ax + q.ay), H = gl_abs(q.bx + q.by);
        const int32_t dax = gl_sgn(q.ax), day = gl_sgn(q.ay), dbx = gl_sgn(q.bx), dby = gl_sgn(q.by);
        if (H == 1 || W == 1) {   // a line: along a (H == 1 is asked first there, too), else along b; in pieces of at most max_area
            const int32_t len = H == 1 ? W : H, sx = H == 1 ? dax : dbx, sy = H == 1 ? day : dby;
            for (int32_t o = 0; o < len; o += (int32_t)max_area) {
                const int32_t l = (int32_t)std::min<int64_t>((int64_t)max_area, len - o);
                if (H == 1) leaf(q.x + sx * o, q.y + sy * o, dax * l, day * l, q.bx, q.by, (uint32_t)l);
                else leaf(q.x + sx * o, q.y + sy * o, q.ax, q.ay, dbx * l, dby * l, (uint32_t)l);
            }
            continue;
        }
        if ((uint64_t)W * (uint64_t)H <= max_area) { leaf(q.x, q.y, q.ax, q.ay, q.bx, q.by, (uint32_t)(W * H)); continue; }
        int32_t ax2 = q.ax >> 1, ay2 = q.ay >> 1, bx2 = q.bx >> 1, by2 = q.by >> 1;
        const int32_t w2 = gl_abs(ax2 + ay2), h2 = gl_abs(bx2 + by2);
        if (2 * (int64_t)W > 3 * (int64_t)H) {
            if ((w2 & 1) && W > 2) { ax2 += dax; ay2 += day; }
            stack.push_back(Node{q.x + ax2, q.y + ay2, q.ax - ax2, q.ay - ay2, q.bx, q.by});
            stack.push_back(Node{q.x, q.y, ax2, ay2, q.bx, q.by});
        } else {
            if ((h2 & 1) && H > 2) { bx2 += dbx; by2 += dby; }
            stack.push_back(Node{q.x + (q.ax - dax) + (bx2 - dbx), q.y + (q.ay - day) + (by2 - dby), -bx2, -by2, -(q.ax - ax2), -(q.ay - ay2)});
            stack.push_back(Node{q.x + bx2, q.y + by2, q.ax, q.ay, q.bx - bx2, q.by - by2});
            stack.push_back(Node{q.x, q.y, bx2, by2, ax2, ay2});
        }
    }
    if (at != n) return c->fail(CNIIC_ERR_HIP, "hilbert: the leaves of a %ux%u image cover %llu positions", w, h, (unsigned long long)at);
    const uint32_t nleaf = (uint32_t)d0.size();
    d0.push_back((uint32_t)n);   // (n < 2^32: check_dims)
    std::vector<uint32_t> idx((n >> shift) + 1);
    for (uint64_t j = 0, k = 0; j < idx.size(); j++) {
        while (k + 1 < nleaf && d0[k + 1] <= (j << shift)) k++;
        idx[j] = (uint32_t)k;
    }
    host_trace().mark("hilbert: leaves of the image size (host)");
    auto en = std::make_unique<ScanLeavesEntry>();
    en->w = w; en->h = h; en->stamp = ++cache->clock;
    DevBuf cls_d, bad_d;
    {
        DevPool *saved = current_pool();
        current_pool() = nullptr;   // live as long as the context keeps the entry, not recycled
        hipError_t e = en->idx.alloc(idx.size() * 4);
        if (e == hipSuccess) e = en->d0.alloc(d0.size() * 4);
        if (e == hipSuccess) e = en->rec.alloc((uint64_t)nleaf * sizeof(int4));
        if (e == hipSuccess) e = en->lut.alloc(std::max<uint64_t>(total, 1) * 4);
        if (e == hipSuccess) e = en->hdr.alloc(sizeof(ScanLeavesDev));
        current_pool() = saved;
        if (e != hipSuccess) return c->fail(CNIIC_ERR_NOMEM, "hilbert: no memory for the leaves of a %ux%u image", w, h);
    }
    CNIIC_HIP_TRY(c, cls_d.alloc(cls.size() * sizeof(LeafClass)));
    CNIIC_HIP_TRY(c, bad_d.alloc(4));
    CNIIC_HIP_TRY(c, hipMemsetAsync(bad_d.p, 0, 4, c->stream));
    const ScanLeavesDev L{en->idx.as<uint32_t>(), en->d0.as<uint32_t>(), en->rec.as<int4>(), en->lut.as<uint32_t>(), shift, nleaf};
    CNIIC_HIP_TRY(c, hipMemcpyAsync(en->idx.p, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(en->d0.p, d0.data(), d0.size() * 4, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(en->rec.p, rec.data(), (uint64_t)nleaf * sizeof(int4), hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(cls_d.p, cls.data(), cls.size() * sizeof(LeafClass), hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(en->hdr.p, &L, sizeof L, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_scan_leaf_lut, dim3((uint32_t)ceil_div(total, (uint64_t)256)), dim3(256), 0, c->stream, (const LeafClass *)cls_d.as<LeafClass>(), (uint32_t)cls.size(),
                       (uint32_t)total, en->lut.as<uint32_t>());
    hipLaunchKernelGGL(k_scan_leaf_check, dim3(256), dim3(256), 0, c->stream, L, w, h, n, bad_d.as<uint32_t>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint32_t bad = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&bad, bad_d.p, 4, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));   // (the host vectors above are pageable: they must outlive the copies)
    if (bad) return c->fail(CNIIC_ERR_HIP, "hilbert: the leaves of a %ux%u image failed their self-check", w, h);
    host_trace().mark("hilbert: class tables + self-check");
    if (cache->e.size() >= kScanLeavesKept) {   // the size used longest ago makes room
        size_t old = 0;
        for (size_t i = 1; i < cache->e.size(); i++) if (cache->e[i]->stamp < cache->e[old]->stamp) old = i;
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        cache->e.erase(cache->e.begin() + (long)old);
    }
    *hdr_d = en->hdr.as<ScanLeavesDev>();
    cache->e.push_back(std::move(en));
    return CNIIC_OK;
}

int scan_select(Ctx *c, uint32_t w, uint32_t h, ScanSel *sel) {
    if (c->scan_xy.p && c->scan_w == w && c->scan_h == h) { *sel = ScanSel{0u, reinterpret_cast<const HilbertLut *>(c->scan_xy.p), 0u}; return CNIIC_OK; }
    const HilbertLut *lut = nullptr;
    CNIIC_TRY(hilbert_lut(c, &lut));
    const uint32_t order = pow2_order(w, h);
    if (!order) {
        const char *e = test_env("CNIIC_SCAN_LEAVES_MIN");   // pixels from which an image size gets its leaves (tests: 0; never: a huge number)
        const uint64_t min_px = e ? strtoull(e, nullptr, 10) : kScanLeavesMinPx;
        if ((uint64_t)w * h >= std::max<uint64_t>(min_px, 2)) {
            const ScanLeavesDev *hdr = nullptr;
            CNIIC_TRY(scan_leaves_get(c, w, h, &hdr));
            *sel = ScanSel{0u, reinterpret_cast<const HilbertLut *>(hdr), kScanLeavesBit};
            return CNIIC_OK;
        }
    }
    *sel = ScanSel{order, order ? lut : nullptr, order};
    return CNIIC_OK;
}

// cniic_ctx_set_scan: xy_d = w h positions (x, y); every pixel exactly once, or the scan is refused
__global__ void k_scan_mark(const uint2 *__restrict__ xy, uint64_t n, uint32_t w, uint32_t h, uint32_t *__restrict__ seen, uint32_t *__restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += stride) {
        const uint2 v = xy[d];
        if (v.x >= w || v.y >= h) { *bad = 1u; continue; }
        const uint64_t p = (uint64_t)v.y * w + v.x;
        if (atomicOr(&seen[p >> 5], 1u << (p & 31)) & (1u << (p & 31))) *bad = 1u;   // visited twice (so another pixel never)
    }
}
int scan_inject(Ctx *c, uint32_t w, uint32_t h, const uint32_t *xy, bool xy_dev) {
    c->scan_xy.release();
    c->scan_w = c->scan_h = 0;
    if (!xy) return CNIIC_OK;
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    if (!n) return c->fail(CNIIC_ERR_BAD_ARG, "set_scan: empty image");
    DevBuf buf, seen, bad;
    {
        // (a buffer of the context's own: it outlives the call's scratch pool scope because the context keeps it)
        void *p = nullptr;
        CNIIC_HIP_TRY(c, hipMalloc(&p, n * 8));
        buf.p = p; buf.bytes = buf.cap = n * 8; buf.pool = nullptr; buf.owned = true;
    }
    CNIIC_HIP_TRY(c, hipMemcpyAsync(buf.p, xy, n * 8, xy_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, seen.alloc(((n + 31) / 32) * 4));
    CNIIC_HIP_TRY(c, bad.alloc(4));
    CNIIC_HIP_TRY(c, hipMemsetAsync(seen.p, 0, ((n + 31) / 32) * 4, c->stream));
    CNIIC_HIP_TRY(c, hipMemsetAsync(bad.p, 0, 4, c->stream));
    hipLaunchKernelGGL(k_scan_mark, dim3(hgrid(n)), dim3(256), 0, c->stream, (const uint2 *)buf.as<uint2>(), n, w, h, seen.as<uint32_t>(), bad.as<uint32_t>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    uint32_t b = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&b, bad.p, 4, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (b) return c->fail(CNIIC_ERR_BAD_ARG, "set_scan: not a scan of the %u x %u image (a position outside it, or a pixel visited twice)", w, h);
    c->scan_xy = std::move(buf);
    c->scan_w = w; c->scan_h = h;
    return CNIIC_OK;
}

int hilbert_xy(Ctx *c, uint32_t w, uint32_t h, uint32_t *xy_d) {
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    ScanSel sel;
    CNIIC_TRY(scan_select(c, w, h, &sel));
    hipLaunchKernelGGL(k_hilbert_xy, dim3(hgrid(n)), dim3(256), 0, c->stream, w, h, sel.korder, sel.arg, xy_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int hilbert_linearize(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint8_t *out_d) {
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    ScanSel sel;
    CNIIC_TRY(scan_select(c, w, h, &sel));
    if (move_by_tiles(sel.order, rgb_d, out_d))
        hipLaunchKernelGGL(k_hilbert_move_p2<false>, dim3((uint32_t)std::min<uint64_t>(n >> 12, 256 * 8)), dim3(256), 0, c->stream, rgb_d, sel.order, sel.arg,
                           out_d);
    else if ((sel.korder & kScanLeavesBit) && (reinterpret_cast<uintptr_t>(out_d) & 3) == 0)
        hipLaunchKernelGGL(k_hilbert_move_leaves<false>, dim3(hgrid(n >> 2)), dim3(256), 0, c->stream, rgb_d, w, h, reinterpret_cast<const ScanLeavesDev *>(sel.arg), out_d);
    else
        hipLaunchKernelGGL(k_hilbert_move<false>, dim3(hgrid(n)), dim3(256), 0, c->stream, rgb_d, w, h, sel.korder, sel.arg, out_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int hilbert_scatter(Ctx *c, const uint8_t *lin_d, uint32_t w, uint32_t h, uint8_t *rgb_out_d) {
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    ScanSel sel;
    CNIIC_TRY(scan_select(c, w, h, &sel));
    if (move_by_tiles(sel.order, lin_d, rgb_out_d))
        hipLaunchKernelGGL(k_hilbert_move_p2<true>, dim3((uint32_t)std::min<uint64_t>(n >> 12, 256 * 8)), dim3(256), 0, c->stream, lin_d, sel.order, sel.arg,
                           rgb_out_d);
    else if ((sel.korder & kScanLeavesBit) && (reinterpret_cast<uintptr_t>(lin_d) & 3) == 0)
        hipLaunchKernelGGL(k_hilbert_move_leaves<true>, dim3(hgrid(n >> 2)), dim3(256), 0, c->stream, lin_d, w, h, reinterpret_cast<const ScanLeavesDev *>(sel.arg), rgb_out_d);
    else
        hipLaunchKernelGGL(k_hilbert_move<true>, dim3(hgrid(n)), dim3(256), 0, c->stream, lin_d, w, h, sel.korder, sel.arg, rgb_out_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// FromDiff + the scatter along the scan in one pass over the decoded symbols (2^n squares from 64 x 64, 16-byte aligned buffers;
// chunk_off_d: delta_undiff_prefix's exclusive channel sums per 4096 symbols).  *fused = false: not this image -- the caller
// goes through the linearised colours (delta_undiff_dev + hilbert_scatter).
int hilbert_undiff_scatter(Ctx *c, const uint32_t *keys_d, const int32_t *chunk_off_d, uint32_t w, uint32_t h, uint8_t *rgb_out_d, uint32_t *bad_d,
                           bool *fused) {
    *fused = false;
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    ScanSel sel;
    CNIIC_TRY(scan_select(c, w, h, &sel));
    if (!n || !move_by_tiles(sel.order, keys_d, rgb_out_d)) return CNIIC_OK;
    hipLaunchKernelGGL((k_hilbert_move_p2<true, true>), dim3((uint32_t)std::min<uint64_t>(n >> 12, 256 * 8)), dim3(256), 0, c->stream,
                       reinterpret_cast<const uint8_t *>(keys_d), sel.order, sel.arg, rgb_out_d, chunk_off_d, bad_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    *fused = true;
    return CNIIC_OK;
}

int hilbert_delta(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t *syms_d, uint32_t *table_d) {
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    ScanSel sel;
    CNIIC_TRY(scan_select(c, w, h, &sel));
    const HilbertLut *lut = sel.arg;
    const uint64_t nruns = ceil_div(n, kDeltaRun);
    // persistent blocks: one 1024-thread block per CU keeps the LDS bins private for as long as possible
    const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(nruns, kDeltaThreads), 1), 256);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {  // 128 KiB of dynamic LDS per block
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hilbert_delta<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(kHotBins * 4));
    });
    ScopedKernelTimer timer(c, "hilbert_delta");
    const uint32_t order = sel.order;
    // Measured at 16384^2: without the histogram the lane-per-position kernel takes 0.75 ms against 0.89 ms; with it 1.30 against
    // 1.14 ms (one block per CU for the 128 KiB of bins either way), so the fused encode keeps the 4-positions-per-thread kernel.
    // CNIIC_HILBERT_LANE_SCAN=1 / 0 forces one or the other (tests run both).
    const char *ls = test_env("CNIIC_HILBERT_LANE_SCAN");
    const bool lane_scan = ls ? atoi(ls) != 0 : table_d == nullptr;
    if (order >= 3 && lane_scan) {  // 2^n squares from 8 x 8: one position per lane, the shared levels walked once per group of 64
        static std::once_flag attr2;
        std::call_once(attr2, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hilbert_delta_p2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kHotBins * 4));
        });
        const uint32_t ngroups = (uint32_t)(n >> 6);
        const uint32_t g2 = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(ngroups, kDeltaThreads / 64), 1), table_d ? 256 : 1024);
        if (table_d)
            hipLaunchKernelGGL(k_hilbert_delta_p2<true>, dim3(g2), dim3(kDeltaThreads), kHotBins * 4, c->stream, rgb_d, order, lut, syms_d, table_d);
        else
            hipLaunchKernelGGL(k_hilbert_delta_p2<false>, dim3(g2), dim3(kDeltaThreads), 0, c->stream, rgb_d, order, lut, syms_d, table_d);
    } else if (table_d)
        hipLaunchKernelGGL(k_hilbert_delta<true>, dim3(grid), dim3(kDeltaThreads), kHotBins * 4, c->stream, rgb_d, w, h, sel.korder, lut,
                           syms_d, table_d);
    else
        hipLaunchKernelGGL(k_hilbert_delta<false>, dim3(std::min<uint32_t>(grid * 4, 1024)), dim3(kDeltaThreads), 0, c->stream, rgb_d, w, h,
                           sel.korder, lut, syms_d, table_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    timer.stop(1);
    return CNIIC_OK;
}

}  // namespace cniic
