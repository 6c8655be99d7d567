// k_delta.hip -- the `delta` codec's encoder (reference: src/codec/hilbertc.rs:405-415 Delta::encode =
// hilbert::linearize (hilbert.rs:10-12) -> DiffStream (hilbertc.rs:449-477) -> huf::encode_all (huf.rs:22-43)).
//
// Almost every difference of two neighbouring pixels lies in the cube [-16, 15]^3, so a symbol travels between the passes
// as ONE 16-bit word -- its index in that cube -- instead of the 27-bit key in 32 bits:
//     0 .. 32767   index in the cube (dr + 16) << 10 | (dg + 16) << 5 | (db + 16)
//     0x8000 + r   outside the cube ("cold", about 1 % of a photograph's symbols), the r-th such symbol of its 512-symbol
//                  chunk: the key is entry r of the chunk's 64 in a side array (a chunk with more sends the call to the
//                  32-bit route)
//     0x8040       padding behind the last symbol (the stream is a whole number of chunks)
// Passes (HBM bytes per pixel): gather 3 + 2, histogram 2, code lengths 2, pack 2 + the payload; the 32-bit route they
// replace (hilbert_delta + huff_pack_code32_hot) moved 3 + 4, 4 + 4 and 4 + payload.
//   k_delta_gather_p2   2^n squares: a block stages one 64 x 64 tile of the image -- 4096 consecutive scan positions --
//                       through LDS with 48-byte row reads and walks it in scan order from there
//   k_delta_gather_any  any rectangle: eight scan positions per thread (a wave = a chunk), one pixel read each
//   k_delta_hist16      the cube's counts in LDS bins, added to the dense 2^27-bin table once per block
//   k_delta_count16     bits per 512-symbol chunk from a table of code lengths in LDS; turns the chunk's cold keys into
//                       their (length, code) words on the way
//   k_delta_write16     the pack: (length, code) of the cube in LDS, one wave per chunk with a bit image of its own --
//                       no block barrier in the loop; k_delta_edges joins the words neighbouring chunks share
// Between histogram and code lengths the tree: compaction of the touched pages, leaves sorted on the GPU, merged on the
// host, codes and the serialised decoder from huff_tree_codes (k_huff.hip) -- see encode_delta (codec.cpp).
// The dense table is kept clean between calls and swept by 4096-entry pages that a flag marks as touched, so that an
// image with few distinct differences does not pay for 2^27 entries three times per call.
#include <mutex>

#include "common.hpp"
#include "device_utils.hpp"
#include "hilbert_scan.hpp"

namespace cniic {

constexpr uint32_t kHot = 32 * 32 * 32;
constexpr uint32_t kCold16 = 0x8000u, kPad16 = 0x8040u;  // kCold16 + r, r < 64
#ifndef CNIIC_COUNT_WAVES
#define CNIIC_COUNT_WAVES 8
#endif
#ifndef CNIIC_HIST_FLY
#define CNIIC_HIST_FLY 4
#endif
#ifndef CNIIC_COUNT_BATCH
#define CNIIC_COUNT_BATCH 4
#endif
#ifndef CNIIC_HIST_PIPE
#define CNIIC_HIST_PIPE 0
#endif
constexpr int kChunk16 = 512;  // symbols per wave and step of the count and the pack: one 16-byte read per lane

// DiffStream::next (hilbertc.rs:458-476) on two r | g << 8 | b << 16 pixels: the packed SignedColor key and the cube index
__device__ __forceinline__ uint32_t delta_key(uint32_t px, uint32_t prev, uint32_t &hot) {
    const int32_t dr = (int32_t)(px & 255) - (int32_t)(prev & 255), dg = (int32_t)((px >> 8) & 255) - (int32_t)((prev >> 8) & 255),
                  db = (int32_t)((px >> 16) & 255) - (int32_t)((prev >> 16) & 255);
    const uint32_t hr = (uint32_t)(dr + 16), hg = (uint32_t)(dg + 16), hb = (uint32_t)(db + 16);
    hot = (hr | hg | hb) < 32u ? (hr << 10) | (hg << 5) | hb : kCold16;
    return ((uint32_t)(dr + 255) << 18) | ((uint32_t)(dg + 255) << 9) | (uint32_t)(db + 255);
}
__device__ __forceinline__ uint32_t hot_to_key(uint32_t i) {
    return (((i >> 10) + 255 - 16) << 18) | ((((i >> 5) & 31) + 255 - 16) << 9) | ((i & 31) + 255 - 16);
}
constexpr uint32_t kColdPerChunk = 64;
__device__ __forceinline__ uint32_t lanes_before(uint64_t mask) {  // set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---------------------------------------------------------------- pass 1: gather + differences
// 2^n squares of side >= 64 (rows of 16-byte aligned 48-byte pieces).  A tile is 4096 consecutive positions; wave v of
// the block walks positions 1024 v .. 1024 v + 1023 of it, 64 at a time: an 8 x 8 block of pixels.  In LDS a pixel is
// three 10-bit fields  b + 528 | (g + 528) << 10 | (r + 528) << 20, so that ONE subtraction gives the three differences
// (fields c - p + 528, no borrow between them), one mask test says whether they all lie in [-16, 15] and three shifts make
// the cube index.  The tile is kept block by block (64 words per 8 x 8 block, the row inside a block XOR-ed with a number
// that differs between the blocks a wave writes together): a step's 64 reads are the 64 words of one block, its address
// the block's base (wave-uniform, scalar) plus the lane's place in the block for the orientation the curve has there --
// four possibilities, six bits each, packed in one register per lane for the whole kernel.  The predecessor of a lane's
// pixel is the lane before; the wave's last pixel is carried to its next step.
constexpr uint32_t kField = 528;                                      // c - p + 528 in [273, 783]: ten bits, never negative
constexpr uint32_t kFields = 1u | (1u << 10) | (1u << 20);
__device__ __forceinline__ uint32_t px_fields(uint32_t px) {         // r | g << 8 | b << 16 (bits 24..31: anything)
    return ((px >> 16) & 255u) + (((px >> 8) & 255u) << 10) + ((px & 255u) << 20) + kField * kFields;
}
__global__ __launch_bounds__(256) void k_delta_gather_p2(const uint8_t *__restrict__ rgb, uint32_t order, const HilbertLut *__restrict__ lut,
                                                         uint16_t *__restrict__ hot16, uint32_t *__restrict__ table, uint8_t *__restrict__ pages,
                                                         uint32_t *__restrict__ coldkeys, uint8_t *__restrict__ chunk_cold,
                                                         uint32_t *__restrict__ overflow /* = 1 when a chunk has more cold symbols than fit */) {
    __shared__ __align__(16) uint32_t s_tile[64 * 64];
    __shared__ uint16_t s_l4[1024];
    __shared__ uint8_t s_l1[16];
    __shared__ __align__(16) uint8_t s_l3[4 * 64];  // three levels from state s for six bits q: x:3 | y:3 << 3 | end state << 6
    __shared__ uint32_t s_cold[4 * kColdPerChunk];  // per wave: the differences (three fields) of the chunk's cold symbols so far
    const uint32_t w = 1u << order;
    const Scan sc = load_scan(w, w, order, lut, s_l4, s_l1);
    {
        uint32_t st = threadIdx.x >> 6, x = 0, y = 0;
        for (int lv = 2; lv >= 0; lv--) {
            const uint32_t e = s_l1[st * 4 + ((threadIdx.x >> (2 * lv)) & 3)];
            x = (x << 1) | (e & 1); y = (y << 1) | ((e >> 1) & 1); st = e >> 2;
        }
        s_l3[threadIdx.x] = (uint8_t)(x | (y << 3) | (st << 6));
    }
    __syncthreads();
    const uint64_t n = (uint64_t)w * w;
    const uint32_t ntiles = (uint32_t)(n >> 12);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t row = threadIdx.x >> 2, seg = threadIdx.x & 3;
    // the lane's place (y3 * 8 + x3) inside a block the curve crosses in orientation 0 .. 3
    const uint32_t places = (s_l3[lane] & 63u) | ((s_l3[64 + lane] & 63u) << 6) | ((s_l3[128 + lane] & 63u) << 12) | ((s_l3[192 + lane] & 63u) << 18);
    // a block's word base and the XOR of its rows, from x:3 | y:3 << 3
    auto blk_base = [](uint32_t e) { return (e & 63u) << 6; };
    auto blk_xor = [](uint32_t e) { return (((e >> 1) & 3u) | (((e >> 3) & 1u) << 2)) << 3; };
    constexpr uint32_t kC = kField * kFields, kMask = 0x3e0u * kFields, kHotBits = 0x200u * kFields;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // where the tile lies and in which orientation the curve enters it: the levels above the tile
        uint32_t st = 0, tx = 0, ty = 0, rem = order - 6;
        while (rem >= 4) {
            const uint32_t e = s_l4[st * 256 + ((tile >> (2 * (rem - 4))) & 255)];
            tx = (tx << 4) | (e & 15); ty = (ty << 4) | ((e >> 4) & 15); st = e >> 8; rem -= 4;
        }
        while (rem >= 1) {
            const uint32_t e = s_l1[st * 4 + ((tile >> (2 * (rem - 1))) & 3)];
            tx = (tx << 1) | (e & 1); ty = (ty << 1) | ((e >> 1) & 1); st = e >> 2; rem -= 1;
        }
        st = (uint32_t)__builtin_amdgcn_readfirstlane((int)st);
        // (reading the next tile while this one is walked made the kernel slower, 441 against 394 us at 16384^2: the walk's
        // stores sit between a read and its use, and the wait for the read waits for them all.  The kernel is bound by
        // instruction issue anyway: ~1000 per wave and tile.)
        const uint4 *src = reinterpret_cast<const uint4 *>(rgb + ((uint64_t)((ty << 6) + row) * w + (tx << 6) + seg * 16) * 3);
        const uint4 q0 = src[0], q1 = src[1], q2 = src[2];
        uint32_t carried = kC;  // the pixel before the wave's first one; START = (0, 0, 0) hilbertc.rs:445
        if (wave == 0 && tile > 0) {
            uint32_t x, y;
            sc.xy((uint64_t)tile * 4096 - 1, x, y);
            carried = px_fields(px_le24(rgb, (uint64_t)y * w + x, n));
        }
        const uint32_t q[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
        __syncthreads();  // the walk of the tile before is over
        {
            const uint32_t yhi = row >> 3, y3 = row & 7;
#pragma unroll
            for (int g = 0; g < 4; g++) {  // pixels 4 g .. 4 g + 3 of the piece: bytes 12 g .. 12 g + 11
                const uint32_t a = q[3 * g], b = q[3 * g + 1], c = q[3 * g + 2];
                const uint4 px4 = make_uint4(px_fields(a), px_fields((a >> 24) | (b << 8)), px_fields((b >> 16) | (c << 16)), px_fields(c >> 8));
                const uint32_t e = (seg * 2 + (g >> 1)) | (yhi << 3);  // the block of pixels 8 (g / 2) .. of the piece
                *reinterpret_cast<uint4 *>(&s_tile[blk_base(e) + (((y3 << 3) ^ blk_xor(e)) | ((g & 1) << 2))]) = px4;
            }
        }
        __syncthreads();
        // the wave's sixteen blocks: x:3 | y:3 << 3 | orientation << 6, a byte each (wave-uniform: scalar registers)
        const uint4 ev = *reinterpret_cast<const uint4 *>(&s_l3[st * 64 + wave * 16]);
        const uint32_t es[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)ev.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)ev.y),
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)ev.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)ev.w)};
        if (wave > 0) {  // (uniform per wave) the last pixel of the block before
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_l3[st * 64 + wave * 16 - 1]);
            const uint32_t place = s_l3[(e >> 6) * 64 + 63] & 63u;
            carried = s_tile[blk_base(e) + (place ^ blk_xor(e))];
        }
        uint16_t *o = hot16 + (uint64_t)tile * 4096 + wave * 1024 + lane;
        uint32_t crank = 0;  // cold symbols of the chunk so far (a wave's 1024 positions are two chunks)
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            const uint32_t e = (es[j >> 2] >> (8 * (j & 3))) & 255u;
            const uint32_t place = (places >> (6 * (e >> 6))) & 63u;
            const uint32_t t = s_tile[blk_base(e) + (place ^ blk_xor(e))];
            const uint32_t d = t - wave_prev_lane(t, carried) + kC;  // fields c - p + 528
            carried = (uint32_t)__builtin_amdgcn_readlane((int)t, 63);
            const bool cold = (d & kMask) != kHotBits;              // some field outside [512, 543]
            uint32_t hot = ((d >> 10) & 0x7c00u) | ((d >> 5) & 0x3e0u) | (d & 31u);
            const uint64_t cm = __builtin_amdgcn_ballot_w64(cold);
            if (cm) {  // (one step in two: one lane in 90 is cold) the lane's differences go to the wave's list in LDS
                if (cold) {
                    const uint32_t r = crank + lanes_before(cm);
                    if (r < kColdPerChunk) s_cold[wave * kColdPerChunk + r] = d;
                    hot = kCold16 + min(r, kColdPerChunk - 1);
                }
                crank += (uint32_t)__popcll(cm);
            }
            o[j * 64] = (uint16_t)hot;
            if ((j & 7) == 7) {  // the chunk's cold symbols, once: keys to the side array (a row), counts to the table
                const uint32_t ch = tile * 8 + wave * 2 + (j >> 3);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane < min(crank, kColdPerChunk)) {
                    const uint32_t dd = s_cold[wave * kColdPerChunk + lane];
                    const uint32_t key = ((((dd >> 20) & 1023u) - (kField - 255)) << 18) | ((((dd >> 10) & 1023u) - (kField - 255)) << 9) |
                                         ((dd & 1023u) - (kField - 255));
                    // (a photograph's chunk holds five such symbols of five kinds: one atomic each.  A chunk with sixteen or more is not a
                    // photograph's -- a pattern of few colours, whose keys repeat: those add together, atomic_count.  crank is wave-uniform.)
                    if (crank >= 16) atomic_count(table, key);
                    else atomicAdd(&table[key], 1u);
                    pages[key >> kPageShift] = 1;
                    coldkeys[(uint64_t)ch * kColdPerChunk + lane] = key;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    chunk_cold[ch] = (uint8_t)min(crank, 255u);
                    if (crank > kColdPerChunk) *overflow = 1;
                }
                crank = 0;
            }
        }
    }
}

// any rectangle: eight consecutive positions per thread, one pixel read each -- a wave's step is one chunk
__global__ __launch_bounds__(256) void k_delta_gather_any(const uint8_t *__restrict__ rgb, uint32_t w, uint32_t h, uint32_t order,
                                                          const HilbertLut *__restrict__ lut, uint16_t *__restrict__ hot16,
                                                          uint32_t *__restrict__ table, uint8_t *__restrict__ pages, uint32_t *__restrict__ coldkeys,
                                                          uint8_t *__restrict__ chunk_cold, uint32_t *__restrict__ overflow) {
    __shared__ uint16_t s_l4[1024];
    __shared__ uint8_t s_l1[16];
    const Scan sc = load_scan(w, h, order, lut, s_l4, s_l1);
    const uint64_t n = (uint64_t)w * h;
    const uint32_t nchunks = (uint32_t)((n + kChunk16 - 1) / kChunk16);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nw = gridDim.x * 4;
    for (uint32_t ch = blockIdx.x * 4 + (threadIdx.x >> 6); ch < nchunks; ch += nw) {
        const uint64_t d0 = (uint64_t)ch * kChunk16 + lane * 8;
        uint32_t px[8];
        ScanCursor cu;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            px[i] = 0;
            if (d0 + i < n) {
                uint32_t x, y;
                sc.xy_seq(cu, d0 + i, x, y);
                px[i] = px_le24(rgb, (uint64_t)y * w + x, n);
            }
        }
        uint32_t prev = wave_prev_lane(px[7], 0u);  // (the lane before holds the run before)
        if (lane == 0) {
            prev = 0;  // START = (0, 0, 0) hilbertc.rs:445
            if (d0 > 0) {
                uint32_t x, y;
                sc.xy(d0 - 1, x, y);
                prev = px_le24(rgb, (uint64_t)y * w + x, n);
            }
        }
        uint32_t hot[8], key[8], mine = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            hot[i] = kPad16;
            key[i] = 0;
            if (d0 + i < n) {
                key[i] = delta_key(px[i], prev, hot[i]);
                prev = px[i];
                mine += hot[i] == kCold16;
            }
        }
        const uint32_t incl = wave_inclusive_scan(mine);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total) {
            uint32_t r = incl - mine;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (hot[i] == kCold16) {
                    if (total >= 16) atomic_count(table, key[i]);   // (as in the tile gather: total is the chunk's count, wave-uniform)
                    else atomicAdd(&table[key[i]], 1u);
                    pages[key[i] >> kPageShift] = 1;
                    if (r < kColdPerChunk) coldkeys[(uint64_t)ch * kColdPerChunk + r] = key[i];
                    hot[i] = kCold16 + min(r, kColdPerChunk - 1);
                    r++;
                }
        }
        if (lane == 0) {
            chunk_cold[ch] = (uint8_t)min(total, 255u);
            if (total > kColdPerChunk) *overflow = 1;
        }
        // (the stream is a whole number of chunks)
        *reinterpret_cast<uint4 *>(hot16 + d0) = make_uint4(hot[0] | (hot[1] << 16), hot[2] | (hot[3] << 16), hot[4] | (hot[5] << 16), hot[6] | (hot[7] << 16));
    }
}

// ---------------------------------------------------------------- pass 2: utils::count_freqs (utils.rs:4-16 via huf.rs:30)
__global__ __launch_bounds__(1024) void k_delta_hist16(const uint16_t *__restrict__ hot16, uint64_t nvec /* 8 symbols each */,
                                                       uint32_t *__restrict__ table, uint8_t *__restrict__ pages) {
    extern __shared__ uint32_t s_bins[];  // [kHot]
    for (uint32_t i = threadIdx.x; i < kHot; i += 1024) s_bins[i] = 0;
    __syncthreads();
    const uint4 *v = reinterpret_cast<const uint4 *>(hot16);
    const uint64_t stride = (uint64_t)gridDim.x * 1024;
    auto count8 = [&](const uint4 q) {
        const uint32_t s[8] = {q.x & 0xffffu, q.x >> 16, q.y & 0xffffu, q.y >> 16, q.z & 0xffffu, q.z >> 16, q.w & 0xffffu, q.w >> 16};
        uint32_t run = 1;  // equal neighbours (flat areas) make one addition
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (i < 7 && s[i] == s[i + 1]) { run++; continue; }
            if (s[i] < kHot) atomicAdd(&s_bins[s[i]], run);
            run = 1;
        }
    };
    uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    constexpr int kFly = CNIIC_HIST_FLY;   // reads in flight per thread (one block per CU: kFly x 16 KiB in flight)
#if CNIIC_HIST_PIPE
    // (measuring builds) the next kFly reads are asked for before the current ones are counted
    if (i + (kFly - 1) * stride < nvec) {
        uint4 q[kFly];
#pragma unroll
        for (int f = 0; f < kFly; f++) q[f] = v[i + f * stride];
        i += kFly * stride;
        for (; i + (kFly - 1) * stride < nvec; i += kFly * stride) {
            uint4 qn[kFly];
#pragma unroll
            for (int f = 0; f < kFly; f++) qn[f] = v[i + f * stride];
#pragma unroll
            for (int f = 0; f < kFly; f++) count8(q[f]);
#pragma unroll
            for (int f = 0; f < kFly; f++) q[f] = qn[f];
        }
#pragma unroll
        for (int f = 0; f < kFly; f++) count8(q[f]);
    }
#else
    for (; i + (kFly - 1) * stride < nvec; i += kFly * stride) {
        uint4 q[kFly];
#pragma unroll
        for (int f = 0; f < kFly; f++) q[f] = v[i + f * stride];
#pragma unroll
        for (int f = 0; f < kFly; f++) count8(q[f]);
    }
#endif
    for (; i < nvec; i += stride) count8(v[i]);
    __syncthreads();
#ifdef CNIIC_HIST_NOFLUSH   // (measuring builds: what the block's 32768 additions to the table cost; the result is wrong)
    if (blockIdx.x) return;
#endif
    for (uint32_t b = threadIdx.x; b < kHot; b += 1024) {
        const uint32_t cnt = s_bins[b];
        if (cnt) {
            const uint32_t key = hot_to_key(b);
            atomicAdd(&table[key], cnt);
            pages[key >> kPageShift] = 1;
        }
    }
}

// ---------------------------------------------------------------- the table's pages
// every touched page back to zero (and its flag): the table is clean again for the next call
__global__ __launch_bounds__(256) void k_delta_clean_pages(uint32_t *__restrict__ table, uint8_t *__restrict__ pages) {
    if (!pages[blockIdx.x]) return;
    uint4 *p = reinterpret_cast<uint4 *>(table + ((uint64_t)blockIdx.x << kPageShift));
    for (uint32_t i = threadIdx.x; i < (1u << kPageShift) / 4; i += 256) p[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (threadIdx.x == 0) pages[blockIdx.x] = 0;
}

// ---------------------------------------------------------------- codes: dense[key] and, inside the cube, hot[] / hotlen[]
__global__ void k_delta_fill_codes(const uint32_t *__restrict__ keys, const uint8_t *__restrict__ len, const uint64_t *__restrict__ code, uint64_t U,
                                   uint32_t *__restrict__ dense, uint32_t *__restrict__ hot, uint8_t *__restrict__ hotlen, uint32_t inline_max) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += stride) {
        const uint32_t v = len[i] <= inline_max ? ((uint32_t)len[i] << 26) | (uint32_t)code[i] : (kEscape << 26) | (uint32_t)i;
        const uint32_t k = keys[i];
        dense[k] = v;
        const uint32_t hr = (k >> 18) - (255 - 16), hg = ((k >> 9) & 511) - (255 - 16), hb = (k & 511) - (255 - 16);
        if ((hr | hg | hb) < 32u) {
            const uint32_t hx = (hr << 10) | (hg << 5) | hb;
            hot[hx] = v;
            hotlen[hx] = len[i];
        }
    }
}

// ---------------------------------------------------------------- pass 3: bits per chunk
constexpr int kCountBatch = CNIIC_COUNT_BATCH;  // chunks whose reads a wave has in flight together
constexpr int kCountWaves = CNIIC_COUNT_WAVES;   // waves per block: four blocks per CU (32 KiB of lengths each), so 8 -> 32 waves per CU.  The kernel is a chain of four dependent
                                                 // round trips per batch (symbols -> cold keys -> their words -> escaped lengths): round 4, twice the waves = twice the chains in flight
__global__ __launch_bounds__(kCountWaves * 64) void k_delta_count16(const uint16_t *__restrict__ hot16, uint32_t nchunks, const uint8_t *__restrict__ hotlen,
                                                       uint32_t *__restrict__ coldcodes /* in: keys */, const uint8_t *__restrict__ chunk_cold,
                                                       const uint32_t *__restrict__ dense, const uint8_t *__restrict__ len,
                                                       uint32_t *__restrict__ chunk_bits) {
    __shared__ __align__(16) uint8_t s_len[kHot];
    for (uint32_t i = threadIdx.x; i < kHot / 16; i += kCountWaves * 64) reinterpret_cast<uint4 *>(s_len)[i] = reinterpret_cast<const uint4 *>(hotlen)[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nw = gridDim.x * kCountWaves;
    for (uint32_t ch0 = blockIdx.x * kCountWaves + (threadIdx.x >> 6); ch0 < nchunks; ch0 += nw * kCountBatch) {
        uint4 q[kCountBatch];
        uint32_t ncold[kCountBatch];
#pragma unroll
        for (int b = 0; b < kCountBatch; b++) {
            const uint32_t ch = ch0 + b * nw;
            q[b] = ch < nchunks ? reinterpret_cast<const uint4 *>(hot16)[(uint64_t)ch * 64 + lane] : make_uint4(kPad16, kPad16, kPad16, kPad16);
            ncold[b] = ch < nchunks ? chunk_cold[ch] : 0;
        }
        // the chunks' cold symbols: lane r takes the r-th (the sum does not care whose it is) and turns its key into the len << 26 | code
        // word the pack will want (a pass of its own: 31 us).  Key -> word -> (escaped) length is three dependent reads: asked for the
        // whole batch at a time -- chunk by chunk they were 8-12 round trips to memory per batch and held the kernel at 2 TB/s.
        uint32_t ck[kCountBatch], cv[kCountBatch], cl[kCountBatch];
#pragma unroll
        for (int b = 0; b < kCountBatch; b++) {
            const uint32_t ch = ch0 + b * nw;
            ck[b] = (ch < nchunks && lane < ncold[b]) ? coldcodes[(uint64_t)ch * kColdPerChunk + lane] : 0u;
        }
#pragma unroll
        for (int b = 0; b < kCountBatch; b++) {
            const uint32_t ch = ch0 + b * nw;
            cv[b] = (ch < nchunks && lane < ncold[b]) ? dense[ck[b]] : 0u;
        }
#pragma unroll
        for (int b = 0; b < kCountBatch; b++) {
            const uint32_t ch = ch0 + b * nw;
            const bool cold = ch < nchunks && lane < ncold[b];
            cl[b] = cold ? ((cv[b] >> 26) == kEscape ? (uint32_t)len[cv[b] & 0x3ffffffu] : cv[b] >> 26) : 0u;
            if (cold) coldcodes[(uint64_t)ch * kColdPerChunk + lane] = cv[b];
        }
#pragma unroll
        for (int b = 0; b < kCountBatch; b++) {
            const uint32_t ch = ch0 + b * nw;
            if (ch >= nchunks) break;
            const uint32_t s[8] = {q[b].x & 0xffffu, q[b].x >> 16, q[b].y & 0xffffu, q[b].y >> 16, q[b].z & 0xffffu, q[b].z >> 16, q[b].w & 0xffffu, q[b].w >> 16};
            uint32_t bits = cl[b];
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (s[i] < kHot) bits += s_len[s[i]];
            bits = wave_reduce_sum(bits);
            if (lane == 0) chunk_bits[ch] = bits;
        }
    }
}

// ---------------------------------------------------------------- pass 4: the payload (huf.rs:37-41)
// One wave per chunk, no block barrier in the loop.  LDS: the cube's len << 26 | code words, then per wave 65 words for
// the chunk's cold symbols (entry r behind the cube's, so that a symbol's word is ONE read at index s, or s + the wave's
// offset from 0x8000 on; entry 64 = 0 for padding) and a bit image.  A code (<= 26 bits) goes into the image as a 64-bit
// window over two words, OR-ed in without a branch.
constexpr int kWriteWaves = 16;
constexpr uint32_t kWriteCold = kColdPerChunk + 1;
constexpr uint32_t kWriteImg = 440;  // words of a wave's bit image: 27 bits per symbol on average (a denser chunk goes to memory piece by piece)
__global__ __launch_bounds__(kWriteWaves * 64) void k_delta_write16(const uint16_t *__restrict__ hot16, uint32_t nchunks, const uint32_t *__restrict__ hot,
                                                                    const uint8_t *__restrict__ len, const uint64_t *__restrict__ code,
                                                                    const uint64_t *__restrict__ chunk_off, uint32_t *__restrict__ out_words,
                                                                    uint64_t bit_base, uint32_t img_cap, const uint32_t *__restrict__ coldcodes, uint2 *__restrict__ edge) {
    extern __shared__ uint32_t s_mem[];
    uint32_t *s_hot = s_mem;  // [kHot], then [kWriteWaves][kWriteCold], then [kWriteWaves][kWriteImg]
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cold_at = kHot + wave * kWriteCold;  // this wave's cold words
    uint32_t *img = s_mem + kHot + kWriteWaves * kWriteCold + wave * kWriteImg;
    for (uint32_t i = threadIdx.x; i < kHot / 4; i += kWriteWaves * 64) reinterpret_cast<uint4 *>(s_hot)[i] = reinterpret_cast<const uint4 *>(hot)[i];
    if (lane == 0) s_hot[cold_at + kColdPerChunk] = 0;
    __syncthreads();
    const uint32_t nw = gridDim.x * kWriteWaves;
    const uint32_t cap = min(kWriteImg, img_cap);
    const uint32_t woff = cold_at - kCold16;  // symbol 0x8000 + r -> word cold_at + r
    // One block per CU (the table) and sixteen waves: a wave keeps the reads of the NEXT four chunks in flight while it packs
    // four (one chunk at a time left the kernel waiting for memory: 7 us per chunk and wave).
    constexpr int kBatch = 4;
    const uint32_t ch0 = blockIdx.x * kWriteWaves + wave;
    uint4 qn[kBatch];
    uint32_t ccn[kBatch];
    uint64_t offn[kBatch];
    auto fetch = [&](uint32_t first) {
#pragma unroll
        for (int j = 0; j < kBatch; j++) {
            const uint32_t c2 = first + j * nw;
            if (c2 < nchunks) {
                qn[j] = reinterpret_cast<const uint4 *>(hot16)[(uint64_t)c2 * 64 + lane];
                ccn[j] = coldcodes[(uint64_t)c2 * kColdPerChunk + lane];  // lane r: the r-th cold symbol's word (past the chunk's count: anything, never used)
                offn[j] = chunk_off[c2];
            }
        }
    };
    fetch(ch0);
    for (uint32_t chb = ch0; chb < nchunks; chb += kBatch * nw) {
        uint4 qc[kBatch];
        uint32_t ccc[kBatch];
        uint64_t offc[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; j++) { qc[j] = qn[j]; ccc[j] = ccn[j]; offc[j] = offn[j]; }
        if (chb + kBatch * (uint64_t)nw < nchunks) fetch(chb + kBatch * nw);
#pragma unroll
      for (int jb = 0; jb < kBatch; jb++) {
        const uint32_t ch = chb + jb * nw;
        if (ch >= nchunks) break;
        const uint4 q = qc[jb];
        const uint64_t g0 = bit_base + offc[jb];
        s_hot[cold_at + lane] = ccc[jb];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t s[8] = {q.x & 0xffffu, q.x >> 16, q.y & 0xffffu, q.y >> 16, q.z & 0xffffu, q.z >> 16, q.w & 0xffffu, q.w >> 16};
        uint32_t v[8], bits = 0, vmax = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            v[i] = s_hot[s[i] < kCold16 ? s[i] : s[i] + woff];
            vmax = max(vmax, v[i]);
            bits += v[i] >> 26;
        }
        const bool escapes = __builtin_amdgcn_ballot_w64((vmax >> 26) == kEscape) != 0;  // (uniform) a code longer than 26 bits in the chunk
        uint32_t l[8];
        if (escapes) {
            bits = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t L = v[i] >> 26;
                l[i] = L == kEscape ? (uint32_t)len[v[i] & 0x3ffffffu] : L;
                bits += l[i];
            }
        }
        const uint32_t incl = wave_inclusive_scan(bits);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total == 0) {  // (a single-symbol alphabet: zero-length codes, huf.rs:140-142)
            if (lane == 0) edge[ch] = make_uint2(0u, 0u);
            continue;
        }
        const uint32_t skew = (uint32_t)(g0 & 31);
        const uint32_t nwords = (skew + total + 31) >> 5;
        const uint64_t w0 = g0 >> 5;
        uint32_t pos = skew + incl - bits;
        auto code_of = [&](int i) -> uint64_t { return (v[i] >> 26) == kEscape ? code[v[i] & 0x3ffffffu] : (uint64_t)(v[i] & 0x3ffffffu); };
        const bool direct = nwords + 2 > cap;
        const uint32_t tail = (skew + total) & 31;
        if (direct) {
            // A chunk too dense for the image (codes longer than 26 bits; the tests' cap): the words it fills alone are cleared
            // and the pieces OR-ed into memory, those of its first and last word collected for edge[] like everybody's.
            for (uint32_t i = lane; i < nwords; i += 64)
                if (!((i == 0 && skew) || (i == nwords - 1 && tail))) out_words[w0 + i] = 0;
            __threadfence();
            if (!escapes) {
#pragma unroll
                for (int i = 0; i < 8; i++) l[i] = v[i] >> 26;
            }
            uint32_t e_first = 0, e_last = 0;
#pragma unroll 1
            for (int i = 0; i < 8; i++) {
                if (l[i])
                    pack_pieces(pos, l[i], code_of(i), [&](uint32_t wi, uint32_t piece) {
                        if (wi == 0 && skew) e_first |= piece;
                        else if (wi == nwords - 1 && tail) e_last |= piece;
                        else if (piece) atomicOr(&out_words[w0 + wi], __builtin_bswap32(piece));
                    });
                pos += l[i];
            }
            e_first = wave_reduce_dpp(e_first, 0u, [](uint32_t a, uint32_t b) { return a | b; });
            e_last = wave_reduce_dpp(e_last, 0u, [](uint32_t a, uint32_t b) { return a | b; });
            if (lane == 0) edge[ch] = make_uint2(__builtin_bswap32(e_first), nwords > 1 || !skew ? __builtin_bswap32(e_last) : 0u);
            continue;
        }
        for (uint32_t i = lane; i < nwords + 2; i += 64) img[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (escapes) {  // symbol by symbol
#pragma unroll 1
            for (int i = 0; i < 8; i++) {
                if (l[i]) pack_put<false>(img, pos, l[i], code_of(i));
                pos += l[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {  // bits [pos, pos + L) of the image = the code: the 64-bit window that starts at pos's word
                const uint32_t L = v[i] >> 26;
                const uint64_t win = (uint64_t)(v[i] & 0x3ffffffu) << ((64 - (pos & 31) - L) & 63);  // (L = 0: the code is 0 too)
                uint32_t *wp = img + (pos >> 5);
                atomicOr(wp, (uint32_t)(win >> 32));
                atomicOr(wp + 1, (uint32_t)win);
                pos += L;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // The words the chunk fills alone go out as they are; its first and last word, when a neighbour chunk has bits in
        // them, go to edge[] and k_delta_edges joins them (an OR into memory per chunk boundary made this kernel 0.84 ms
        // instead of 0.31 at 16384^2).
        for (uint32_t i = lane; i < nwords; i += 64) {
            const uint32_t o = __builtin_bswap32(img[i]);  // MSB-first bit order -> big-endian bytes
            if (!((i == 0 && skew) || (i == nwords - 1 && tail))) out_words[w0 + i] = o;
        }
        if (lane == 0)
            edge[ch] = make_uint2(skew ? __builtin_bswap32(img[0]) : 0u, tail && (nwords > 1 || !skew) ? __builtin_bswap32(img[nwords - 1]) : 0u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
}

// The words two chunks share: chunk b's first word when it starts inside one (with the last word of the chunk before: every
// chunk but the last holds 512 symbols of at least one bit, so its first and last word differ), and the stream's last word.
__global__ __launch_bounds__(256) void k_delta_edges(const uint2 *__restrict__ edge, const uint64_t *__restrict__ chunk_off,
                                                     const uint32_t *__restrict__ chunk_bits, uint32_t nchunks, uint32_t *__restrict__ out_words,
                                                     uint64_t bit_base) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nchunks) return;
    const uint32_t total = chunk_bits[b];
    if (total == 0) return;
    const uint64_t g0 = bit_base + chunk_off[b], g1 = g0 + total;
    if (g0 & 31) out_words[g0 >> 5] = edge[b].x | (b > 0 ? edge[b - 1].y : 0u);
    if (b == nchunks - 1 && (g1 & 31) && ((g1 >> 5) != (g0 >> 5) || !(g0 & 31))) out_words[g1 >> 5] = edge[b].y;
}

// ---------------------------------------------------------------- host
// the context's 2^27-bin table of SignedColor counts + a flag per 4096-entry page; all zero between calls
int delta_table(Ctx *c, uint32_t **table_d, uint8_t **pages_d) {
    const uint64_t bytes = (1ull << 27) * 4, npages = (1ull << 27) >> kPageShift;
    if (!c->dense27.p) {
        DevPool *saved = current_pool();
        current_pool() = nullptr;  // live as long as the context
        hipError_t e = c->dense27.alloc(bytes);
        if (e == hipSuccess) e = c->dense27_pages.alloc(npages);
        current_pool() = saved;
        if (e != hipSuccess) { c->dense27.release(); c->dense27_pages.release(); return c->fail(CNIIC_ERR_HIP, "delta: hipMalloc of the symbol table failed"); }
        c->dense27_clean = false;
    }
    if (!c->dense27_clean) {  // first use, or a call that failed half-way
        CNIIC_HIP_TRY(c, hipMemsetAsync(c->dense27.p, 0, bytes, c->stream));
        CNIIC_HIP_TRY(c, hipMemsetAsync(c->dense27_pages.p, 0, npages, c->stream));
    }
    c->dense27_clean = false;  // until delta_table_clean
    *table_d = c->dense27.as<uint32_t>();
    *pages_d = c->dense27_pages.as<uint8_t>();
    return CNIIC_OK;
}

int delta_table_clean(Ctx *c) {
    hipLaunchKernelGGL(k_delta_clean_pages, dim3((1u << 27) >> kPageShift), dim3(256), 0, c->stream, c->dense27.as<uint32_t>(),
                       c->dense27_pages.as<uint8_t>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    c->dense27_clean = true;
    return CNIIC_OK;
}

uint64_t delta_stream_len(uint64_t n) { return ceil_div(n, (uint64_t)kChunk16) * kChunk16; }

// hot16_d: delta_stream_len(n) u16; coldkeys_d: 64 u32 and chunk_cold_d: a byte per 512 symbols; the cube's counts and the
// cold symbols' into table_d / pages_d; *overflow_d = 1 when a chunk has more than 64 cold symbols (a counter of them all,
// one atomic per wave on one address, took 0.2 ms: same-address atomics are served 26 ns apart)
int delta_gather_hist(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint16_t *hot16_d, uint32_t *table_d, uint8_t *pages_d,
                      uint32_t *coldkeys_d, uint8_t *chunk_cold_d, uint32_t *overflow_d) {
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    if (w >= (1u << 30) || h >= (1u << 30) || n >= (1ull << 32)) return c->fail(CNIIC_ERR_BAD_ARG, "hilbert: image %ux%u too large", w, h);
    ScanSel sel;
    CNIIC_TRY(scan_select(c, w, h, &sel));
    const HilbertLut *lut = sel.arg;
    const uint64_t npad = delta_stream_len(n);
    ScopedKernelTimer timer(c, "delta_gather");  // (bench.py --config c5 takes the gather's roofline from this one)
    const uint32_t order = sel.order;
    const char *force = test_env("CNIIC_DELTA_GATHER");  // "any": the per-position kernel on 2^n squares too (tests)
    if (order >= 6 && (reinterpret_cast<uintptr_t>(rgb_d) & 15) == 0 && !(force && force[0] == 'a')) {
        const uint32_t ntiles = (uint32_t)(n >> 12);
        hipLaunchKernelGGL(k_delta_gather_p2, dim3(std::min<uint32_t>(ntiles, 256 * 8)), dim3(256), 0, c->stream, rgb_d, order, lut, hot16_d, table_d,
                           pages_d, coldkeys_d, chunk_cold_d, overflow_d);
    } else {
        const uint32_t grid = (uint32_t)std::min<uint64_t>(ceil_div(npad / kChunk16, (uint64_t)4), 256 * 8);
        hipLaunchKernelGGL(k_delta_gather_any, dim3(grid), dim3(256), 0, c->stream, rgb_d, w, h, sel.korder, lut, hot16_d, table_d, pages_d, coldkeys_d,
                           chunk_cold_d, overflow_d);
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    timer.stop(1);
    ScopedKernelTimer timer_h(c, "delta_hist");
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_delta_hist16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kHot * 4));
    });
    // a block adds its 32768 bins to the table once: at least 2^17 symbols each
    const uint32_t hgrid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(npad >> 17, 1), 256);
    hipLaunchKernelGGL(k_delta_hist16, dim3(hgrid), dim3(1024), kHot * 4, c->stream, (const uint16_t *)hot16_d, npad / 8, table_d, pages_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    timer_h.stop(1);
    return CNIIC_OK;
}

// The payload at bit_base of out_d (4-byte aligned; NOT cleared before: every word of the payload is stored exactly once, and
// the bytes before bit_base in its first word come out as zero -- the header goes there afterwards).  At least one code has
// a length above zero.  keys_d / len_d / code_d: the U distinct symbols and their codes; dense_d: the table (overwritten at
// the U keys).  Nothing waits: *total_d (device) = the bits written, once the stream has run; scratch lives in `keep`.
// The pack in two halves: the first (codes into the tables, bits per chunk, their offsets) needs the codes but neither the output nor the
// payload's size, so the encoder enqueues it BEFORE it waits for that size -- the wait (a round trip to the host: 25-40 us of idle GPU at
// 16384^2) then passes while k_delta_count16 runs.
int delta_pack16_count(Ctx *c, const uint16_t *hot16_d, uint64_t n, uint32_t *coldkeys_d, const uint8_t *chunk_cold_d, uint32_t *dense_d,
                       const uint32_t *keys_d, const uint8_t *len_d, const uint64_t *code_d, uint64_t U, uint64_t *total_d, DeltaPackScratch *keep) {
    const uint64_t nchunks64 = delta_stream_len(n) / kChunk16;
    if (nchunks64 > 0x7fffffffull || U >= (1ull << 26)) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: too many symbols");
    const uint32_t nchunks = (uint32_t)nchunks64;
    CNIIC_HIP_TRY(c, keep->cb.alloc((uint64_t)nchunks * 4));
    CNIIC_HIP_TRY(c, keep->co.alloc((uint64_t)nchunks * 8));
    CNIIC_HIP_TRY(c, keep->edge.alloc((uint64_t)nchunks * 8));
    CNIIC_HIP_TRY(c, keep->hot.alloc((uint64_t)kHot * 4));
    CNIIC_HIP_TRY(c, keep->hotlen.alloc(kHot));
    constexpr uint32_t kWriteLds = (kHot + kWriteWaves * (kWriteCold + kWriteImg)) * 4;
    static_assert(kWriteLds <= 160 * 1024, "one block per CU");
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_delta_write16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWriteLds);
    });
    // codes of up to 26 bits sit in the word itself; longer ones (nearly never) send the pack to the per-rank tables -- tests
    // lower the limit (CNIIC_TEST_INLINE_CODE_BITS) so that ordinary images take that way
    const char *im = test_env("CNIIC_TEST_INLINE_CODE_BITS");
    const uint32_t inline_max = im ? std::min<uint32_t>((uint32_t)atoi(im), 26u) : 26u;
    hipLaunchKernelGGL(k_delta_fill_codes, dim3((uint32_t)std::min<uint64_t>(ceil_div(U, 256), 2048)), dim3(256), 0, c->stream, keys_d, len_d, code_d, U,
                       dense_d, keep->hot.as<uint32_t>(), keep->hotlen.as<uint8_t>(), inline_max);
    hipLaunchKernelGGL(k_delta_count16, dim3(std::min<uint32_t>(ceil_div(nchunks, (uint32_t)kCountWaves), 256 * 4)), dim3(kCountWaves * 64), 0, c->stream, hot16_d, nchunks,
                       (const uint8_t *)keep->hotlen.as<uint8_t>(), coldkeys_d, chunk_cold_d, (const uint32_t *)dense_d, len_d, keep->cb.as<uint32_t>());
    CNIIC_TRY(pack_scan(c, keep->cb.as<uint32_t>(), nchunks, keep->co.as<uint64_t>(), total_d));
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int delta_pack16_write(Ctx *c, const uint16_t *hot16_d, uint64_t n, const uint32_t *coldkeys_d, const uint8_t *len_d, const uint64_t *code_d, uint8_t *out_d,
                       uint64_t bit_base, DeltaPackScratch *keep) {
    if (reinterpret_cast<uintptr_t>(out_d) & 3) return c->fail(CNIIC_ERR_BAD_ARG, "huff_pack: output must be 4-byte aligned");
    const uint32_t nchunks = (uint32_t)(delta_stream_len(n) / kChunk16);
    constexpr uint32_t kWriteLds = (kHot + kWriteWaves * (kWriteCold + kWriteImg)) * 4;
    hipLaunchKernelGGL(k_delta_write16, dim3(std::min<uint32_t>(ceil_div(nchunks, (uint32_t)kWriteWaves), 256)), dim3(kWriteWaves * 64), kWriteLds,
                       c->stream, hot16_d, nchunks, (const uint32_t *)keep->hot.as<uint32_t>(), len_d, code_d,
                       (const uint64_t *)keep->co.as<uint64_t>(), reinterpret_cast<uint32_t *>(out_d), bit_base, pack_img_cap(),
                       (const uint32_t *)coldkeys_d, keep->edge.as<uint2>());
    hipLaunchKernelGGL(k_delta_edges, dim3(ceil_div(nchunks, 256u)), dim3(256), 0, c->stream, (const uint2 *)keep->edge.as<uint2>(),
                       (const uint64_t *)keep->co.as<uint64_t>(), (const uint32_t *)keep->cb.as<uint32_t>(), nchunks, reinterpret_cast<uint32_t *>(out_d), bit_base);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

}  // namespace cniic
