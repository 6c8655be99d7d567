// device_utils.hpp -- small device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cniic {

__device__ __forceinline__ uint32_t rgb_key(const uint8_t *p) {
    return ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
}

// v holds r in byte 0, g in byte 1, b in byte 2 (byte 3 ignored) -> r<<16|g<<8|b
__device__ __forceinline__ uint32_t key_from_le24(uint32_t v) { return __builtin_bswap32(v) >> 8; }

// pixel idx of an N-pixel interleaved RGB8 buffer as ONE (unaligned) dword load instead of three byte loads --
// sub-dword loads go through the texture addresser several times slower; the last pixel, whose dword would
// end one byte past the buffer, is read by bytes
__device__ __forceinline__ uint32_t rgb_key_at(const uint8_t *rgb, uint64_t idx, uint64_t N) {
    if (idx + 1 < N) {
        uint32_t v;
        __builtin_memcpy(&v, rgb + 3 * idx, 4);
        return key_from_le24(v);
    }
    return rgb_key(rgb + 3 * idx);
}

// 16 interleaved RGB pixels (48 B, 16-B aligned) -> 16 packed keys
__device__ __forceinline__ void load16px_keys(const uint4 *p, uint32_t key[16]) {
    uint4 q0 = p[0], q1 = p[1], q2 = p[2];
    uint32_t w[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
    for (int g = 0; g < 4; g++) {  // 4 pixels per 3 dwords
        uint32_t a = w[3 * g], b = w[3 * g + 1], c = w[3 * g + 2];
        key[4 * g + 0] = key_from_le24(a);
        key[4 * g + 1] = key_from_le24((a >> 24) | (b << 8));
        key[4 * g + 2] = key_from_le24((b >> 16) | (c << 16));
        key[4 * g + 3] = key_from_le24(c >> 8);
    }
}

// ---- wave-wide reductions on the DPP data path (row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 and 31): six VALU
// instructions with a few cycles of latency each; the shuffle form (__shfl_down / __shfl_xor) compiles to six dependent
// ds_bpermute_b32, each a trip through the LDS crossbar (~100 cycles).  The result is in EVERY lane (read from lane 63).
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_move(uint32_t identity, uint32_t v) {  // lanes without a source keep `identity`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
template <class Op>
__device__ __forceinline__ uint32_t wave_reduce_dpp(uint32_t v, uint32_t identity, Op op) {
    v = op(v, dpp_move<0x111, 0xf, 0xf>(identity, v));  // row_shr:1
    v = op(v, dpp_move<0x112, 0xf, 0xf>(identity, v));  // row_shr:2
    v = op(v, dpp_move<0x114, 0xf, 0xe>(identity, v));  // row_shr:4  (lanes 4..15 of a row)
    v = op(v, dpp_move<0x118, 0xf, 0xc>(identity, v));  // row_shr:8  (lanes 8..15): lane 15 of a row holds the row
    v = op(v, dpp_move<0x142, 0xa, 0xf>(identity, v));  // row_bcast:15 into rows 1 and 3
    v = op(v, dpp_move<0x143, 0xc, 0xf>(identity, v));  // row_bcast:31 into rows 2 and 3: lane 63 holds the wave
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t v) {
    return wave_reduce_dpp(v, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}
__device__ __forceinline__ uint64_t wave_reduce_sum64(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    auto step = [&](uint32_t mlo, uint32_t mhi) { const uint64_t t = (((uint64_t)hi << 32) | lo) + (((uint64_t)mhi << 32) | mlo); lo = (uint32_t)t; hi = (uint32_t)(t >> 32); };
    step(dpp_move<0x111, 0xf, 0xf>(0u, lo), dpp_move<0x111, 0xf, 0xf>(0u, hi));
    step(dpp_move<0x112, 0xf, 0xf>(0u, lo), dpp_move<0x112, 0xf, 0xf>(0u, hi));
    step(dpp_move<0x114, 0xf, 0xe>(0u, lo), dpp_move<0x114, 0xf, 0xe>(0u, hi));
    step(dpp_move<0x118, 0xf, 0xc>(0u, lo), dpp_move<0x118, 0xf, 0xc>(0u, hi));
    step(dpp_move<0x142, 0xa, 0xf>(0u, lo), dpp_move<0x142, 0xa, 0xf>(0u, hi));
    step(dpp_move<0x143, 0xc, 0xf>(0u, lo), dpp_move<0x143, 0xc, 0xf>(0u, hi));
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
}
__device__ __forceinline__ uint32_t wave_reduce_min(uint32_t v) {
    return wave_reduce_dpp(v, 0xffffffffu, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
}
__device__ __forceinline__ uint32_t wave_reduce_max(uint32_t v) {
    return wave_reduce_dpp(v, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
}

// 64-bit minimum / maximum of the wave (argmin keys "distance << 32 | index"), in every lane
template <bool MAX> __device__ __forceinline__ uint64_t wave_reduce_minmax64(uint64_t v) {
    const uint32_t idl = MAX ? 0u : 0xffffffffu;
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    auto step = [&](uint32_t mlo, uint32_t mhi) {
        const uint64_t a = ((uint64_t)hi << 32) | lo, b = ((uint64_t)mhi << 32) | mlo;
        const uint64_t t = MAX ? (a > b ? a : b) : (a < b ? a : b);
        lo = (uint32_t)t; hi = (uint32_t)(t >> 32);
    };
    step(dpp_move<0x111, 0xf, 0xf>(idl, lo), dpp_move<0x111, 0xf, 0xf>(idl, hi));
    step(dpp_move<0x112, 0xf, 0xf>(idl, lo), dpp_move<0x112, 0xf, 0xf>(idl, hi));
    step(dpp_move<0x114, 0xf, 0xe>(idl, lo), dpp_move<0x114, 0xf, 0xe>(idl, hi));
    step(dpp_move<0x118, 0xf, 0xc>(idl, lo), dpp_move<0x118, 0xf, 0xc>(idl, hi));
    step(dpp_move<0x142, 0xa, 0xf>(idl, lo), dpp_move<0x142, 0xa, 0xf>(idl, hi));
    step(dpp_move<0x143, 0xc, 0xf>(idl, lo), dpp_move<0x143, 0xc, 0xf>(idl, hi));
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
}
__device__ __forceinline__ uint64_t wave_reduce_min64(uint64_t v) { return wave_reduce_minmax64<false>(v); }
__device__ __forceinline__ uint64_t wave_reduce_max64(uint64_t v) { return wave_reduce_minmax64<true>(v); }

// inclusive scan across the 64 lanes of a wave: the same six DPP steps with every bank enabled (row_shr leaves the first
// lanes of a row without a source, they add 0; the two row broadcasts carry the totals of the rows before)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += dpp_move<0x111, 0xf, 0xf>(0u, v);
    v += dpp_move<0x112, 0xf, 0xf>(0u, v);
    v += dpp_move<0x114, 0xf, 0xf>(0u, v);
    v += dpp_move<0x118, 0xf, 0xf>(0u, v);
    v += dpp_move<0x142, 0xa, 0xf>(0u, v);
    v += dpp_move<0x143, 0xc, 0xf>(0u, v);
    return v;
}

template <bool MAX> __device__ __forceinline__ uint64_t wave_inclusive_scan64(uint64_t v) {  // sum, or running maximum
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    auto step = [&](uint32_t mlo, uint32_t mhi) {
        const uint64_t a = ((uint64_t)hi << 32) | lo, b = ((uint64_t)mhi << 32) | mlo;
        const uint64_t t = MAX ? (a > b ? a : b) : a + b;
        lo = (uint32_t)t; hi = (uint32_t)(t >> 32);
    };
    step(dpp_move<0x111, 0xf, 0xf>(0u, lo), dpp_move<0x111, 0xf, 0xf>(0u, hi));
    step(dpp_move<0x112, 0xf, 0xf>(0u, lo), dpp_move<0x112, 0xf, 0xf>(0u, hi));
    step(dpp_move<0x114, 0xf, 0xf>(0u, lo), dpp_move<0x114, 0xf, 0xf>(0u, hi));
    step(dpp_move<0x118, 0xf, 0xf>(0u, lo), dpp_move<0x118, 0xf, 0xf>(0u, hi));
    step(dpp_move<0x142, 0xa, 0xf>(0u, lo), dpp_move<0x142, 0xa, 0xf>(0u, hi));
    step(dpp_move<0x143, 0xc, 0xf>(0u, lo), dpp_move<0x143, 0xc, 0xf>(0u, hi));
    return ((uint64_t)hi << 32) | lo;
}
// value of the lane before (lane 0: `first`): wave_shr:1
__device__ __forceinline__ uint32_t wave_prev_lane(uint32_t v, uint32_t first) { return dpp_move<0x138, 0xf, 0xf>(first, v); }

// block-wide sum; result valid in thread 0 (and broadcast to all through LDS)
template <int THREADS> __device__ __forceinline__ uint32_t block_reduce_sum(uint32_t v) {
    __shared__ uint32_t sh[THREADS / 64];
    v = wave_reduce_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; i++) t += sh[i];
    __syncthreads();
    return t;
}

// table[key] += 1 for every calling lane, with the lanes of the wave that hold the same key as the first (then the second) lane still to do
// adding TOGETHER: atomics on one address are served tens of nanoseconds apart, so a flat image (one colour, one difference) or a two-colour
// pattern made 16.8 M of them 0.1 s (round 4: `delta` on a 4096^2 checkerboard took 110 ms instead of 0.4).  Eight rounds cover up to eight
// dominant keys at ~8 instructions each; whoever is left adds alone.  Safe under divergence: the ballots see the calling lanes only.
__device__ __forceinline__ void atomic_count(uint32_t *table, uint32_t key) {
    bool todo = true;
    int small = 0;
#pragma unroll 1
    for (int r = 0; r < 8; r++) {   // (a call with one or two lanes -- the ordinary case of a rare symbol -- leaves after as many rounds)
        const unsigned long long act = __ballot(todo);
        if (!act) return;
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(act));
        const bool same = todo && key == k;
        const unsigned long long sm = __ballot(same);
        if (same) {
            // (the lane's number from the hardware, not from threadIdx: any block shape)
            if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(sm)) atomicAdd(&table[k], (uint32_t)__popcll(sm));
            todo = false;
        }
        // (a photograph's wave holds some fifty different keys -- a leader that speaks for less than an eighth of the lanes still
        // to do says the rounds will not pay: the others add alone, after two rounds instead of eight)
#ifndef CNIIC_ATOMIC_COUNT_EIGHT   // (a measuring build keeps all eight rounds: profiles/r05_hist_atomic_count_gate.txt)
        if ((uint32_t)__popcll(sm) * 8u < (uint32_t)__popcll(act) && ++small == 2) break;   // (twice: one odd lane in front of a flat wave must not send 63 lanes to one address)
#endif
    }
    if (todo) atomicAdd(&table[key], 1u);
}

// ... and, where a block has 512 bytes of LDS to spare, a 64-slot cache in front of it: a slot belongs to the first key that hashes to it for the
// block's life and counts it there; the block adds its slots to the table once (cold_flush, behind a barrier).  An image whose differences are ALL
// rare symbols of a few kinds (stripes, a dither pattern, 2^24 colours in scan order) then costs two LDS atomics a symbol instead of one global
// atomic per key and WAVE on two or three addresses (8192^2 stripes: 14 ms of `delta` encode).  Slots start at 0xffffffff (no 27-bit key).
__device__ __forceinline__ void cold_count(uint32_t *table, uint32_t *ck, uint32_t *cc, uint32_t key) {
    const uint32_t h = (key * 2654435761u) >> 26;
    const uint32_t old = atomicCAS(&ck[h], 0xffffffffu, key);
    if (old == 0xffffffffu || old == key) atomicAdd(&cc[h], 1u);
    else atomic_count(table, key);
}
__device__ __forceinline__ void cold_flush(uint32_t *table, const uint32_t *ck, const uint32_t *cc) {
    if (threadIdx.x < 64 && ck[threadIdx.x] != 0xffffffffu && cc[threadIdx.x]) atomicAdd(&table[ck[threadIdx.x]], cc[threadIdx.x]);
}

// block-wide exclusive scan in thread order; wsum = LDS scratch of THREADS/64 words
template <int THREADS> __device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wsum) {
    uint32_t inc = wave_inclusive_scan(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; i++)
        if (i < wid) off += wsum[i];
    __syncthreads();
    return off + inc - v;
}

// ---- colour-space cells of the cluster-colors K-means (k_kmeans_rgbw.hip); the compaction counts them on its way
// (kCellShift, kCellsPerDim, kNumCells: common.hpp)
// Cell id = super-cell (4x4x4 cells = a 32^3 cube of colours; 9 bits, r-major) << 6 | cell within it (6 bits,
// r-major): the 64 cells of a super-cell are consecutive, so a wave walking its cell range changes
// super-cell rarely.
constexpr int kSuperShift = 6;
constexpr uint32_t kSupersPerDim = kCellsPerDim / 4;
__device__ __forceinline__ uint32_t cell_of(uint32_t key) {
    const uint32_t rc = ((key >> 16) & 255) >> kCellShift, gc = ((key >> 8) & 255) >> kCellShift, bc = (key & 255) >> kCellShift;
    const uint32_t sup = ((rc >> 2) * kSupersPerDim + (gc >> 2)) * kSupersPerDim + (bc >> 2);
    return (sup << kSuperShift) | ((rc & 3) << 4) | ((gc & 3) << 2) | (bc & 3);
}

// ---- index of the colours that occur anywhere (several GPUs, each holding only its own image's colours): the
// "point list" of the reference is then the ascending list of ALL occupied keys, known only as a bitmap plus a
// popcount prefix per 64-key word.  rank = position of a key in that list, select = the key at a position.
struct GIdx {
    const unsigned long long *bits;  // [2^18] occupancy, bit (key & 63) of word key >> 6; null: not in use
    const uint32_t *wprefix;         // [2^18] occupied keys before each word
    uint64_t U;                      // occupied keys in all
};
__device__ __forceinline__ uint32_t gidx_rank(const GIdx &g, uint32_t key) {
    const uint32_t w = key >> 6;
    return g.wprefix[w] + (uint32_t)__popcll(g.bits[w] & ((1ull << (key & 63)) - 1ull));
}
__device__ __forceinline__ uint32_t gidx_select(const GIdx &g, uint64_t idx) {
    uint32_t a = 0, b = 1u << 18;  // the word holding the idx-th occupied key = the last one whose prefix is <= idx
    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (g.wprefix[mid] <= idx) a = mid; else b = mid; }
    unsigned long long word = g.bits[a];
    for (uint32_t r = (uint32_t)(idx - g.wprefix[a]); r; r--) word &= word - 1;
    return (a << 6) | (uint32_t)(__ffsll((long long)word) - 1);
}

__host__ __device__ __forceinline__ uint64_t splitmix_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// index of the point stolen for empty cluster c at iteration iter: the deterministic stand-in
// for rand::thread_rng in the reference's empty-cluster branch (src/kmeans.rs:123-133)
__host__ __device__ __forceinline__ uint64_t reseed_index(uint64_t seed, uint64_t iter, uint32_t c, uint64_t n) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (iter * 65536ULL + (uint64_t)c + 1ULL);
    return splitmix_mix(z) % n;
}

// kmeans.rs:61-78 init_assignment expressed per point index
__host__ __device__ __forceinline__ uint32_t init_label(uint64_t i, uint64_t n, uint32_t K) {
    uint64_t ppc = n / K;
    uint64_t c = (n - 1 - i) / ppc;
    return (uint32_t)(c > (uint64_t)K - 1 ? (uint64_t)K - 1 : c);
}

// floor(a / b) for a < 2^53 and 0 < b < 2^32 with a quotient below 2^32: a double quotient is within one of it and is
// corrected exactly (the 64-bit integer division it replaces is ~150 instructions, three per cluster and launch)
__device__ __forceinline__ uint32_t div_floor_small(unsigned long long a, unsigned long long b) {
    if ((b >> 32) || (a >> 53)) return (uint32_t)(a / b);  // (never at the sizes this library accepts on one GPU)
    uint32_t q = (uint32_t)((double)a / (double)b);
    const uint32_t b32 = (uint32_t)b;
    if ((unsigned long long)q * b32 > a) q--;
    else if ((unsigned long long)(q + 1) * b32 <= a) q++;
    return q;
}

// the same for n <= 2^24 points with ppc = n / K and rcp = 1.0f / ppc given: both operands are exact floats and
// the float quotient is within one of the integer one (two 64-bit divisions per point are ~300 instructions)
__device__ __forceinline__ uint32_t init_label24(uint32_t i, uint32_t n, uint32_t K, uint32_t ppc, float rcp) {
    const uint32_t x = n - 1 - i;
    uint32_t q = (uint32_t)((float)x * rcp);
    if (q * ppc > x) q--;
    else if ((q + 1) * ppc <= x) q++;
    return min(q, K - 1);
}

// ---- bit packing (k_huff.hip, k_delta.hip)
constexpr uint32_t kEscape = 63;  // length field of a len << 26 | code word that redirects to the per-rank tables (U < 2^26)

// one symbol's code as up to three pieces of 32-bit words: emit(word index, bits to OR in; MSB-first inside the word)
template <typename Emit>
__device__ __forceinline__ void pack_pieces(uint32_t pos, uint32_t L, uint64_t cd, Emit emit) {
    // place bits [pos, pos+L) MSB-first: word w bit (31 - b)
    const uint32_t w = pos >> 5, b = pos & 31, room = 32 - b;  // room: bits left in word w
    if (L <= room) {
        emit(w, (uint32_t)(cd << (room - L)));
    } else {
        const uint32_t rem = L - room;  // bits after the first word
        emit(w, (uint32_t)(cd >> rem));
        if (rem <= 32) emit(w + 1, (uint32_t)(cd << (32 - rem)));
        else { emit(w + 1, (uint32_t)(cd >> (rem - 32))); emit(w + 2, (uint32_t)(cd << (64 - rem))); }
    }
}
// ... into a bit image (LDS) or straight into the output words (memory; big-endian bit order, pre-zeroed)
template <bool DIRECT>
__device__ __forceinline__ void pack_put(uint32_t *words, uint32_t pos, uint32_t L, uint64_t cd) {
    pack_pieces(pos, L, cd, [&](uint32_t i, uint32_t v) { atomicOr(&words[i], DIRECT ? __builtin_bswap32(v) : v); });
}

}  // namespace cniic
