// k_trieparse.hip -- Dec::deserialize (src/huf.rs:323-348) on gfx950, for decoders too large for one host core.
//
// `hufman` on a photograph ships one leaf record per distinct colour: 6.8 M leaves = 82 MB of pre-order trie in front of the
// payload, which one core parses in 25-50 ms (the whole decode of the payload takes 4).  The serialisation is a byte string of
// records -- tag 1 = a branch (1 byte), tag 0 = a leaf (1 + S bytes: the tag and the symbol, S = 11 for Rgb<u8>, 6 for
// SignedColor; src/ser.rs:188-214) -- and which bytes are tags is only known by reading from the start.  In parallel:
//
//   1. k_tp_maps    the bytes are cut into chunks.  Where the first tag of a chunk lies is one of R = 1 + S offsets; for EACH of
//                   them one thread walks the chunk and notes where the walk leaves it (the first-tag offset of the next chunk),
//                   how many nodes and leaves it met, and the lowest value its count of open subtrees reached.  A chunk is a map
//                   offset -> offset of R nibbles.
//   2. k_tp_chain_* one block composes the maps (a scan over function composition) -> every chunk's true first-tag offset, its
//                   first node and leaf number, the number P of subtrees still open in front of it (the parser's stack depth),
//                   and the chunk in which P reaches 0: the end of the trie.
//   3. k_tp_emit    every chunk is walked once more from its true offset: per node P and leaf / branch, per leaf its symbol;
//                   tags other than 0 / 1 and malformed symbols are the reference's None (huf.rs:343-345, ser.rs:216-222).
//   4. k_tp_levels, k_tp_parents   the right child of branch i is the first later node with P <= P[i] (everything between
//                   them, its left subtree, has P > P[i]); found through three levels of 64-way minima.  The left child is i + 1.
//   5. k_tp_leafcodes   every leaf walks to the root: depth and path = the table of leaves k_hdecode.hip searches (codes left
//                   aligned, ascending in pre-order).
// The host only reads a few counters.  Same table as huff_parse_leaves (tests compare the decoded images).
#include "common.hpp"
#include "device_utils.hpp"
#include "huff_host.hpp"

namespace cniic {

constexpr uint32_t kTpChunk = 1024;     // bytes per chunk
constexpr int kTpThreads = 256;         // 16 chunks x 16 lanes (R <= 12 of them walk)
constexpr uint32_t kTpPad = 4;          // LDS row padding (bank spread)
constexpr uint32_t kTpMaxP = 120;       // P is kept in 7 bits next to the leaf flag; deeper stacks are "too deep" (host walk)

struct TpInfo { int32_t dP, minP; uint32_t nodes, leaves; };
struct TpTotals {
    uint32_t end_chunk;      // chunk in which the trie ends (0xffffffff: it does not end inside the stream)
    uint32_t nodes_upper, leaves_upper;   // bounds for the allocation of the per-node / per-leaf arrays
    uint32_t nodes, leaves;  // exact (k_tp_emit)
    uint64_t end_pos;        // byte position (from the stream's start) just behind the trie: where the payload begins
    uint32_t err;            // 1: malformed (a tag other than 0 / 1, a bad symbol, truncated)
    uint32_t too_deep;       // a stack deeper than kTpMaxP or a leaf deeper than kLeafMaxLen
    uint32_t max_len, min_len;
};

__device__ __forceinline__ uint32_t tp_byte(const uint8_t *__restrict__ b, uint64_t nbytes, uint64_t at) { return at < nbytes ? b[at] : 0xffu; }

// chunk c = bytes [pos0 + c * kTpChunk, ...) of the stream
template <int R>
__global__ __launch_bounds__(kTpThreads) void k_tp_maps(const uint8_t *__restrict__ b, uint64_t nbytes, uint64_t pos0, uint32_t nchunks,
                                                        unsigned long long *__restrict__ maps, TpInfo *__restrict__ info) {
    __shared__ uint8_t s_b[16][kTpChunk + kTpPad];
    const uint32_t c0 = blockIdx.x * 16;
    for (uint32_t i = threadIdx.x; i < 16 * kTpChunk; i += kTpThreads) {
        const uint32_t ci = i / kTpChunk, off = i % kTpChunk;
        s_b[ci][off] = (uint8_t)tp_byte(b, nbytes, pos0 + (uint64_t)(c0 + ci) * kTpChunk + off);
    }
    __syncthreads();
    const uint32_t ci = threadIdx.x >> 4, o = threadIdx.x & 15, c = c0 + ci;
    uint32_t exit_o = 0;
    if (c < nchunks && o < (uint32_t)R) {
        const uint64_t cbeg = pos0 + (uint64_t)c * kTpChunk;
        const uint32_t limit = (uint32_t)min((uint64_t)kTpChunk, nbytes > cbeg ? nbytes - cbeg : 0ull);
        uint32_t p = o, nodes = 0, leaves = 0;
        int32_t P = 0, minP = 1 << 30;
        while (p < limit) {
            nodes++;
            if (s_b[ci][p] == 0) { p += R; leaves++; P--; minP = min(minP, P); }
            else { p += 1; P++; }
        }
        exit_o = p >= kTpChunk ? p - kTpChunk : 0;
        info[(size_t)c * R + o] = TpInfo{P, minP, nodes, leaves};
    }
    // the chunk's map: nibble o = where a walk that enters at offset o leaves (lanes 16 ci .. 16 ci + R - 1 hold the nibbles)
    unsigned long long m = 0;
#pragma unroll
    for (int k = 0; k < R; k++) m |= (unsigned long long)(uint32_t)__shfl((int)exit_o, (int)((threadIdx.x & 48) | k), 64) << (4 * k);
    if (c < nchunks && o == 0) maps[c] = m;
}

__device__ __forceinline__ uint32_t tp_nib(unsigned long long m, uint32_t o) { return (uint32_t)(m >> (4 * o)) & 15u; }

// The chunks' maps composed in order -- in three kernels since round 3.  As ONE block it took 1.17 ms for the 133 K chunks of a
// 6.8 M-leaf decoder, and 1.0 ms of that was the block's CU pulling 17 MB of cache lines through its own memory pipe to pick one 16-byte
// record per chunk out of R (twice); picked by a grid of threads, one per chunk, the same records are 2 MB of consecutive memory.
//   k_tp_chain_entries  (one block) the maps composed per group of chunks, the groups' entries by one thread, every chunk's entry offset
//   k_tp_pick           (grid) picked[c] = info[c][entry[c]]
//   k_tp_bases_*        (grid) exclusive sums over the chunks: every chunk's first node / leaf number and P; the trie's end
constexpr uint32_t kTpB = 8;   // loads asked for eight at a time: a lone block has nobody else to hide a round trip to memory behind
template <int R>
__global__ __launch_bounds__(1024) void k_tp_chain_entries(const unsigned long long *__restrict__ maps, uint32_t nchunks, uint8_t *__restrict__ entry) {
    __shared__ unsigned long long s_gmap[1024];
    __shared__ uint8_t s_gentry[1025];
    const uint32_t j = threadIdx.x;
    const uint32_t G = (nchunks + 1023) / 1024;
    const uint32_t c_lo = min(j * G, nchunks), c_hi = min(c_lo + G, nchunks);
    unsigned long long f = 0;
#pragma unroll
    for (int k = 0; k < R; k++) f |= (unsigned long long)k << (4 * k);  // identity
    for (uint32_t c0 = c_lo; c0 < c_hi; c0 += kTpB) {
        unsigned long long mm[kTpB];
#pragma unroll
        for (uint32_t u = 0; u < kTpB; u++) mm[u] = c0 + u < c_hi ? maps[c0 + u] : 0ull;
#pragma unroll
        for (uint32_t u = 0; u < kTpB; u++)
            if (c0 + u < c_hi) {
                unsigned long long g = 0;
#pragma unroll
                for (int k = 0; k < R; k++) g |= (unsigned long long)tp_nib(mm[u], tp_nib(f, k)) << (4 * k);
                f = g;
            }
    }
    s_gmap[j] = f;
    __syncthreads();
    if (j == 0) {
        uint32_t e = 0;  // the trie's first tag is the first byte of chunk 0
        for (uint32_t g = 0; g < 1024; g++) { s_gentry[g] = (uint8_t)e; e = tp_nib(s_gmap[g], e); }
    }
    __syncthreads();
    uint32_t e = s_gentry[j];
    for (uint32_t c0 = c_lo; c0 < c_hi; c0 += kTpB) {
        unsigned long long mm[kTpB];
#pragma unroll
        for (uint32_t u = 0; u < kTpB; u++) mm[u] = c0 + u < c_hi ? maps[c0 + u] : 0ull;
#pragma unroll
        for (uint32_t u = 0; u < kTpB; u++)
            if (c0 + u < c_hi) { entry[c0 + u] = (uint8_t)e; e = tp_nib(mm[u], e); }
    }
}
template <int R>
__global__ __launch_bounds__(256) void k_tp_pick(const TpInfo *__restrict__ info, const uint8_t *__restrict__ entry, uint32_t nchunks, TpInfo *__restrict__ picked) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c < nchunks) picked[c] = info[(size_t)c * R + entry[c]];
}
// every chunk's first node / leaf number and P = exclusive sums over the chunks of (nodes, leaves, dP): a grid-wide scan in three steps
// (as one block -- one CU storing 400 K scattered words -- this was 0.37 ms)
__global__ __launch_bounds__(1024) void k_tp_bases_local(const TpInfo *__restrict__ picked, uint32_t nchunks, uint32_t *__restrict__ nodebase, uint32_t *__restrict__ leafbase,
                                                         int32_t *__restrict__ pbase, uint32_t *__restrict__ blk /* [3][nblk] */, uint32_t nblk) {
    __shared__ uint32_t wsum[1024 / 64];
    const uint32_t c = blockIdx.x * 1024 + threadIdx.x;
    TpInfo in = {0, 1 << 30, 0, 0};
    if (c < nchunks) in = picked[c];
    const uint32_t en = block_exclusive_scan<1024>(in.nodes, wsum), el = block_exclusive_scan<1024>(in.leaves, wsum);
    const uint32_t ep = block_exclusive_scan<1024>((uint32_t)in.dP, wsum);   // (two's complement: the sum of signed values)
    if (c < nchunks) { nodebase[c] = en; leafbase[c] = el; pbase[c] = (int32_t)ep; }
    if (threadIdx.x == 1023) { blk[blockIdx.x] = en + in.nodes; blk[nblk + blockIdx.x] = el + in.leaves; blk[2 * nblk + blockIdx.x] = ep + (uint32_t)in.dP; }
}
__global__ __launch_bounds__(1024) void k_tp_bases_blocks(uint32_t *__restrict__ blk, uint32_t nblk) {   // exclusive sums over the blocks, in place (one block)
    __shared__ uint32_t wsum[1024 / 64];
    for (uint32_t q = 0; q < 3; q++) {
        uint32_t carry = 0;
        for (uint32_t b0 = 0; b0 < nblk; b0 += 1024) {
            const uint32_t i = b0 + threadIdx.x;
            const uint32_t v = i < nblk ? blk[q * nblk + i] : 0u;
            const uint32_t ex = block_exclusive_scan<1024>(v, wsum);
            if (i < nblk) blk[q * nblk + i] = carry + ex;
            __shared__ uint32_t s_tot;
            if (threadIdx.x == 1023) s_tot = ex + v;
            __syncthreads();
            carry += s_tot;
            __syncthreads();
        }
    }
}
__global__ __launch_bounds__(1024) void k_tp_bases_add(const TpInfo *__restrict__ picked, uint32_t nchunks, const uint32_t *__restrict__ blk, uint32_t nblk,
                                                       uint32_t *__restrict__ nodebase, uint32_t *__restrict__ leafbase, int32_t *__restrict__ pbase, TpTotals *__restrict__ tot) {
    const uint32_t c = blockIdx.x * 1024 + threadIdx.x;
    if (c >= nchunks) return;
    const uint32_t nb = nodebase[c] + blk[blockIdx.x], lb = leafbase[c] + blk[nblk + blockIdx.x];
    const int32_t P = 1 + pbase[c] + (int32_t)blk[2 * nblk + blockIdx.x];   // one subtree -- the whole trie -- is open in front of the first tag
    nodebase[c] = nb; leafbase[c] = lb; pbase[c] = P;
    const int32_t mp = picked[c].minP;
    if (mp != (1 << 30) && P + mp <= 0) atomicMin(&tot->end_chunk, c);   // the first chunk in which the count of open subtrees reaches 0: the trie ends there
}
__global__ void k_tp_bases_end(const TpInfo *__restrict__ picked, const uint32_t *__restrict__ nodebase, const uint32_t *__restrict__ leafbase, TpTotals *__restrict__ tot) {
    const uint32_t c = tot->end_chunk;
    if (c == 0xffffffffu) return;
    tot->nodes_upper = nodebase[c] + picked[c].nodes;
    tot->leaves_upper = leafbase[c] + picked[c].leaves;
}

__device__ __forceinline__ bool tp_symbol(const uint8_t *__restrict__ b, uint64_t nbytes, uint64_t at, int sym_kind, uint32_t &key) {
    if (sym_kind == CNIIC_SYM_RGB) {  // u64 length (must be 3) + 3 bytes (ser.rs:210-222)
        if (at + 11 > nbytes) return false;
        uint64_t len = 0;
        for (int i = 0; i < 8; i++) len |= (uint64_t)b[at + i] << (8 * i);
        if (len != 3) return false;
        key = ((uint32_t)b[at + 8] << 16) | ((uint32_t)b[at + 9] << 8) | b[at + 10];
        return true;
    }
    if (at + 6 > nbytes) return false;   // three i16, each a difference of two u8 channels (hilbertc.rs:561-571)
    key = 0;
    for (int i = 0; i < 3; i++) {
        const int32_t v = (int16_t)(uint16_t)(b[at + 2 * i] | (b[at + 2 * i + 1] << 8));
        if (v < -255 || v > 255) return false;
        key = (key << 9) | (uint32_t)(v + 255);
    }
    return true;
}

// one thread per chunk up to the trie's end: node records (P | leaf << 7), leaf symbols, the exact totals
template <int R>
__global__ __launch_bounds__(64) void k_tp_emit(const uint8_t *__restrict__ b, uint64_t nbytes, uint64_t pos0, int sym_kind, const uint8_t *__restrict__ entry,
                                                const uint32_t *__restrict__ nodebase, const uint32_t *__restrict__ leafbase, const int32_t *__restrict__ pbase,
                                                uint8_t *__restrict__ node, uint32_t *__restrict__ leafkey, uint32_t *__restrict__ leafnode,
                                                TpTotals *__restrict__ tot) {
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c > tot->end_chunk) return;
    const uint64_t cbeg = pos0 + (uint64_t)c * kTpChunk;
    uint64_t p = cbeg + entry[c];
    const uint64_t cend = min(cbeg + kTpChunk, nbytes);
    uint32_t ni = nodebase[c], li = leafbase[c];
    int32_t P = pbase[c];
    bool bad = false, deep = false;
    while (p < cend) {
        const uint32_t tag = b[p];
        if (tag > 1) bad = true;
        if (P > (int32_t)kTpMaxP) deep = true;
        if (tag == 0) {
            uint32_t key = 0;
            if (!tp_symbol(b, nbytes, p + 1, sym_kind, key)) bad = true;
            node[ni] = (uint8_t)(min(P, (int32_t)kTpMaxP) | 128);
            leafkey[li] = key; leafnode[li] = ni;
            ni++; li++; P--; p += R;
            if (P == 0) {  // the trie is complete: what follows is the payload
                tot->nodes = ni; tot->leaves = li; tot->end_pos = p;
                break;
            }
        } else {
            node[ni] = (uint8_t)min(P, (int32_t)kTpMaxP);
            ni++; P++; p += 1;
        }
    }
    if (bad) tot->err = 1u;
    if (deep) tot->too_deep = 1u;
}

// 64-way minima of P, level by level (level 0 = the node records themselves)
__global__ __launch_bounds__(256) void k_tp_levels(const uint8_t *__restrict__ src, uint32_t n_src, bool src_is_nodes, uint8_t *__restrict__ dst, uint32_t n_dst) {
    const uint32_t k = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;   // one wave per destination entry
    if (k >= n_dst) return;
    const uint32_t i = k * 64 + lane;
    uint32_t v = 255;
    if (i < n_src) v = src_is_nodes ? (src[i] & 127u) : src[i];
    v = wave_reduce_min(v);
    if (lane == 0) dst[k] = (uint8_t)v;
}

struct TpLevels { const uint8_t *m[4]; uint32_t n[4]; int count; };   // m[0] = minima of 64 nodes, m[1] of 64 x 64, ...

// parent[i] = parent's node number << 1 | (1 if i is the right child)
__global__ __launch_bounds__(256) void k_tp_parents(const uint8_t *__restrict__ node, uint32_t m, TpLevels L, uint32_t *__restrict__ parent) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    if (i == 0) parent[0] = 0xffffffffu;
    const uint32_t rec = node[i];
    if (rec & 128u) return;   // a leaf has no children
    const uint32_t H = rec & 127u;
    if (i + 1 < m) parent[i + 1] = i << 1;
    // the right child: the first j >= i + 2 with P[j] <= H
    uint32_t j = i + 2;
    bool found = false;
    while (j < m && (j & 63u)) { if ((node[j] & 127u) <= H) { found = true; break; } j++; }
    if (!found && j < m) {
        // j is a multiple of 64: climb while nothing in the rest of the current group of the level qualifies
        int lv = 0;
        uint32_t k = j >> 6;   // index at level lv
        for (;;) {
            bool hit = false;
            while (k < L.n[lv]) {
                if (L.m[lv][k] <= H) { hit = true; break; }
                k++;
                if (!(k & 63u) && lv + 1 < L.count) break;   // end of this group of 64: one level up
            }
            if (hit) {  // descend to the first qualifying child, level by level
                while (lv > 0) {
                    lv--;
                    k <<= 6;
                    while (L.m[lv][k] > H) k++;
                }
                j = k << 6;
                while ((node[j] & 127u) > H) j++;
                found = true;
                break;
            }
            if (k >= L.n[lv] || lv + 1 >= L.count) break;
            k >>= 6;
            lv++;
        }
    }
    if (found && j < m) parent[j] = (i << 1) | 1u;
}

// (grid-stride, a few thousand blocks, ONE pair of atomics per block: a pair per wave -- 10^5 waves for 6.8 M leaves -- on the
// same two words is served 26 ns apart and made this kernel 2.4 ms instead of 0.2)
__global__ __launch_bounds__(256) void k_tp_leafcodes(const uint32_t *__restrict__ leafnode, const uint32_t *__restrict__ parent, uint32_t n,
                                                      unsigned long long *__restrict__ code, uint8_t *__restrict__ len, TpTotals *__restrict__ tot) {
    __shared__ uint32_t s_mx[4], s_mn[4];
    uint32_t mx = 0, mn = 0xffffffffu;
    bool deep = false;
    for (uint32_t l = blockIdx.x * 256 + threadIdx.x; l < n; l += gridDim.x * 256) {
        unsigned long long acc = 0;
        uint32_t nd = leafnode[l], d = 0;
        while (nd != 0 && d <= kLeafMaxLen) {
            const uint32_t pr = parent[nd];
            acc |= (unsigned long long)(pr & 1u) << d;   // the deepest bit is the code's last
            nd = pr >> 1;
            d++;
        }
        code[l] = d ? acc << (64 - d) : 0ull;
        len[l] = (uint8_t)d;
        deep |= d > kLeafMaxLen;
        mx = max(mx, d); mn = min(mn, d);
    }
    mx = wave_reduce_max(mx); mn = wave_reduce_min(mn);
    if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_mn[threadIdx.x >> 6] = mn; }
    if (deep) tot->too_deep = 1u;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(&tot->max_len, max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3])));
        atomicMin(&tot->min_len, min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3])));
    }
}

// stream_d: the whole stream in HBM, the serialised decoder from byte pos0.  On success (*status == 0) the table of leaves is
// in tab (code u64[n] | key u32[n] | len u8[n]; offsets in *off_key / *off_len), *n_leaves, *max_len, and *payload_pos = where the
// payload begins.  *status: 1 = malformed / truncated (the reference's None), 2 = too deep for the table (the host's node walk).
int huff_parse_leaves_dev(Ctx *c, int sym_kind, const uint8_t *stream_d, uint64_t nbytes, uint64_t pos0, DevBuf *tab, uint64_t *n_leaves,
                          uint64_t *off_key, uint64_t *off_len, uint32_t *max_len, uint64_t *payload_pos, int *status) {
    *status = 1;
    if (pos0 >= nbytes) return CNIIC_OK;
    const int R = 1 + huff_symbol_size(sym_kind);
    if (R != 12 && R != 7) return c->fail(CNIIC_ERR_BAD_ARG, "trie parse: unknown symbol kind");
    const uint64_t span = nbytes - pos0;
    // node, leaf and dP totals are 32-bit scans; a node is at least one byte, so below 4 GiB of span they cannot wrap.  Anything longer
    // (no histogram makes such a decoder; a crafted stream of branch tags could) goes to the host's walk, which caps its leaves
    if (span >= 0xfff00000ull) { *status = 2; return CNIIC_OK; }
    const uint32_t nchunks = (uint32_t)ceil_div(span, kTpChunk);
    DevBuf maps, info, entry, nodebase, leafbase, pbase, tot_d, picked;
    CNIIC_HIP_TRY(c, maps.alloc((uint64_t)nchunks * 8));
    CNIIC_HIP_TRY(c, info.alloc((uint64_t)nchunks * R * sizeof(TpInfo)));
    CNIIC_HIP_TRY(c, entry.alloc(nchunks));
    CNIIC_HIP_TRY(c, nodebase.alloc((uint64_t)nchunks * 4));
    CNIIC_HIP_TRY(c, leafbase.alloc((uint64_t)nchunks * 4));
    CNIIC_HIP_TRY(c, pbase.alloc((uint64_t)nchunks * 4));
    CNIIC_HIP_TRY(c, picked.alloc((uint64_t)nchunks * sizeof(TpInfo)));
    CNIIC_HIP_TRY(c, tot_d.alloc(sizeof(TpTotals)));
    CNIIC_HIP_TRY(c, ctx_pinned_u(c));
    TpTotals *th = reinterpret_cast<TpTotals *>(c->pinned_u + 4200);   // (slots of this function's own)
    TpTotals init{};
    init.end_chunk = 0xffffffffu; init.min_len = 0xffffffffu;
    *th = init;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(tot_d.p, th, sizeof(TpTotals), hipMemcpyHostToDevice, c->stream));
    const uint32_t g16 = (nchunks + 15) / 16;
    if (R == 12) {
        hipLaunchKernelGGL(k_tp_maps<12>, dim3(g16), dim3(kTpThreads), 0, c->stream, stream_d, nbytes, pos0, nchunks, maps.as<unsigned long long>(), info.as<TpInfo>());
        hipLaunchKernelGGL(k_tp_chain_entries<12>, dim3(1), dim3(1024), 0, c->stream, (const unsigned long long *)maps.as<unsigned long long>(), nchunks, entry.as<uint8_t>());
        hipLaunchKernelGGL(k_tp_pick<12>, dim3(ceil_div(nchunks, 256u)), dim3(256), 0, c->stream, (const TpInfo *)info.as<TpInfo>(), (const uint8_t *)entry.as<uint8_t>(), nchunks, picked.as<TpInfo>());
    } else {
        hipLaunchKernelGGL(k_tp_maps<7>, dim3(g16), dim3(kTpThreads), 0, c->stream, stream_d, nbytes, pos0, nchunks, maps.as<unsigned long long>(), info.as<TpInfo>());
        hipLaunchKernelGGL(k_tp_chain_entries<7>, dim3(1), dim3(1024), 0, c->stream, (const unsigned long long *)maps.as<unsigned long long>(), nchunks, entry.as<uint8_t>());
        hipLaunchKernelGGL(k_tp_pick<7>, dim3(ceil_div(nchunks, 256u)), dim3(256), 0, c->stream, (const TpInfo *)info.as<TpInfo>(), (const uint8_t *)entry.as<uint8_t>(), nchunks, picked.as<TpInfo>());
    }
    {
        const uint32_t nblk = ceil_div(nchunks, 1024u);
        DevBuf blk;
        CNIIC_HIP_TRY(c, blk.alloc((uint64_t)3 * nblk * 4));
        hipLaunchKernelGGL(k_tp_bases_local, dim3(nblk), dim3(1024), 0, c->stream, (const TpInfo *)picked.as<TpInfo>(), nchunks, nodebase.as<uint32_t>(), leafbase.as<uint32_t>(),
                           pbase.as<int32_t>(), blk.as<uint32_t>(), nblk);
        hipLaunchKernelGGL(k_tp_bases_blocks, dim3(1), dim3(1024), 0, c->stream, blk.as<uint32_t>(), nblk);
        hipLaunchKernelGGL(k_tp_bases_add, dim3(nblk), dim3(1024), 0, c->stream, (const TpInfo *)picked.as<TpInfo>(), nchunks, (const uint32_t *)blk.as<uint32_t>(), nblk,
                           nodebase.as<uint32_t>(), leafbase.as<uint32_t>(), pbase.as<int32_t>(), tot_d.as<TpTotals>());
        hipLaunchKernelGGL(k_tp_bases_end, dim3(1), dim3(1), 0, c->stream, (const TpInfo *)picked.as<TpInfo>(), (const uint32_t *)nodebase.as<uint32_t>(),
                           (const uint32_t *)leafbase.as<uint32_t>(), tot_d.as<TpTotals>());
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipMemcpyAsync(th, tot_d.p, sizeof(TpTotals), hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (th->end_chunk == 0xffffffffu) return CNIIC_OK;   // the trie does not end inside the stream: None
    const uint32_t m_up = th->nodes_upper, n_up = th->leaves_upper, endc = th->end_chunk;
    if (n_up == 0 || n_up >= 0xfffffff0u) return CNIIC_OK;
    DevBuf node, leafkey, leafnode, parent;
    CNIIC_HIP_TRY(c, node.alloc((uint64_t)m_up + 64));
    CNIIC_HIP_TRY(c, leafkey.alloc((uint64_t)n_up * 4));
    CNIIC_HIP_TRY(c, leafnode.alloc((uint64_t)n_up * 4));
    const uint32_t ge = (endc + 1 + 63) / 64;
    if (R == 12)
        hipLaunchKernelGGL(k_tp_emit<12>, dim3(ge), dim3(64), 0, c->stream, stream_d, nbytes, pos0, sym_kind, (const uint8_t *)entry.as<uint8_t>(), (const uint32_t *)nodebase.as<uint32_t>(),
                           (const uint32_t *)leafbase.as<uint32_t>(), (const int32_t *)pbase.as<int32_t>(), node.as<uint8_t>(), leafkey.as<uint32_t>(), leafnode.as<uint32_t>(), tot_d.as<TpTotals>());
    else
        hipLaunchKernelGGL(k_tp_emit<7>, dim3(ge), dim3(64), 0, c->stream, stream_d, nbytes, pos0, sym_kind, (const uint8_t *)entry.as<uint8_t>(), (const uint32_t *)nodebase.as<uint32_t>(),
                           (const uint32_t *)leafbase.as<uint32_t>(), (const int32_t *)pbase.as<int32_t>(), node.as<uint8_t>(), leafkey.as<uint32_t>(), leafnode.as<uint32_t>(), tot_d.as<TpTotals>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipMemcpyAsync(th, tot_d.p, sizeof(TpTotals), hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (th->err || th->leaves == 0 || th->nodes != 2 * th->leaves - 1) return CNIIC_OK;
    if (th->too_deep) { *status = 2; return CNIIC_OK; }
    const uint32_t m = th->nodes, n = th->leaves;
    // minima of P over 64, 64^2, ... nodes
    DevBuf lv[4];
    TpLevels L{};
    {
        const uint8_t *src = node.as<uint8_t>();
        uint32_t ns = m;
        bool nodes_level = true;
        for (int k = 0; k < 4; k++) {
            const uint32_t nd = (ns + 63) / 64;
            CNIIC_HIP_TRY(c, lv[k].alloc((uint64_t)nd + 64));
            hipLaunchKernelGGL(k_tp_levels, dim3((nd * 64 + 255) / 256), dim3(256), 0, c->stream, src, ns, nodes_level, lv[k].as<uint8_t>(), nd);
            L.m[k] = lv[k].as<uint8_t>(); L.n[k] = nd; L.count = k + 1;
            src = lv[k].as<uint8_t>(); ns = nd; nodes_level = false;
            if (nd <= 64) break;
        }
    }
    CNIIC_HIP_TRY(c, parent.alloc((uint64_t)m * 4));
    hipLaunchKernelGGL(k_tp_parents, dim3((m + 255) / 256), dim3(256), 0, c->stream, (const uint8_t *)node.as<uint8_t>(), m, L, parent.as<uint32_t>());
    *off_key = (uint64_t)n * 8;
    *off_len = *off_key + (uint64_t)n * 4;
    CNIIC_HIP_TRY(c, tab->alloc(*off_len + n));
    uint8_t *tb = tab->as<uint8_t>();
    CNIIC_HIP_TRY(c, hipMemcpyAsync(tb + *off_key, leafkey.p, (uint64_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_tp_leafcodes, dim3(std::min<uint32_t>((n + 255) / 256, 4096u)), dim3(256), 0, c->stream, (const uint32_t *)leafnode.as<uint32_t>(), (const uint32_t *)parent.as<uint32_t>(), n,
                       reinterpret_cast<unsigned long long *>(tb), tb + *off_len, tot_d.as<TpTotals>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipMemcpyAsync(th, tot_d.p, sizeof(TpTotals), hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (th->too_deep) { *status = 2; return CNIIC_OK; }
    *n_leaves = n;
    *max_len = th->max_len;
    *payload_pos = th->end_pos;
    *status = 0;
    return CNIIC_OK;
}

}  // namespace cniic
