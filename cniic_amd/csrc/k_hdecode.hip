// k_hdecode.hip -- DecStream (src/huf.rs:187-206, 366-374) on gfx950: Huffman decoding of a whole payload in
// parallel, plus FromDiff (src/codec/hilbertc.rs:482-509) as a prefix sum.
//
// The stream has no markers, so where a symbol starts is only known by decoding everything before it.  Huffman
// codes resynchronise: a decoder started at a wrong bit position falls into step with the true symbol
// boundaries after a few symbols.  The payload is cut into subsequences of kHdSub bits, one per thread:
//   pass 0      every thread decodes from the first bit of its subsequence (a guess, exact only for thread 0)
//               to the first symbol boundary at or past its end, and reports that boundary and how many
//               symbols started inside
//   pass r > 0  thread t restarts from the boundary thread t-1 reported in pass r-1
//   until no boundary changes.  Thread 0 is exact from the start and exactness moves at least one thread per
//   pass, so an unchanged pass is the exact chain from bit 0; resynchronisation makes that 2-4 passes instead of
//   one per thread.  Codes that refuse to settle within kHdMaxPasses fall back to the host decoder.
//   Then: exclusive scan of the symbol counts -> every thread decodes once more and writes its symbols.
// The walk itself is the reference's trie walk, kHdLut bits at a time through a table held in LDS.
#include "common.hpp"
#include "device_utils.hpp"
#include "huff_host.hpp"

namespace cniic {

constexpr int kHdLut = 12;
constexpr uint32_t kHdSub = 1024;       // bits per thread
constexpr int kHdThreads = 256;
constexpr int kHdMaxPasses = 48;
constexpr uint32_t kHdLeaf = 0xffffffffu;

// n <= 32 bits starting at bit `at` of an MSB-first stream held as words (zero padded past the end)
__device__ __forceinline__ uint32_t hd_peek(const uint32_t *__restrict__ w, uint64_t at, int n) {
    const uint64_t i = at >> 5;
    const uint64_t v = ((uint64_t)__builtin_bswap32(w[i]) << 32) | __builtin_bswap32(w[i + 1]);
    return (uint32_t)((v << (at & 31)) >> (64 - n));
}

// one symbol from bit `at`: false when the stream ends inside it (DecStream yields None)
__device__ __forceinline__ bool hd_symbol(const uint32_t *__restrict__ w, uint64_t nbits, const uint2 *__restrict__ nodes,
                                          const uint2 *lut, uint64_t &at, uint32_t &key) {
    if (at >= nbits) return false;
    const uint2 hop = lut[hd_peek(w, at, kHdLut)];
    uint64_t p = at + hop.y;
    if (p > nbits) return false;
    uint2 node = nodes[hop.x];
    while (node.x != kHdLeaf) {
        if (p >= nbits) return false;
        node = nodes[hd_peek(w, p, 1) ? node.y : node.x];
        p++;
    }
    key = node.y;
    at = p;
    return true;
}

__global__ __launch_bounds__(kHdThreads) void k_hd_pass(const uint32_t *__restrict__ w, uint64_t nbits, const uint2 *__restrict__ nodes,
                                                        const uint2 *__restrict__ lut_g, uint64_t nsub,
                                                        const uint64_t *__restrict__ end_prev /* null: pass 0 */,
                                                        const uint64_t *__restrict__ end_prev2 /* the pass before that; null: passes 0, 1 */,
                                                        uint64_t *__restrict__ end_out, uint32_t *__restrict__ count,
                                                        uint32_t *__restrict__ changed) {
    __shared__ uint2 lut[1 << kHdLut];
    for (uint32_t i = threadIdx.x; i < (1u << kHdLut); i += kHdThreads) lut[i] = lut_g[i];
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * kHdThreads + threadIdx.x;
    if (t >= nsub) return;
    const uint64_t lo = t * kHdSub, hi = min(lo + kHdSub, nbits);
    uint64_t at = !end_prev ? lo : (t ? end_prev[t - 1] : 0);
    // a thread whose start did not move since the pass before ends where it ended then (and counted what it counted): from
    // the third pass on only the few subsequences still out of step decode again (every pass decoding everything made ten
    // passes over a 6.8 M-leaf code cost 63 ms)
    if (end_prev) {
        const uint64_t before = end_prev2 ? (t ? end_prev2[t - 1] : 0) : lo;
        if (before == at) { end_out[t] = end_prev[t]; return; }
    }
    uint32_t cnt = 0, key;
    while (at < hi) {
        if (!hd_symbol(w, nbits, nodes, lut, at, key)) { at = nbits; break; }  // nothing decodable from here on
        cnt++;
    }
    if (end_prev && end_prev[t] != at) *changed = 1u;
    end_out[t] = at;
    count[t] = cnt;
}

// single block: off[t] = symbols before subsequence t; *total = all symbols
__global__ __launch_bounds__(1024) void k_hd_offsets(const uint32_t *__restrict__ count, uint64_t nsub, uint64_t *__restrict__ off,
                                                     uint64_t *__restrict__ total) {
    __shared__ uint64_t sh[1024];
    const uint64_t per = (nsub + 1023) / 1024;
    const uint64_t lo = threadIdx.x * per, hi = min(lo + per, nsub);
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; i++) s += count[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint64_t v = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = sh[threadIdx.x] - s;
    for (uint64_t i = lo; i < hi; i++) { off[i] = run; run += count[i]; }
    if (threadIdx.x == 1023) *total = sh[1023];
}

__global__ __launch_bounds__(kHdThreads) void k_hd_write(const uint32_t *__restrict__ w, uint64_t nbits, const uint2 *__restrict__ nodes,
                                                         const uint2 *__restrict__ lut_g, uint64_t nsub, const uint64_t *__restrict__ end,
                                                         const uint64_t *__restrict__ off, uint64_t nsyms, uint32_t *__restrict__ keys) {
    __shared__ uint2 lut[1 << kHdLut];
    for (uint32_t i = threadIdx.x; i < (1u << kHdLut); i += kHdThreads) lut[i] = lut_g[i];
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * kHdThreads + threadIdx.x;
    if (t >= nsub) return;
    const uint64_t hi = min((t + 1) * kHdSub, nbits);
    uint64_t at = t ? end[t - 1] : 0, idx = off[t];
    uint32_t key;
    while (at < hi && idx < nsyms) {  // the reference reads exactly nsyms symbols; what the padding decodes to is dropped
        if (!hd_symbol(w, nbits, nodes, lut, at, key)) break;
        keys[idx++] = key;
    }
}

__global__ void k_hd_fill(uint32_t *__restrict__ keys, uint64_t n, uint32_t key) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) keys[i] = key;
}

// packed RGB keys -> interleaved bytes
__global__ void k_keys_to_rgb(const uint32_t *__restrict__ keys, uint64_t n, uint8_t *__restrict__ rgb) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t k = keys[i];
        rgb[3 * i] = (uint8_t)(k >> 16); rgb[3 * i + 1] = (uint8_t)(k >> 8); rgb[3 * i + 2] = (uint8_t)k;
    }
}

// nodes_h: (l, r) per node, leaves as (kHdLeaf, key); payload: host bytes.  keys_d receives nsyms symbols.
// *status: 0 = decoded, 1 = the stream ends early (None), 2 = did not settle (caller decodes on the host)
int huff_decode_dev(Ctx *c, const std::vector<TrieNode> &nodes_h, const uint8_t *payload, uint64_t payload_bytes, uint64_t nsyms,
                    uint32_t *keys_d, int *status) {
    *status = 0;
    if (nsyms == 0) return CNIIC_OK;
    static_assert(sizeof(TrieNode) == sizeof(uint2) && kTrieLeaf == kHdLeaf, "TrieNode is copied to the device as uint2");
    if (nodes_h[0].l == kHdLeaf) {  // one symbol, zero-length code, no payload (huf.rs:140-142)
        hipLaunchKernelGGL(k_hd_fill, dim3(1024), dim3(256), 0, c->stream, keys_d, nsyms, nodes_h[0].r);
        CNIIC_HIP_TRY(c, hipGetLastError());
        return CNIIC_OK;
    }
    const uint64_t nbits = payload_bytes * 8;
    if (nbits == 0) { *status = 1; return CNIIC_OK; }
    // the reference's walk, kHdLut bits at a time: node reached from the root and bits used
    std::vector<uint2> lut(1u << kHdLut);
    for (uint32_t pre = 0; pre < (1u << kHdLut); pre++) {
        uint32_t nd = 0, used = 0;
        while (used < (uint32_t)kHdLut && nodes_h[nd].l != kHdLeaf) {
            nd = ((pre >> (kHdLut - 1 - used)) & 1) ? nodes_h[nd].r : nodes_h[nd].l;
            used++;
        }
        lut[pre] = make_uint2(nd, used);
    }
    const uint64_t nsub = ceil_div(nbits, kHdSub);
    const uint64_t words = ceil_div(payload_bytes, 4) + 4;  // zero padding: hd_peek reads one word ahead
    DevBuf w_d, nodes_d, lut_d, end_a, end_b, end_c, count, off, tot, changed;
    CNIIC_HIP_TRY(c, w_d.alloc(words * 4));
    CNIIC_HIP_TRY(c, nodes_d.alloc(nodes_h.size() * 8));
    CNIIC_HIP_TRY(c, lut_d.alloc(lut.size() * 8));
    CNIIC_HIP_TRY(c, end_a.alloc(nsub * 8));
    CNIIC_HIP_TRY(c, end_b.alloc(nsub * 8));
    CNIIC_HIP_TRY(c, end_c.alloc(nsub * 8));
    CNIIC_HIP_TRY(c, count.alloc(nsub * 4));
    CNIIC_HIP_TRY(c, off.alloc(nsub * 8));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    CNIIC_HIP_TRY(c, changed.alloc(4));
    CNIIC_HIP_TRY(c, hipMemsetAsync(static_cast<uint8_t *>(w_d.p) + (words - 5) * 4, 0, 20, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(w_d.p, payload, payload_bytes, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(nodes_d.p, nodes_h.data(), nodes_h.size() * 8, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(lut_d.p, lut.data(), lut.size() * 8, hipMemcpyHostToDevice, c->stream));
    const uint32_t grid = (uint32_t)ceil_div(nsub, kHdThreads);
    uint64_t *bufs[3] = {end_a.as<uint64_t>(), end_b.as<uint64_t>(), end_c.as<uint64_t>()};
    uint64_t *cur = bufs[0], *prev = nullptr, *prev2 = nullptr;
    bool settled = false;
    for (int pass = 0; pass < kHdMaxPasses; pass++) {
        if (prev) CNIIC_HIP_TRY(c, hipMemsetAsync(changed.p, 0, 4, c->stream));
        hipLaunchKernelGGL(k_hd_pass, dim3(grid), dim3(kHdThreads), 0, c->stream, w_d.as<uint32_t>(), nbits, nodes_d.as<uint2>(),
                           lut_d.as<uint2>(), nsub, (const uint64_t *)prev, (const uint64_t *)prev2, cur, count.as<uint32_t>(), changed.as<uint32_t>());
        CNIIC_HIP_TRY(c, hipGetLastError());
        if (prev) {
            uint32_t ch = 1;
            CNIIC_HIP_TRY(c, hipMemcpyAsync(&ch, changed.p, 4, hipMemcpyDeviceToHost, c->stream));
            CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (!ch) { settled = true; break; }
        }
        prev2 = prev;
        prev = cur;
        cur = bufs[(pass + 1) % 3];
    }
    if (!settled) { *status = 2; CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream)); return CNIIC_OK; }
    // `cur` holds the settled boundaries (equal to prev's)
    hipLaunchKernelGGL(k_hd_offsets, dim3(1), dim3(1024), 0, c->stream, count.as<uint32_t>(), nsub, off.as<uint64_t>(), tot.as<uint64_t>());
    uint64_t total = 0;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&total, tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (total < nsyms) { *status = 1; return CNIIC_OK; }
    hipLaunchKernelGGL(k_hd_write, dim3(grid), dim3(kHdThreads), 0, c->stream, w_d.as<uint32_t>(), nbits, nodes_d.as<uint2>(),
                       lut_d.as<uint2>(), nsub, (const uint64_t *)cur, off.as<uint64_t>(), nsyms, keys_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));  // the scratch buffers go back to the pool
    return CNIIC_OK;
}

int keys_to_rgb(Ctx *c, const uint32_t *keys_d, uint64_t n, uint8_t *rgb_d) {
    if (!n) return CNIIC_OK;
    hipLaunchKernelGGL(k_keys_to_rgb, dim3((uint32_t)std::min<uint64_t>(ceil_div(n, 256), 8192)), dim3(256), 0, c->stream, keys_d, n, rgb_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// ---------------------------------------------------------------- FromDiff (hilbertc.rs:482-509): c_i = c_{i-1} + s_i, c_{-1} = 0
// per channel, as a prefix sum of the signed differences; a value outside 0..255 is the reference's
// `try_into().unwrap()` failure (:505-506), reported through *bad.
constexpr int kUdThreads = 256, kUdPer = 16;
constexpr uint32_t kUdChunk = kUdThreads * kUdPer;

__device__ __forceinline__ void ud_unpack(uint32_t key, int32_t d[3]) {
    d[0] = (int32_t)((key >> 18) & 511) - 255; d[1] = (int32_t)((key >> 9) & 511) - 255; d[2] = (int32_t)(key & 511) - 255;
}

__global__ __launch_bounds__(kUdThreads) void k_ud_sums(const uint32_t *__restrict__ keys, uint64_t n, int32_t *__restrict__ chunk_sum) {
    __shared__ int32_t sh[3][kUdThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kUdChunk + (uint64_t)threadIdx.x * kUdPer;
    int32_t s[3] = {0, 0, 0};
    for (int j = 0; j < kUdPer; j++)
        if (base + j < n) { int32_t d[3]; ud_unpack(keys[base + j], d); s[0] += d[0]; s[1] += d[1]; s[2] += d[2]; }
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        int32_t v = s[ch];
        v = (int32_t)wave_reduce_sum((uint32_t)v);
        if ((threadIdx.x & 63) == 0) sh[ch][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) chunk_sum[3 * (size_t)blockIdx.x + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

// single block: exclusive scan of the chunk sums, in place, per channel
__global__ __launch_bounds__(1024) void k_ud_scan(int32_t *__restrict__ chunk_sum, uint32_t nchunks) {
    __shared__ int32_t sh[3][1024];
    const uint32_t per = (nchunks + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, nchunks);
    int32_t s[3] = {0, 0, 0};
    for (uint32_t i = lo; i < hi; i++)
        for (int ch = 0; ch < 3; ch++) s[ch] += chunk_sum[3 * (size_t)i + ch];
    for (int ch = 0; ch < 3; ch++) sh[ch][threadIdx.x] = s[ch];
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        int32_t v[3];
        for (int ch = 0; ch < 3; ch++) v[ch] = threadIdx.x >= o ? sh[ch][threadIdx.x - o] : 0;
        __syncthreads();
        for (int ch = 0; ch < 3; ch++) sh[ch][threadIdx.x] += v[ch];
        __syncthreads();
    }
    int32_t run[3];
    for (int ch = 0; ch < 3; ch++) run[ch] = sh[ch][threadIdx.x] - s[ch];
    for (uint32_t i = lo; i < hi; i++)
        for (int ch = 0; ch < 3; ch++) { const int32_t v = chunk_sum[3 * (size_t)i + ch]; chunk_sum[3 * (size_t)i + ch] = run[ch]; run[ch] += v; }
}

__global__ __launch_bounds__(kUdThreads) void k_ud_apply(const uint32_t *__restrict__ keys, uint64_t n, const int32_t *__restrict__ chunk_off,
                                                         uint8_t *__restrict__ lin, uint32_t *__restrict__ bad) {
    __shared__ int32_t wsum[3][kUdThreads / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * kUdChunk + (uint64_t)threadIdx.x * kUdPer;
    int32_t d[kUdPer][3], s[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < kUdPer; j++) {
        d[j][0] = d[j][1] = d[j][2] = 0;
        if (base + j < n) ud_unpack(keys[base + j], d[j]);
        s[0] += d[j][0]; s[1] += d[j][1]; s[2] += d[j][2];
    }
    int32_t run[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {  // exclusive scan of the per-thread sums across the block
        int32_t inc = s[ch];
        inc = (int32_t)wave_inclusive_scan((uint32_t)inc);
        if (lane == 63) wsum[ch][wid] = inc;
        run[ch] = inc - s[ch];
    }
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        for (int i = 0; i < wid; i++) run[ch] += wsum[ch][i];
        run[ch] += chunk_off[3 * (size_t)blockIdx.x + ch];
    }
    bool oob = false;
#pragma unroll
    for (int j = 0; j < kUdPer; j++) {
        if (base + j < n) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                run[ch] += d[j][ch];
                oob |= run[ch] < 0 || run[ch] > 255;
                lin[3 * (base + j) + ch] = (uint8_t)run[ch];
            }
        }
    }
    if (oob) *bad = 1u;
}

// keys_d: n packed SignedColor symbols in scan order -> lin_d: n colours (3 B each); *bad_h != 0: a colour left 0..255
int delta_undiff_dev(Ctx *c, const uint32_t *keys_d, uint64_t n, uint8_t *lin_d, uint32_t *bad_h) {
    *bad_h = 0;
    if (!n) return CNIIC_OK;
    const uint64_t nchunks64 = ceil_div(n, kUdChunk);
    if (nchunks64 > 0x7fffffffull) return c->fail(CNIIC_ERR_BAD_ARG, "undiff: too many symbols");
    const uint32_t nchunks = (uint32_t)nchunks64;
    DevBuf sums, bad;
    CNIIC_HIP_TRY(c, sums.alloc((uint64_t)nchunks * 12));
    CNIIC_HIP_TRY(c, bad.alloc(4));
    CNIIC_HIP_TRY(c, hipMemsetAsync(bad.p, 0, 4, c->stream));
    hipLaunchKernelGGL(k_ud_sums, dim3(nchunks), dim3(kUdThreads), 0, c->stream, keys_d, n, sums.as<int32_t>());
    hipLaunchKernelGGL(k_ud_scan, dim3(1), dim3(1024), 0, c->stream, sums.as<int32_t>(), nchunks);
    hipLaunchKernelGGL(k_ud_apply, dim3(nchunks), dim3(kUdThreads), 0, c->stream, keys_d, n, (const int32_t *)sums.as<int32_t>(), lin_d,
                       bad.as<uint32_t>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipMemcpyAsync(bad_h, bad.p, 4, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

}  // namespace cniic
