// k_hdecode.hip -- DecStream (src/huf.rs:187-206, 366-374) on gfx950: Huffman decoding of a whole payload in
// parallel, plus FromDiff (src/codec/hilbertc.rs:482-509) as a prefix sum.
//
// The stream has no markers, so where a symbol starts is only known by decoding everything before it.  Huffman
// codes resynchronise: a decoder started at a wrong bit position falls into step with the true symbol
// boundaries after a few symbols.  The payload is cut into subsequences of kHdSub bits, one per thread:
//   pass 0      every thread starts `warm` bits BEFORE its subsequence (a guess; exact for thread 0), decodes up to the
//               first symbol boundary inside it -- by then it has almost always fallen into step -- and notes that boundary
//               (start[t]); then on to the first boundary at or past the subsequence's end (end[t]), counting the symbols
//               that start in between
//   pass r > 0  thread t compares the boundary thread t-1 reported (end[t-1]) with the start it assumed: equal -> nothing
//               to do; different -> it decodes again from end[t-1]
//   until a pass changes nothing.  Thread 0 is exact from the start, and a thread whose start equals its predecessor's
//   end is exact if the predecessor is, so an unchanged pass is the exact chain from bit 0.  With the warm-up the first
//   check already finds every thread in step on ordinary streams: 1.25 decodes of the payload instead of one per pass.
//   Then: exclusive scan of the symbol counts -> every thread decodes once more and writes its symbols.
// Round 3: the walk is no longer the reference's node-by-node trie walk (one dependent random read per bit beyond the
// first twelve: 6 ms per pass over a 6.8 M-leaf code).  The decoder is the table of leaves in pre-order = ascending code
// order (huff_parse_leaves): the symbol at a bit position is the LAST leaf whose left-aligned code is <= the next 64 bits.
// A 12-bit table in LDS answers directly when one leaf covers the whole prefix (every code of up to 12 bits), else with
// the range of leaves to search; codes longer than that go through a second table on the top 20 bits (in L2) and a
// binary search over a handful of neighbouring leaves.  The block's stretch of the stream is staged in LDS, byte-swapped
// once, so a symbol costs two or three LDS reads and a shift.  Same symbols as the walk, bit for bit (tests).
#include "common.hpp"
#include "hilbert_scan.hpp"
#include "device_utils.hpp"
#include "huff_host.hpp"

namespace cniic {

constexpr int kHdLut = 12;
constexpr uint32_t kHdSub = 512;        // bits per thread: short enough that a 4096^2 image fills the machine (2.6 x 10^5 threads for 16 MB)
constexpr uint32_t kHdWarm = 480;       // bits staged before a block's first subsequence (a multiple of 32): the longest warm-up a launch may ask for (HdStream::warm)
constexpr int kHdThreads = 256;
constexpr int kHdMaxPasses = 48;
constexpr int kHdBlindChecks = 2;
constexpr uint32_t kHdDirect = 0x80000000u;   // second-table entry: x = symbol key, y = kHdDirect | length; else x = first leaf, y = leaves after it
constexpr uint32_t kHdSub3 = 0x40000000u;     // ... or x = first entry of the prefix's own third table, y = kHdSub3 | e: the next e bits pick its entry (key, length)
constexpr uint32_t kHd3MaxBits = 6;           // third tables of up to 2^6 entries (a prefix whose longest code runs further keeps the bisection)
constexpr uint32_t kHd3MaxLeaves = 96;        // leaves under a prefix its builder is willing to walk
constexpr uint32_t kHdTail = 4;               // words staged past the block's last subsequence (a symbol may run kLeafMaxLen bits over)
constexpr uint32_t kHdStageWords = kHdThreads * (kHdSub / 32) + kHdWarm / 32 + kHdTail;
// The staged stream is kept TRANSPOSED (round 5): word i of the stretch at (i mod 16) x kHdSwz + i / 16.  Thread t reads the words
// 16 t + k of its own subsequence; word for word in stream order the 64 lanes of a wave -- each topping up its window from its own
// subsequence -- hit two of the LDS's banks, a 32-way conflict on every read of every step (per block of pass 0: 47 us for 45 symbols, 24
// waves a CU queueing for the one LDS).  Transposed, lanes t, t + 1, ... at the same k read neighbouring words.
constexpr uint32_t kHdSwz = kHdStageWords / 16 + 1, kHdStageAlloc = 16 * kHdSwz;
template <uint32_t SWS> __device__ __forceinline__ uint32_t hd_sw(uint32_t i) { return (i & 15u) * SWS + (i >> 4); }
// first table (LDS, 4 bytes an entry -- an 8-byte entry per lane is two LDS passes and twice the footprint): a code of up to
// kHdLut bits answers key << 5 | length << 1 | 1 (keys have at most 27 bits); anything longer answers 0 and goes to the second table
constexpr size_t kHdLds = (4u << kHdLut) + kHdStageAlloc * 4;   // 16 KiB + 16.5 KiB

struct HdTables {
    const uint64_t *code;   // [n] left-aligned codes, ascending
    const uint32_t *key;    // [n]
    const uint8_t *len;     // [n]
    const uint32_t *lut1;   // [1 << kHdLut]
    const uint2 *lut2;      // [1 << bits2], or null when no code is longer than kHdLut bits
    uint32_t bits2;
    uint32_t n;
    const uint2 *lut3;      // third tables (k_hd_build_lut3), or null
    uint32_t use1;          // is the first table asked at all?  (the host's choice, see hd_use_first_table: without it the table is not
                            // even staged and a block needs half the LDS)
};
// The first table answers codes of up to kHdLut bits.  The differences of a photograph have next to none (1.6 % of the symbols of `delta`
// at 2048^2, 84 of 42 K leaves): every symbol paid an LDS look-up that said "ask the next table", and the table's 16 KiB held a CU to three
// blocks where six fit -- for a walk that is a chain of dependent reads, twice the waves in flight.  A stream whose symbols average more
// than 12.5 bits goes straight to the second table (which answers the short codes too).
__host__ inline bool hd_use_first_table(uint64_t payload_bits, uint64_t nsyms, uint32_t bits2) {
    return bits2 == 0 || payload_bits * 2 < nsyms * 25;
}

// the stream as the decoder sees it: 32-bit words from a 4-byte aligned address; the payload's first bit is bit `bit0`
// of that word sequence and its last one bit `nbits` - 1 (positions below are in that frame)
struct HdStream {
    const uint32_t *w;
    uint64_t nwords;   // words that may be read
    uint64_t bit0, nbits;
    uint32_t warm;     // bits of warm-up before a subsequence in pass 0 (at most kHdWarm)
};

// table entries for the `bits` top bits p: all 64-bit windows that begin with p lie between base and top
__global__ void k_hd_build_lut(const uint64_t *__restrict__ code, const uint32_t *__restrict__ key, const uint8_t *__restrict__ len, uint32_t n,
                               uint32_t bits, uint32_t *__restrict__ lut1, uint2 *__restrict__ lut2) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (1u << bits)) return;
    const uint64_t base = (uint64_t)p << (64 - bits), top = base | ((1ull << (64 - bits)) - 1ull);
    auto last_le = [&](uint64_t v) -> uint32_t {  // largest i with code[i] <= v (code[0] = 0: exists)
        uint32_t a = 0, b = n;                    // invariant: code[a] <= v, (b == n or code[b] > v)
        while (b - a > 1) { const uint32_t m = a + (b - a) / 2; if (code[m] <= v) a = m; else b = m; }
        return a;
    };
    // (the end of this entry's range from the start of the next one's: the lane above has it -- one search per entry instead of two)
    const uint32_t lo = last_le(base);
    uint32_t lo_next = (uint32_t)__shfl_down((int)lo, 1, 64);
    if ((threadIdx.x & 63) == 63 || p + 1 == (1u << bits)) lo_next = p + 1 == (1u << bits) ? n - 1 : last_le(top + 1);
    const uint32_t hi = p + 1 == (1u << bits) ? n - 1 : (code[lo_next] == top + 1 ? lo_next - 1 : lo_next);
    if (lut1) lut1[p] = lo == hi ? (key[lo] << 5) | ((uint32_t)len[lo] << 1) | 1u : 0u;
    else lut2[p] = lo == hi ? make_uint2(key[lo], kHdDirect | len[lo]) : make_uint2(lo, hi - lo);
}

// The third tables (round 5).  A second-table entry that holds a RANGE of leaves sent its lane into a bisection of the code array
// (two or three dependent reads) and then to the leaf's length and key (one more): 1.3 % of the symbols of a photograph's `delta` --
// and so 57 % of the steps of a wave of 64 lanes, each waiting for its slowest lane.  A prefix whose longest code runs at most
// kHd3MaxBits past the second table's bits gets a table of its own, 2^e entries (key, length), picked by the next e bits: ONE more read.
// Entries are dealt out of a pool by an atomic cursor (a prefix that finds the pool empty keeps its range).
__global__ __launch_bounds__(256) void k_hd_build_lut3(const uint64_t *__restrict__ code, const uint32_t *__restrict__ key, const uint8_t *__restrict__ len, uint32_t bits,
                                                       uint2 *__restrict__ lut2, uint2 *__restrict__ lut3, uint32_t cap3, uint32_t *__restrict__ cursor) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= (1u << bits)) return;
    const uint2 e = lut2[p];
    if (e.y & kHdDirect) return;
    const uint32_t lo = e.x, nl = e.y + 1;
    if (nl > kHd3MaxLeaves) return;
    uint32_t L = 0;
    for (uint32_t i = 0; i < nl; i++) L = max(L, (uint32_t)len[lo + i]);
    if (L <= bits || L - bits > kHd3MaxBits) return;
    const uint32_t eb = L - bits, sz = 1u << eb;
    const uint32_t at = atomicAdd(cursor, sz);
    if (at + sz > cap3) return;
    uint32_t a = lo;
    for (uint32_t j = 0; j < sz; j++) {
        const uint64_t win = ((uint64_t)p << (64 - bits)) | ((uint64_t)j << (64 - bits - eb));
        while (a + 1 < lo + nl && code[a + 1] <= win) a++;   // the last leaf whose code is <= the window (windows ascend with j)
        lut3[at + j] = make_uint2(key[a], (uint32_t)len[a]);
    }
    lut2[p] = make_uint2(at, kHdSub3 | eb);
}

// The same second table for a decoder of a million leaves, built from the LEAVES' side (round 3: 2^24 entries by binary search were
// 0.5 ms of a 4.5 ms decode).  Prefix p's first leaf lo(p) = the last leaf whose code is <= p's first window: leaf i owns the prefixes
// whose first window lies in its span [code[i], code[i + 1]), i.e. p in [ceil(code[i] / 2^S), ceil(code[i + 1] / 2^S)) -- every prefix
// has exactly one owner.  k_hd_lut_owner: leaf i writes itself into lo[] for its prefixes (a leaf that owns more than kHdOwnMax of
// them -- a short code -- leaves the rest to k_hd_lut_owner_big, a block per such leaf); k_hd_lut_entries: hi(p) from lo(p + 1) as above.
constexpr uint32_t kHdOwnMax = 64;
__device__ __forceinline__ uint64_t hd_first_prefix(uint64_t code, uint32_t S) { return (code >> S) + ((code & ((1ull << S) - 1ull)) ? 1ull : 0ull); }
__global__ __launch_bounds__(256) void k_hd_lut_owner(const uint64_t *__restrict__ code, uint32_t n, uint32_t bits, uint32_t *__restrict__ lo,
                                                      uint32_t *__restrict__ big, uint32_t *__restrict__ nbig, uint32_t big_cap) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t S = 64 - bits;
    const uint64_t a = hd_first_prefix(code[i], S), b = i + 1 < n ? hd_first_prefix(code[i + 1], S) : (1ull << bits);
    const uint64_t m = min(b, a + kHdOwnMax);
    for (uint64_t p = a; p < m; p++) lo[p] = i;
    if (b > m) { const uint32_t at = atomicAdd(nbig, 1u); if (at < big_cap) big[at] = i; }
}
__global__ __launch_bounds__(256) void k_hd_lut_owner_big(const uint64_t *__restrict__ code, uint32_t n, uint32_t bits, uint32_t *__restrict__ lo,
                                                          const uint32_t *__restrict__ big, const uint32_t *__restrict__ nbig, uint32_t big_cap) {
    const uint32_t nb = min(*nbig, big_cap), S = 64 - bits;
    for (uint32_t q = blockIdx.x; q < nb; q += gridDim.x) {
        const uint32_t i = big[q];
        const uint64_t a = hd_first_prefix(code[i], S) + kHdOwnMax, b = i + 1 < n ? hd_first_prefix(code[i + 1], S) : (1ull << bits);
        for (uint64_t p = a + threadIdx.x; p < b; p += 256) lo[p] = i;
    }
}
__global__ __launch_bounds__(256) void k_hd_lut_entries(const uint64_t *__restrict__ code, const uint32_t *__restrict__ key, const uint8_t *__restrict__ len, uint32_t n,
                                                        uint32_t bits, const uint32_t *__restrict__ lo_of, uint2 *__restrict__ lut2) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= (1u << bits)) return;
    const uint32_t lo = lo_of[p];
    uint32_t hi = n - 1;
    if (p + 1 < (1u << bits)) {
        const uint32_t ln = lo_of[p + 1];
        hi = code[ln] == ((uint64_t)(p + 1) << (64 - bits)) ? ln - 1 : ln;
    }
    lut2[p] = lo == hi ? make_uint2(key[lo], kHdDirect | len[lo]) : make_uint2(lo, hi - lo);
}

struct HdSym { uint32_t key, len; };

// the leaf of the 64-bit window `win` (its top 33 bits are stream bits, or all of it when WIDE)
template <bool KEY>
__device__ __forceinline__ HdSym hd_lookup(const HdTables &T, const uint32_t *lut_s, uint64_t win) {
    if (T.use1) {
        const uint32_t e1 = lut_s[win >> (64 - kHdLut)];
        if (e1) return HdSym{e1 >> 5, (e1 >> 1) & 15u};
    }
    const uint2 e = T.lut2[win >> (64 - T.bits2)];   // (a code longer than kHdLut bits exists, so the second table does)
    if (e.y & kHdDirect) return HdSym{e.x, e.y & 0xffu};
    if (e.y & kHdSub3) {
        const uint2 f = T.lut3[e.x + (uint32_t)((win << T.bits2) >> (64 - (e.y & 63u)))];
        return HdSym{f.x, f.y};
    }
    uint32_t a = e.x, b = e.x + e.y + 1;  // code[a] <= win; the answer is in [a, b)
    while (b - a > 1) { const uint32_t m = a + (b - a) / 2; if (T.code[m] <= win) a = m; else b = m; }
    return HdSym{KEY ? T.key[a] : 0u, (uint32_t)T.len[a]};
}

// Block prologue: the first table and the block's stretch of the stream into LDS (stage[i] = word i of the stretch, MSB-first).
// Returns the bit position (stream frame) of stage word 0.
__device__ __forceinline__ uint64_t hd_stage(const HdStream &S, const uint32_t *__restrict__ lut_g, uint32_t *lut_s, uint32_t *stage) {
    if (lut_g)
        for (uint32_t i = threadIdx.x; i < (1u << kHdLut); i += kHdThreads) lut_s[i] = lut_g[i];
    const uint64_t blk_bit = (uint64_t)blockIdx.x * kHdThreads * kHdSub;
    const int64_t w0 = (int64_t)(blk_bit / 32) - (int64_t)(kHdWarm / 32);   // negative for block 0: those words read as zero
    // (every word of the thread asked for before the first one is waited for: as a loop of load, swap, store the seventeen round trips
    // to memory came one behind the other, 11 us of a block's 146)
    constexpr uint32_t PER = (kHdStageWords + kHdThreads - 1) / kHdThreads;
    uint32_t wv[PER];
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) {
        const uint32_t i = threadIdx.x + r * kHdThreads;
        const int64_t wi = w0 + i;
        const bool in = i < kHdStageWords && wi >= 0 && (uint64_t)wi < S.nwords;
        wv[r] = S.w[in ? wi : 0];   // (word 0 exists: the payload is not empty; what must read as zero is zeroed below)
        if (!in) wv[r] = 0u;
    }
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) {
        const uint32_t i = threadIdx.x + r * kHdThreads;
        if (i < kHdStageWords) stage[hd_sw<kHdSwz>(i)] = __builtin_bswap32(wv[r]);
    }
    __syncthreads();
    return (uint64_t)(w0 * 32);  // (two's complement: position - base stays right for block 0)
}

// A thread's view of the staged stream: the next bits left-aligned in a register, topped up a word at a time -- one LDS read per
// 32 bits consumed instead of two or three per symbol.  WIDE (codes of more than 32 bits): the window is rebuilt from three
// words for every symbol, all 64 bits valid.
template <bool WIDE, uint32_t SWS = kHdSwz>
struct HdBits {
    const uint32_t *stage;
    uint64_t buf;      // !WIDE: bits from position `at`, the top nb of them valid
    uint32_t nb, wi;
    uint32_t rel;      // WIDE: at - base
    __device__ __forceinline__ uint32_t word(uint32_t i) const { return stage[hd_sw<SWS>(i)]; }
    __device__ __forceinline__ void seek(const uint32_t *st, uint32_t r) {
        stage = st;
        if (WIDE) { rel = r; return; }
        const uint32_t i = r >> 5, s = r & 31;
        buf = (((uint64_t)word(i) << 32) | word(i + 1)) << s;
        nb = 64 - s;
        wi = i + 2;
    }
    __device__ __forceinline__ uint64_t window() {
        if (WIDE) {
            const uint32_t i = rel >> 5, s = rel & 31;
            const uint64_t hi = ((uint64_t)word(i) << 32) | word(i + 1);
            return s ? (hi << s) | (word(i + 2) >> (32 - s)) : hi;
        }
        if (nb < 33) { buf |= (uint64_t)word(wi++) << (32 - nb); nb += 32; }
        return buf;
    }
    __device__ __forceinline__ void skip(uint32_t len) {
        if (WIDE) { rel += len; return; }
        buf <<= len;
        nb -= len;
    }
};

// decode from `at` until the first symbol boundary at or past `until`; cnt (COUNT) = the symbols met on the way.
// A symbol that would end past the stream's last bit ends the walk there (at = nbits): DecStream yields None.
// STORE (round 5): the symbols met are KEPT -- symbol number i of the walk goes to col[64 i] while i < kHdKeep, col = the
// subsequence's column of the kept-symbol rows (hd_keep_col) -- so that the decode that used to follow the offsets scan (every
// subsequence once more, each lane writing its own 150 bytes somewhere: 2.5 bytes moved per byte written, profiles/traffic.json
// round 5) becomes a copy (k_hd_compact).
constexpr uint32_t kHdKeep = 64;   // symbols kept per subsequence (a 512-bit subsequence of codes shorter than 8 bits has more: those are decoded again)
// The rows of a wave's 64 subsequences: symbols 4 g .. 4 g + 3 of subsequence t as ONE 16-byte group at [t / 64][g][t % 64] -- a lane stores
// a group every fourth symbol (a wave: 1 KiB in one piece).  A store per symbol cost more than its instruction: stores count in vmcnt
// like loads on gfx9, so the wait for the NEXT symbol's table entry was also a wait for the previous symbol's store to be acknowledged.
typedef uint32_t hd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t *hd_keep_col(uint32_t *keep, uint64_t t) { return keep + (t >> 6) * (uint64_t)(kHdKeep * 64) + (t & 63) * 4; }
__device__ __forceinline__ uint32_t hd_keep_at(uint32_t j) { return (j >> 2) * 256u + (j & 3u); }   // symbol j of a subsequence, from its hd_keep_col
template <bool WIDE, bool COUNT, bool STORE = false, uint32_t SWS = kHdSwz>
__device__ __forceinline__ void hd_run(const HdTables &T, const uint32_t *lut_s, const uint32_t *stage, uint64_t base, uint64_t nbits,
                                       uint64_t &at, uint64_t until, uint32_t &cnt, uint32_t *col = nullptr) {
    if (at >= until) return;
    HdBits<WIDE, SWS> B;
    B.seek(stage, (uint32_t)(at - base));
    uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    while (at < until) {
        if (at >= nbits) { at = nbits; break; }
        const HdSym sy = hd_lookup<STORE>(T, lut_s, B.window());
        if (at + sy.len > nbits) { at = nbits; break; }
        if (STORE) {
            const uint32_t q = cnt & 3u;
            k0 = q == 0 ? sy.key : k0; k1 = q == 1 ? sy.key : k1; k2 = q == 2 ? sy.key : k2; k3 = q == 3 ? sy.key : k3;
            if (q == 3 && cnt < kHdKeep) __builtin_nontemporal_store(hd_u32x4{k0, k1, k2, k3}, reinterpret_cast<hd_u32x4 *>(col + (size_t)(cnt >> 2) * 256));
        }
        if (COUNT) cnt++;
        at += sy.len;
        B.skip(sy.len);
    }
    if (STORE && (cnt & 3u) && cnt < kHdKeep) __builtin_nontemporal_store(hd_u32x4{k0, k1, k2, k3}, reinterpret_cast<hd_u32x4 *>(col + (size_t)(cnt >> 2) * 256));   // the last, partial group
}

// pass 0 (end_prev == null) and the checking passes.
//
// How fast a decoder that starts at a wrong bit falls into step depends on the code: the differences of a photograph (14 bits a
// symbol) need ~300 bits, the near-fixed-length code of 256 equally popular cluster colours far longer -- after a 128-bit
// warm-up 59 % of the `delta` subsequences are still out of step, and each further 512 bits cures 4 in 5 of them.  A pass per
// cure (every block staging again, a whole wave decoding for one lane) cost eight checks and seven decodes' worth of time.  So
// the BLOCK settles itself before it leaves: its threads' ends sit in LDS, whoever starts elsewhere than its predecessor ended
// goes on a work list, the list is decoded by as many threads as it has entries (dense waves, the stream still staged), and so
// on until the list is empty.  What remains between launches is the first thread of a block against the block before it.
constexpr int kHdMaxRounds = 320;   // (a cure moves at least one thread forward for good: 256 rounds settle any block)
constexpr int kHdRoundsShort = 24;
// of a block's out-of-step list still there after three rounds: the stream does not self-synchronise well enough for the settle loop.
// By measurement (tools/all_probe.py, photograph and uniform noise, `hufman` and `delta`, 512^2 .. 4096^2): a round is a chain of ~25
// dependent look-ups (12 us when every code goes to the second table) whatever the stream's size, the phase maps are 32 decodes of
// EVERYTHING -- short streams give up early, long ones late, streams of more than 2^19 subsequences (32 MiB of payload) never.
// (round 4) ... never, that is, unless the stream does not fall into step AT ALL: codes of one length (every colour once, a ramp of equally frequent
// colours) cure exactly one subsequence a round, 48 passes of 320 rounds later the phase maps ran anyway -- 425 ms for a 4096^2 image of 2^24
// different colours, 51 ms for a ramp.  A list that has kept 97 per cent of its length over three rounds is such a stream at any size.
__host__ inline uint32_t hd_hopeless_pct(uint64_t nsub) { return nsub <= (1ull << 16) ? 35u : nsub <= (1ull << 19) ? 60u : 97u; }
constexpr uint64_t kHdPhasesMaxSub = 1ull << 16;   // streams of up to this many subsequences (4 MiB) go to k_hd_phase_maps when the blind checks have not settled them
// -DCNIIC_HD_PHASES (a measuring build, tools/build_variant.sh): per block of pass 0, thread 0's wall clock (100 MHz) spent staging, in the
// first decode (until the whole block is through it), in the settle rounds, and the number of rounds; summed over the blocks
#ifdef CNIIC_HD_PHASES
__device__ unsigned long long g_hd_phase[8];
#define HD_T(i) do { if (threadIdx.x == 0 && !end_prev) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_hd_phase[i], now_ - t_ph_); t_ph_ = now_; } } while (0)
#define HD_C(i, v) do { if (threadIdx.x == 0 && !end_prev) atomicAdd(&g_hd_phase[i], (unsigned long long)(v)); } while (0)
#else
#define HD_T(i) do {} while (0)
#define HD_C(i, v) do {} while (0)
#endif
template <bool WIDE>
__global__ __launch_bounds__(kHdThreads) void k_hd_pass(HdStream S, HdTables Tg, uint64_t nsub, const uint64_t *__restrict__ end_prev,
                                                        uint64_t *__restrict__ end_out, uint64_t *__restrict__ start, uint32_t *__restrict__ count,
                                                        uint32_t *__restrict__ changed, int max_rounds, uint32_t hopeless_min, uint32_t hopeless_pct,
                                                        uint32_t *__restrict__ keep /* the kept-symbol rows (hd_keep_col), or null */) {
    extern __shared__ __align__(16) uint32_t hd_lds[];
    // (hopeless_min != 0: pass 0 counts in changed[5] the blocks whose threads do not fall into step -- see the settle loop -- and once
    // that many have said so, the checks and the write behind it return at once: the host takes k_hd_phase_maps after its one look)
    if (hopeless_min && end_prev && changed[5] >= hopeless_min) return;
    const HdTables &T = Tg;
    __shared__ unsigned long long s_end[kHdThreads], s_nstart[kHdThreads], s_nend[kHdThreads];
    __shared__ uint32_t s_ncnt[kHdThreads];
    __shared__ uint16_t s_list[kHdThreads];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_pred0;
    uint32_t *lut_s = hd_lds;
    uint32_t *stage = hd_lds + (T.use1 ? (1u << kHdLut) : 0u);   // (without the first table the launch brought LDS for the stream alone)
    static_assert(kHdSub - kHdWarm >= 32, "a warm-up never starts before the stream's first bit");
    const uint32_t tid = threadIdx.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * kHdThreads, t = t0 + tid;
    const bool live = t < nsub;
    const uint64_t lo = max(t * kHdSub, S.bit0), hi = min((t + 1) * kHdSub, S.nbits);
    uint64_t my_start = 0, my_end = 0, base = 0;
    uint32_t my_cnt = 0;
    uint64_t pred0 = 0;      // where the thread before the block's first one ended (pass 0: not known yet -- the first thread trusts its warm-up)
#ifdef CNIIC_HD_PHASES
    unsigned long long t_ph_ = wall_clock64();
#endif
    if (!end_prev) {
        base = hd_stage(S, T.use1 ? T.lut1 : nullptr, lut_s, stage);
        HD_T(0);
        if (live) {
            uint64_t at = S.bit0;
            if (t) {       // warm-up: from kHdWarm bits before the subsequence to the first boundary inside it
                at = t * kHdSub - S.warm;
                uint32_t dummy = 0;
                hd_run<WIDE, false>(T, lut_s, stage, base, S.nbits, at, lo, dummy);
            }
            my_start = at;
            if (keep) hd_run<WIDE, true, true>(T, lut_s, stage, base, S.nbits, at, hi, my_cnt, hd_keep_col(keep, t));
            else hd_run<WIDE, true>(T, lut_s, stage, base, S.nbits, at, hi, my_cnt);
            my_end = at;
        }
        pred0 = my_start;    // (read by thread 0 only)
    } else {
        // a check: who is out of step with the thread before?  Nobody, as a rule -- then the block is done without staging anything
        bool redo = false;
        if (live) {
            const uint64_t s = t ? end_prev[t - 1] : S.bit0;
            my_start = start[t]; my_end = end_prev[t]; my_cnt = count[t];
            redo = s != my_start;
            if (tid == 0) pred0 = s;
            if (!redo) end_out[t] = my_end;   // in step with the thread before: what it found stands (rewritten below if a cure moves it)
        }
        if (changed[2]) {  // measuring runs (CNIIC_HD_STATS): threads out of step / blocks that stage, per check
            const int nredo = __syncthreads_count(redo);
            if (tid == 0 && nredo) { atomicAdd(&changed[3], (uint32_t)nredo); atomicAdd(&changed[4], 1u); }
            if (!nredo) return;
        } else if (!__syncthreads_or(redo)) return;
        base = hd_stage(S, T.use1 ? T.lut1 : nullptr, lut_s, stage);
    }
    // ---- the block settles itself
    s_end[tid] = live ? my_end : ~0ull;
    if (tid == 0) s_pred0 = pred0;
    const uint64_t end_at_entry = my_end;
    // (max_rounds < kHdMaxRounds -- a short stream of codes of at most 32 bits: a block that is not in step after kHdRoundsShort rounds, where
    // an ordinary stream needs a handful, belongs to a stream that does not fall into step at all; it stops curing one subsequence per
    // round and says so, and the host takes k_hd_phase_maps.  Such a block used to spend its 256 rounds, 2 ms, in each of three passes.)
    bool gave_up = true;
    uint32_t nl0 = 0;
    for (int round = 0; round < max_rounds; round++) {
        if (tid == 0) s_n = 0;
        __syncthreads();                                   // s_end of the round before is in place
        if (round == 0) HD_T(1); else HD_T(2);
        const uint64_t pred = tid ? s_end[tid - 1] : s_pred0;
        const bool redo = live && pred != my_start;
        if (redo) s_list[atomicAdd(&s_n, 1u)] = (uint16_t)tid;
        __syncthreads();
        const uint32_t nl = s_n;
        HD_C(4, nl);
        if (nl == 0) { gave_up = false; HD_C(3, round); HD_C(5, 1); break; }
        // Does this stream fall into step at all?  Every round re-decodes the listed subsequences from where their predecessors ended; on
        // an ordinary stream that cures four in five of them, on a near-fixed-length code next to none, and the loop becomes a chain of
        // one cure per round.  Curing a fraction p per round costs list / p decodes and ln(list) / p rounds per pass, the phase maps 32
        // decodes and no chain: a block whose list has kept hopeless_pct per cent of its length over three rounds stops here and says so.
        if (hopeless_min && !end_prev) {
            if (round == 0) nl0 = nl;
            else if ((round == 3 && nl0 >= 64 && nl * 100 > nl0 * hopeless_pct) ||
                     (round == kHdRoundsShort && nl0 >= 64 && nl * 2 > nl0)) {   // (round 4: ... or half of it over 24 rounds: codes of nearly one length
                                                                                 // cure a few entries a round, and the block would chain through all 256)
                if (tid == 0) atomicAdd(&changed[5], 1u);
                break;
            }
        }
        if (tid < nl) {                                    // entry tid of the list: subsequence j again, from where its predecessor ended
            const uint32_t j = s_list[tid];
            const uint64_t tj = t0 + j, hj = min((tj + 1) * kHdSub, S.nbits);
            uint64_t at = j ? s_end[j - 1] : s_pred0;
            uint32_t cn = 0;
            s_nstart[j] = at;
            if (keep) hd_run<WIDE, true, true>(T, lut_s, stage, base, S.nbits, at, hj, cn, hd_keep_col(keep, tj));
            else hd_run<WIDE, true>(T, lut_s, stage, base, S.nbits, at, hj, cn);
            s_nend[j] = at;
            s_ncnt[j] = cn;
        }
        __syncthreads();                                   // every s_end[j - 1] has been read
        if (redo) { my_start = s_nstart[tid]; my_end = s_nend[tid]; my_cnt = s_ncnt[tid]; s_end[tid] = my_end; }
    }
    HD_T(2);
    if (!live) return;
    start[t] = my_start;
    end_out[t] = my_end;
    count[t] = my_cnt;
    // Only a moved END matters to anybody outside the block.  If no end moves in a pass, every thread's start equals its
    // predecessor's end and the chain from the first bit is exact.
    if (end_prev && (my_end != end_at_entry || gave_up)) *changed = 1u;
}

// ---------------------------------------------------------------- streams that do not fall into step (round 3)
// A code whose words are nearly all the same length -- uniform noise over a small alphabet: `delta` on the 512 x 512 U image, 33 K
// symbols of 15 bits -- does not self-synchronise: a decoder that starts on a wrong bit stays wrong, the settle loop above cures one
// subsequence per round, and the decode is a chain (18-25 ms for 262 K symbols, nine checks).  For such a stream (codes of at most 32
// bits) every subsequence is decoded from EACH of the 32 bit positions its first symbol can start at (k_hd_phase_maps: where the walk
// enters the next subsequence and how many symbols it met), the maps are composed along the stream (k_hd_phase_chain: per group of
// subsequences, then the 512 groups by one thread, then inside the groups with the entries known), and every subsequence's true
// start and count fall out -- what k_hd_write needs.  32 times the first pass's work, no chain: taken when the blind checks have not
// settled the stream (CNIIC_HD_PHASES=1: always, for the tests).
constexpr uint32_t kHpGroups = 512;   // groups of subsequences whose maps k_hd_phase_chain composes (one thread each)
// (a subsequence is entered at most max_len - 1 bits past its start -- the overshoot of the symbol that straddles its boundary -- so
// only `phases` = max(16, longest code) of the 32 are walked: a block's 1024 threads take 1024 / phases subsequences)
constexpr uint32_t kHpSubsMax = 64, kHpThreads = 1024, kHpStageWords = kHpSubsMax * (kHdSub / 32) + kHdTail + 2, kHpSwz = kHpStageWords / 16 + 1;
__global__ __launch_bounds__(kHpThreads) void k_hd_phase_maps(HdStream S, HdTables Tg, uint64_t nsub, uint8_t *__restrict__ maps /* [nsub][32] */,
                                                              uint16_t *__restrict__ cnts /* [nsub][32] */, uint32_t phases, uint32_t subs_per_block) {
    const HdTables &T = Tg;
    __shared__ uint32_t lut_s[1u << kHdLut];
    __shared__ uint32_t stage[16 * kHpSwz];   // (transposed like k_hd_pass's: one HdBits)
    if (T.use1)
        for (uint32_t i = threadIdx.x; i < (1u << kHdLut); i += kHpThreads) lut_s[i] = T.lut1[i];
    const uint64_t t0 = (uint64_t)blockIdx.x * subs_per_block, base = t0 * kHdSub, w0 = base / 32;
    const uint32_t nstage = subs_per_block * (kHdSub / 32) + kHdTail + 2;
    for (uint32_t i = threadIdx.x; i < nstage; i += kHpThreads) stage[hd_sw<kHpSwz>(i)] = w0 + i < S.nwords ? __builtin_bswap32(S.w[w0 + i]) : 0u;
    __syncthreads();
    const uint32_t sub = threadIdx.x / phases, o = threadIdx.x - sub * phases;
    const uint64_t t = t0 + sub;
    if (sub >= subs_per_block || t >= nsub) return;
    const uint64_t next = (t + 1) * kHdSub, hi = min(next, S.nbits);
    uint64_t at = t * kHdSub + o;
    uint32_t cn = 0;
    if (at < hi) hd_run<false, true, false, kHpSwz>(T, lut_s, stage, base, S.nbits, at, hi, cn);
    maps[t * 32 + o] = (uint8_t)(at >= next ? min<uint64_t>(at - next, 31) : 0);   // (a walk that ends with the stream enters nothing)
    cnts[t * 32 + o] = (uint16_t)cn;
}
__global__ __launch_bounds__(kHpGroups) void k_hd_phase_chain(const uint8_t *__restrict__ maps, const uint16_t *__restrict__ cnts, uint64_t nsub, uint64_t nbits, uint32_t bit0,
                                                         uint64_t *__restrict__ start, uint32_t *__restrict__ count) {
    __shared__ uint8_t s_row[kHpGroups][36];   // a thread's current map (rows padded: bank spread)
    __shared__ uint8_t s_g[kHpGroups][32];     // the groups' composed maps
    __shared__ uint8_t s_entry[kHpGroups + 1];
    const uint32_t j = threadIdx.x;
    const uint64_t G = (nsub + kHpGroups - 1) / kHpGroups, lo = min<uint64_t>(j * G, nsub), hi = min<uint64_t>(lo + G, nsub);
    uint8_t f[32];
#pragma unroll
    for (int k = 0; k < 32; k++) f[k] = (uint8_t)k;   // identity
    for (uint64_t t = lo; t < hi; t++) {
        const uint4 a = reinterpret_cast<const uint4 *>(maps + t * 32)[0], b = reinterpret_cast<const uint4 *>(maps + t * 32)[1];
        uint32_t *row = reinterpret_cast<uint32_t *>(s_row[j]);
        row[0] = a.x; row[1] = a.y; row[2] = a.z; row[3] = a.w; row[4] = b.x; row[5] = b.y; row[6] = b.z; row[7] = b.w;
#pragma unroll
        for (int k = 0; k < 32; k++) f[k] = s_row[j][f[k]];
    }
#pragma unroll
    for (int k = 0; k < 32; k++) s_g[j][k] = f[k];
    __syncthreads();
    if (j == 0) {
        uint32_t e = bit0;   // the payload's first symbol starts at its first bit
        for (uint32_t g = 0; g < kHpGroups; g++) { s_entry[g] = (uint8_t)e; e = s_g[g][e]; }
    }
    __syncthreads();
    uint32_t e = s_entry[j];
    for (uint64_t t = lo; t < hi; t++) {
        start[t] = min(t * kHdSub + e, nbits);
        count[t] = cnts[t * 32 + e];
        e = maps[t * 32 + e];
    }
}

// FromDiff's chunks (below): the channel sums of the differences of every 4096 symbols are what its prefix scan starts from.  Where the
// decoder writes packed differences (MODE 0) it can add them up on its way -- chunk_sum[3 chunk + channel], zero before the launch --
// and save the undiff a pass over all the symbols (1.07 GB read at 16384^2).
constexpr int kUdThreads = 256, kUdPer = 16;
constexpr uint32_t kUdChunk = kUdThreads * kUdPer;
static_assert(kUdChunk == 4096, "chunk of a position p: p >> 12");
__device__ __forceinline__ void ud_unpack(uint32_t key, int32_t d[3]) {
    d[0] = (int32_t)((key >> 18) & 511) - 255; d[1] = (int32_t)((key >> 9) & 511) - 255; d[2] = (int32_t)(key & 511) - 255;
}

// every thread decodes its symbols once more and writes them: MODE 0 = packed keys (u32 each), 1 = RGB bytes (3 each).
// Four symbols leave together (16 / 12 bytes at an aligned address) where the symbol index allows: a store per symbol is one
// L2 request per lane, 16.7 M of them at 4096^2.
template <bool WIDE, int MODE>
__global__ __launch_bounds__(kHdThreads) void k_hd_write(HdStream S, HdTables Tg, uint64_t nsub, const uint64_t *__restrict__ start,
                                                         const uint64_t *__restrict__ off, uint64_t nsyms, void *__restrict__ out,
                                                         const uint32_t *__restrict__ hopeless /* null, or the count and its bound: see k_hd_pass */, uint32_t hopeless_min,
                                                         const uint32_t *__restrict__ only_long = nullptr /* counts: only the subsequences of more than kHdKeep symbols (the others were copied) */,
                                                         int32_t *__restrict__ chunk_sum = nullptr /* MODE 0: see above */) {
    extern __shared__ __align__(16) uint32_t hd_lds[];
    if (hopeless && *hopeless >= hopeless_min) return;
    const HdTables &T = Tg;
    uint32_t *lut_s = hd_lds;
    uint32_t *stage = hd_lds + (T.use1 ? (1u << kHdLut) : 0u);
    const uint64_t t = (uint64_t)blockIdx.x * kHdThreads + threadIdx.x;
    if (only_long && !__syncthreads_or(t < nsub && only_long[t] > kHdKeep)) return;   // (the rule: nobody -- nothing is staged)
    const uint64_t base = hd_stage(S, T.use1 ? T.lut1 : nullptr, lut_s, stage);
    if (t >= nsub) return;
    if (only_long && only_long[t] <= kHdKeep) return;
    const uint64_t hi = min((t + 1) * kHdSub, S.nbits);
    uint64_t at = start[t], idx = off[t], widx = idx;   // idx: next symbol to be stored, widx: next to be decoded
    if (at >= hi) return;
    uint32_t *keys = static_cast<uint32_t *>(out);
    uint8_t *rgb = static_cast<uint8_t *>(out);
    HdBits<WIDE> B;
    B.seek(stage, (uint32_t)(at - base));
    int32_t cs[3] = {0, 0, 0};          // the thread's differences summed, of the chunk it is in (a subsequence spans two at most)
    uint64_t cs_chunk = idx >> 12;
    auto cs_flush = [&]() {
        if (cs[0]) atomicAdd(&chunk_sum[3 * cs_chunk], cs[0]);
        if (cs[1]) atomicAdd(&chunk_sum[3 * cs_chunk + 1], cs[1]);
        if (cs[2]) atomicAdd(&chunk_sum[3 * cs_chunk + 2], cs[2]);
        cs[0] = cs[1] = cs[2] = 0;
    };
    auto next = [&](uint32_t &key) -> bool {   // the reference reads exactly nsyms symbols; what the padding decodes to is dropped
        if (at >= hi || widx >= nsyms) return false;
        const HdSym sy = hd_lookup<true>(T, lut_s, B.window());
        if (at + sy.len > S.nbits) { at = hi; return false; }
        at += sy.len;
        B.skip(sy.len);
        if (MODE == 0 && chunk_sum) {
            if ((widx >> 12) != cs_chunk) { cs_flush(); cs_chunk = widx >> 12; }
            int32_t d[3];
            ud_unpack(sy.key, d);
            cs[0] += d[0]; cs[1] += d[1]; cs[2] += d[2];
        }
        widx++;
        key = sy.key;
        return true;
    };
    auto put1 = [&](uint32_t key) {
        if (MODE == 0) keys[idx] = key;
        else { rgb[3 * idx] = (uint8_t)(key >> 16); rgb[3 * idx + 1] = (uint8_t)(key >> 8); rgb[3 * idx + 2] = (uint8_t)key; }
        idx++;
    };
    uint32_t k0, k1, k2, k3;
    while ((idx & 3) && next(k0)) put1(k0);
    for (;;) {
        if (!next(k0)) break;
        if (!next(k1)) { put1(k0); break; }
        if (!next(k2)) { put1(k0); put1(k1); break; }
        if (!next(k3)) { put1(k0); put1(k1); put1(k2); break; }
        // (next() advanced `at` and widx only: idx is still the first of the four)
        if (MODE == 0) {
            *reinterpret_cast<uint4 *>(keys + idx) = make_uint4(k0, k1, k2, k3);
        } else {  // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3, little-endian words
            uint32_t *dst = reinterpret_cast<uint32_t *>(rgb + 3 * idx);
            const uint32_t w0 = ((k0 >> 16) & 255) | (((k0 >> 8) & 255) << 8) | ((k0 & 255) << 16) | (((k1 >> 16) & 255) << 24);
            const uint32_t w1 = ((k1 >> 8) & 255) | ((k1 & 255) << 8) | (((k2 >> 16) & 255) << 16) | (((k2 >> 8) & 255) << 24);
            const uint32_t w2 = (k2 & 255) | (((k3 >> 16) & 255) << 8) | (((k3 >> 8) & 255) << 16) | ((k3 & 255) << 24);
            dst[0] = w0; dst[1] = w1; dst[2] = w2;
        }
        idx += 4;
    }
    if (MODE == 0 && chunk_sum) cs_flush();
}

// The kept symbols to their places (round 5): block b = the subsequences [256 b, 256 b + 256), whose symbols are the output positions
// [off[256 b], off[256 b + 256)) -- a thread per GROUP of four consecutive positions (16 / 12 bytes at an aligned address), the owner of
// a position found by bisection of the block's 257 offsets in LDS, its symbol read from the owner's column: the loads of a wave fall on
// the few rows its neighbours' lanes read too (cache), the stores are whole lines.  Subsequences of more than kHdKeep symbols are left
// to k_hd_write(only_long).
template <int MODE>
__global__ __launch_bounds__(kHdThreads) void k_hd_compact(const uint32_t *__restrict__ keep, const uint64_t *__restrict__ off, const uint32_t *__restrict__ count, uint64_t nsub,
                                                           uint64_t nsyms, void *__restrict__ out, const uint32_t *__restrict__ hopeless, uint32_t hopeless_min,
                                                           int32_t *__restrict__ chunk_sum /* MODE 0: FromDiff's chunk sums, or null */) {
    if (hopeless && *hopeless >= hopeless_min) return;
    __shared__ unsigned long long s_off[kHdThreads + 1];
    __shared__ uint32_t s_cnt[kHdThreads];
    const uint64_t t0 = (uint64_t)blockIdx.x * kHdThreads;
    const uint32_t nt = (uint32_t)min<uint64_t>(kHdThreads, nsub - t0), tid = threadIdx.x;
    if (tid < nt) { s_off[tid] = off[t0 + tid]; s_cnt[tid] = count[t0 + tid]; }
    __syncthreads();
    if (tid == 0) s_off[nt] = s_off[nt - 1] + s_cnt[nt - 1];
    __syncthreads();
    const uint64_t lo = s_off[0], hi = min(s_off[nt], nsyms);   // (the reference reads exactly nsyms symbols; what the padding decodes to is dropped)
    if (lo >= hi) return;
    uint32_t *keys = static_cast<uint32_t *>(out);
    uint8_t *rgb = static_cast<uint8_t *>(out);
    for (uint64_t g0 = lo >> 2; g0 * 4 < hi; g0 += kHdThreads) {   // (every lane goes round as often as the block's first: the wave sums below want them all)
        const uint64_t g = g0 + tid, p0 = g * 4;
        uint32_t k[4] = {0, 0, 0, 0};
        bool have[4] = {false, false, false, false};
        if (p0 < hi) {
        // owner of the group's first position inside the block: the last subsequence whose offset is <= it (empty ones share an offset: the last wins)
        uint32_t a = 0, b = nt;
        const uint64_t pf = max(p0, lo);
        while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (s_off[m] <= pf) a = m; else b = m; }
        uint32_t own = a;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t p = p0 + u;
            if (p < lo || p >= hi) continue;
            while (p >= s_off[own + 1]) own++;          // (p < s_off[nt]: ends at a subsequence that holds it)
            if (s_cnt[own] > kHdKeep) continue;         // decoded again by k_hd_write(only_long)
            k[u] = hd_keep_col(const_cast<uint32_t *>(keep), t0 + own)[hd_keep_at((uint32_t)(p - s_off[own]))];
            have[u] = true;
        }
        }
        if (MODE == 0 && chunk_sum) {   // a wave's 256 positions lie in one chunk of 4096, or in two
            int32_t d[3] = {0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (have[u]) { int32_t e[3]; ud_unpack(k[u], e); d[0] += e[0]; d[1] += e[1]; d[2] += e[2]; }
            const uint32_t ch = (uint32_t)(p0 >> 12), ch0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch);
            const bool first = ch == ch0;
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int32_t a = (int32_t)wave_reduce_sum((uint32_t)(first ? d[q] : 0)), b = (int32_t)wave_reduce_sum((uint32_t)(first ? 0 : d[q]));
                if (lane == 0 && a) atomicAdd(&chunk_sum[3 * (size_t)ch0 + q], a);
                if (lane == 0 && b) atomicAdd(&chunk_sum[3 * (size_t)(ch0 + 1) + q], b);
            }
        }
        if (have[0] && have[1] && have[2] && have[3]) {
            if (MODE == 0) *reinterpret_cast<uint4 *>(keys + p0) = make_uint4(k[0], k[1], k[2], k[3]);
            else {
                uint32_t *dst = reinterpret_cast<uint32_t *>(rgb + 3 * p0);
                dst[0] = ((k[0] >> 16) & 255) | (((k[0] >> 8) & 255) << 8) | ((k[0] & 255) << 16) | (((k[1] >> 16) & 255) << 24);
                dst[1] = ((k[1] >> 8) & 255) | ((k[1] & 255) << 8) | (((k[2] >> 16) & 255) << 16) | (((k[2] >> 8) & 255) << 24);
                dst[2] = (k[2] & 255) | (((k[3] >> 16) & 255) << 8) | (((k[3] >> 8) & 255) << 16) | ((k[3] & 255) << 24);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!have[u]) continue;
                const uint64_t p = p0 + u;
                if (MODE == 0) keys[p] = k[u];
                else { rgb[3 * p] = (uint8_t)(k[u] >> 16); rgb[3 * p + 1] = (uint8_t)(k[u] >> 8); rgb[3 * p + 2] = (uint8_t)k[u]; }
            }
        }
    }
}

__global__ void k_hd_fill(uint32_t *__restrict__ keys, uint64_t n, uint32_t key) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) keys[i] = key;
}
__global__ void k_hd_fill_rgb(uint8_t *__restrict__ rgb, uint64_t n, uint32_t key) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        rgb[3 * i] = (uint8_t)(key >> 16); rgb[3 * i + 1] = (uint8_t)(key >> 8); rgb[3 * i + 2] = (uint8_t)key;
    }
}

// packed RGB keys -> interleaved bytes
__global__ void k_keys_to_rgb(const uint32_t *__restrict__ keys, uint64_t n, uint8_t *__restrict__ rgb) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t k = keys[i];
        rgb[3 * i] = (uint8_t)(k >> 16); rgb[3 * i + 1] = (uint8_t)(k >> 8); rgb[3 * i + 2] = (uint8_t)k;
    }
}

struct LeafMeta { uint32_t max_len; };

// The decoder as a table of leaves already in HBM (tab_d: code u64[n] | key u32[n] at off_key | len u8[n] at off_len).
// payload: the bit stream, in HOST memory or -- payload_dev -- anywhere in HBM (any alignment: it is read as words from the 4-byte
// boundary below it).  mode 0: out_d receives nsyms packed keys (u32, 16-byte aligned); mode 1: nsyms RGB triples (u8, 4-byte aligned).
// *status: 0 = decoded, 1 = the stream ends early (None), 2 = did not settle (caller decodes on the host)
int huff_decode_tables_dev(Ctx *c, const uint8_t *tab_d, uint64_t n, uint64_t off_key, uint64_t off_len, uint32_t max_len, uint32_t first_key,
                           const uint8_t *payload, bool payload_dev, uint64_t payload_bytes, uint64_t nsyms, int mode, void *out_d, int *status, UdSums *sums) {
    *status = 0;
    if (sums) sums->filled = false;
    if (nsyms == 0) return CNIIC_OK;
    if (n == 0 || max_len > kLeafMaxLen) return c->fail(CNIIC_ERR_BAD_ARG, "huff_decode_dev: no usable leaf table");
    if (n == 1) {  // one symbol, zero-length code, no payload (huf.rs:140-142)
        if (mode == 0) hipLaunchKernelGGL(k_hd_fill, dim3(1024), dim3(256), 0, c->stream, static_cast<uint32_t *>(out_d), nsyms, first_key);
        else hipLaunchKernelGGL(k_hd_fill_rgb, dim3(1024), dim3(256), 0, c->stream, static_cast<uint8_t *>(out_d), nsyms, first_key);
        CNIIC_HIP_TRY(c, hipGetLastError());
        return CNIIC_OK;
    }
    if (payload_bytes == 0) { *status = 1; return CNIIC_OK; }
    if ((reinterpret_cast<uintptr_t>(out_d) & (mode == 0 ? 15u : 3u)) != 0) return c->fail(CNIIC_ERR_BAD_ARG, "huff_decode_dev: misaligned output");
    // ---- the stream as aligned words
    DevBuf w_d;
    HdStream S{};
    if (payload_dev) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(payload);
        S.w = reinterpret_cast<const uint32_t *>(a & ~uintptr_t(3));
        S.bit0 = (a & 3) * 8;
    } else {
        CNIIC_HIP_TRY(c, w_d.alloc((payload_bytes + 3) & ~3ull));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(w_d.p, payload, payload_bytes, hipMemcpyHostToDevice, c->stream));
        S.w = w_d.as<uint32_t>();
        S.bit0 = 0;
    }
    S.nbits = S.bit0 + payload_bytes * 8;
    S.nwords = ceil_div(S.nbits, 32);
    // The warm-up.  A thread that has not fallen into step by the time it enters its subsequence is decoded again in the block's settle
    // loop, and a RUN of such threads one round per member: every round is a chain as long as the whole first decode with a handful of
    // lanes at work.  Codes of 14 bits need ~300 bits to fall into step: 128 bits of warm-up left a third of the threads for the rounds.
    S.warm = 128;
    if (payload_bytes * 8 >= nsyms * 12) S.warm = 384;   // long codes (the differences of a photograph)
    if (const char *e = test_env("CNIIC_HD_WARM")) S.warm = std::min<uint32_t>(kHdWarm, std::max(32, atoi(e)) & ~31u);
    // ---- the look-up tables
    // The second table, measured (round 3): 2^18 entries = 2 MiB, which stays in an XCD's L2 -- `delta` 16384^2 (54 K leaves, 14.5 bits a
    // symbol) decodes in 12.3 / 10.0 / 8.8 / 11.0 / 17.9 ms with 14 / 16 / 18 / 20 / 24 bits; for a decoder of a million leaves and more
    // 2^24 entries = 128 MiB -- there even 20 bits leave half a dozen leaves per entry, i.e. a few more dependent reads for EVERY symbol:
    // `hufman` 4096^2 (6.8 M leaves) 7.0 / 5.3 / 4.6 / 4.6 ms with 20 / 22 / 24 / 26.  CNIIC_HD_LUT2_BITS: the cap, for measurements.
    const uint32_t cap2 = test_env("CNIIC_HD_LUT2_BITS") ? (uint32_t)atoi(test_env("CNIIC_HD_LUT2_BITS")) : (n >= (1u << 20) ? 24u : 18u);
    const uint32_t bits2 = max_len > (uint32_t)kHdLut ? std::min<uint32_t>(max_len, std::max(cap2, (uint32_t)kHdLut + 1)) : 0u;
    DevBuf lut1_d, lut2_d, lut3_d, cnt_d;
    CNIIC_HIP_TRY(c, lut1_d.alloc((4ull << kHdLut)));
    CNIIC_HIP_TRY(c, cnt_d.alloc(4));   // the third tables' cursor
    CNIIC_HIP_TRY(c, hipMemsetAsync(cnt_d.p, 0, 4, c->stream));
    bool use1 = hd_use_first_table(payload_bytes * 8, nsyms, bits2);
    if (const char *e = test_env("CNIIC_HD_LUT1")) use1 = bits2 == 0 || atoi(e) != 0;   // (tests: either way round)
    HdTables T{reinterpret_cast<const uint64_t *>(tab_d), reinterpret_cast<const uint32_t *>(tab_d + off_key), tab_d + off_len, lut1_d.as<uint32_t>(), nullptr, bits2, (uint32_t)n,
               nullptr, use1 ? 1u : 0u};
    if (use1)
        hipLaunchKernelGGL(k_hd_build_lut, dim3((1u << kHdLut) / 256), dim3(256), 0, c->stream, T.code, T.key, T.len, T.n, (uint32_t)kHdLut, lut1_d.as<uint32_t>(),
                           (uint2 *)nullptr);
    DevBuf lo_d, big_d;
    if (bits2) {
        CNIIC_HIP_TRY(c, lut2_d.alloc(8ull << bits2));
        if (bits2 > 20 && !test_env("CNIIC_HD_LUT2_SEARCH")) {   // a large table: from the leaves' side (see k_hd_lut_owner)
            const uint32_t big_cap = (1u << bits2) / kHdOwnMax + 1;   // (leaves that own more than kHdOwnMax prefixes: at most this many)
            CNIIC_HIP_TRY(c, lo_d.alloc(4ull << bits2));
            CNIIC_HIP_TRY(c, big_d.alloc(((uint64_t)big_cap + 1) * 4));
            CNIIC_HIP_TRY(c, hipMemsetAsync(big_d.p, 0, 4, c->stream));
            hipLaunchKernelGGL(k_hd_lut_owner, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, c->stream, T.code, T.n, bits2, lo_d.as<uint32_t>(), big_d.as<uint32_t>() + 1,
                               big_d.as<uint32_t>(), big_cap);
            hipLaunchKernelGGL(k_hd_lut_owner_big, dim3(512), dim3(256), 0, c->stream, T.code, T.n, bits2, lo_d.as<uint32_t>(), (const uint32_t *)big_d.as<uint32_t>() + 1,
                               (const uint32_t *)big_d.as<uint32_t>(), big_cap);
            hipLaunchKernelGGL(k_hd_lut_entries, dim3((1u << bits2) / 256), dim3(256), 0, c->stream, T.code, T.key, T.len, T.n, bits2, (const uint32_t *)lo_d.as<uint32_t>(),
                               lut2_d.as<uint2>());
        } else {
            hipLaunchKernelGGL(k_hd_build_lut, dim3((1u << bits2) / 256), dim3(256), 0, c->stream, T.code, T.key, T.len, T.n, bits2, (uint32_t *)nullptr,
                               lut2_d.as<uint2>());
        }
        T.lut2 = lut2_d.as<uint2>();
        if (!(test_env("CNIIC_HD_LUT3") && !atoi(test_env("CNIIC_HD_LUT3")))) {   // the third tables (k_hd_build_lut3) out of a pool of 4 n + 2^16 entries, 2^24 at most
            const uint32_t cap3 = (uint32_t)std::min<uint64_t>(4 * n + 65536, 1ull << 24);
            if (lut3_d.alloc((uint64_t)cap3 * 8) == hipSuccess) {
                hipLaunchKernelGGL(k_hd_build_lut3, dim3((1u << bits2) / 256), dim3(256), 0, c->stream, T.code, T.key, T.len, bits2, lut2_d.as<uint2>(), lut3_d.as<uint2>(), cap3,
                                   cnt_d.as<uint32_t>());
                T.lut3 = lut3_d.as<uint2>();
            } else (void)hipGetLastError();
        }
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    const LeafMeta lt{max_len};
    // ---- boundaries
    const uint64_t nsub = ceil_div(S.nbits, kHdSub);
    if (nsub > 0xffffffffull) return c->fail(CNIIC_ERR_BAD_ARG, "huff_decode_dev: payload too long");
    DevBuf start_d, end_a, end_b, count, off, tot, changed;
    CNIIC_HIP_TRY(c, start_d.alloc(nsub * 8));
    CNIIC_HIP_TRY(c, end_a.alloc(nsub * 8));
    CNIIC_HIP_TRY(c, end_b.alloc(nsub * 8));
    CNIIC_HIP_TRY(c, count.alloc(nsub * 4));
    CNIIC_HIP_TRY(c, off.alloc(nsub * 8));
    CNIIC_HIP_TRY(c, tot.alloc(8));
    CNIIC_HIP_TRY(c, changed.alloc(32));   // [0] an end moved in this check; [2] statistics wanted, [3] threads out of step, [4] blocks that staged
    CNIIC_HIP_TRY(c, hipMemsetAsync(changed.p, 0, 32, c->stream));
    const bool hd_stats = test_env("CNIIC_HD_STATS") != nullptr;
    if (hd_stats) { const uint32_t one = 1; CNIIC_HIP_TRY(c, hipMemcpyAsync(changed.as<uint32_t>() + 2, &one, 4, hipMemcpyHostToDevice, c->stream)); }
    CNIIC_HIP_TRY(c, ctx_pinned_u(c));
    volatile uint64_t *pin = reinterpret_cast<volatile uint64_t *>(c->pinned_u) + 4096;   // (slots of this function's own) [0] total symbols, [1] did the last pass move an end?
    const uint32_t grid = (uint32_t)ceil_div(nsub, kHdThreads);
    const size_t lds = use1 ? kHdLds : (size_t)kHdStageAlloc * 4;
    const bool wide = lt.max_len > 32;
    const char *ph_env = test_env("CNIIC_HD_PHASES");   // 1: the phase maps whatever the blind checks say (tests); 0: never
    const bool phases_ok = !wide && !(ph_env && !atoi(ph_env)), phases_force = phases_ok && ph_env && atoi(ph_env);
    const uint64_t phases_max_sub = test_env("CNIIC_HD_PHASES_MAX_SUB") ? strtoull(test_env("CNIIC_HD_PHASES_MAX_SUB"), nullptr, 10) : kHdPhasesMaxSub;   // (for measurements)
    const bool phases_first = phases_ok && (nsub <= phases_max_sub || phases_force);   // a short stream: instead of more checks
    int pass_rounds = phases_first ? kHdRoundsShort : kHdMaxRounds;
    // a stream that does not fall into step is found out by pass 0 itself (k_hd_pass: blocks whose lists do not shrink count themselves in
    // changed[5]); from a quarter of the blocks on, the checks and the write behind pass 0 return at once and the one look sends the host
    // to the phase maps -- instead of three passes that cure one subsequence a round first (1024^2 photograph, `hufman`: 3.2 -> 1.6 ms)
    const uint32_t hopeless_pct = test_env("CNIIC_HD_HOPELESS_PCT") ? (uint32_t)atoi(test_env("CNIIC_HD_HOPELESS_PCT")) : hd_hopeless_pct(nsub);   // (0: no such verdict)
    const uint32_t hopeless_min = phases_ok && !phases_force && hopeless_pct ? std::max(1u, grid / 4) : 0u;
    // the symbols the passes meet are kept (hd_run<STORE>): kHdKeep words per subsequence, 4 x the payload; without the memory for it
    // (or CNIIC_HD_KEEP=0, testing build) the write decodes everything once more, as until round 4.  Only streams whose subsequences
    // hold 48 symbols on average at most: the 8-bit codes of 256 cluster colours put 65 into 512 bits, every other subsequence would be
    // decoded again anyway and the kept rows were a loss (4096^2 cluster-colors: 0.30 -> 0.38 ms).  CNIIC_HD_KEEP=1: whatever the average.
    DevBuf keep_d;
    uint32_t *keep_p = nullptr;
    const bool keep_forced = test_env("CNIIC_HD_KEEP") && atoi(test_env("CNIIC_HD_KEEP"));
    if (!(test_env("CNIIC_HD_KEEP") && !atoi(test_env("CNIIC_HD_KEEP"))) && (keep_forced || nsyms * kHdSub <= payload_bytes * 8 * 48)) {
        if (keep_d.alloc(ceil_div(nsub, 64) * (uint64_t)(kHdKeep * 64) * 4) == hipSuccess) keep_p = keep_d.as<uint32_t>();
        else (void)hipGetLastError();
    }
    auto pass = [&](const uint64_t *prev, uint64_t *cur) {
        if (wide) hipLaunchKernelGGL(k_hd_pass<true>, dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, prev, cur, start_d.as<uint64_t>(), count.as<uint32_t>(), changed.as<uint32_t>(), kHdMaxRounds, 0u, 0u, keep_p);
        else hipLaunchKernelGGL(k_hd_pass<false>, dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, prev, cur, start_d.as<uint64_t>(), count.as<uint32_t>(), changed.as<uint32_t>(), pass_rounds, hopeless_min, hopeless_pct, keep_p);
    };
    auto write = [&](bool guarded, bool kept) -> int {   // guarded: nothing to write if pass 0 has given the stream up; kept: the passes' symbols stand in keep_d
        CNIIC_TRY(pack_scan(c, count.as<uint32_t>(), (uint32_t)nsub, off.as<uint64_t>(), tot.as<uint64_t>()));
        const uint64_t *st = start_d.as<uint64_t>(), *of = off.as<uint64_t>();
        const uint32_t *hp = guarded && hopeless_min ? changed.as<uint32_t>() + 5 : nullptr;
        if (sums) sums->filled = false;
        if (kept && keep_p) {   // a copy, and one more decode of the few subsequences of more than kHdKeep symbols
            const uint32_t *cn = count.as<uint32_t>();
            int32_t *cs = nullptr;   // packed differences: FromDiff's chunk sums on the way (a write that is repeated starts them again)
            if (mode == 0 && sums) {
                const uint64_t cs_bytes = ceil_div(nsyms, kUdChunk) * 12;
                if (sums->buf.bytes < cs_bytes) CNIIC_HIP_TRY(c, sums->buf.alloc(cs_bytes));
                CNIIC_HIP_TRY(c, hipMemsetAsync(sums->buf.p, 0, cs_bytes, c->stream));
                cs = sums->buf.as<int32_t>();
                sums->filled = true;
            }
            if (mode == 0) hipLaunchKernelGGL(k_hd_compact<0>, dim3(grid), dim3(kHdThreads), 0, c->stream, (const uint32_t *)keep_p, of, cn, nsub, nsyms, out_d, hp, hopeless_min, cs);
            else hipLaunchKernelGGL(k_hd_compact<1>, dim3(grid), dim3(kHdThreads), 0, c->stream, (const uint32_t *)keep_p, of, cn, nsub, nsyms, out_d, hp, hopeless_min, (int32_t *)nullptr);
            if (wide) {
                if (mode == 0) hipLaunchKernelGGL((k_hd_write<true, 0>), dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, st, of, nsyms, out_d, (const uint32_t *)nullptr, 0u, cn, cs);
                else hipLaunchKernelGGL((k_hd_write<true, 1>), dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, st, of, nsyms, out_d, (const uint32_t *)nullptr, 0u, cn, (int32_t *)nullptr);
            } else {
                if (mode == 0) hipLaunchKernelGGL((k_hd_write<false, 0>), dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, st, of, nsyms, out_d, hp, hopeless_min, cn, cs);
                else hipLaunchKernelGGL((k_hd_write<false, 1>), dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, st, of, nsyms, out_d, hp, hopeless_min, cn, (int32_t *)nullptr);
            }
            CNIIC_HIP_TRY(c, hipGetLastError());
            return CNIIC_OK;
        }
        if (wide) {
            if (mode == 0) hipLaunchKernelGGL((k_hd_write<true, 0>), dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, st, of, nsyms, out_d, (const uint32_t *)nullptr, 0u);
            else hipLaunchKernelGGL((k_hd_write<true, 1>), dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, st, of, nsyms, out_d, (const uint32_t *)nullptr, 0u);
        } else {
            if (mode == 0) hipLaunchKernelGGL((k_hd_write<false, 0>), dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, st, of, nsyms, out_d, hp, hopeless_min);
            else hipLaunchKernelGGL((k_hd_write<false, 1>), dim3(grid), dim3(kHdThreads), lds, c->stream, S, T, nsub, st, of, nsyms, out_d, hp, hopeless_min);
        }
        CNIIC_HIP_TRY(c, hipGetLastError());
        return CNIIC_OK;
    };
    auto look = [&]() -> int {   // total and flags to the host
        CNIIC_HIP_TRY(c, hipMemcpyAsync(const_cast<uint64_t *>(pin), tot.p, 8, hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(const_cast<uint64_t *>(pin) + 2, changed.as<uint32_t>() + 5, 4, hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(const_cast<uint64_t *>(pin) + 1, changed.p, 4, hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        return CNIIC_OK;
    };
    // Pass 0, then kHdBlindChecks checks and -- taking the last of them to find everybody in step, as it does on ordinary streams --
    // offsets and symbols straight behind: ONE look at the result for the whole decode.  (The first check re-decodes the few per
    // cent of subsequences whose warm-up had not fallen into step; that can move an end and put a neighbour out of step, which the
    // next check repairs; a check that finds its whole block in step stages nothing and costs next to nothing.)  Should the last
    // blind check still have moved an end, checks go on one look at a time and the symbols are written again.
    uint64_t *cur = end_a.as<uint64_t>(), *nxt = end_b.as<uint64_t>();
    pin[1] = 0; pin[2] = 0;
    {
        ScopedKernelTimer t0(c, "hd_pass0");   // (stage timers: CNIIC_OPT_STAGE_TIMERS; they synchronise)
        pass(nullptr, cur);
        t0.stop();
#ifdef CNIIC_HD_PHASES
        {
            unsigned long long ph[8], zero[8] = {0};
            CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
            CNIIC_HIP_TRY(c, hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_hd_phase), sizeof ph));
            CNIIC_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_hd_phase), zero, sizeof zero));
            const double nb = (double)grid;
            fprintf(stderr, "[hd phases] pass 0, %u blocks, warm %u: per block (us) stage %.2f | first decode %.2f | settle rounds %.2f | rounds %.2f, list entries %.1f, blocks that settled %.0f %%\n",
                    grid, S.warm, ph[0] / nb / 100.0, ph[1] / nb / 100.0, ph[2] / nb / 100.0, ph[3] / nb, ph[4] / nb, 100.0 * ph[5] / nb);
        }
#endif
        ScopedKernelTimer t1(c, "hd_check");
        for (int r = 0; r < kHdBlindChecks; r++) {
            CNIIC_HIP_TRY(c, hipMemsetAsync(changed.p, 0, 4, c->stream));
            pass(cur, nxt);
            std::swap(cur, nxt);
            if (hd_stats) {
                uint32_t st[5];
                CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
                CNIIC_HIP_TRY(c, hipMemcpy(st, changed.p, 20, hipMemcpyDeviceToHost));
                fprintf(stderr, "[hd] check %d: %u of %llu threads out of step, %u of %u blocks staged, an end moved: %u\n", r + 1, st[3], (unsigned long long)nsub, st[4], grid, st[0]);
                CNIIC_HIP_TRY(c, hipMemsetAsync(changed.as<uint32_t>() + 3, 0, 8, c->stream));
            }
        }
        t1.stop();
        ScopedKernelTimer t2(c, "hd_write");
        CNIIC_TRY(write(true, true));
        t2.stop();
    }
    CNIIC_TRY(look());
    auto phases = [&]() -> int {   // every phase of every subsequence (see k_hd_phase_maps) -> start_d, count
        DevBuf maps_d, cnts_d;
        // (96 bytes a subsequence: 400 MB for a 256 MiB payload.  Without the memory the caller decodes on the host -- slow, not an error)
        if (maps_d.alloc(nsub * 32) != hipSuccess || cnts_d.alloc(nsub * 64) != hipSuccess) { (void)hipGetLastError(); *status = 2; return CNIIC_OK; }
        const uint32_t nph = std::min(32u, std::max(16u, lt.max_len)), spb = kHpThreads / nph;   // (the other entries of a map are never asked for)
        CNIIC_HIP_TRY(c, hipMemsetAsync(maps_d.p, 0, nsub * 32, c->stream));
        hipLaunchKernelGGL(k_hd_phase_maps, dim3((uint32_t)ceil_div(nsub, (uint64_t)spb)), dim3(kHpThreads), 0, c->stream, S, T, nsub, maps_d.as<uint8_t>(), cnts_d.as<uint16_t>(), nph, spb);
        if (S.bit0 >= nph)   // the first subsequence alone is entered where the payload begins, which may be any bit of its first word
            hipLaunchKernelGGL(k_hd_phase_maps, dim3(1), dim3(kHpThreads), 0, c->stream, S, T, (uint64_t)1, maps_d.as<uint8_t>(), cnts_d.as<uint16_t>(), 32u, 32u);
        hipLaunchKernelGGL(k_hd_phase_chain, dim3(1), dim3(kHpGroups), 0, c->stream, (const uint8_t *)maps_d.as<uint8_t>(), (const uint16_t *)cnts_d.as<uint16_t>(), nsub, S.nbits,
                           (uint32_t)S.bit0, start_d.as<uint64_t>(), count.as<uint32_t>());
        CNIIC_HIP_TRY(c, hipGetLastError());
        if (hd_stats) fprintf(stderr, "[hd] not in step: every phase of every subsequence\n");
        CNIIC_TRY(write(false, false));   // (the maps count; they keep nothing)
        CNIIC_TRY(look());
        return CNIIC_OK;
    };
    const bool hopeless = hopeless_min && (uint32_t)pin[2] >= hopeless_min;
    if (hd_stats) fprintf(stderr, "[hd] blocks of pass 0 that did not fall into step: %u of %u%s\n", (uint32_t)pin[2], grid, hopeless ? " -- given up" : "");
    if (hopeless || (phases_first && ((uint32_t)pin[1] || phases_force))) {
        CNIIC_TRY(phases());   // not in step after the blind checks: no more checks one subsequence at a time
        if (*status == 2) return CNIIC_OK;
    } else if ((uint32_t)pin[1]) {
        bool settled = false;
        for (int r = kHdBlindChecks; r < kHdMaxPasses; r++) {
            CNIIC_HIP_TRY(c, hipMemsetAsync(changed.p, 0, 4, c->stream));
            pass(cur, nxt);
            std::swap(cur, nxt);
            CNIIC_TRY(look());
            if (hd_stats) fprintf(stderr, "[hd] check %d (after a look): an end moved: %u\n", r + 1, (uint32_t)pin[1]);
            if (!(uint32_t)pin[1]) { settled = true; break; }
        }
        if (!settled && !phases_ok) { *status = 2; return CNIIC_OK; }
        if (!settled) { CNIIC_TRY(phases()); if (*status == 2) return CNIIC_OK; }   // (a long stream that kHdMaxPasses checks have not settled)
        else { CNIIC_TRY(write(false, true)); CNIIC_TRY(look()); }
    }
    if (pin[0] < nsyms) { *status = 1; return CNIIC_OK; }
    return CNIIC_OK;
}

// the same from the host's table of leaves (huff_parse_leaves; not too_deep): one pinned block, one copy up (three copies out of
// pageable vectors were ~100 us of a 0.5 ms decode)
int huff_decode_dev(Ctx *c, const LeafTable &lt, const uint8_t *payload, bool payload_dev, uint64_t payload_bytes, uint64_t nsyms,
                    int mode, void *out_d, int *status, UdSums *sums) {
    *status = 0;
    if (sums) sums->filled = false;
    if (nsyms == 0) return CNIIC_OK;
    const uint64_t n = lt.n();
    if (n == 0 || lt.too_deep) return c->fail(CNIIC_ERR_BAD_ARG, "huff_decode_dev: no usable leaf table");
    const uint64_t off_key = n * 8, off_len = off_key + n * 4, tab_bytes = off_len + n;
    DevBuf tab_d;
    CNIIC_HIP_TRY(c, tab_d.alloc(tab_bytes));
    CNIIC_HIP_TRY(c, ctx_pinned_huf(c, tab_bytes));   // (the stream's head, if it was fetched there, has been parsed: lt owns what it said)
    uint8_t *ph = static_cast<uint8_t *>(c->pinned_huf);
    memcpy(ph, lt.code.data(), n * 8);
    memcpy(ph + off_key, lt.key.data(), n * 4);
    memcpy(ph + off_len, lt.len.data(), n);
    CNIIC_HIP_TRY(c, hipMemcpyAsync(tab_d.p, c->pinned_huf, tab_bytes, hipMemcpyHostToDevice, c->stream));
    return huff_decode_tables_dev(c, tab_d.as<uint8_t>(), n, off_key, off_len, lt.max_len, lt.key[0], payload, payload_dev, payload_bytes, nsyms, mode, out_d, status, sums);
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

__global__ __launch_bounds__(kUdThreads) void k_ud_sums(const uint32_t *__restrict__ keys, uint64_t n, int32_t *__restrict__ chunk_sum) {
    __shared__ int32_t sh[3][kUdThreads / 64];
    // (a sum does not care about the order: neighbouring lanes read neighbouring words -- with 16 consecutive words per thread a load
    // touched 64 cache lines and the kernel ran at 1.8 TB/s)
    const uint64_t base = (uint64_t)blockIdx.x * kUdChunk + threadIdx.x;
    int32_t s[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < kUdPer; j++) {
        const uint64_t i = base + (uint64_t)j * kUdThreads;
        if (i < n) { int32_t d[3]; ud_unpack(keys[i], d); s[0] += d[0]; s[1] += d[1]; s[2] += d[2]; }
    }
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

// LEAVES: the colours go straight to their pixels along the leaves of a large rectangle (lin = the image, w its width) instead of
// into a linearised image that a scatter pass then reads again (6 of 17 bytes per pixel and a launch, as on 2^n squares)
template <bool LEAVES>
__global__ __launch_bounds__(kUdThreads) void k_ud_apply(const uint32_t *__restrict__ keys, uint64_t n, const int32_t *__restrict__ chunk_off,
                                                         uint8_t *__restrict__ lin, uint32_t *__restrict__ bad, uint32_t w = 0,
                                                         const ScanLeavesDev *__restrict__ hdr = nullptr) {
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
    Scan sc{};
    ScanCursor cu;
    if (LEAVES) sc.lf = *hdr;
#pragma unroll
    for (int j = 0; j < kUdPer; j++) {
        if (base + j < n) {
            uint64_t at = base + j;
            if (LEAVES) {
                uint32_t x, y;
                sc.xy_seq(cu, base + j, x, y);
                at = (uint64_t)y * w + x;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                run[ch] += d[j][ch];
                oob |= run[ch] < 0 || run[ch] > 255;
                if (!LEAVES) lin[3 * at + ch] = (uint8_t)run[ch];
            }
            if (LEAVES) store_px3(lin + 3 * at, (uint32_t)(run[0] & 255) | ((uint32_t)(run[1] & 255) << 8) | ((uint32_t)(run[2] & 255) << 16));
        }
    }
    if (oob) *bad = 1u;
}

// exclusive channel sums of the differences before every chunk of kUdChunk symbols (sums_d: 3 x int32 per chunk)
int delta_undiff_prefix(Ctx *c, const uint32_t *keys_d, uint64_t n, DevBuf *sums) {
    const uint64_t nchunks64 = ceil_div(n, kUdChunk);
    if (nchunks64 > 0x7fffffffull) return c->fail(CNIIC_ERR_BAD_ARG, "undiff: too many symbols");
    const uint32_t nchunks = (uint32_t)nchunks64;
    CNIIC_HIP_TRY(c, sums->alloc((uint64_t)nchunks * 12));
    hipLaunchKernelGGL(k_ud_sums, dim3(nchunks), dim3(kUdThreads), 0, c->stream, keys_d, n, sums->as<int32_t>());
    hipLaunchKernelGGL(k_ud_scan, dim3(1), dim3(1024), 0, c->stream, sums->as<int32_t>(), nchunks);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

// FromDiff (hilbertc.rs:482-509) and the walk along the scan (hilbertc.rs:426-428): keys_d = the w x h decoded symbols in scan
// order -> rgb_out_d.  2^n squares: one fused pass by tiles (k_hilbert_move_p2<true, true>); other rectangles: the linearised
// colours first, then the per-position scatter.  *bad_h != 0: a colour left 0..255 (the reference's unwrap, hilbertc.rs:505).
int delta_undiff_scatter_dev(Ctx *c, const uint32_t *keys_d, uint32_t w, uint32_t h, uint8_t *rgb_out_d, uint32_t *bad_h, UdSums *made) {
    *bad_h = 0;
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    static_assert(kUdChunk == 4096, "a chunk of the prefix sums is a 64 x 64 tile of the scan");
    DevBuf sums, bad, lin;
    CNIIC_HIP_TRY(c, bad.alloc(4));
    CNIIC_HIP_TRY(c, hipMemsetAsync(bad.p, 0, 4, c->stream));
    if (made && made->filled && ceil_div(n, kUdChunk) <= 0x7fffffffull) {   // the decoder added the chunks up while it wrote the symbols: only the scan is left
        sums.view(made->buf.p, made->buf.bytes);
        hipLaunchKernelGGL(k_ud_scan, dim3(1), dim3(1024), 0, c->stream, sums.as<int32_t>(), (uint32_t)ceil_div(n, kUdChunk));
        CNIIC_HIP_TRY(c, hipGetLastError());
    } else CNIIC_TRY(delta_undiff_prefix(c, keys_d, n, &sums));
    bool fused = false;
    CNIIC_TRY(hilbert_undiff_scatter(c, keys_d, sums.as<int32_t>(), w, h, rgb_out_d, bad.as<uint32_t>(), &fused));
    if (!fused) {
        ScanSel sel;
        CNIIC_TRY(scan_select(c, w, h, &sel));
        if (sel.korder & kScanLeavesBit) {   // a large rectangle: undiff and scatter in one pass along its leaves
            hipLaunchKernelGGL(k_ud_apply<true>, dim3((uint32_t)ceil_div(n, kUdChunk)), dim3(kUdThreads), 0, c->stream, keys_d, n, (const int32_t *)sums.as<int32_t>(),
                               rgb_out_d, bad.as<uint32_t>(), w, reinterpret_cast<const ScanLeavesDev *>(sel.arg));
            CNIIC_HIP_TRY(c, hipGetLastError());
        } else {
            CNIIC_HIP_TRY(c, lin.alloc(n * 3));
            hipLaunchKernelGGL(k_ud_apply<false>, dim3((uint32_t)ceil_div(n, kUdChunk)), dim3(kUdThreads), 0, c->stream, keys_d, n, (const int32_t *)sums.as<int32_t>(),
                               lin.as<uint8_t>(), bad.as<uint32_t>());
            CNIIC_HIP_TRY(c, hipGetLastError());
            CNIIC_TRY(hilbert_scatter(c, lin.as<uint8_t>(), w, h, rgb_out_d));
        }
    }
    CNIIC_HIP_TRY(c, hipMemcpyAsync(bad_h, bad.p, 4, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
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
    hipLaunchKernelGGL(k_ud_apply<false>, dim3(nchunks), dim3(kUdThreads), 0, c->stream, keys_d, n, (const int32_t *)sums.as<int32_t>(), lin_d,
                       bad.as<uint32_t>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipMemcpyAsync(bad_h, bad.p, 4, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

}  // namespace cniic
