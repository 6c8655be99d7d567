// huff_host.cpp -- see huff_host.hpp.  Reference: src/huf.rs, src/ser.rs, src/bit.rs:256-259.
#include "huff_host.hpp"

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "../../include/cniic_hip.h"

namespace cniic {

// huf.rs:58-117 `build`: the two rarest subtrees become the left (first popped) and right (second popped) child of a new
// branch, until one tree is left.  The reference takes them off a BinaryHeap fed in the iteration order of a std HashMap
// (huf.rs:30-31, 96), so which of several EQUALLY rare subtrees comes first is random per process there, and nothing
// depends on it (all such trees cost the same).  One rule is fixed here and in the parity checker (deviation D1):
//     the rarest first; among equally rare ones a leaf before a branch, leaves by ascending symbol key, branches in the
//     order they were made.
// Leaves sorted by (count, key) + branches appended as they are made (their counts never decrease) = the two-queue
// merge: O(n) after the sort, sequential memory, no heap (the heap was 2.3 of the 3.3 ms of a 4096^2 `delta` encode).
namespace {

// stable LSD radix sort of (count << 32 | leaf) by count, 11 bits a pass, only over the bits some count has
void sort_leaves(std::vector<uint64_t> &a, std::vector<uint64_t> &tmp, uint32_t maxc) {
    const size_t n = a.size();
    if (n < 2) return;
    tmp.resize(n);
    uint64_t *src = a.data(), *dst = tmp.data();
    for (int sh = 32; sh < 64 && (maxc >> (sh - 32)); sh += 11) {
        size_t cnt[2049] = {0};
        for (size_t i = 0; i < n; i++) cnt[((src[i] >> sh) & 2047) + 1]++;
        for (int i = 0; i < 2048; i++) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i < n; i++) dst[cnt[(src[i] >> sh) & 2047]++] = src[i];
        std::swap(src, dst);
    }
    if (src != a.data()) memcpy(a.data(), src, n * sizeof(uint64_t));
}

}  // namespace

// counts below 2^32 (every image this library takes has fewer pixels): leaves packed as count << 32 | leaf
static void merge_sorted_u32(const uint64_t *leaf, uint64_t n, uint32_t *left, uint32_t *right, uint32_t *nl, HuffScratch &sc);
static void build_tree_u32(const uint64_t *counts, uint64_t n, uint32_t *left, uint32_t *right, uint32_t *nl, HuffScratch &sc) {
    std::vector<uint64_t> &leaf = sc.leaf;
    leaf.resize(n);
    uint32_t maxc = 0;
    for (uint64_t i = 0; i < n; i++) { leaf[i] = (counts[i] << 32) | i; maxc |= (uint32_t)counts[i]; }
    sort_leaves(leaf, sc.tmp, maxc);  // (count, key): the leaf ids are ascending keys and the sort is stable
    merge_sorted_u32(leaf.data(), n, left, right, nl, sc);
}

// any counts: the same rule through a comparison sort
static void build_tree_u64(const uint64_t *counts, uint64_t n, uint32_t *left, uint32_t *right, uint32_t *nl, HuffScratch &sc) {
    std::vector<uint32_t> order(n);
    for (uint64_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return counts[a] < counts[b]; });
    std::vector<uint64_t> &bfreq = sc.bfreq;
    bfreq.resize(n > 1 ? n - 1 : 0);
    uint64_t li = 0, bi = 0, made = 0;
    while (made + 1 < n) {
        uint32_t node[2], leaves = 0;
        uint64_t f[2];
        for (int k = 0; k < 2; k++) {
            if (li < n && (bi >= made || counts[order[li]] <= bfreq[bi])) { f[k] = counts[order[li]]; node[k] = order[li]; li++; leaves += 1; }
            else { f[k] = bfreq[bi]; node[k] = (uint32_t)(n + bi); if (nl) leaves += nl[bi]; bi++; }
        }
        left[made] = node[0];
        right[made] = node[1];
        if (nl) nl[made] = leaves;
        bfreq[made] = f[0] + f[1];
        made++;
    }
}

// the merge alone: leaf[] = count << 32 | leaf id, sorted by (count, id)
// Round 4: the same picks in the same order, taken in STRETCHES.  While the two rarest subtrees are both leaves -- the leaf queue's next two
// are no heavier than the branch queue's head, which does not change while branches are only appended -- leaves pair off in a loop that
// compares two counts with a register; while they are both branches (strictly lighter than the next leaf: at equal weight a leaf goes
// first) branches pair off likewise; only the mixed pairs take the general step.  The general loop spent its time on the mispredicted
// "which queue" branch of every pick (0.19 ms for the 54 K leaves of a `delta` encode at 16384^2, the GPU idle meanwhile).
static void merge_sorted_u32(const uint64_t *leaf, uint64_t n, uint32_t *left, uint32_t *right, uint32_t *nl, HuffScratch &sc) {
    std::vector<uint64_t> &bfreq = sc.bfreq;
    bfreq.resize(n > 1 ? n - 1 : 0);
    uint64_t li = 0, bi = 0, made = 0;
    constexpr uint64_t kInf = ~0ull;
    while (made + 1 < n) {
        {   // a stretch of leaf pairs
            uint64_t bw = bi < made ? bfreq[bi] : kInf;
            while (li + 1 < n && (leaf[li + 1] >> 32) <= bw) {   // (sorted: leaf[li] <= leaf[li + 1])
                const uint64_t a = leaf[li], b = leaf[li + 1];
                left[made] = (uint32_t)a;
                right[made] = (uint32_t)b;
                if (nl) nl[made] = 2;
                bfreq[made] = (a >> 32) + (b >> 32);
                made++;
                li += 2;
                if (bw == kInf) bw = bfreq[bi];   // (the queue was empty: the branch just made is its head)
            }
            if (made + 1 >= n) break;
        }
        {   // a stretch of branch pairs
            const uint64_t lw = li < n ? leaf[li] >> 32 : kInf;
            while (bi + 1 < made && bfreq[bi + 1] < lw) {   // (non-decreasing: bfreq[bi] <= bfreq[bi + 1]; strict: a leaf of equal weight goes first)
                left[made] = (uint32_t)(n + bi);
                right[made] = (uint32_t)(n + bi + 1);
                if (nl) nl[made] = nl[bi] + nl[bi + 1];
                bfreq[made] = bfreq[bi] + bfreq[bi + 1];
                made++;
                bi += 2;
            }
            if (made + 1 >= n) break;
        }
        uint32_t node[2], leaves = 0;
        uint64_t f[2];
        for (int k = 0; k < 2; k++) {
            if (li < n && (bi >= made || (leaf[li] >> 32) <= bfreq[bi])) { f[k] = leaf[li] >> 32; node[k] = (uint32_t)leaf[li]; li++; leaves += 1; }
            else { f[k] = bfreq[bi]; node[k] = (uint32_t)(n + bi); if (nl) leaves += nl[bi]; bi++; }
        }
        left[made] = node[0];
        right[made] = node[1];
        if (nl) nl[made] = leaves;
        bfreq[made] = f[0] + f[1];
        made++;
    }
}

bool huff_merge_sorted_into(const uint64_t *leaf_sorted, uint64_t n, uint32_t *left, uint32_t *right, uint32_t *nleaves, uint32_t *root,
                            HuffScratch *scratch) {
    if (n == 0 || n > 0x7fffffffull) return false;
    HuffScratch local;
    merge_sorted_u32(leaf_sorted, n, left, right, nleaves, scratch ? *scratch : local);
    *root = n > 1 ? (uint32_t)(2 * n - 2) : 0;
    return true;
}

bool huff_build_tree_into(const uint64_t *counts, uint64_t n, uint32_t *left, uint32_t *right, uint32_t *nleaves, uint32_t *root, HuffScratch *scratch) {
    if (n == 0 || n > 0x7fffffffull) return false;  // huf.rs:99 assert!(min_heap.len() > 0)
    HuffScratch local;
    HuffScratch &sc = scratch ? *scratch : local;
    uint64_t total = 0;
    bool small = true;
    for (uint64_t i = 0; i < n && small; i++) { total += counts[i]; small = total < (1ull << 32) && counts[i] < (1ull << 32); }
    if (small) build_tree_u32(counts, n, left, right, nleaves, sc);
    else build_tree_u64(counts, n, left, right, nleaves, sc);
    *root = n > 1 ? (uint32_t)(2 * n - 2) : 0;
    return true;
}

bool huff_build_tree(const uint64_t *counts, uint64_t n, HuffTree &t, HuffScratch *scratch) {
    if (n == 0 || n > 0x7fffffffull) return false;
    t.nleaf = n;
    t.left.resize(n - 1);  // (every entry is written)
    t.right.resize(n - 1);
    return huff_build_tree_into(counts, n, t.left.data(), t.right.data(), nullptr, &t.root, scratch);
}

bool huff_codes(const HuffTree &t, std::vector<uint8_t> &len, std::vector<uint64_t> &code) {
    len.resize(t.nleaf);
    code.resize(t.nleaf);
    return huff_codes_into(t, len.data(), code.data());
}

bool huff_codes_into(const HuffTree &t, uint8_t *len, uint64_t *code) {
    const uint64_t n = t.nleaf;
    if (n == 1) { len[0] = 0; code[0] = 0; return true; }  // a single symbol: the zero-length code (huf.rs:140-142)
    // An inner node is made after its two children, so it has the larger id and the root the largest: walking the
    // ids downwards meets every parent before its children (Bit::Zero = left, Bit::One = right).  No stack.
    const uint64_t ninner = n - 1;
    std::vector<uint8_t> idepth(ninner, 0);
    std::vector<uint64_t> ibits(ninner, 0);
    bool ok = true;
    for (uint64_t i = ninner; i-- > 0;) {
        const uint32_t d = (uint32_t)idepth[i] + 1;
        const uint64_t b = ibits[i];
        const uint32_t kid[2] = {t.left[i], t.right[i]};
        for (int s = 0; s < 2; s++) {
            const uint64_t cb = (d <= 64 ? (b << 1) : 0) | (uint64_t)s;
            if (kid[s] < n) {
                if (d > 64) { ok = false; len[kid[s]] = 0; code[kid[s]] = 0; continue; }
                len[kid[s]] = (uint8_t)d;
                code[kid[s]] = cb;
            } else {
                idepth[kid[s] - n] = (uint8_t)(d > 255 ? 255 : d);
                ibits[kid[s] - n] = cb;
            }
        }
    }
    return ok;
}

int huff_symbol_size(int sym_kind) {
    switch (sym_kind) {
    case CNIIC_SYM_RGB: return 11;     // ser.rs:210-214: u64 length (3) + 3 bytes
    case CNIIC_SYM_SIGNED: return 6;   // hilbertc.rs:561-565 -> ser.rs:188-195: three i16 LE
    }
    return -1;
}

void put_u32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); }
void put_u64(std::vector<uint8_t> &o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
bool get_u32(const uint8_t *b, uint64_t n, uint64_t &pos, uint32_t &v) {
    if (pos + 4 > n) return false;
    v = 0;
    for (int i = 0; i < 4; i++) v |= (uint32_t)b[pos + i] << (8 * i);
    pos += 4;
    return true;
}
bool get_u64(const uint8_t *b, uint64_t n, uint64_t &pos, uint64_t &v) {
    if (pos + 8 > n) return false;
    v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)b[pos + i] << (8 * i);
    pos += 8;
    return true;
}

static void put_symbol(std::vector<uint8_t> &o, int kind, uint32_t key) {
    if (kind == CNIIC_SYM_RGB) {
        put_u64(o, 3);
        o.push_back((uint8_t)(key >> 16)); o.push_back((uint8_t)(key >> 8)); o.push_back((uint8_t)key);
    } else {
        for (int i = 0; i < 3; i++) {
            int v = (int)((key >> (18 - 9 * i)) & 511) - 255;
            uint16_t u = (uint16_t)(int16_t)v;
            o.push_back((uint8_t)u); o.push_back((uint8_t)(u >> 8));
        }
    }
}

void huff_serialize_tree(const HuffTree &t, int sym_kind, const uint32_t *keys, std::vector<uint8_t> &out) {
    const uint64_t base = out.size();
    out.resize(base + huff_tree_bytes(sym_kind, t.nleaf));
    huff_serialize_tree_into(t, sym_kind, keys, out.data() + base);
}

uint64_t huff_tree_bytes(int sym_kind, uint64_t n) { return n * (1 + (uint64_t)huff_symbol_size(sym_kind)) + (n - 1); }

void huff_serialize_tree_into(const HuffTree &t, int sym_kind, const uint32_t *keys, uint8_t *o) {
    const uint64_t n = t.nleaf;
    // the size is known (n leaves of 1 + S bytes, n - 1 branch tags): written through a pointer, not byte by byte
    std::vector<uint32_t> st;
    st.reserve(128);
    st.push_back(t.root);
    while (!st.empty()) {
        const uint32_t nd = st.back();
        st.pop_back();
        if (nd < n) {
            *o++ = 0;  // SER_ENUM_LEAF huf.rs:296
            const uint32_t key = keys[nd];
            if (sym_kind == CNIIC_SYM_RGB) {  // ser.rs:210-214: u64 length (3) + 3 bytes
                o[0] = 3; o[1] = o[2] = o[3] = o[4] = o[5] = o[6] = o[7] = 0;
                o[8] = (uint8_t)(key >> 16); o[9] = (uint8_t)(key >> 8); o[10] = (uint8_t)key;
                o += 11;
            } else {                          // hilbertc.rs:561-565 -> ser.rs:188-195: three i16 LE
                for (int i = 0; i < 3; i++) {
                    const uint16_t u = (uint16_t)(int16_t)((int)((key >> (18 - 9 * i)) & 511) - 255);
                    *o++ = (uint8_t)u; *o++ = (uint8_t)(u >> 8);
                }
            }
        } else {
            *o++ = 1;  // SER_ENUM_BRANCH huf.rs:297
            st.push_back(t.right[nd - n]);
            st.push_back(t.left[nd - n]);
        }
    }
}

uint64_t huff_stream_size(int sym_kind, const uint64_t *counts, const uint8_t *len, uint64_t n) {
    uint64_t bits = 0;
    for (uint64_t i = 0; i < n; i++) bits += counts[i] * len[i];
    return n * (uint64_t)(1 + huff_symbol_size(sym_kind)) + (n - 1) + (bits + 7) / 8;
}

// ---------------------------------------------------------------- decode
static bool get_symbol(const uint8_t *b, uint64_t n, uint64_t &pos, int kind, uint32_t &key) {
    if (kind == CNIIC_SYM_RGB) {
        uint64_t len;
        if (!get_u64(b, n, pos, len) || len != 3 || pos + 3 > n) return false;  // ser.rs:216-222
        key = ((uint32_t)b[pos] << 16) | ((uint32_t)b[pos + 1] << 8) | b[pos + 2];
        pos += 3;
        return true;
    }
    if (pos + 6 > n) return false;
    key = 0;
    for (int i = 0; i < 3; i++) {
        int16_t v = (int16_t)(uint16_t)(b[pos] | (b[pos + 1] << 8));
        pos += 2;
        if (v < -255 || v > 255) return false;  // cannot be a delta of two u8 channels
        key = (key << 9) | (uint32_t)(v + 255);
    }
    return true;
}

// Dec::deserialize (huf.rs:323-348): the pre-order trie at bytes[pos..] as flat nodes, root first
bool huff_parse_trie(int sym_kind, const uint8_t *bytes, uint64_t nbytes, uint64_t &pos, std::vector<TrieNode> &nodes) {
    nodes.clear();
    struct Pending { size_t node; int filled; };
    std::vector<Pending> pend;
    bool have_root = false;
    for (;;) {
        if (pos >= nbytes) return false;
        const uint8_t tag = bytes[pos++];
        TrieNode nd{0, 0};
        if (tag == 0) {
            nd.l = kTrieLeaf;
            if (!get_symbol(bytes, nbytes, pos, sym_kind, nd.r)) return false;
        } else if (tag != 1) {
            return false;  // huf.rs:343-345
        }
        const size_t id = nodes.size();
        if (id >= 0xfffffffeull) return false;
        nodes.push_back(nd);
        if (have_root) {
            Pending &p = pend.back();
            if (p.filled == 0) { nodes[p.node].l = (uint32_t)id; p.filled = 1; }
            else { nodes[p.node].r = (uint32_t)id; pend.pop_back(); }
        }
        have_root = true;
        if (tag == 1) pend.push_back({id, 0});
        if (pend.empty()) break;
    }
    return true;
}

// Dec::deserialize again, keeping only what a table-driven decoder needs: the leaves in pre-order with their paths.  The parse
// IS a depth-first walk, so path and depth are at hand when a leaf is met; pending right children wait on a stack.
// Returns false exactly where huff_parse_trie does (truncated, a tag other than 0 / 1, a malformed symbol).
bool huff_parse_leaves(int sym_kind, const uint8_t *bytes, uint64_t nbytes, uint64_t &pos, LeafTable &t) {
    t.code.clear(); t.key.clear(); t.len.clear();
    t.max_len = 0; t.min_len = 0xffffffffu; t.too_deep = false;
    struct Pend { uint64_t code; uint32_t depth; };
    std::vector<Pend> stack;
    const uint64_t pos0 = pos;
    uint64_t code = 0;
    uint32_t depth = 0;
    for (;;) {
        if (pos >= nbytes) return false;
        const uint8_t tag = bytes[pos++];
        if (tag == 1) {
            if (depth >= kLeafMaxLen) { t.too_deep = true; pos = pos0; return true; }  // the caller parses nodes instead (huff_parse_trie)
            stack.push_back({code | (1ull << (63 - depth)), depth + 1});  // right child: this path + a 1
            depth++;                                                      // left child: this path + a 0
        } else if (tag == 0) {
            uint32_t key;
            if (!get_symbol(bytes, nbytes, pos, sym_kind, key)) return false;
            if (t.code.size() >= 0xfffffff0ull) return false;
            t.code.push_back(code); t.key.push_back(key); t.len.push_back((uint8_t)depth);
            t.max_len = std::max(t.max_len, depth); t.min_len = std::min(t.min_len, depth);
            if (stack.empty()) break;
            code = stack.back().code; depth = stack.back().depth;
            stack.pop_back();
        } else {
            return false;  // huf.rs:343-345
        }
    }
    return true;
}

// bit_reader MsbFirst (bit.rs:256-259) + BinTrie::lookup (huf.rs:187-206).  The walk is the reference's, taken
// kLut bits at a time: lut[prefix] = the node reached from the root by those bits and how many of them were
// used (a leaf may be reached early); a symbol longer than kLut bits goes on bit by bit from that node.
bool huff_decode_host(const std::vector<TrieNode> &nodes, const uint8_t *p, uint64_t payload_bytes, uint64_t nsyms,
                      uint32_t *keys_out, uint64_t *bytes_used) {
    const uint64_t nbits = payload_bytes * 8;
    uint64_t bp = 0;
    if (bytes_used) *bytes_used = 0;
    if (nodes[0].l == kTrieLeaf) {  // single symbol: zero-length code, no payload (huf.rs:140-142)
        for (uint64_t i = 0; i < nsyms; i++) keys_out[i] = nodes[0].r;
        return true;
    }
    constexpr int kLut = 12;
    struct Hop { uint32_t node; uint32_t used; };
    std::vector<Hop> lut(1u << kLut);
    for (uint32_t pre = 0; pre < (1u << kLut); pre++) {
        uint32_t nd = 0, used = 0;
        while (used < (uint32_t)kLut && nodes[nd].l != kTrieLeaf) {
            nd = ((pre >> (kLut - 1 - used)) & 1) ? nodes[nd].r : nodes[nd].l;
            used++;
        }
        lut[pre] = {nd, used};
    }
    auto peek = [&](uint64_t at) -> uint32_t {  // kLut bits from bit position `at`, zero-filled past the end
        uint32_t v = 0;
        const uint64_t byte = at >> 3;
        for (int k = 0; k < 3; k++) v = (v << 8) | (byte + k < payload_bytes ? p[byte + k] : 0u);
        return (v >> (24 - kLut - (at & 7))) & ((1u << kLut) - 1);
    };
    for (uint64_t i = 0; i < nsyms; i++) {
        const Hop hp = lut[peek(bp)];
        uint32_t nd = hp.node;
        if (bp + hp.used > nbits) return false;  // EOF inside the symbol -> None
        bp += hp.used;
        while (nodes[nd].l != kTrieLeaf) {
            if (bp >= nbits) return false;
            const int bit = (p[bp >> 3] >> (7 - (bp & 7))) & 1;
            bp++;
            nd = bit ? nodes[nd].r : nodes[nd].l;
        }
        keys_out[i] = nodes[nd].r;
    }
    if (bytes_used) *bytes_used = (bp + 7) / 8;
    return true;
}

bool huff_decode_symbols(int sym_kind, const uint8_t *bytes, uint64_t nbytes, uint64_t &pos, uint64_t nsyms,
                         uint32_t *keys_out) {
    std::vector<TrieNode> nodes;
    if (!huff_parse_trie(sym_kind, bytes, nbytes, pos, nodes)) return false;
    uint64_t used = 0;
    if (!huff_decode_host(nodes, bytes + pos, nbytes - pos, nsyms, keys_out, &used)) return false;
    pos += used;
    return true;
}

}  // namespace cniic
