// huff_host.cpp -- see huff_host.hpp.  Reference: src/huf.rs, src/ser.rs, src/bit.rs:256-259.
#include "huff_host.hpp"

#include <cstddef>
#include <cstdlib>
#include <utility>

#include "../../include/cniic_hip.h"

namespace cniic {

namespace {

// One heap entry: huf.rs:63-66 `Suffix { freq, tree }` with Ord reversed on freq (huf.rs:80-85),
// so "greater" means "smaller frequency" and the max-heap pops the rarest subtree first.
// FreqT = u32 whenever the total fits (every image this library takes: < 2^32 pixels): 8-byte entries, half the
// memory the sift loops walk; u64 otherwise.
template <typename FreqT> struct SuffixT {
    FreqT freq;
    uint32_t node;
};
template <typename S> inline bool le(const S &a, const S &b) { return a.freq >= b.freq; }  // a <= b in reversed order
template <typename S> inline bool lt(const S &a, const S &b) { return a.freq > b.freq; }

// std::collections::BinaryHeap restated: the exact sift routines decide how equal frequencies are
// ordered, hence the tree shape.
template <typename Suffix> class RustMaxHeap {
  public:
    explicit RustMaxHeap(std::vector<Suffix> v) : d_(std::move(v)) {  // From<Vec<T>> -> rebuild()
        for (size_t n = d_.size() / 2; n > 0;) sift_down_range(--n, d_.size());
    }
    size_t size() const { return d_.size(); }
    void push(Suffix s) {
        d_.push_back(s);
        sift_up(0, d_.size() - 1);
    }
    Suffix pop() {
        Suffix item = d_.back();
        d_.pop_back();
        if (!d_.empty()) {
            std::swap(item, d_[0]);
            sift_down_to_bottom();
        }
        return item;
    }

  private:
    std::vector<Suffix> d_;
    void sift_up(size_t start, size_t pos) {
        Suffix hole = d_[pos];
        while (pos > start) {
            size_t parent = (pos - 1) / 2;
            if (le(hole, d_[parent])) break;
            d_[pos] = d_[parent];
            pos = parent;
        }
        d_[pos] = hole;
    }
    void sift_down_range(size_t pos, size_t end) {
        Suffix hole = d_[pos];
        size_t child = 2 * pos + 1;
        while (end >= 2 && child <= end - 2) {
            if (le(d_[child], d_[child + 1])) child++;  // the greater of the two children
            if (!lt(hole, d_[child])) { d_[pos] = hole; return; }  // hole >= child: in order
            d_[pos] = d_[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (end >= 1 && child == end - 1 && lt(hole, d_[child])) {
            d_[pos] = d_[child];
            pos = child;
        }
        d_[pos] = hole;
    }
    void sift_down_to_bottom() {
        const size_t end = d_.size();
        size_t pos = 0;
        Suffix hole = d_[0];
        size_t child = 1;
        Suffix *d = d_.data();
        // every level's address depends on the compare of the level above: the walk is a chain of cache misses once
        // the array outgrows L1.  The four grandchildren of (child, child + 1) are one contiguous run, and so are
        // their eight children: both are requested before the compare that picks one of them.
        while (end >= 2 && child <= end - 2) {
            __builtin_prefetch(d + 4 * child + 3);
            __builtin_prefetch(d + 4 * child + 3 + 64 / sizeof(Suffix) - 1);
            child += le(d[child], d[child + 1]) ? 1 : 0;
            d[pos] = d[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (end >= 1 && child == end - 1) {
            d[pos] = d[child];
            pos = child;
        }
        d[pos] = hole;
        sift_up(0, pos);
    }
};

}  // namespace

template <typename FreqT> static void build_tree_with(const uint64_t *counts, uint64_t n, HuffTree &t) {
    using Suffix = SuffixT<FreqT>;
    std::vector<Suffix> items(n);
    for (uint64_t i = 0; i < n; i++) items[i] = Suffix{(FreqT)counts[i], (uint32_t)i};
    RustMaxHeap<Suffix> heap(std::move(items));
    uint32_t next = (uint32_t)n;
    while (heap.size() > 1) {  // huf.rs:100-110
        Suffix l = heap.pop();
        Suffix r = heap.pop();
        t.left[next - n] = l.node;
        t.right[next - n] = r.node;
        heap.push(Suffix{(FreqT)(l.freq + r.freq), next});
        next++;
    }
    t.root = heap.pop().node;
}

bool huff_build_tree(const uint64_t *counts, uint64_t n, HuffTree &t) {
    if (n == 0 || n > 0x7fffffffull) return false;  // huf.rs:99 assert!(min_heap.len() > 0)
    t.nleaf = n;
    t.left.assign(n > 1 ? n - 1 : 0, 0);
    t.right.assign(n > 1 ? n - 1 : 0, 0);
    uint64_t total = 0;
    bool small = true;
    for (uint64_t i = 0; i < n && small; i++) { total += counts[i]; small = total < (1ull << 32) && counts[i] < (1ull << 32); }
    if (small) build_tree_with<uint32_t>(counts, n, t);
    else build_tree_with<uint64_t>(counts, n, t);
    return true;
}

bool huff_codes(const HuffTree &t, std::vector<uint8_t> &len, std::vector<uint64_t> &code) {
    const uint64_t n = t.nleaf;
    len.assign(n, 0);
    code.assign(n, 0);
    if (n == 1) return true;  // a single symbol: the zero-length code (huf.rs:140-142)
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
                if (d > 64) { ok = false; continue; }
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
    const uint64_t n = t.nleaf;
    // the size is known (n leaves of 1 + S bytes, n - 1 branch tags): written through a pointer, not byte by byte
    const uint64_t S = (uint64_t)huff_symbol_size(sym_kind), base = out.size();
    out.resize(base + n * (1 + S) + (n - 1));
    uint8_t *o = out.data() + base;
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
