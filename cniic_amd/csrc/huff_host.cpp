// huff_host.cpp -- see huff_host.hpp.  Reference: src/huf.rs, src/ser.rs, src/bit.rs:256-259.
#include "huff_host.hpp"

#include <utility>

#include "../../include/cniic_hip.h"

namespace cniic {

namespace {

// One heap entry: huf.rs:63-66 `Suffix { freq, tree }` with Ord reversed on freq (huf.rs:80-85),
// so "greater" means "smaller frequency" and the max-heap pops the rarest subtree first.
struct Suffix {
    uint64_t freq;
    uint32_t node;
};
inline bool le(const Suffix &a, const Suffix &b) { return a.freq >= b.freq; }  // a <= b in reversed order
inline bool lt(const Suffix &a, const Suffix &b) { return a.freq > b.freq; }

// std::collections::BinaryHeap restated: the exact sift routines decide how equal frequencies are
// ordered, hence the tree shape.
class RustMaxHeap {
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
        while (end >= 2 && child <= end - 2) {
            if (le(d_[child], d_[child + 1])) child++;
            d_[pos] = d_[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (end >= 1 && child == end - 1) {
            d_[pos] = d_[child];
            pos = child;
        }
        d_[pos] = hole;
        sift_up(0, pos);
    }
};

}  // namespace

bool huff_build_tree(const uint64_t *counts, uint64_t n, HuffTree &t) {
    if (n == 0 || n > 0x7fffffffull) return false;  // huf.rs:99 assert!(min_heap.len() > 0)
    t.nleaf = n;
    t.left.assign(n > 1 ? n - 1 : 0, 0);
    t.right.assign(n > 1 ? n - 1 : 0, 0);
    std::vector<Suffix> items(n);
    for (uint64_t i = 0; i < n; i++) items[i] = Suffix{counts[i], (uint32_t)i};
    RustMaxHeap heap(std::move(items));
    uint32_t next = (uint32_t)n;
    while (heap.size() > 1) {  // huf.rs:100-110
        Suffix l = heap.pop();
        Suffix r = heap.pop();
        t.left[next - n] = l.node;
        t.right[next - n] = r.node;
        heap.push(Suffix{l.freq + r.freq, next});
        next++;
    }
    t.root = heap.pop().node;
    return true;
}

bool huff_codes(const HuffTree &t, std::vector<uint8_t> &len, std::vector<uint64_t> &code) {
    const uint64_t n = t.nleaf;
    len.assign(n, 0);
    code.assign(n, 0);
    struct Fr { uint32_t node; uint32_t depth; uint64_t bits; };
    std::vector<Fr> st;
    st.push_back({t.root, 0, 0});
    bool ok = true;
    while (!st.empty()) {
        Fr f = st.back();
        st.pop_back();
        if (f.node < n) {
            if (f.depth > 64) { ok = false; continue; }
            len[f.node] = (uint8_t)f.depth;
            code[f.node] = f.bits;
        } else {
            st.push_back({t.right[f.node - n], f.depth + 1, (f.bits << 1) | 1});  // Bit::One  = right
            st.push_back({t.left[f.node - n], f.depth + 1, f.bits << 1});         // Bit::Zero = left
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
    std::vector<uint32_t> st;
    st.push_back(t.root);
    while (!st.empty()) {
        uint32_t nd = st.back();
        st.pop_back();
        if (nd < n) {
            out.push_back(0);  // SER_ENUM_LEAF huf.rs:296
            put_symbol(out, sym_kind, keys[nd]);
        } else {
            out.push_back(1);  // SER_ENUM_BRANCH huf.rs:297
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
