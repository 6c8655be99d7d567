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

bool huff_decode_symbols(int sym_kind, const uint8_t *bytes, uint64_t nbytes, uint64_t &pos, uint64_t nsyms,
                         uint32_t *keys_out) {
    // flat trie: child[2*i], child[2*i+1] for branches; leaves carry the key
    struct Node { int64_t l = -1, r = -1; uint32_t key = 0; bool leaf = false; };
    std::vector<Node> nodes;
    struct Pending { size_t node; int filled; };
    std::vector<Pending> pend;
    bool have_root = false;
    for (;;) {
        if (pos >= nbytes) return false;
        uint8_t tag = bytes[pos++];
        Node nd;
        if (tag == 0) {
            nd.leaf = true;
            if (!get_symbol(bytes, nbytes, pos, sym_kind, nd.key)) return false;
        } else if (tag != 1) {
            return false;  // huf.rs:343-345
        }
        size_t id = nodes.size();
        nodes.push_back(nd);
        if (have_root) {
            Pending &p = pend.back();
            if (p.filled == 0) { nodes[p.node].l = (int64_t)id; p.filled = 1; }
            else { nodes[p.node].r = (int64_t)id; pend.pop_back(); }
        }
        have_root = true;
        if (tag == 1) pend.push_back({id, 0});
        if (pend.empty()) break;
    }
    // bit_reader MsbFirst (bit.rs:256-259) + BinTrie::lookup (huf.rs:187-206)
    const uint8_t *p = bytes + pos;
    const uint64_t nbits = (nbytes - pos) * 8;
    uint64_t bp = 0;
    for (uint64_t i = 0; i < nsyms; i++) {
        size_t nd = 0;
        while (!nodes[nd].leaf) {
            if (bp >= nbits) return false;  // EOF -> None
            int bit = (p[bp >> 3] >> (7 - (bp & 7))) & 1;
            bp++;
            nd = (size_t)(bit ? nodes[nd].r : nodes[nd].l);
        }
        keys_out[i] = nodes[nd].key;
    }
    pos += (bp + 7) / 8;
    return true;
}

}  // namespace cniic
