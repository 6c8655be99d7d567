// huff_host.hpp -- host side of huf::encode_all / decode_all (reference src/huf.rs): code
// construction from a histogram, decoder (trie) serialisation, and stream decoding.  The
// per-symbol work (histogram, bit-pack) runs on the GPU; what is here is O(alphabet), not O(pixels),
// except decode_symbols, which walks the bit stream (Huffman decoding is serial in the stream).
#pragma once
#include <cstdint>
#include <vector>

namespace cniic {

struct HuffTree {
    // nodes 0..n-1 are leaves (leaf i = i-th symbol of the ascending-key histogram);
    // nodes n..2n-2 are branches (huf.rs:167-171 BinTrie)
    uint64_t nleaf = 0;
    std::vector<uint32_t> left, right;  // per branch
    uint32_t root = 0;
};

// huf.rs:58-117 build(): min-heap merge.  Heap = Rust std BinaryHeap semantics, item order = the
// given (ascending key) order.
struct HuffScratch { std::vector<uint64_t> leaf, tmp, bfreq; };  // kept between calls: no 54 MB of fresh pages per array and call
bool huff_build_tree(const uint64_t *counts, uint64_t n, HuffTree &t, HuffScratch *scratch = nullptr);
// the same into arrays of the caller's (n - 1 entries each; pinned memory, say); nleaves (optional): leaves below every branch
bool huff_build_tree_into(const uint64_t *counts, uint64_t n, uint32_t *left, uint32_t *right, uint32_t *nleaves, uint32_t *root,
                          HuffScratch *scratch = nullptr);
// the merge alone, from leaves count << 32 | id already sorted by (count, id) (counts below 2^32)
bool huff_merge_sorted_into(const uint64_t *leaf_sorted, uint64_t n, uint32_t *left, uint32_t *right, uint32_t *nleaves, uint32_t *root,
                            HuffScratch *scratch = nullptr);
// Enc::from(&Dec) (huf.rs:125-135): code length and code bits (MSB-first in the low bits) per leaf.
bool huff_codes(const HuffTree &t, std::vector<uint8_t> &len, std::vector<uint64_t> &code);
bool huff_codes_into(const HuffTree &t, uint8_t *len, uint64_t *code);  // len / code: t.nleaf entries (pinned memory, say)
// BinTrie::serialize (huf.rs:305-321), pre-order: 0+symbol for a leaf, 1+left+right for a branch.
void huff_serialize_tree(const HuffTree &t, int sym_kind, const uint32_t *keys, std::vector<uint8_t> &out);
uint64_t huff_tree_bytes(int sym_kind, uint64_t n);  // bytes of the serialised decoder of n leaves
void huff_serialize_tree_into(const HuffTree &t, int sym_kind, const uint32_t *keys, uint8_t *out);  // huff_tree_bytes() bytes
int  huff_symbol_size(int sym_kind);
// total stream size of encode_all for this histogram
uint64_t huff_stream_size(int sym_kind, const uint64_t *counts, const uint8_t *len, uint64_t n);

// flat decoder trie: branch = (left, right) node indices, leaf = (kTrieLeaf, symbol key); node 0 is the root
constexpr uint32_t kTrieLeaf = 0xffffffffu;
struct TrieNode { uint32_t l, r; };
bool huff_parse_trie(int sym_kind, const uint8_t *bytes, uint64_t nbytes, uint64_t &pos, std::vector<TrieNode> &nodes);
bool huff_decode_host(const std::vector<TrieNode> &nodes, const uint8_t *payload, uint64_t payload_bytes, uint64_t nsyms,
                      uint32_t *keys_out, uint64_t *bytes_used);

// The same deserialisation as a table of LEAVES in pre-order, which is ascending order of their codes read as left-aligned
// binary fractions (a left child comes before its right sibling): code[i] = leaf i's path from the root in the TOP bits of a
// u64 (left = 0, right = 1), len[i] its depth, key[i] its symbol.  That table is all a decoder needs -- the symbol at a bit
// position is the last leaf whose code is <= the next 64 bits -- and the parallel decoder (k_hdecode.hip) searches it instead
// of walking nodes.  too_deep: some leaf lies deeper than kLeafMaxLen (no encoder of fewer than 2^32 symbols makes one; such
// a stream is decoded by the node walk instead).
constexpr uint32_t kLeafMaxLen = 56;
struct LeafTable {
    std::vector<uint64_t> code;
    std::vector<uint32_t> key;
    std::vector<uint8_t> len;
    uint32_t max_len = 0, min_len = 0;
    bool too_deep = false;
    uint64_t n() const { return code.size(); }
};
bool huff_parse_leaves(int sym_kind, const uint8_t *bytes, uint64_t nbytes, uint64_t &pos, LeafTable &t);

// Dec::deserialize + DecStream (huf.rs:323-348, 187-206, 366-374): read the trie at bytes[pos..],
// then decode nsyms symbols into keys_out.  Returns false where the reference yields None.
bool huff_decode_symbols(int sym_kind, const uint8_t *bytes, uint64_t nbytes, uint64_t &pos, uint64_t nsyms,
                         uint32_t *keys_out);

// little-endian writers (src/ser.rs)
void put_u32(std::vector<uint8_t> &o, uint32_t v);
void put_u64(std::vector<uint8_t> &o, uint64_t v);
bool get_u32(const uint8_t *b, uint64_t n, uint64_t &pos, uint32_t &v);
bool get_u64(const uint8_t *b, uint64_t n, uint64_t &pos, uint64_t &v);

}  // namespace cniic
