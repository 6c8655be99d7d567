// codec.cpp -- host-side mirror of the reference's Codec trait (src/codec.rs:14-19) for the four
// codecs on the hot path, driving the HIP kernels.  Same names (Codec::name), same lossless flags,
// same --codec= expressions (FromStr impls), same wire format, same failure points.
//
//   Hufman          src/codec/hufc.rs        dims + huf::encode_all over row-major pixels
//   ClusterColors   src/codec/clusterc.rs:17 dedup -> K-means -> remap -> Hufman
//   VoronoiCluster  src/codec/clusterc.rs:147 5-D K-means, centroids only; Voronoi repaint on decode
//   Delta           src/codec/hilbertc.rs:404 Hilbert gather -> neighbour delta -> huf::encode_all
//   Hilbert{RLE(0)} src/codec/hilbertc.rs:12  Hilbert gather -> exact run-length records
#include <algorithm>
#include <atomic>
#include <thread>

#include "codec.hpp"

#include <algorithm>
#include <cctype>
#include <cstring>
#include <map>
#include <memory>

#include "huff_host.hpp"

namespace cniic {

// ------------------------------------------------------------------ FromStr (codec.rs:41-59)
static bool match_fun_u32(const std::string &s, const char *const *names, uint32_t *arg) {
    // Regex::captures is an unanchored search (clusterc.rs:125-127, 281-283)
    for (size_t p = 0; p < s.size(); p++)
        for (int i = 0; names[i]; i++) {
            const size_t l = strlen(names[i]);
            if (s.compare(p, l, names[i]) != 0 || p + l >= s.size() || s[p + l] != '(') continue;
            size_t q = p + l + 1;
            if (q >= s.size() || !isdigit((unsigned char)s[q])) continue;
            unsigned long long v = 0;
            bool ok = true;
            while (q < s.size() && isdigit((unsigned char)s[q])) {
                v = v * 10 + (unsigned)(s[q] - '0');
                if (v > 0xffffffffull) { ok = false; break; }
                q++;
            }
            if (!ok || q >= s.size() || s[q] != ')') continue;
            *arg = (uint32_t)v;
            return true;
        }
    return false;
}

// Hilbert::from_str (hilbertc.rs:341-397): fun_call named ^[Hh]ilbert$ with one argument, `rle` or `rle(<f64>)`.
// Only the exact method (d == 0.0) is built; rle(d != 0) (a sequential running average) and zip are not.
static bool match_hilbert_rle(const std::string &s) {
    if (s.compare(0, 8, "hilbert(") != 0 && s.compare(0, 8, "Hilbert(") != 0) return false;
    if (s.size() < 10 || s.back() != ')') return false;
    const std::string arg = s.substr(8, s.size() - 9);
    if (arg == "rle") return true;
    if (arg.size() > 5 && arg.compare(0, 4, "rle(") == 0 && arg.back() == ')') {
        const std::string num = arg.substr(4, arg.size() - 5);
        char *end = nullptr;
        const double d = strtod(num.c_str(), &end);
        return end && *end == 0 && !num.empty() && d == 0.0;
    }
    return false;
}

bool parse_codec(const char *expr, CodecDesc *out) {
    if (!expr) return false;
    const std::string s(expr);
    // alternatives in the order of gen_all! (codec.rs:120-127); Hilbert and Zip are out of scope
    static const char *const cc[] = {"cluster-colors", "cluster-col", "clustercolors", "clustercol",
                                     "c-colors", "c-col", "ccolors", "ccol", nullptr};  // c(?:luster)?-?col(?:ors)?
    static const char *const vo[] = {"voronoi", nullptr};
    uint32_t k = 0;
    if (match_fun_u32(s, cc, &k)) { *out = {CODEC_CLUSTER_COLORS, k}; return true; }
    if (match_fun_u32(s, vo, &k)) { *out = {CODEC_VORONOI, k}; return true; }
    if (s == "delta") { *out = {CODEC_DELTA, 0}; return true; }  // prs::expect_name: ^delta$
    if (match_hilbert_rle(s)) { *out = {CODEC_HILBERT_RLE, 0}; return true; }
    if (s.size() == 6) {                                         // hufc.rs:54-59 eq_ignore_ascii_case
        std::string t = s;
        std::transform(t.begin(), t.end(), t.begin(), [](unsigned char ch) { return (char)tolower(ch); });
        if (t == "hufman") { *out = {CODEC_HUFMAN, 0}; return true; }
    }
    return false;
}

std::string codec_name(const CodecDesc &d) {
    switch (d.kind) {
    case CODEC_HUFMAN: return "Hufman";                                   // hufc.rs:42-44
    case CODEC_CLUSTER_COLORS: return "cluster-colors_" + std::to_string(d.arg);  // clusterc.rs:59-61
    case CODEC_VORONOI: return "voronoi_" + std::to_string(d.arg);        // clusterc.rs:191-193
    case CODEC_DELTA: return "delta";                                     // hilbertc.rs:433-435
    case CODEC_HILBERT_RLE: return "hilbert-rle";                         // hilbertc.rs:83-85
    }
    return "";
}

bool codec_is_lossless(const CodecDesc &d) { return d.kind == CODEC_HUFMAN || d.kind == CODEC_DELTA || d.kind == CODEC_HILBERT_RLE; }

// ------------------------------------------------------------------ output assembly
// The encoded stream (host-built header + device-packed payload) is assembled in HBM: directly in
// the caller's buffer when that is 4-byte aligned device memory, otherwise in a staging buffer
// that is copied out once.
struct StreamOut {
    Ctx *c;
    uint8_t *caller;
    uint64_t cap;
    uint64_t *len;
    bool direct = false;
    DevBuf staging;
    uint8_t *dev = nullptr;
    uint64_t total = 0;
    StreamOut(Ctx *ctx, uint8_t *out, uint64_t capacity, uint64_t *len_out) : c(ctx), caller(out), cap(capacity), len(len_out) {}
    int begin(const std::vector<uint8_t> &header, uint64_t payload_bytes) {
        CNIIC_TRY(begin_sized(header.size(), payload_bytes));
        return put_header(header);
    }
    // the header's bytes may follow the payload (put_header): its size is enough to place the payload
    int put_header(const std::vector<uint8_t> &header) {
        if (!header.empty()) CNIIC_HIP_TRY(c, hipMemcpyAsync(dev, header.data(), header.size(), hipMemcpyHostToDevice, c->stream));
        return CNIIC_OK;
    }
    int begin_sized(uint64_t header_bytes, uint64_t payload_bytes, bool zero = true) {
        total = header_bytes + payload_bytes;
        *len = total;
        if (total > cap) return c->fail(CNIIC_ERR_CAPACITY, "encode: stream is %llu bytes, capacity %llu",
                                        (unsigned long long)total, (unsigned long long)cap);
        const uint64_t padded = (total + 3) & ~3ull;
        direct = is_device_ptr(caller) && (reinterpret_cast<uintptr_t>(caller) & 3) == 0 && padded <= cap;
        if (direct) dev = caller;
        else { CNIIC_HIP_TRY(c, staging.alloc(padded + 16)); dev = staging.as<uint8_t>(); }
        if (zero) CNIIC_HIP_TRY(c, hipMemsetAsync(dev, 0, padded, c->stream));
        return CNIIC_OK;
    }
    int finish() {
        if (!direct && total)
            CNIIC_HIP_TRY(c, hipMemcpyAsync(caller, dev, total, is_device_ptr(caller) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        return CNIIC_OK;
    }
};

// ------------------------------------------------------------------ huf::encode_all (huf.rs:22-43)
// Symbols come either as pixels (rgb_d) or as packed keys (syms_d).  table_d holds the dense
// histogram on entry when have_hist, otherwise it is built here.
int huf_encode_all_dev(Ctx *c, int sym_kind, const uint8_t *rgb_d, uint32_t *syms_d, bool syms_scratch, uint64_t n,
                       uint32_t *table_d, bool have_hist, std::vector<uint8_t> &header, uint8_t *out, uint64_t cap,
                       uint64_t *len) {
    if (n == 0) return c->fail(CNIIC_ERR_BAD_ARG, "huf::encode_all on an empty stream (src/huf.rs:99 asserts)");
    const uint32_t bits = sym_kind == CNIIC_SYM_RGB ? 24 : 27;
    host_trace().mark("huf: enter");
    // 1. utils::count_freqs (huf.rs:30)
    if (!have_hist) {
        if (rgb_d) CNIIC_TRY(hist_rgb_dense(c, rgb_d, n, table_d));
        else CNIIC_TRY(hist_syms_dense(c, syms_d, n, table_d, bits));
    }
    CompactPlan plan;
    CNIIC_TRY(hist_compact_count(c, table_d, bits, &plan));
    const uint64_t U = plan.n_unique;
    DevBuf keys_d, counts_d;
    CNIIC_HIP_TRY(c, keys_d.alloc(U * 4));
    CNIIC_HIP_TRY(c, counts_d.alloc(U * 8));
    CNIIC_TRY(hist_compact_write(c, table_d, &plan, keys_d.as<uint32_t>(), counts_d.as<uint64_t>(), nullptr));
    // A large alphabet (a photograph's colours): the host only merges the tree; codes and the serialised decoder are the
    // GPU's (huff_tree_codes).  The counts land in pinned memory, the tree's arrays are written there, the keys stay put.
    const bool gpu_codes = U >= c->opt(CNIIC_OPT_HUF_GPU_CODES_MIN, "CNIIC_HUF_GPU_CODES_MIN", 32768) && U >= 2 && U < (1ull << 30) && n < (1ull << 32);
    std::vector<uint32_t> keys(gpu_codes ? 0 : U);
    std::vector<uint64_t> counts_v(gpu_codes ? 0 : U);
    uint64_t *counts = counts_v.data();
    uint32_t *left_h = nullptr, *right_h = nullptr, *nl_h = nullptr;
    if (gpu_codes) {
        CNIIC_HIP_TRY(c, ctx_pinned_huf(c, U * 8 + 3 * (U - 1) * 4 + 64));
        counts = static_cast<uint64_t *>(c->pinned_huf);
        left_h = reinterpret_cast<uint32_t *>(counts + U);
        right_h = left_h + (U - 1);
        nl_h = right_h + (U - 1);
    } else {
        CNIIC_HIP_TRY(c, hipMemcpyAsync(keys.data(), keys_d.p, U * 4, hipMemcpyDeviceToHost, c->stream));
    }
    // (large alphabet: the leaves come back sorted by (count, key) -- huff_sort_leaves_dev -- and the host only merges)
    DevBuf sort_a, sort_b, len_d, code_d, off_run_d;
    uint64_t nbits_runs = 0;
    bool tree_built = false;
    if (gpu_codes) {
        uint64_t *sorted_d = nullptr;
        CNIIC_HIP_TRY(c, sort_a.alloc(U * 8));
        CNIIC_HIP_TRY(c, sort_b.alloc(U * 8));
        CNIIC_TRY(huff_sort_leaves_dev(c, counts_d.as<uint64_t>(), (uint32_t)U, plan.max_count ? plan.max_count : n, sort_a.as<uint64_t>(), sort_b.as<uint64_t>(), &sorted_d));
        // (round 3) the tree, the codes and the leaves' places in the decoder without the host's merge, when the counts come in runs
        CNIIC_HIP_TRY(c, len_d.alloc(U));
        CNIIC_HIP_TRY(c, code_d.alloc(U * 8));
        CNIIC_HIP_TRY(c, off_run_d.alloc(U * 8));
        CNIIC_TRY(huff_tree_from_runs(c, sorted_d, counts_d.as<uint64_t>(), (uint32_t)U, sym_kind, len_d.as<uint8_t>(), code_d.as<uint64_t>(),
                                      off_run_d.as<uint64_t>(), &nbits_runs, &tree_built));
        if (!tree_built) CNIIC_HIP_TRY(c, hipMemcpyAsync(counts, sorted_d, U * 8, hipMemcpyDeviceToHost, c->stream));
    } else {
        CNIIC_HIP_TRY(c, hipMemcpyAsync(counts, counts_d.p, U * 8, hipMemcpyDeviceToHost, c->stream));
    }
    if (!c->huf_ev) CNIIC_HIP_TRY(c, hipEventCreateWithFlags(&c->huf_ev, hipEventDisableTiming));
    CNIIC_HIP_TRY(c, hipEventRecord(c->huf_ev, c->stream));
    // `delta` symbols on a buffer of ours: nothing to do while the host builds the tree -- the pack looks (length, code) up
    // in an LDS table of the cube of small differences (huff_pack_code32_hot).  Otherwise the GPU meanwhile turns every
    // symbol into its rank in the compacted list -- the one random read per symbol into the dense table (it now holds
    // rank + 1), which needs no code -- in place over the symbol stream when it is ours; the pack then reads that stream
    // and the small per-rank length / code tables.
    const bool hot_route = sym_kind == CNIIC_SYM_SIGNED && syms_d && syms_scratch && U < (1ull << 26) && (reinterpret_cast<uintptr_t>(syms_d) & 15) == 0;
    DevBuf ranks_own;
    uint32_t *ranks = syms_d;
    const bool inline_codes = U < (1ull << 26);  // (len, code) of a symbol in one u32 looked up by rank; else per-rank tables
    // (round 3) with the tree on the GPU there is nothing for that pass to hide behind: the pack's first pass looks every symbol up in the
    // dense table itself, once it holds (length, code) words -- one random read per symbol instead of two (hufman 4096^2: -0.25 ms)
    const bool direct = gpu_codes && !hot_route && inline_codes;
    if (!hot_route) {
        if (!syms_d || !syms_scratch) { CNIIC_HIP_TRY(c, ranks_own.alloc(n * 4 + 16)); ranks = ranks_own.as<uint32_t>(); }
        if (!direct) CNIIC_TRY(huff_rank_stream(c, syms_d, rgb_d, n, table_d, ranks, !inline_codes));
    }
    CNIIC_HIP_TRY(c, hipEventSynchronize(c->huf_ev));
    host_trace().mark("huf: hist + compaction + D2H");
    // build() (huf.rs:31) and the serialised decoder (huf.rs:34)
    if (!c->huf_scratch) c->huf_scratch = std::make_shared<HuffScratch>();
    HuffScratch *scratch = static_cast<HuffScratch *>(c->huf_scratch.get());
    if (!len_d.p) CNIIC_HIP_TRY(c, len_d.alloc(U));
    if (!code_d.p) CNIIC_HIP_TRY(c, code_d.alloc(U * 8));
    uint64_t nbits = 0, header_bytes = 0;
    StreamOut so(c, out, cap, len);
    if (gpu_codes) {
        DevBuf tree_d, off_d, totals_d;
        const uint64_t *off_use = off_run_d.as<uint64_t>();
        if (tree_built) {
            nbits = nbits_runs;
        } else {
            uint32_t root = 0;
            if (!huff_merge_sorted_into(counts /* sorted leaves */, U, left_h, right_h, nl_h, &root, scratch))
                return c->fail(CNIIC_ERR_BAD_ARG, "huffman: cannot build code (alphabet %llu)", (unsigned long long)U);
            host_trace().mark("huf: tree (host)");
            CNIIC_HIP_TRY(c, tree_d.alloc(3 * (U - 1) * 4));
            CNIIC_HIP_TRY(c, off_d.alloc(U * 8));
            CNIIC_HIP_TRY(c, totals_d.alloc(16));
            CNIIC_HIP_TRY(c, hipMemcpyAsync(tree_d.p, left_h, 3 * (U - 1) * 4, hipMemcpyHostToDevice, c->stream));
            const uint32_t *left_d = tree_d.as<uint32_t>(), *right_d = left_d + (U - 1), *nl_d = right_d + (U - 1);
            CNIIC_TRY(huff_tree_codes(c, left_d, right_d, nl_d, counts_d.as<uint64_t>(), (uint32_t)U, root, sym_kind, len_d.as<uint8_t>(),
                                      code_d.as<uint64_t>(), off_d.as<uint64_t>(), totals_d.as<uint64_t>()));
            CNIIC_HIP_TRY(c, ctx_pinned_u(c));
            CNIIC_HIP_TRY(c, hipMemcpyAsync(&c->pinned_u[2], totals_d.p, 16, hipMemcpyDeviceToHost, c->stream));
            CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->pinned_u[3]) return c->fail(CNIIC_ERR_BAD_ARG, "huffman: cannot build code (alphabet %llu)", (unsigned long long)U);
            nbits = c->pinned_u[2];
            off_use = off_d.as<uint64_t>();
        }
        host_trace().mark("huf: codes (GPU)");
        const uint64_t trie_bytes = huff_tree_bytes(sym_kind, U), head = header.size();
        CNIIC_TRY(so.begin_sized(head + trie_bytes, (nbits + 7) / 8));
        CNIIC_TRY(so.put_header(header));
        CNIIC_TRY(huff_tree_serialize_dev(c, keys_d.as<uint32_t>(), off_use, (uint32_t)U, sym_kind, so.dev + head, trie_bytes));
        header_bytes = head + trie_bytes;  // (what the pack below starts behind; the decoder's bytes are on the device)
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));  // tree_d / off_d go back to the pool
        host_trace().mark("huf: serialise trie (GPU)");
    } else {
        HuffTree tree;
        std::vector<uint8_t> clen;
        std::vector<uint64_t> code;
        if (!huff_build_tree(counts, U, tree, scratch) || !huff_codes(tree, clen, code))
            return c->fail(CNIIC_ERR_BAD_ARG, "huffman: cannot build code (alphabet %llu)", (unsigned long long)U);
        host_trace().mark("huf: tree + codes (host)");
        huff_serialize_tree(tree, sym_kind, keys.data(), header);
        host_trace().mark("huf: serialise trie (host)");
        // 3. payload (huf.rs:37-41), packed in place behind the header
        for (uint64_t i = 0; i < U; i++) nbits += counts[i] * clen[i];
        CNIIC_TRY(so.begin(header, (nbits + 7) / 8));
        header_bytes = header.size();
        CNIIC_HIP_TRY(c, hipMemcpyAsync(len_d.p, clen.data(), U, hipMemcpyHostToDevice, c->stream));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(code_d.p, code.data(), U * 8, hipMemcpyHostToDevice, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));  // (clen / code are locals of this branch)
    }
    uint64_t packed_bits = 0;
    ScopedKernelTimer timer(c, "huff_pack");
    if (hot_route) {
        CNIIC_TRY(huff_pack_code32_hot(c, syms_d, n, table_d, keys_d.as<uint32_t>(), len_d.as<uint8_t>(), code_d.as<uint64_t>(), U, syms_d, so.dev,
                                       header_bytes * 8, &packed_bits));
    } else if (direct) {
        CNIIC_TRY(huff_pack_code32(c, syms_d, rgb_d, n, table_d, keys_d.as<uint32_t>(), len_d.as<uint8_t>(), code_d.as<uint64_t>(), U, ranks,
                                   so.dev, header_bytes * 8, &packed_bits));
    } else if (inline_codes) {
        DevBuf code32;
        CNIIC_HIP_TRY(c, code32.alloc(U * 4));
        CNIIC_TRY(huff_pack_code32(c, ranks, nullptr, n, code32.as<uint32_t>(), nullptr, len_d.as<uint8_t>(), code_d.as<uint64_t>(), U, ranks,
                                   so.dev, header_bytes * 8, &packed_bits));
    } else {
        CNIIC_TRY(huff_pack_ranks(c, ranks, n, len_d.as<uint8_t>(), code_d.as<uint64_t>(), so.dev, header_bytes * 8, &packed_bits));
    }
    timer.stop(1);
    host_trace().mark("huf: pack");
    if (packed_bits != nbits)
        return c->fail(CNIIC_ERR_HIP, "huffman: packed %llu bits, histogram predicts %llu", (unsigned long long)packed_bits,
                       (unsigned long long)nbits);
    const int rc_fin = so.finish();
    host_trace().mark("huf: finish");
    host_trace().dump();
    return rc_fin;
}

// ------------------------------------------------------------------ Hufman::encode (hufc.rs:12-17)
static int encode_hufman(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint8_t *out, uint64_t cap, uint64_t *len) {
    const uint64_t n = (uint64_t)w * h;
    std::vector<uint8_t> header;
    put_u32(header, w);  // img.dimensions().serialize (hufc.rs:13)
    put_u32(header, h);
    uint32_t *table = nullptr;
    CNIIC_TRY(dense_table(c, 24, &table));
    return huf_encode_all_dev(c, CNIIC_SYM_RGB, rgb_d, nullptr, false, n, table, false, header, out, cap, len);
}

// ------------------------------------------------------------------ ClusterColors::encode (clusterc.rs:18-53)
// Split in two so that a multi-GPU caller can all-reduce between the pieces:
//   cc_prepare  dense colour counts -> distinct colours (ascending key = point order, clusterc.rs:21-24)
//               + K-means state (kmeans::init, kmeans.rs:80-90)
//   (K-means loop: km_rgbw_run on one GPU, or assign / all-reduce / update driven by the caller)
//   cc_finish   clusters -> colour lookup -> Hufman.encode of the reduced image (clusterc.rs:31-52)
CcSession::~CcSession() { if (km) km_rgbw_destroy(km); }
HostTrace &host_trace() { static thread_local HostTrace t; return t; }


int cc_prepare(Ctx *c, uint32_t *table_counts_d, uint32_t K, const cniic_kmeans_opts *opts, uint32_t shard, uint32_t nshards,
               void *partials_dev, CcSession **out, const uint32_t *occ_d) {
    if (K == 0) return c->fail(CNIIC_ERR_BAD_ARG, "cluster-colors(0)");
    auto s = std::make_unique<CcSession>();
    s->c = c; s->K = K; s->table = table_counts_d;
    CompactPlan plan;
    DevBuf cell_count;  // occupied bins per K-means colour cell, counted by the compaction on its way
    CNIIC_HIP_TRY(c, cell_count.alloc((uint64_t)kNumCells * 4));
    CNIIC_HIP_TRY(c, hipMemsetAsync(cell_count.p, 0, (uint64_t)kNumCells * 4, c->stream));
    CNIIC_TRY(hist_compact_count(c, table_counts_d, 24, &plan, cell_count.as<uint32_t>()));
    host_trace().mark("compact_count+sync");
    const uint64_t U = plan.n_unique;
    s->U = U;
    uint64_t Ug = 0;
    if (occ_d) {  // the reference's point list is the union's; this rank holds its own share of it
        CNIIC_TRY(gidx_build(c, occ_d, s->gbits, s->gprefix, &Ug));
        s->local_points = true;
    }
    if ((occ_d ? Ug : U) / K == 0 || U == 0)
        return c->fail(CNIIC_ERR_TOO_FEW_POINTS, "kmeans: %llu distinct colours for %u clusters (src/kmeans.rs:68)",
                       (unsigned long long)(occ_d ? Ug : U), K);
    CNIIC_HIP_TRY(c, s->keys_d.alloc(U * 4));
    CNIIC_HIP_TRY(c, s->weight_d.alloc(U * 4));
    CNIIC_TRY(hist_compact_write(c, table_counts_d, &plan, s->keys_d.as<uint32_t>(), nullptr, s->weight_d.as<uint32_t>()));
    host_trace().mark("compact_write enq");
    // kmeans::cluster (clusterc.rs:28); the table now maps key -> rank + 1
    CNIIC_TRY(km_rgbw_create(c, s->keys_d.as<uint32_t>(), s->weight_d.as<uint32_t>(), U, shard, nshards, K, opts, partials_dev,
                             table_counts_d, &s->km, cell_count.as<uint32_t>(), occ_d ? s->gbits.p : nullptr,
                             occ_d ? s->gprefix.as<uint32_t>() : nullptr, Ug));
    host_trace().mark("km_create");
    *out = s.release();
    return CNIIC_OK;
}

int cc_prepare_image(Ctx *c, const uint8_t *rgb_d, uint64_t npx, uint32_t K, const cniic_kmeans_opts *opts, CcSession **out) {
    if (K == 0) return c->fail(CNIIC_ERR_BAD_ARG, "cluster-colors(0)");
    auto s = std::make_unique<CcSession>();
    s->c = c; s->K = K; s->sp_mode = true;
    CNIIC_TRY(sp_build(c, rgb_d, npx, &s->sp));   // count_freqs (clusterc.rs:21): distinct colours per cell, occupancy bitmap
    // kmeans::cluster (clusterc.rs:28): the point list is known as the bitmap and its cell-major copy is written below.
    // The number of distinct colours is still on the device: the state is sized for the most there can be, its
    // set-up kernels read the count there, and the host only waits for it after all of them are enqueued.
    const uint64_t Umax = std::min<uint64_t>(npx, 1ull << 24);
    const uint64_t *U_dev = s->sp.total.as<uint64_t>();
    CNIIC_TRY(km_rgbw_create(c, nullptr, nullptr, Umax, 0, 1, K, opts, nullptr, nullptr, &s->km, s->sp.cell_count.as<uint32_t>(), s->sp.bits.p,
                             s->sp.wprefix.as<uint32_t>(), Umax, true, U_dev));
    uint32_t *cell_start, *ckeys, *cweight;
    km_rgbw_cell_arrays(s->km, &cell_start, &ckeys, &cweight);
    CNIIC_TRY(sp_emit(c, &s->sp, cell_start, ckeys, cweight, km_rgbw_labels_internal(s->km, nullptr), km_rgbw_is_wide(s->km), K, s->sp.bits.p,
                      s->sp.wprefix.as<uint32_t>(), 0, U_dev));
    CNIIC_TRY(sp_wait_count(c, &s->sp));
    s->U = s->sp.U;
    if (s->U / K == 0)
        return c->fail(CNIIC_ERR_TOO_FEW_POINTS, "kmeans: %llu distinct colours for %u clusters (src/kmeans.rs:68)", (unsigned long long)s->U, K);
    CNIIC_TRY(km_rgbw_set_points(s->km, s->U));
    host_trace().mark("km_create + emit enq");
    *out = s.release();
    return CNIIC_OK;
}

int cc_image_begin(Ctx *c, const uint8_t *rgb_d, uint64_t npx, CcSession **out) {
    auto s = std::make_unique<CcSession>();
    s->c = c; s->sp_mode = true; s->local_points = true;
    CNIIC_TRY(sp_build(c, rgb_d, npx, &s->sp));
    *out = s.release();
    return CNIIC_OK;
}

int cc_image_create(CcSession *s, const uint32_t *occ_d, uint32_t K, const cniic_kmeans_opts *opts, void *partials_dev) {
    Ctx *c = s->c;
    if (!s->sp_mode || s->km) return c->fail(CNIIC_ERR_BAD_ARG, "cc_image_create: needs a session from cc_image_begin, once");
    if (K == 0) return c->fail(CNIIC_ERR_BAD_ARG, "cluster-colors(0)");
    s->K = K;
    // The reference's point list is the colours of ALL images: the summed occupancy as a bitmap + prefix.  Its length (and
    // this image's own colour count) stay on the device while the state is set up -- sized for the most there can be, the
    // set-up kernels read the counts where they are -- and the host fetches both once everything is enqueued.
    CNIIC_HIP_TRY(c, ctx_pinned_u(c));
    uint64_t *Ug_h = c->pinned_u + 1;
    CNIIC_TRY(gidx_build(c, occ_d, s->gbits, s->gprefix, Ug_h, &s->gtotal));
    const uint64_t Umax = std::min<uint64_t>(s->sp.npx, 1ull << 24);
    const uint64_t *Ug_dev = s->gtotal.as<uint64_t>();
    CNIIC_TRY(km_rgbw_create(c, nullptr, nullptr, Umax, 0, 1, K, opts, partials_dev, nullptr, &s->km, s->sp.cell_count.as<uint32_t>(), s->gbits.p,
                             s->gprefix.as<uint32_t>(), 1ull << 24, true, Ug_dev));
    uint32_t *cell_start, *ckeys, *cweight;
    km_rgbw_cell_arrays(s->km, &cell_start, &ckeys, &cweight);
    CNIIC_TRY(sp_emit(c, &s->sp, cell_start, ckeys, cweight, km_rgbw_labels_internal(s->km, nullptr), km_rgbw_is_wide(s->km), K, s->gbits.p,
                      s->gprefix.as<uint32_t>(), 0, Ug_dev));
    // this image's own colour count, fetched again here: the context's pinned slot sp_build copied it to may have been
    // rewritten since by another session of the same context (a second cniic_cc_image_begin, a plain encode)
    CNIIC_HIP_TRY(c, hipMemcpyAsync(c->pinned_u + 2, s->sp.total.p, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));  // (the all-reduced occupancy had to arrive anyway)
    const uint64_t Ug = *Ug_h;
    s->sp.U = c->pinned_u[2];
    s->U = s->sp.U;
    if (Ug / K == 0 || s->U == 0)
        return c->fail(CNIIC_ERR_TOO_FEW_POINTS, "kmeans: %llu distinct colours for %u clusters (src/kmeans.rs:68)", (unsigned long long)Ug, K);
    return km_rgbw_set_points(s->km, s->U, Ug);
}

int cc_finish(CcSession *s, const uint8_t *rgb_d, uint32_t w, uint32_t h, const uint32_t *local_counts_d, uint8_t *out,
              uint64_t cap, uint64_t *len, cniic_kmeans_stats *stats) {
    Ctx *c = s->c;
    const uint32_t K = s->K;
    const uint64_t U = s->U, n = (uint64_t)w * h;
    KmRgbwState *km = s->km;
    std::vector<uint8_t> cent(3 * (size_t)K);
    std::vector<uint64_t> members(K), wsum(K);
    cniic_kmeans_stats st{};
    // The result block starts towards the host; behind it goes everything that needs no code table - the
    // colour -> label table and the label of every pixel (the one random read per pixel) - so that the GPU
    // is busy while the host waits for the block and builds the tree.
    const bool wide = km_rgbw_is_wide(km);
    DevBuf lab_d, key2label, pixlab;
    CNIIC_HIP_TRY(c, pixlab.alloc(n * (wide ? 2 : 1) + 16));
    DevBuf lw;
    bool lw_early = false;
    // (twice only if a persistent K-means launch that nobody had waited for turns out to have given up: km_rgbw_result_end has run the
    // launch-per-iteration loop by then and answers kKmRetry -- what was enqueued on the labels is enqueued again)
    for (int attempt = 0;; attempt++) {
    CNIIC_TRY(km_rgbw_result_begin(km));
    if (s->sp_mode) {  // every pixel's label from the partition: no table of 2^24 entries, no random read
        uint32_t *cell_start, *ckeys, *cweight;
        km_rgbw_cell_arrays(km, &cell_start, &ckeys, &cweight);
        if (s->local_points && K <= 4096) {
            // shared palette: THIS image's pixels per cluster (below) -- asked for first, so that the answer travels while
            // the pixel labels are computed instead of stalling the stream after them
            CNIIC_HIP_TRY(c, ctx_pinned_u(c));
            CNIIC_HIP_TRY(c, lw.alloc((uint64_t)K * 8));
            CNIIC_HIP_TRY(c, hipMemsetAsync(lw.p, 0, (uint64_t)K * 8, c->stream));
            CNIIC_TRY(local_cluster_weights(c, ckeys, km_rgbw_labels_internal(km, nullptr), wide, U, nullptr, K, lw.as<uint64_t>(), cweight));
            CNIIC_HIP_TRY(c, hipMemcpyAsync(c->pinned_u + 8, lw.p, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
            if (!c->u_ev) CNIIC_HIP_TRY(c, hipEventCreateWithFlags(&c->u_ev, hipEventDisableTiming));
            CNIIC_HIP_TRY(c, hipEventRecord(c->u_ev, c->stream));
            lw_early = true;
        }
        CNIIC_TRY(sp_pixel_labels(c, &s->sp, rgb_d, cell_start, ckeys, km_rgbw_labels_internal(km, nullptr), wide, pixlab.p));
    } else {
        CNIIC_HIP_TRY(c, lab_d.alloc(U * (wide ? 2 : 1)));
        CNIIC_TRY(km_rgbw_labels_canonical(km, lab_d.p));
        CNIIC_HIP_TRY(c, key2label.alloc((1ull << 24) * (wide ? 2 : 1)));
        CNIIC_TRY(scatter_labels_by_key(c, s->keys_d.as<uint32_t>(), lab_d.p, wide, U, key2label.p));
        CNIIC_TRY(pixel_labels(c, rgb_d, n, key2label.p, wide, pixlab.p));
    }
    host_trace().mark("labels + pixel labels enq");
    const int rc_res = km_rgbw_result_end(km, cent.data(), members.data(), wsum.data(), &st);
    if (rc_res == kKmRetry && attempt == 0) continue;
    CNIIC_TRY(rc_res);
    break;
    }
    host_trace().mark("km_result");
    if (stats) *stats = st;
    // check_enough_active_clusters (kmeans.rs:41-57)
    uint64_t min_cc = (uint64_t)(0.99 * (double)K);
    if (U < min_cc) min_cc = U;
    if (st.active < min_cc)
        return c->fail(CNIIC_ERR_FEW_ACTIVE, "Not enough active clusters: requested %u, got %llu (min allowed: %llu)", K,
                       (unsigned long long)st.active, (unsigned long long)min_cc);
    if (lw_early) {
        CNIIC_HIP_TRY(c, hipEventSynchronize(c->u_ev));
        for (uint32_t k = 0; k < K; k++) { wsum[k] = c->pinned_u[8 + k]; members[k] = wsum[k] ? 1 : 0; }
    } else if (local_counts_d || s->local_points) {
        // shared palette over several images: THIS image's pixels per cluster (its reduced image is
        // what Hufman.encode sees, clusterc.rs:52), from its own colour counts
        CNIIC_HIP_TRY(c, lw.alloc((uint64_t)K * 8));
        CNIIC_HIP_TRY(c, hipMemsetAsync(lw.p, 0, (uint64_t)K * 8, c->stream));
        if (s->sp_mode) {  // cell-major colours, labels and pixel counts of this image: any common order will do
            uint32_t *cell_start, *ckeys, *cweight;
            km_rgbw_cell_arrays(km, &cell_start, &ckeys, &cweight);
            CNIIC_TRY(local_cluster_weights(c, ckeys, km_rgbw_labels_internal(km, nullptr), wide, U, nullptr, K, lw.as<uint64_t>(), cweight));
        } else {
            CNIIC_TRY(local_cluster_weights(c, s->keys_d.as<uint32_t>(), lab_d.p, wide, U, local_counts_d, K, lw.as<uint64_t>(),
                                            s->weight_d.as<uint32_t>()));
        }
        CNIIC_HIP_TRY(c, hipMemcpyAsync(wsum.data(), lw.p, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (uint32_t k = 0; k < K; k++) members[k] = wsum[k] ? 1 : 0;
    }
    // Histogram of the colour-reduced image = per-centroid-colour sum of member weights
    // (what count_freqs inside Hufman.encode would find, clusterc.rs:52 -> huf.rs:30).
    std::vector<std::pair<uint32_t, uint64_t>> kc;
    kc.reserve(K);
    for (uint32_t k = 0; k < K; k++)
        if (members[k]) kc.emplace_back(((uint32_t)cent[3 * k] << 16) | ((uint32_t)cent[3 * k + 1] << 8) | cent[3 * k + 2], wsum[k]);
    std::sort(kc.begin(), kc.end());  // ascending colour, equal colours (two clusters with one mean) adjacent in cluster order
    std::vector<uint32_t> skeys;
    std::vector<uint64_t> scounts;
    for (auto &e : kc) {
        if (!skeys.empty() && skeys.back() == e.first) scounts.back() += e.second;
        else { skeys.push_back(e.first); scounts.push_back(e.second); }
    }
    host_trace().mark("map");
    HuffTree tree;
    std::vector<uint8_t> slen;
    std::vector<uint64_t> scode;
    if (!huff_build_tree(scounts.data(), scounts.size(), tree) || !huff_codes(tree, slen, scode))
        return c->fail(CNIIC_ERR_BAD_ARG, "huffman: cannot build code");
    host_trace().mark("build+codes");
    std::vector<uint8_t> header;
    put_u32(header, w);
    put_u32(header, h);
    huff_serialize_tree(tree, CNIIC_SYM_RGB, skeys.data(), header);
    uint64_t nbits = 0;
    for (size_t i = 0; i < scounts.size(); i++) nbits += scounts[i] * slen[i];
    // per-cluster code; every pixel reaches it through a dense colour -> cluster-label table:
    // reduced_colors.get(original_colour) (clusterc.rs:43-47) fused with Enc::encode (huf.rs:137-148)
    std::vector<uint8_t> clen(K, 0);
    std::vector<uint64_t> ccode(K, 0);
    for (uint32_t k = 0; k < K; k++) {
        if (!members[k]) continue;
        uint32_t key = ((uint32_t)cent[3 * k] << 16) | ((uint32_t)cent[3 * k + 1] << 8) | cent[3 * k + 2];
        size_t si = std::lower_bound(skeys.begin(), skeys.end(), key) - skeys.begin();
        clen[k] = slen[si];
        ccode[k] = scode[si];
    }
    host_trace().mark("tree+codes (host)");
    StreamOut so(c, out, cap, len);
    CNIIC_TRY(so.begin(header, (nbits + 7) / 8));
    host_trace().mark("so.begin");
    DevBuf clen_d, ccode_d;
    CNIIC_HIP_TRY(c, clen_d.alloc(K));
    CNIIC_HIP_TRY(c, ccode_d.alloc((uint64_t)K * 8));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(clen_d.p, clen.data(), K, hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(ccode_d.p, ccode.data(), (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
    uint64_t packed_bits = 0;
    {
        ScopedKernelTimer t(c, "huff_pack");
        CNIIC_TRY(huff_pack_labels(c, pixlab.p, n, wide, K, clen_d.as<uint8_t>(), ccode_d.as<uint64_t>(), so.dev,
                                   (uint64_t)header.size() * 8, &packed_bits));
        t.stop(1);
    }
    host_trace().mark("pack (+sync)");
    if (packed_bits != nbits)
        return c->fail(CNIIC_ERR_HIP, "cluster-colors: packed %llu bits, histogram predicts %llu",
                       (unsigned long long)packed_bits, (unsigned long long)nbits);
    const int rc_fin = so.finish();
    host_trace().mark("so.finish");
    return rc_fin;
}

// A batch of F equally sized frames (contiguous in rgb_d) coded with ONE palette -- north_star config 4: the K-means ran over the
// union of all the pixels (of all ranks); every frame is then its own Hufman stream (clusterc.rs:31-52 per frame: the reduced
// frame's own histogram, tree and payload), written at out + f * stride.  One pass gives every pixel's label, one kernel the
// pixels per (frame, cluster), the host builds the F small trees on a few threads and the F packs are enqueued back to back.
int cc_finish_frames(CcSession *s, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t F, uint8_t *out, uint64_t stride, uint64_t *lens,
                     cniic_kmeans_stats *stats) {
    Ctx *c = s->c;
    const uint32_t K = s->K;
    const uint64_t U = s->U, npf = (uint64_t)w * h, n = npf * F;
    KmRgbwState *km = s->km;
    host_trace().mark("frames: enter");
    if (!F || !npf) return c->fail(CNIIC_ERR_BAD_ARG, "cc_finish_frames: empty batch");
    if (stride & 3) return c->fail(CNIIC_ERR_BAD_ARG, "cc_finish_frames: the stride between streams must be a multiple of 4");
    if (s->sp_mode && s->sp.npx != n) return c->fail(CNIIC_ERR_BAD_ARG, "cc_finish_frames: the session was opened on %llu pixels, the batch has %llu",
                                                     (unsigned long long)s->sp.npx, (unsigned long long)n);
    std::vector<uint8_t> cent(3 * (size_t)K);
    std::vector<uint64_t> members(K), wsum(K);
    cniic_kmeans_stats st{};
    CNIIC_TRY(km_rgbw_result_begin(km));
    const bool wide = km_rgbw_is_wide(km);
    const uint64_t lb = wide ? 2 : 1;
    DevBuf lab_d, key2label, pixlab, pixlab_al, cnt_d;
    host_trace().mark("frames: result_begin");
    CNIIC_HIP_TRY(c, pixlab.alloc(n * lb + 16));
    host_trace().mark("frames: alloc pixel labels");
    if (s->sp_mode) {
        uint32_t *cell_start, *ckeys, *cweight;
        km_rgbw_cell_arrays(km, &cell_start, &ckeys, &cweight);
        CNIIC_TRY(sp_pixel_labels(c, &s->sp, rgb_d, cell_start, ckeys, km_rgbw_labels_internal(km, nullptr), wide, pixlab.p));
        host_trace().mark("frames: pixel labels enqueued");
    } else {
        CNIIC_HIP_TRY(c, lab_d.alloc(U * lb));
        CNIIC_TRY(km_rgbw_labels_canonical(km, lab_d.p));
        CNIIC_HIP_TRY(c, key2label.alloc((1ull << 24) * lb));
        CNIIC_TRY(scatter_labels_by_key(c, s->keys_d.as<uint32_t>(), lab_d.p, wide, U, key2label.p));
        CNIIC_TRY(pixel_labels(c, rgb_d, n, key2label.p, wide, pixlab.p));
    }
    // frames whose label run does not start on a 16-byte boundary are moved apart (the pack reads 16 labels per load)
    const void *labs = pixlab.p;
    uint64_t lab_stride = npf;
    if ((npf * lb) & 15) {
        lab_stride = (npf + 15) & ~15ull;
        CNIIC_HIP_TRY(c, pixlab_al.alloc(lab_stride * lb * F + 16));
        CNIIC_HIP_TRY(c, hipMemcpy2DAsync(pixlab_al.p, lab_stride * lb, pixlab.p, npf * lb, npf * lb, F, hipMemcpyDeviceToDevice, c->stream));
        labs = pixlab_al.p;
    }
    CNIIC_HIP_TRY(c, cnt_d.alloc((uint64_t)F * K * 4));
    CNIIC_TRY(frame_label_hist(c, labs, npf, lab_stride, F, wide, K, cnt_d.as<uint32_t>()));
    const bool gpu_trees = !wide && K <= 256 && c->opt(CNIIC_OPT_FRAME_TREES_HOST, "CNIIC_FRAME_TREES_HOST", 0) == 0;
    std::vector<uint32_t> cnt(gpu_trees ? 0 : (size_t)F * K);
    if (!gpu_trees) CNIIC_HIP_TRY(c, hipMemcpyAsync(cnt.data(), cnt_d.p, cnt.size() * 4, hipMemcpyDeviceToHost, c->stream));
    host_trace().mark("frames: hist enqueued");
    CNIIC_TRY(km_rgbw_result_end(km, cent.data(), members.data(), wsum.data(), &st));
    if (stats) *stats = st;
    uint64_t min_cc = (uint64_t)(0.99 * (double)K);  // check_enough_active_clusters (kmeans.rs:41-57)
    if (U < min_cc) min_cc = U;
    if (st.active < min_cc)
        return c->fail(CNIIC_ERR_FEW_ACTIVE, "Not enough active clusters: requested %u, got %llu (min allowed: %llu)", K,
                       (unsigned long long)st.active, (unsigned long long)min_cc);
    host_trace().mark("frames: result_end (sync)");
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    host_trace().mark("frames: wait for the histograms");
    if (gpu_trees) {
        // ---- K <= 256: codes, code tables and stream headers of all frames by one kernel (k_frame_trees); the host sees the
        // lengths (it owes them to the caller and must hold them against the stride before anything is packed) and nothing else
        const bool direct = is_device_ptr(out) && (reinterpret_cast<uintptr_t>(out) & 3) == 0;
        DevBuf staging, clen_d, ccode_d, meta_d;
        uint8_t *dev = out;
        if (!direct) { CNIIC_HIP_TRY(c, staging.alloc(stride * F + 16)); dev = staging.as<uint8_t>(); }
        CNIIC_HIP_TRY(c, hipMemsetAsync(dev, 0, stride * F, c->stream));
        CNIIC_HIP_TRY(c, clen_d.alloc((size_t)F * K));
        CNIIC_HIP_TRY(c, ccode_d.alloc((size_t)F * K * 8));
        CNIIC_HIP_TRY(c, meta_d.alloc((size_t)F * 24 + 8));  // bit base, payload bits, stream length per frame; error word
        uint64_t *bb_d = meta_d.as<uint64_t>(), *nb_d = bb_d + F, *ln_d = nb_d + F;
        uint32_t *err_d = reinterpret_cast<uint32_t *>(ln_d + F);
        CNIIC_HIP_TRY(c, hipMemsetAsync(err_d, 0, 8, c->stream));
        CNIIC_TRY(frame_trees(c, cnt_d.as<uint32_t>(), km_rgbw_centroids_dev(km), F, K, w, h, dev, stride, clen_d.as<uint8_t>(), ccode_d.as<uint64_t>(), bb_d, nb_d, ln_d, err_d));
        std::vector<uint64_t> meta((size_t)F * 3 + 1);
        CNIIC_HIP_TRY(c, hipMemcpyAsync(meta.data(), meta_d.p, (size_t)F * 24 + 8, hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        host_trace().mark("frames: trees, codes, headers (GPU) + lengths back");
        const uint32_t err = (uint32_t)meta[(size_t)F * 3];
        if (err & 3u) return c->fail(CNIIC_ERR_BAD_ARG, "huffman: cannot build code");
        for (uint32_t f = 0; f < F; f++) {
            lens[f] = meta[2 * (size_t)F + f];
            if ((err & 4u) || ((lens[f] + 3) & ~3ull) > stride)
                return c->fail(CNIIC_ERR_CAPACITY, "encode: stream of frame %u is %llu bytes, %llu between streams", f, (unsigned long long)lens[f], (unsigned long long)stride);
        }
        std::vector<uint64_t> totals(F, 0);
        CNIIC_TRY(huff_pack_labels_frames(c, labs, npf, lab_stride, F, wide, K, clen_d.as<uint8_t>(), ccode_d.as<uint64_t>(), dev, stride, nullptr, totals.data(), bb_d));
        host_trace().mark("frames: pack (+sync)");
        for (uint32_t f = 0; f < F; f++)
            if (totals[f] != meta[(size_t)F + f])
                return c->fail(CNIIC_ERR_HIP, "cluster-colors: frame %u packed %llu bits, its histogram predicts %llu", f, (unsigned long long)totals[f], (unsigned long long)meta[(size_t)F + f]);
        if (!direct) {
            CNIIC_HIP_TRY(c, hipMemcpyAsync(out, dev, stride * F, is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
            CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        host_trace().dump();
        return CNIIC_OK;
    }
    // ---- per frame on the host: histogram of the reduced frame = its pixels per centroid COLOUR (two clusters with one mean
    // are one symbol), tree, serialised decoder, per-cluster code
    std::vector<std::vector<uint8_t>> headers(F);
    std::vector<uint8_t> clen((size_t)F * K, 0);
    std::vector<uint64_t> ccode((size_t)F * K, 0), nbits(F, 0);
    std::atomic<int> bad{0};
    auto one_frame = [&](uint32_t f) {
        const uint32_t *fc = cnt.data() + (size_t)f * K;
        std::vector<std::pair<uint32_t, uint64_t>> kc;
        kc.reserve(K);
        for (uint32_t k = 0; k < K; k++)
            if (fc[k]) kc.emplace_back(((uint32_t)cent[3 * k] << 16) | ((uint32_t)cent[3 * k + 1] << 8) | cent[3 * k + 2], fc[k]);
        std::sort(kc.begin(), kc.end());
        std::vector<uint32_t> skeys;
        std::vector<uint64_t> scounts;
        for (auto &e : kc) {
            if (!skeys.empty() && skeys.back() == e.first) scounts.back() += e.second;
            else { skeys.push_back(e.first); scounts.push_back(e.second); }
        }
        HuffTree tree;
        std::vector<uint8_t> slen;
        std::vector<uint64_t> scode;
        if (!huff_build_tree(scounts.data(), scounts.size(), tree) || !huff_codes(tree, slen, scode)) { bad = 1; return; }
        std::vector<uint8_t> &hd = headers[f];
        put_u32(hd, w);
        put_u32(hd, h);
        huff_serialize_tree(tree, CNIIC_SYM_RGB, skeys.data(), hd);
        uint64_t nb = 0;
        for (size_t i = 0; i < scounts.size(); i++) nb += scounts[i] * slen[i];
        nbits[f] = nb;
        for (uint32_t k = 0; k < K; k++) {
            if (!fc[k]) continue;
            const uint32_t key = ((uint32_t)cent[3 * k] << 16) | ((uint32_t)cent[3 * k + 1] << 8) | cent[3 * k + 2];
            const size_t si = std::lower_bound(skeys.begin(), skeys.end(), key) - skeys.begin();
            clen[(size_t)f * K + k] = slen[si];
            ccode[(size_t)f * K + k] = scode[si];
        }
    };
    {
        const uint32_t nthr = std::max(1u, std::min({F, 16u, std::thread::hardware_concurrency()}));
        std::atomic<uint32_t> next{0};
        auto work = [&]() { for (uint32_t f; (f = next.fetch_add(1)) < F;) one_frame(f); };
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < nthr; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    host_trace().mark("frames: trees, codes, headers (host threads)");
    if (bad) return c->fail(CNIIC_ERR_BAD_ARG, "huffman: cannot build code");
    uint64_t hmax = 0;
    for (uint32_t f = 0; f < F; f++) {
        lens[f] = headers[f].size() + (nbits[f] + 7) / 8;
        if (((lens[f] + 3) & ~3ull) > stride)
            return c->fail(CNIIC_ERR_CAPACITY, "encode: stream of frame %u is %llu bytes, %llu between streams", f, (unsigned long long)lens[f], (unsigned long long)stride);
        hmax = std::max<uint64_t>(hmax, headers[f].size());
    }
    hmax = (hmax + 3) & ~3ull;
    // ---- output: device memory is written in place, a host buffer through a staging copy
    const bool direct = is_device_ptr(out) && (reinterpret_cast<uintptr_t>(out) & 3) == 0;
    DevBuf staging, hdr_d, clen_d, ccode_d;
    uint8_t *dev = out;
    if (!direct) { CNIIC_HIP_TRY(c, staging.alloc(stride * F + 16)); dev = staging.as<uint8_t>(); }
    CNIIC_HIP_TRY(c, hipMemsetAsync(dev, 0, stride * F, c->stream));
    std::vector<uint8_t> hdr_all(hmax * F, 0);
    std::vector<uint64_t> bit_base(F), totals(F, 0);
    for (uint32_t f = 0; f < F; f++) { memcpy(hdr_all.data() + hmax * f, headers[f].data(), headers[f].size()); bit_base[f] = headers[f].size() * 8; }
    CNIIC_HIP_TRY(c, hdr_d.alloc(hdr_all.size()));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(hdr_d.p, hdr_all.data(), hdr_all.size(), hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpy2DAsync(dev, stride, hdr_d.p, hmax, hmax, F, hipMemcpyDeviceToDevice, c->stream));  // (zero padding behind a header = the pre-zeroed payload)
    CNIIC_HIP_TRY(c, clen_d.alloc(clen.size()));
    CNIIC_HIP_TRY(c, ccode_d.alloc(ccode.size() * 8));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(clen_d.p, clen.data(), clen.size(), hipMemcpyHostToDevice, c->stream));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(ccode_d.p, ccode.data(), ccode.size() * 8, hipMemcpyHostToDevice, c->stream));
    CNIIC_TRY(huff_pack_labels_frames(c, labs, npf, lab_stride, F, wide, K, clen_d.as<uint8_t>(), ccode_d.as<uint64_t>(), dev, stride, bit_base.data(), totals.data()));
    host_trace().mark("frames: copies + pack (+sync)");
    for (uint32_t f = 0; f < F; f++)
        if (totals[f] != nbits[f])
            return c->fail(CNIIC_ERR_HIP, "cluster-colors: frame %u packed %llu bits, its histogram predicts %llu", f, (unsigned long long)totals[f], (unsigned long long)nbits[f]);
    if (!direct) {
        CNIIC_HIP_TRY(c, hipMemcpyAsync(out, dev, stride * F, is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    host_trace().dump();
    return CNIIC_OK;
}

// images of at least this many pixels take the super-cell partition (k_points.hip); CNIIC_SP_MIN_PIXELS overrides (tests: 0)
static uint64_t sp_min_pixels(const Ctx *c) { return c->opt(CNIIC_OPT_SP_MIN_PIXELS, "CNIIC_SP_MIN_PIXELS", 1ull << 20); }

static int encode_cluster_colors(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K,
                                 const cniic_kmeans_opts *opts, uint8_t *out, uint64_t cap, uint64_t *len,
                                 cniic_kmeans_stats *stats) {
    const uint64_t n = (uint64_t)w * h;
    host_trace().mark("enter");
    if (n >= sp_min_pixels(c) && (reinterpret_cast<uintptr_t>(rgb_d) & 15) == 0 && !(opts && (opts->flags & CNIIC_KM_BRUTE_FORCE))) {
        CcSession *raw = nullptr;
        CNIIC_TRY(cc_prepare_image(c, rgb_d, n, K, opts, &raw));
        std::unique_ptr<CcSession> s(raw);
        CNIIC_TRY(km_rgbw_run(s->km, nullptr, /*may_defer=*/true));   // (cc_finish looks at how a persistent launch ended where it fetches the result)
        host_trace().mark("km_run");
        const int rc_all = cc_finish(s.get(), rgb_d, w, h, nullptr, out, cap, len, stats);
        host_trace().dump();
        return rc_all;
    }
    // count_freqs over the pixels (clusterc.rs:21)
    uint32_t *table = nullptr;
    CNIIC_TRY(dense_table(c, 24, &table));
    {
        ScopedKernelTimer t(c, "hist_rgb");
        CNIIC_TRY(hist_rgb_dense(c, rgb_d, n, table));
        t.stop(1);
    }
    host_trace().mark("hist (+timer sync)");
    CcSession *raw = nullptr;
    CNIIC_TRY(cc_prepare(c, table, K, opts, 0, 1, nullptr, &raw));
    std::unique_ptr<CcSession> s(raw);
    CNIIC_TRY(km_rgbw_run(s->km, nullptr, /*may_defer=*/true));
    host_trace().mark("km_run");
    const int rc_all = cc_finish(s.get(), rgb_d, w, h, nullptr, out, cap, len, stats);
    host_trace().dump();
    return rc_all;
}

// ------------------------------------------------------------------ VoronoiCluster::encode (clusterc.rs:148-166)
static int encode_voronoi(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K, const cniic_kmeans_opts *opts,
                          uint8_t *out, uint64_t cap, uint64_t *len, cniic_kmeans_stats *stats) {
    if (K == 0) return c->fail(CNIIC_ERR_BAD_ARG, "voronoi(0)");
    std::vector<cniic_colorpos> cent(K);
    cniic_kmeans_stats st{};
    CNIIC_TRY(km_xyrgb_run(c, rgb_d, w, h, K, opts, cent.data(), nullptr, nullptr, &st));
    if (stats) *stats = st;
    const uint64_t N = (uint64_t)w * h;
    uint64_t min_cc = (uint64_t)(0.99 * (double)K);
    if (N < min_cc) min_cc = N;
    if (st.active < min_cc)
        return c->fail(CNIIC_ERR_FEW_ACTIVE, "Not enough active clusters: requested %u, got %llu (min allowed: %llu)", K,
                       (unsigned long long)st.active, (unsigned long long)min_cc);
    std::vector<uint8_t> header;
    put_u32(header, w);            // clusterc.rs:156-158
    put_u32(header, h);
    put_u64(header, K);            // clusterc.rs:161 (usize -> u64)
    for (uint32_t k = 0; k < K; k++) {  // ColorPos::serialize clusterc.rs:250-257
        put_u32(header, cent[k].x);
        put_u32(header, cent[k].y);
        put_u64(header, 3);        // Rgb<u8> as a length-prefixed slice (ser.rs:210-214)
        header.push_back(cent[k].rgb[0]); header.push_back(cent[k].rgb[1]); header.push_back(cent[k].rgb[2]);
    }
    StreamOut so(c, out, cap, len);
    CNIIC_TRY(so.begin(header, 0));
    return so.finish();
}

// ------------------------------------------------------------------ Delta::encode (hilbertc.rs:405-415)
// The 32-bit route: symbols as packed keys (hilbert_delta + huf_encode_all_dev).  Taken when many differences fall
// outside the cube [-16, 15]^3 (a noisy image), or with CNIIC_DELTA_ROUTE=32.
static int encode_delta_syms32(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, std::vector<uint8_t> &header, uint8_t *out, uint64_t cap,
                               uint64_t *len) {
    const uint64_t n = (uint64_t)w * h;
    uint32_t *table = nullptr;
    CNIIC_TRY(dense_table(c, 27, &table));
    DevBuf syms;
    CNIIC_HIP_TRY(c, syms.alloc(n * 4));
    // one fused pass: Hilbert gather + DiffStream + count_freqs; the symbol stream is kept for the
    // second (bit-pack) pass instead of recomputing the scan as the reference does (huf.rs:30,38)
    CNIIC_TRY(hilbert_delta(c, rgb_d, w, h, syms.as<uint32_t>(), table));
    return huf_encode_all_dev(c, CNIIC_SYM_SIGNED, nullptr, syms.as<uint32_t>(), true, n, table, true, header, out, cap, len);
}

static int encode_delta(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint8_t *out, uint64_t cap, uint64_t *len) {
    const uint64_t n = (uint64_t)w * h;
    std::vector<uint8_t> header;
    put_u32(header, w);
    put_u32(header, h);
    if (n == 0) return c->fail(CNIIC_ERR_BAD_ARG, "delta: empty image (src/huf.rs:99 asserts)");
    if (c->opt(CNIIC_OPT_DELTA_ROUTE, "CNIIC_DELTA_ROUTE", 0) == 32)  // the 32-bit route whatever the image (tests)
        return encode_delta_syms32(c, rgb_d, w, h, header, out, cap, len);
    host_trace().mark("delta: enter");
    // 1. linearize + DiffStream + count_freqs (hilbertc.rs:408-410, huf.rs:30): the symbols as a 16-bit stream (k_delta.hip)
    uint32_t *table = nullptr;
    uint8_t *pages = nullptr;
    CNIIC_TRY(delta_table(c, &table, &pages));
    DevBuf hot16, coldkeys, chunk_cold, small;
    const uint64_t nchunks = delta_stream_len(n) / 512;
    CNIIC_HIP_TRY(c, hot16.alloc(delta_stream_len(n) * 2));
    CNIIC_HIP_TRY(c, coldkeys.alloc(nchunks * 64 * 4));  // (written where cold symbols are)
    CNIIC_HIP_TRY(c, chunk_cold.alloc(nchunks));
    CNIIC_HIP_TRY(c, small.alloc(32));  // u64 [0]: a chunk with more than 64 cold symbols, [2]: bits packed
    CNIIC_HIP_TRY(c, hipMemsetAsync(small.p, 0, 32, c->stream));
    CNIIC_TRY(delta_gather_hist(c, rgb_d, w, h, hot16.as<uint16_t>(), table, pages, coldkeys.as<uint32_t>(), chunk_cold.as<uint8_t>(),
                                small.as<uint32_t>()));
    CNIIC_HIP_TRY(c, ctx_pinned_u(c));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&c->pinned_u[2], small.p, 8, hipMemcpyDeviceToHost, c->stream));
    host_trace().mark("delta: gather + hist enqueued");
    // (stage timers, bench.py --config c5: everything between the histogram and the pack -- compaction, the leaves' sort, the host's merge with
    // the GPU idle, the codes -- as ONE stage, so that the stages account for the whole call)
    ScopedKernelTimer timer_tree(c, "delta_tree");
    CompactPlan plan;
    CNIIC_TRY(hist_compact_count(c, table, 27, &plan, nullptr, pages));  // (waits for the stream)
    host_trace().mark("delta: ... + count of the distinct (wait)");
    const uint64_t U = plan.n_unique;
    const bool overflow = c->pinned_u[2] != 0;
    if (host_trace().on) fprintf(stderr, "[host] delta: %llu symbols, %llu distinct%s\n", (unsigned long long)n, (unsigned long long)U,
                                 overflow ? " (a chunk with more than 64 symbols outside the cube: the 32-bit route)" : "");
    if (U >= (1ull << 26) || overflow) {
        CNIIC_TRY(delta_table_clean(c));
        return encode_delta_syms32(c, rgb_d, w, h, header, out, cap, len);
    }
    DevBuf keys_d, counts_d;
    CNIIC_HIP_TRY(c, keys_d.alloc(U * 4));
    CNIIC_HIP_TRY(c, counts_d.alloc(U * 8));
    CNIIC_TRY(hist_compact_write(c, table, &plan, keys_d.as<uint32_t>(), counts_d.as<uint64_t>(), nullptr));
    // Counts (and codes) cross the bus through pinned memory.  From 32768 distinct symbols on -- any photograph -- the host only
    // merges the tree; codes, lengths and the serialised decoder come from the GPU (huff_tree_codes: 0.1 ms of host work less,
    // and no 0.13-0.28 ms of writing the decoder out beside the pack).  Below: [counts u64 | code u64 | len u8], the keys and
    // the decoder in ordinary memory (the host reads the keys at random and writes the decoder byte by byte: 0.20 ms in pinned
    // memory, 0.12 there).
    const uint64_t decoder_bytes = huff_tree_bytes(CNIIC_SYM_SIGNED, U), header_bytes = header.size() + decoder_bytes;
    const bool gpu_codes = U >= c->opt(CNIIC_OPT_HUF_GPU_CODES_MIN, "CNIIC_HUF_GPU_CODES_MIN", 32768) && U >= 2;
    const uint64_t off_code = U * 8, off_len = off_code + U * 8;
    CNIIC_HIP_TRY(c, ctx_pinned_huf(c, gpu_codes ? U * 8 + 3 * (U - 1) * 4 + 64 : off_len + U));
    uint8_t *const pin = static_cast<uint8_t *>(c->pinned_huf);
    uint64_t *const counts = reinterpret_cast<uint64_t *>(pin);
    std::vector<uint32_t> keys_v(gpu_codes ? 0 : U);
    DevBuf sort_a, sort_b, len_d, code_d, tree_d, off_d;
    uint64_t nbits = 0;
    bool tree_built = false;
    if (gpu_codes) {  // the leaves come back sorted by (count, key): the host only merges
        uint64_t *sorted_d = nullptr;
        CNIIC_HIP_TRY(c, sort_a.alloc(U * 8));
        CNIIC_HIP_TRY(c, sort_b.alloc(U * 8));
        CNIIC_TRY(huff_sort_leaves_dev(c, counts_d.as<uint64_t>(), (uint32_t)U, plan.max_count ? plan.max_count : n, sort_a.as<uint64_t>(), sort_b.as<uint64_t>(), &sorted_d));
        // (round 3) the tree, the codes and the leaves' places in the decoder without the host's merge, when the counts come in runs
        CNIIC_HIP_TRY(c, len_d.alloc(U));
        CNIIC_HIP_TRY(c, code_d.alloc(U * 8));
        CNIIC_HIP_TRY(c, off_d.alloc(U * 8));
        CNIIC_TRY(huff_tree_from_runs(c, sorted_d, counts_d.as<uint64_t>(), (uint32_t)U, CNIIC_SYM_SIGNED, len_d.as<uint8_t>(), code_d.as<uint64_t>(),
                                      off_d.as<uint64_t>(), &nbits, &tree_built));
        if (!tree_built) CNIIC_HIP_TRY(c, hipMemcpyAsync(counts, sorted_d, U * 8, hipMemcpyDeviceToHost, c->stream));
    } else {
        CNIIC_HIP_TRY(c, hipMemcpyAsync(keys_v.data(), keys_d.p, U * 4, hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(counts, counts_d.p, U * 8, hipMemcpyDeviceToHost, c->stream));
    }
    CNIIC_HIP_TRY(c, ctx_spin_sync(c));
    host_trace().mark("delta: compaction + D2H (wait)");
    // 2. build() (huf.rs:31); 3. the payload (huf.rs:37-41) behind the serialised decoder (huf.rs:34), whose size follows
    //    from U alone: U leaves of 1 + 6 bytes and U - 1 branch tags
    if (!c->huf_scratch) c->huf_scratch = std::make_shared<HuffScratch>();
    HuffScratch *hscratch = static_cast<HuffScratch *>(c->huf_scratch.get());
    if (!len_d.p) CNIIC_HIP_TRY(c, len_d.alloc(U));
    if (!code_d.p) CNIIC_HIP_TRY(c, code_d.alloc(U * 8));
    HuffTree tree;
    DeltaPackScratch scratch;
    bool counted = false;   // the first half of the pack is already in the stream
    if (gpu_codes && !tree_built) {
        uint32_t *left_h = reinterpret_cast<uint32_t *>(counts + U), *right_h = left_h + (U - 1), *nl_h = right_h + (U - 1), root = 0;
        if (!huff_merge_sorted_into(counts /* sorted leaves */, U, left_h, right_h, nl_h, &root, hscratch))
            return c->fail(CNIIC_ERR_BAD_ARG, "huffman: cannot build code (alphabet %llu)", (unsigned long long)U);
        host_trace().mark("delta: tree (host)");
        CNIIC_HIP_TRY(c, tree_d.alloc(3 * (U - 1) * 4));
        if (!off_d.p) CNIIC_HIP_TRY(c, off_d.alloc(U * 8));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(tree_d.p, left_h, 3 * (U - 1) * 4, hipMemcpyHostToDevice, c->stream));
        const uint32_t *left_d = tree_d.as<uint32_t>(), *right_d = left_d + (U - 1), *nl_d = right_d + (U - 1);
        CNIIC_TRY(huff_tree_codes(c, left_d, right_d, nl_d, counts_d.as<uint64_t>(), (uint32_t)U, root, CNIIC_SYM_SIGNED, len_d.as<uint8_t>(),
                                  code_d.as<uint64_t>(), off_d.as<uint64_t>(), small.as<uint64_t>() + 2));  // small[2] bits, [3] too long
        CNIIC_HIP_TRY(c, hipMemcpyAsync(&c->pinned_u[4], small.as<uint64_t>() + 2, 16, hipMemcpyDeviceToHost, c->stream));
        // (round 4) the payload's size is on its way to the host: the first half of the pack, which wants the codes and nothing else, is
        // enqueued behind it, and the host waits for the size while the GPU counts
        if (!c->res_ev) CNIIC_HIP_TRY(c, hipEventCreateWithFlags(&c->res_ev, hipEventDisableTiming));
        CNIIC_HIP_TRY(c, hipEventRecord(c->res_ev, c->stream));
        if (!c->timers) {   // (with the stage timers on the whole pack is timed as one stage below)
            CNIIC_HIP_TRY(c, hipMemsetAsync(small.as<uint64_t>() + 2, 0, 8, c->stream));
            CNIIC_TRY(delta_pack16_count(c, hot16.as<uint16_t>(), n, coldkeys.as<uint32_t>(), chunk_cold.as<uint8_t>(), table, keys_d.as<uint32_t>(),
                                         len_d.as<uint8_t>(), code_d.as<uint64_t>(), U, small.as<uint64_t>() + 2, &scratch));
            counted = true;
        }
        {
            hipError_t e;
            while ((e = hipEventQuery(c->res_ev)) == hipErrorNotReady) {}
            CNIIC_HIP_TRY(c, e);
        }
        if (c->pinned_u[5]) return c->fail(CNIIC_ERR_BAD_ARG, "huffman: cannot build code (alphabet %llu)", (unsigned long long)U);
        nbits = c->pinned_u[4];
        host_trace().mark("delta: codes (GPU)");
    } else if (!gpu_codes) {
        uint64_t *const code = reinterpret_cast<uint64_t *>(pin + off_code);
        uint8_t *const clen = pin + off_len;
        if (!huff_build_tree(counts, U, tree, hscratch) || !huff_codes_into(tree, clen, code))
            return c->fail(CNIIC_ERR_BAD_ARG, "huffman: cannot build code (alphabet %llu)", (unsigned long long)U);
        host_trace().mark("delta: tree + codes (host)");
        for (uint64_t i = 0; i < U; i++) nbits += counts[i] * clen[i];
        CNIIC_HIP_TRY(c, hipMemcpyAsync(len_d.p, clen, U, hipMemcpyHostToDevice, c->stream));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(code_d.p, code, U * 8, hipMemcpyHostToDevice, c->stream));
    }
    StreamOut so(c, out, cap, len);
    CNIIC_TRY(so.begin_sized(header_bytes, (nbits + 7) / 8, /*zero=*/false));  // (the pack stores every word of the payload)
    if (!counted) CNIIC_HIP_TRY(c, hipMemsetAsync(small.as<uint64_t>() + 2, 0, 8, c->stream));
    timer_tree.stop(1);
    if (nbits) {  // (a single symbol: the zero-length code and no payload, huf.rs:140-142)
        ScopedKernelTimer timer(c, "huff_pack");   // (with the count already enqueued this times the second half alone; bench.py's stage figure says so)
        if (!counted)
            CNIIC_TRY(delta_pack16_count(c, hot16.as<uint16_t>(), n, coldkeys.as<uint32_t>(), chunk_cold.as<uint8_t>(), table, keys_d.as<uint32_t>(),
                                         len_d.as<uint8_t>(), code_d.as<uint64_t>(), U, small.as<uint64_t>() + 2, &scratch));
        CNIIC_TRY(delta_pack16_write(c, hot16.as<uint16_t>(), n, coldkeys.as<uint32_t>(), len_d.as<uint8_t>(), code_d.as<uint64_t>(), so.dev, header_bytes * 8,
                                     &scratch));
        timer.stop(1);
    }
    ScopedKernelTimer timer_fin(c, "delta_finish");   // (the table's sweep, the header, the decoder, the last wait)
    CNIIC_HIP_TRY(c, hipMemcpyAsync(&c->pinned_u[3], small.as<uint64_t>() + 2, 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_TRY(delta_table_clean(c));
    host_trace().mark("delta: pack enqueued");
    // the decoder goes in AFTER the pack, whose first word comes out with zeros where the decoder's last bytes are
    if (gpu_codes) {
        const uint64_t head = header.size();
        CNIIC_TRY(so.put_header(header));
        CNIIC_TRY(huff_tree_serialize_dev(c, keys_d.as<uint32_t>(), off_d.as<uint64_t>(), (uint32_t)U, CNIIC_SYM_SIGNED, so.dev + head, decoder_bytes));
    } else {
        huff_serialize_tree(tree, CNIIC_SYM_SIGNED, keys_v.data(), header);  // (the GPU packs meanwhile)
        host_trace().mark("delta: serialise trie (host)");
        if (header.size() != header_bytes) return c->fail(CNIIC_ERR_HIP, "delta: decoder of %llu bytes, expected %llu", (unsigned long long)header.size(),
                                                          (unsigned long long)header_bytes);
        CNIIC_TRY(so.put_header(header));
    }
    host_trace().mark("delta: header");
    const int rc_fin = so.finish();  // (waits for the stream)
    timer_fin.stop(1);
    host_trace().mark("delta: pack + finish");
    host_trace().dump();
    if (rc_fin != CNIIC_OK) return rc_fin;
    if (c->pinned_u[3] != nbits)
        return c->fail(CNIIC_ERR_HIP, "huffman: packed %llu bits, histogram predicts %llu", (unsigned long long)c->pinned_u[3], (unsigned long long)nbits);
    return CNIIC_OK;
}

// ------------------------------------------------------------------ Hilbert{RLE(0.0)}::encode (hilbertc.rs:26-39)
static int encode_hilbert_rle(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint8_t *out, uint64_t cap, uint64_t *len) {
    const uint64_t n = (uint64_t)w * h;
    std::vector<uint8_t> header;
    put_u32(header, w);  // img.dimensions().serialize (:27)
    put_u32(header, h);
    StreamOut so(c, out, cap, len);
    if (n == 0) {
        CNIIC_TRY(so.begin(header, 0));
        return so.finish();
    }
    DevBuf lin;
    CNIIC_HIP_TRY(c, lin.alloc(n * 3));
    CNIIC_TRY(hilbert_linearize(c, rgb_d, w, h, lin.as<uint8_t>()));  // hilbert::linearize (:29)
    RlePlan plan;
    CNIIC_TRY(rle_plan(c, lin.as<uint8_t>(), n, &plan));             // rle_exact (:34)
    CNIIC_TRY(so.begin_sized(header.size(), plan.nruns * 12, /*zero=*/false));  // count.serialize + color.serialize per run (:35-36):
    CNIIC_TRY(so.put_header(header));                                            // three whole words each, nothing left to clear
    CNIIC_TRY(rle_emit(c, lin.as<uint8_t>(), &plan, reinterpret_cast<uint32_t *>(so.dev + 8)));
    return so.finish();
}

int codec_encode(Ctx *c, const CodecDesc &d, const uint8_t *rgb_d, uint32_t w, uint32_t h, const cniic_kmeans_opts *opts,
                 uint8_t *out, uint64_t cap, uint64_t *len, cniic_kmeans_stats *stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    if ((uint64_t)w * h >= (1ull << 32)) return c->fail(CNIIC_ERR_BAD_ARG, "image too large");
    struct TimersScope {  // stage timers for this call when the options ask for profiling
        Ctx *c; bool saved;
        TimersScope(Ctx *ctx, bool on) : c(ctx), saved(ctx->timers) { c->timers = saved || on; }
        ~TimersScope() { c->timers = saved; }
    } timers_scope(c, opts && (opts->flags & CNIIC_KM_PROFILE));
    switch (d.kind) {
    case CODEC_HUFMAN: return encode_hufman(c, rgb_d, w, h, out, cap, len);
    case CODEC_CLUSTER_COLORS: return encode_cluster_colors(c, rgb_d, w, h, d.arg, opts, out, cap, len, stats);
    case CODEC_VORONOI: return encode_voronoi(c, rgb_d, w, h, d.arg, opts, out, cap, len, stats);
    case CODEC_DELTA: return encode_delta(c, rgb_d, w, h, out, cap, len);
    case CODEC_HILBERT_RLE: return encode_hilbert_rle(c, rgb_d, w, h, out, cap, len);
    }
    return c->fail(CNIIC_ERR_BAD_ARG, "unknown codec");
}

// ------------------------------------------------------------------ decode
// below this many symbols the parallel decoder's fixed cost is not worth it (CNIIC_GPU_DECODE_MIN overrides: tests)
static uint64_t gpu_decode_min_symbols(const Ctx *c) { return c->opt(CNIIC_OPT_GPU_DECODE_MIN, "CNIIC_GPU_DECODE_MIN", 1ull << 14); }

static int put_image(Ctx *c, const uint8_t *src, bool src_dev, uint64_t bytes, uint8_t *dst) {
    if (!bytes) return CNIIC_OK;
    const bool dst_dev = is_device_ptr(dst);
    hipMemcpyKind kind = src_dev ? (dst_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost)
                                 : (dst_dev ? hipMemcpyHostToDevice : hipMemcpyHostToHost);
    CNIIC_HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, kind, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

// The stream may live in host memory or in HBM (bytes_dev).  Only the HEAD of a device-resident stream comes to the host -- the
// dimensions and, for the Huffman codecs, the serialised decoder, which is parsed there (O(alphabet)); the payload is decoded
// where it lies.  head_h / head_n: the part of the stream the host can read (all of it for a host stream).
struct StreamHead {
    const uint8_t *p = nullptr;
    uint64_t n = 0;
};
// the first `want` bytes of the stream where the host can read them: the stream itself, or (device-resident) a copy in the
// context's pinned block (a pageable landing buffer costs the copy a staging pass and ~40 us)
static int stream_head(Ctx *c, const uint8_t *bytes, bool bytes_dev, uint64_t nbytes, uint64_t want, StreamHead *h) {
    want = std::min(want, nbytes);
    // (a host stream is all there, but the decoder is looked for in its head only, like a device stream's: one that does not end
    // there has hundreds of thousands of leaves and is parsed faster on the GPU than by this core)
    if (!bytes_dev) { h->p = bytes; h->n = std::max(h->n, want); return CNIIC_OK; }
    if (h->n >= want) return CNIIC_OK;
    CNIIC_HIP_TRY(c, ctx_pinned_huf(c, want));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(c->pinned_huf, bytes, want, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    h->p = static_cast<const uint8_t *>(c->pinned_huf);
    h->n = want;
    return CNIIC_OK;
}

constexpr uint64_t kTrieSecondLook = 512ull << 10;   // bytes of a `delta` stream the host parses before the GPU is asked (~7 10^4 leaves)
int codec_decode(Ctx *c, const CodecDesc &d, const uint8_t *bytes, uint64_t nbytes, uint8_t *rgb_out, uint64_t cap,
                 uint32_t *w, uint32_t *h) {
    const bool bytes_dev = is_device_ptr(bytes);
    StreamHead head;
    // (the serialised decoder of a Huffman stream is at most a few per cent of it, for the images these codecs are meant for)
    CNIIC_TRY(stream_head(c, bytes, bytes_dev, nbytes, std::min<uint64_t>(std::max<uint64_t>(nbytes / 48, 8192), 4ull << 20), &head));
    uint64_t pos = 0;
    if (!get_u32(head.p, head.n, pos, *w) || !get_u32(head.p, head.n, pos, *h))  // create_image_buffer_standard codec.rs:22-26
        return c->fail(CNIIC_ERR_DECODE, "decode: truncated dimensions");
    const uint64_t n = (uint64_t)*w * *h;
    if (n >= (1ull << 32)) return c->fail(CNIIC_ERR_DECODE, "decode: image too large");
    if (n * 3 > cap) return c->fail(CNIIC_ERR_CAPACITY, "decode: image needs %llu bytes, capacity %llu",
                                    (unsigned long long)(n * 3), (unsigned long long)cap);
    switch (d.kind) {
    case CODEC_HUFMAN:
    case CODEC_CLUSTER_COLORS:   // clusterc.rs:55-57 delegates to Hufman.decode (hufc.rs:19-40)
    case CODEC_DELTA: {          // hilbertc.rs:417-431
        const bool delta = d.kind == CODEC_DELTA;
        const int sym_kind = delta ? CNIIC_SYM_SIGNED : CNIIC_SYM_RGB;
        const char *bad_stream = delta ? "delta: cannot decode the difference stream" : "Failed to decode symbol";
        host_trace().mark("decode: enter");
        if (!c->trie_scratch) c->trie_scratch = std::make_shared<LeafTable>();
        LeafTable &lt = *static_cast<LeafTable *>(c->trie_scratch.get());
        // the decoder: parsed from what the host holds of the stream; a device-resident stream whose decoder is longer than
        // that is fetched further (a failed parse of a TRUNCATED head says nothing: only the whole stream's verdict counts)
        uint64_t tpos = pos;
        bool parsed = huff_parse_leaves(sym_kind, head.p, head.n, tpos, lt);
        const bool dst_dev = is_device_ptr(rgb_out);
        int status = 2;
        DevBuf keys_d, lin_d, img_d, stream_up, tab_big;
        UdSums ud_sums;   // (delta: FromDiff's chunk sums, when the decoder added them up on its way)
        uint8_t *dst = rgb_out;
        auto need_image = [&]() -> int {   // where the pixels are produced: the caller's image if it is in HBM and word-aligned
            if (!img_d.p && (!dst_dev || (reinterpret_cast<uintptr_t>(rgb_out) & 3))) { CNIIC_HIP_TRY(c, img_d.alloc(n * 3)); dst = img_d.as<uint8_t>(); }
            return CNIIC_OK;
        };
        const bool force_gpu_parse = test_env("CNIIC_TEST_TRIE_GPU") != nullptr;   // tests: every decoder through k_trieparse.hip
        // A `delta` decoder of a photograph has 4-6 10^4 leaves (0.3-0.5 MB): more than the head that was looked at, far fewer than the
        // GPU parse needs to pay for its launches and waits (0.49 ms at 4 10^4 leaves; this core parses them in 0.15) -- a second, longer
        // look before the stream goes to k_trieparse.hip.  (`hufman` decoders that outgrow the first look are ten times that size.)
        if (!parsed && !force_gpu_parse && head.n < nbytes) {
            const char *e2 = test_env("CNIIC_TRIE_HOST_SECOND");
            // (whatever the stream's length: the decoder's share of it grows as the image shrinks -- 57 % at 512^2 -- and a photograph's
            // alphabet stays under 6 10^4 differences at any size.  Uniform noise, whose decoder is most of the stream, paid for the look in
            // vain -- 0.33 ms; see `worth`.)
            // A stream of more than 6 bytes a pixel is mostly decoder (the payload has at most 27 bits a symbol): more leaves than the look
            // could hold, unless the whole stream fits into it.
            const bool worth = nbytes <= kTrieSecondLook || nbytes <= 6 * n;
            const uint64_t second = e2 ? strtoull(e2, nullptr, 10) : delta && worth ? std::min<uint64_t>(kTrieSecondLook, nbytes) : 0;
            if (second > head.n) {
                CNIIC_TRY(stream_head(c, bytes, bytes_dev, nbytes, second, &head));
                tpos = pos;
                parsed = huff_parse_leaves(sym_kind, head.p, head.n, tpos, lt);
                host_trace().mark("decode: second look at the decoder (host)");
            }
        }
        if (force_gpu_parse || (!parsed && head.n < nbytes)) {
            // The decoder is longer than the head of the stream that was looked at (4 MiB at most): an alphabet of hundreds of
            // thousands of symbols -- `hufman` on a photograph.  Parsed on the GPU (k_trieparse.hip), from the stream in HBM.
            bool done_dev = false;
            if (n >= gpu_decode_min_symbols(c)) {
                const uint8_t *sd = bytes;
                if (!bytes_dev) {
                    CNIIC_HIP_TRY(c, stream_up.alloc(nbytes + 16));
                    CNIIC_HIP_TRY(c, hipMemcpyAsync(stream_up.p, bytes, nbytes, hipMemcpyHostToDevice, c->stream));
                    sd = stream_up.as<uint8_t>();
                }
                uint64_t nl = 0, off_key = 0, off_len = 0, ppos = 0;
                uint32_t max_len = 0;
                int pst = 1;
                CNIIC_TRY(huff_parse_leaves_dev(c, sym_kind, sd, nbytes, pos, &tab_big, &nl, &off_key, &off_len, &max_len, &ppos, &pst));
                host_trace().mark("decode: parse the decoder (GPU)");
                if (pst == 1) return c->fail(CNIIC_ERR_DECODE, bad_stream);
                if (pst == 0) {
                    if (!n) return CNIIC_OK;
                    CNIIC_TRY(need_image());
                    if (delta) CNIIC_HIP_TRY(c, keys_d.alloc(n * 4));
                    uint32_t key0 = 0;   // (a one-symbol alphabet is filled in, not decoded: its key is wanted here)
                    if (nl == 1) {
                        CNIIC_HIP_TRY(c, hipMemcpyAsync(&key0, tab_big.as<uint8_t>() + off_key, 4, hipMemcpyDeviceToHost, c->stream));
                        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
                    }
                    CNIIC_TRY(huff_decode_tables_dev(c, tab_big.as<uint8_t>(), nl, off_key, off_len, max_len, key0, sd + ppos, true, nbytes - ppos, n,
                                                     delta ? 0 : 1, delta ? keys_d.p : (void *)dst, &status, delta ? &ud_sums : nullptr));
                    done_dev = status != 2;
                }
            }
            if (!done_dev) {  // the host's way: the whole stream there
                if (bytes_dev) head.n = 0;
                CNIIC_TRY(stream_head(c, bytes, bytes_dev, nbytes, nbytes, &head));
                tpos = pos;
                parsed = huff_parse_leaves(sym_kind, head.p, head.n, tpos, lt);
                if (!parsed) return c->fail(CNIIC_ERR_DECODE, bad_stream);
                lt.too_deep = true;   // (whatever was tried on the GPU did not work out: the node walk below)
            }
        } else {
            if (!parsed) return c->fail(CNIIC_ERR_DECODE, bad_stream);
            host_trace().mark("decode: parse the decoder (host)");
            if (!lt.too_deep) {
                if (!n) return CNIIC_OK;
                CNIIC_TRY(need_image());
                if (delta) CNIIC_HIP_TRY(c, keys_d.alloc(n * 4));
                // symbols on the GPU (parallel, self-synchronising), straight from the stream where it lies; small inputs and codes
                // that do not settle go through the host walk (same answers)
                if (n >= gpu_decode_min_symbols(c))
                    CNIIC_TRY(huff_decode_dev(c, lt, bytes + tpos, bytes_dev, nbytes - tpos, n, delta ? 0 : 1, delta ? keys_d.p : (void *)dst, &status, delta ? &ud_sums : nullptr));
            }
        }
        if (status == 2) {  // the node walk on the host: needs the whole stream there
            ud_sums.filled = false;
            if (bytes_dev) head.n = 0;  // (the pinned block the head was fetched into has served other purposes since: fetch again)
            CNIIC_TRY(stream_head(c, bytes, bytes_dev, nbytes, nbytes, &head));
            std::vector<TrieNode> trie;
            if (!huff_parse_trie(sym_kind, head.p, head.n, pos, trie)) return c->fail(CNIIC_ERR_DECODE, bad_stream);
            if (!n) return CNIIC_OK;
            CNIIC_TRY(need_image());
            CNIIC_HIP_TRY(c, keys_d.alloc(n * 4));
            std::vector<uint32_t> keys(n);
            if (!huff_decode_host(trie, head.p + pos, head.n - pos, n, keys.data(), nullptr)) status = 1;
            else {
                status = 0;
                CNIIC_HIP_TRY(c, hipMemcpyAsync(keys_d.p, keys.data(), n * 4, hipMemcpyHostToDevice, c->stream));
                CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (!delta) CNIIC_TRY(keys_to_rgb(c, keys_d.as<uint32_t>(), n, dst));
            }
        }
        if (status == 1) return c->fail(CNIIC_ERR_DECODE, bad_stream);
        host_trace().mark("decode: symbols");
        if (delta) {
            uint32_t bad = 0;  // START (hilbertc.rs:445); FromDiff (hilbertc.rs:496-508); then follow the traversal (hilbertc.rs:426-428)
            ScopedKernelTimer tu(c, "undiff_scatter");
            CNIIC_TRY(delta_undiff_scatter_dev(c, keys_d.as<uint32_t>(), *w, *h, dst, &bad, &ud_sums));
            tu.stop();
            if (bad) return c->fail(CNIIC_ERR_DECODE, "delta: colour out of range (hilbertc.rs:505)");
        }
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        host_trace().mark("decode: pixels");
        const int rc_put = dst == rgb_out ? CNIIC_OK : put_image(c, dst, true, n * 3, rgb_out);
        host_trace().mark("decode: image out");
        host_trace().dump();
        return rc_put;
    }
    case CODEC_HILBERT_RLE: {  // hilbertc.rs:53-79: RleDecoder (:304-337) zipped with hilbert::iter
        if (!n) return CNIIC_OK;
        // pixels the stream does not reach stay zero (ImageBuffer::new); RepCount::deserialize(..)? ends it quietly
        const uint64_t body = nbytes - pos, R = body / 12, tail = body % 12;
        DevBuf rec_d, lin_d, img_d;
        const uint8_t *recs = bytes + pos;   // (a stream in HBM whose records are 4-byte aligned is read where it lies: 3.2 GB at 16384^2)
        if (!bytes_dev || (reinterpret_cast<uintptr_t>(recs) & 3)) {
            CNIIC_HIP_TRY(c, rec_d.alloc(std::max<uint64_t>(R * 12, 16)));
            if (R) CNIIC_HIP_TRY(c, hipMemcpyAsync(rec_d.p, bytes + pos, R * 12, bytes_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
            recs = rec_d.as<uint8_t>();
        }
        CNIIC_HIP_TRY(c, lin_d.alloc(n * 3));
        int status = 0;
        CNIIC_TRY(rle_expand_dev(c, recs, R, tail, n, lin_d.as<uint8_t>(), &status));
        if (status)
            return c->fail(CNIIC_ERR_DECODE, "hilbert-rle: bad run record (assert!(count > 0) / unwrap, hilbertc.rs:327-328)");
        uint8_t *dst = rgb_out;
        const bool dst_dev = is_device_ptr(rgb_out);
        if (!dst_dev) { CNIIC_HIP_TRY(c, img_d.alloc(n * 3)); dst = img_d.as<uint8_t>(); }
        CNIIC_TRY(hilbert_scatter(c, lin_d.as<uint8_t>(), *w, *h, dst));  // follow the traversal (:58-61)
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!dst_dev) return put_image(c, dst, true, n * 3, rgb_out);
        return CNIIC_OK;
    }
    case CODEC_VORONOI: {  // clusterc.rs:168-189
        CNIIC_TRY(stream_head(c, bytes, bytes_dev, nbytes, nbytes, &head));  // 16 + 19 K bytes: parsed on the host
        const uint8_t *hb = head.p;
        uint64_t K;
        if (!get_u64(hb, nbytes, pos, K)) return c->fail(CNIIC_ERR_DECODE, "voronoi: truncated");
        if (K > (nbytes - pos) / 19) return c->fail(CNIIC_ERR_DECODE, "voronoi: truncated centroid list");
        std::vector<cniic_colorpos> cent(K);
        for (uint64_t k = 0; k < K; k++) {
            uint64_t l;
            if (!get_u32(hb, nbytes, pos, cent[k].x) || !get_u32(hb, nbytes, pos, cent[k].y) ||
                !get_u64(hb, nbytes, pos, l) || l != 3 || pos + 3 > nbytes)
                return c->fail(CNIIC_ERR_DECODE, "voronoi: bad centroid");
            memcpy(cent[k].rgb, hb + pos, 3);
            cent[k].pad = 0;
            pos += 3;
        }
        if (!n) return CNIIC_OK;
        if (K == 0) return c->fail(CNIIC_ERR_DECODE, "voronoi: no centroids (min_by_key().unwrap(), clusterc.rs:184)");
        if (K > 0xffffffffull) return c->fail(CNIIC_ERR_DECODE, "voronoi: too many centroids");
        DevBuf cent_d, img_d;
        CNIIC_HIP_TRY(c, cent_d.alloc(K * sizeof(cniic_colorpos)));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(cent_d.p, cent.data(), K * sizeof(cniic_colorpos), hipMemcpyHostToDevice, c->stream));
        uint8_t *dst = rgb_out;
        const bool dst_dev = is_device_ptr(rgb_out);
        if (!dst_dev) { CNIIC_HIP_TRY(c, img_d.alloc(n * 3)); dst = img_d.as<uint8_t>(); }
        bool small_coords = *w <= (1u << 14) && *h <= (1u << 14);  // then the pruned repaint is exact (k_misc.hip)
        for (uint64_t k = 0; k < K && small_coords; k++) small_coords = cent[k].x < (1u << 14) && cent[k].y < (1u << 14);
        CNIIC_TRY(voronoi_paint(c, cent_d.as<cniic_colorpos>(), (uint32_t)K, *w, *h, dst, small_coords));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!dst_dev) return put_image(c, dst, true, n * 3, rgb_out);
        return CNIIC_OK;
    }
    }
    return c->fail(CNIIC_ERR_BAD_ARG, "unknown codec");
}

}  // namespace cniic
