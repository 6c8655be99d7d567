// codec.hpp -- the reference's Codec surface (src/codec.rs:14-19) for the hot-path codecs.
#pragma once
#include <string>
#include <vector>

#include "common.hpp"

namespace cniic {

enum CodecKind { CODEC_HUFMAN = 1, CODEC_CLUSTER_COLORS = 2, CODEC_VORONOI = 3, CODEC_DELTA = 4, CODEC_HILBERT_RLE = 5 };

struct CodecDesc {
    int      kind;
    uint32_t arg;  // K for cluster-colors / voronoi
};

bool        parse_codec(const char *expr, CodecDesc *out);  // AnyCodec::from_str (codec.rs:41-59)
std::string codec_name(const CodecDesc &d);                 // Codec::name
bool        codec_is_lossless(const CodecDesc &d);          // Codec::is_lossless

// rgb_d is device memory; out / rgb_out may be host or device.
int codec_encode(Ctx *c, const CodecDesc &d, const uint8_t *rgb_d, uint32_t w, uint32_t h, const cniic_kmeans_opts *opts,
                 uint8_t *out, uint64_t cap, uint64_t *len, cniic_kmeans_stats *stats);
// bytes is HOST memory.
int codec_decode(Ctx *c, const CodecDesc &d, const uint8_t *bytes, uint64_t nbytes, uint8_t *rgb_out, uint64_t cap,
                 uint32_t *w, uint32_t *h);

// cluster-colors in pieces (see codec.cpp)
struct CcSession {
    Ctx *c = nullptr;
    uint32_t K = 0;
    uint32_t *table = nullptr;   // dense colour table: counts on entry of cc_prepare, key -> rank + 1 afterwards
    uint64_t U = 0;
    DevBuf keys_d, weight_d;
    DevBuf gbits, gprefix, gtotal;  // shared palette over several images: index of the colours that occur in ANY of them (+ their number, on the device)
    bool local_points = false;   // ... and the points of this session are this image's colours only
    SpPlan sp;                   // large images: the pixels partitioned by colour super-cell (k_points.hip) instead of the dense table
    bool sp_mode = false;
    KmRgbwState *km = nullptr;
    ~CcSession();
};
// occ_d (optional): summed occupancy nibbles of all ranks (occupancy_pack); the table then holds THIS image's counts
int cc_prepare(Ctx *c, uint32_t *table_counts_d, uint32_t K, const cniic_kmeans_opts *opts, uint32_t shard, uint32_t nshards,
               void *partials_dev, CcSession **out, const uint32_t *occ_d = nullptr);
// the same from the image itself, through the super-cell partition (no dense table; 16-byte aligned rgb_d)
int cc_prepare_image(Ctx *c, const uint8_t *rgb_d, uint64_t npx, uint32_t K, const cniic_kmeans_opts *opts, CcSession **out);
// shared palette over several images, through the partition: begin (this image's pixels), the caller sums the occupancy of
// all ranks (sp_occupancy), create (K-means state over this image's colours placed in the list of all colours)
int cc_image_begin(Ctx *c, const uint8_t *rgb_d, uint64_t npx, CcSession **out);
int cc_image_create(CcSession *s, const uint32_t *occ_d, uint32_t K, const cniic_kmeans_opts *opts, void *partials_dev);
int cc_finish(CcSession *s, const uint8_t *rgb_d, uint32_t w, uint32_t h, const uint32_t *local_counts_d, uint8_t *out,
              uint64_t cap, uint64_t *len, cniic_kmeans_stats *stats);
// a batch of F frames coded with the session's one palette: F Hufman streams, stream f at out + f * stride, its length in lens[f]
int cc_finish_frames(CcSession *s, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t F, uint8_t *out, uint64_t stride, uint64_t *lens,
                     cniic_kmeans_stats *stats);

// header carries any prefix already serialised (image dimensions); the decoder trie is appended
// to it and the whole stream lands in out[0..*len)  (out: host or device memory).
// syms_scratch: the symbol stream is ours and may be overwritten.
int huf_encode_all_dev(Ctx *c, int sym_kind, const uint8_t *rgb_d, uint32_t *syms_d, bool syms_scratch, uint64_t n, uint32_t *table_d,
                       bool have_hist, std::vector<uint8_t> &header, uint8_t *out, uint64_t cap, uint64_t *len);

}  // namespace cniic
