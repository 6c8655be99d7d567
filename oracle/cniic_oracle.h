/*
 * cniic_oracle.h -- CPU restatement of hkapp/cniic's per-pixel compression hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load liboracle.so.  The product (cniic_amd/libcniic_hip.so)
 * never links, loads or calls anything in this directory.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * checkout, e.g. src/kmeans.rs:330-416).  The reference is Rust; no Rust toolchain exists in the
 * build image, so the reference itself cannot be compiled (oracle/_ref is therefore absent) and
 * the oracle is pinned by the reference's own unit-test vectors (tests/test_oracle_kat.py).
 *
 * Parity status
 *   - bit I/O, serialisation, Huffman, K-means arithmetic, delta rule: pinned by the reference's
 *     known-answer tests (src/bit.rs:261-493, src/huf.rs:376-540, src/kmeans.rs:446-581,
 *     src/codec/clusterc.rs:299-338, README.md:150-175).
 *   - Hilbert scan (src/hilbert.rs:40-43): PARITY UNPINNED.  The reference delegates to the
 *     un-vendored crate zhang_hilbert 0.1.1 (Cargo.toml:15) whose source is not available and
 *     for which the reference holds no test vector.  This oracle freezes its own scan (the
 *     generalised Hilbert curve, see hilbert.c); only the reference's consistency contract
 *     (a bijective scan shared by encode and decode, hilbertc.rs:410,420) is met.
 *
 * Documented deviations that make the reference's non-deterministic parts deterministic:
 *   D1  HashMap iteration order (utils.rs:4-16) -> ascending packed symbol key.
 *   D2  rand::thread_rng empty-cluster reseed (kmeans.rs:123-133) -> splitmix64(seed, iter, cluster).
 *   D3  sort_unstable_by tie order (kmeans.rs:185) -> stable (ties keep previous list order).
 */
#ifndef CNIIC_ORACLE_H
#define CNIIC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (shared numbering with include/cniic_hip.h) ---- */
#define ORC_OK                  0
#define ORC_ERR_BAD_ARG        -1
#define ORC_ERR_TOO_FEW_POINTS -2  /* kmeans.rs:68  assert!(points_per_cluster > 0) */
#define ORC_ERR_FEW_ACTIVE     -3  /* kmeans.rs:54  "Not enough active clusters" */
#define ORC_ERR_DECODE         -6  /* Option::None from a decode path */
#define ORC_ERR_NOMEM          -7
#define ORC_ERR_CAPACITY       -8  /* caller's output buffer too small */

/* ---- growable byte buffer (stands in for the reference's io::Write sink) ---- */
typedef struct {
    uint8_t *data;
    size_t   len;
    size_t   cap;
} orc_buf;

void orc_buf_init(orc_buf *b);
void orc_buf_free(orc_buf *b);
int  orc_buf_put(orc_buf *b, const void *p, size_t n);

/* ---- src/ser.rs ---- */
int orc_ser_u8(orc_buf *b, uint8_t v);     /* ser.rs:17-21   */
int orc_ser_u16(orc_buf *b, uint16_t v);   /* ser.rs:31-35   */
int orc_ser_i16(orc_buf *b, int16_t v);    /* ser.rs:49-53   */
int orc_ser_u32(orc_buf *b, uint32_t v);   /* ser.rs:67-71   */
int orc_ser_u64(orc_buf *b, uint64_t v);   /* ser.rs:87-91   */
int orc_ser_rgb(orc_buf *b, const uint8_t rgb[3]); /* ser.rs:210-214 -> 164-172: u64 len=3 + 3 bytes */

/* byte-stream reader (stands in for Iterator<Item=u8>) */
typedef struct {
    const uint8_t *p;
    size_t n;
    size_t pos;
} orc_rd;
int orc_de_u8(orc_rd *r, uint8_t *v);
int orc_de_u16(orc_rd *r, uint16_t *v);
int orc_de_i16(orc_rd *r, int16_t *v);
int orc_de_u32(orc_rd *r, uint32_t *v);
int orc_de_u64(orc_rd *r, uint64_t *v);
int orc_de_rgb(orc_rd *r, uint8_t rgb[3]); /* ser.rs:216-222 */

/* ---- src/bit.rs: MSB-first IoBitWriter ---- */
typedef struct {
    orc_buf *out;
    uint8_t  curr_bits;
    uint8_t  bit_count;
} orc_bitw;
void orc_bitw_init(orc_bitw *w, orc_buf *out);        /* bit.rs:193-201 */
int  orc_bitw_bit(orc_bitw *w, int bit);              /* bit.rs:210-220 */
int  orc_bitw_byte(orc_bitw *w, uint8_t n);           /* bit.rs:222-240 */
int  orc_bitw_code(orc_bitw *w, const uint8_t *full_bytes, size_t nfull,
                   uint8_t partial_byte, uint8_t partial_count); /* bit.rs:164-178 */
int  orc_bitw_pad_and_flush(orc_bitw *w);             /* bit.rs:243-253 */
uint8_t orc_bit_mask(uint8_t nbits);                  /* bit.rs:103-105 */
int  orc_bit_nth(uint8_t byte, uint8_t idx, int msb_first); /* bit.rs:70-86 */

/* ---- symbols: one 32-bit key per symbol + a kind that fixes the wire format ---- */
#define ORC_SYM_CHAR   0  /* u8 / ascii char, 1 byte (ser.rs:129-135); used by the KATs      */
#define ORC_SYM_RGB    1  /* Rgb<u8>, key = r<<16|g<<8|b, 11 bytes (ser.rs:210-214)          */
#define ORC_SYM_SIGNED 2  /* SignedColor([i16;3]) (hilbertc.rs:513-516), key =                */
                          /* (dr+255)<<18|(dg+255)<<9|(db+255), 6 bytes (ser.rs:188-195)      */

/* utils.rs:4-16 count_freqs.  Output sorted by ascending key (deviation D1).
 * keys/counts must hold cap entries; *n_unique receives the number of distinct symbols. */
int orc_count_freqs(const uint32_t *syms, uint64_t n, uint32_t *keys, uint64_t *counts,
                    uint64_t cap, uint64_t *n_unique);

/* huf.rs:58-117 build: code lengths (and codes, MSB-first in the low bits of a u64 when
 * len <= 64) for each of the n (key,count) pairs given in ascending key order.  The heap is
 * Rust's std BinaryHeap (from_iter / pop / push) restated, so the tree shape is what the
 * reference would build from this item order. */
int orc_huf_build(const uint64_t *counts, uint64_t n, uint32_t *lens, uint64_t *codes);

/* huf.rs:22-43 encode_all: serialise decoder trie (huf.rs:299-321) then the bit-packed payload. */
int orc_huf_encode_all(int sym_kind, const uint32_t *syms, uint64_t n, orc_buf *out);
/* huf.rs:46-53 + 366-374: deserialise trie, then pull nsyms symbols (caller knows the count). */
int orc_huf_decode_all(int sym_kind, orc_rd *in, uint32_t *syms, uint64_t nsyms);
/* size-only: bytes encode_all would emit for this histogram (pure function of the counts). */
int orc_huf_size(int sym_kind, const uint64_t *counts, uint64_t n, uint64_t *nbytes);

/* ---- src/kmeans.rs ---- */
#define ORC_PT_TOY2   0  /* (i32,i32) of the reference's test module, kmeans.rs:451-477       */
#define ORC_PT_RGBW   1  /* ColorCount, clusterc.rs:68-114 (D=3 + weight)                      */
#define ORC_PT_XYRGB  2  /* ColorPos,   clusterc.rs:200-248 (D=5: x,y,r,g,b)                   */

#define ORC_KM_MODE_R 0  /* reference-faithful: pruned search over truncated neighbour lists   */
#define ORC_KM_MODE_L 1  /* exact Lloyd with the reference's init / tie / mean rules           */

typedef struct {
    uint64_t iterations;        /* kmeans.rs:33                          */
    uint64_t moved_last;        /* kmeans.rs:401 of the last iteration   */
    uint64_t obvious_stay;      /* kmeans.rs:404 summed over the run     */
    uint64_t neighbour_cutoff;  /* kmeans.rs:405 summed over the run     */
    uint64_t tested_neighbours; /* kmeans.rs:386 summed over the run     */
    uint64_t empty_reseeds;     /* kmeans.rs:117-134 occurrences         */
    uint64_t dist_evals;        /* point-to-centroid distance evaluations */
} orc_km_stats;

int orc_pt_dim(int kind);

/* kmeans.rs:21-39 cluster().  pts is n x D int32 (row-major), weight is n u32 (RGBW only, may be
 * NULL otherwise).  Outputs: centroids K x D int32; labels[n] = final cluster of each input point
 * (input order); members[K] = number of points per cluster; radii[K] (optional) = certainty radius
 * (kmeans.rs:257-259, mode R only). max_iters = 0 means "no cap" (the reference has none). */
int orc_kmeans(int kind, int mode, const int32_t *pts, const uint32_t *weight, uint64_t n,
               uint32_t K, uint64_t seed, uint64_t max_iters,
               int32_t *centroids, uint32_t *labels, uint64_t *members, double *radii,
               orc_km_stats *stats);

/* One exact-Lloyd step from given centroids and labels: labels updated in place,
 * sums[K x D] / wsum[K] (sum of weights, or member count when unweighted) / members[K] / *changed
 * filled.  Centroids are NOT updated (use orc_kmeans_finalize). */
int orc_kmeans_step(int kind, const int32_t *pts, const uint32_t *weight, uint64_t n, uint32_t K,
                    const int32_t *centroids, uint32_t *labels,
                    uint64_t *sums, uint64_t *wsum, uint64_t *members, uint64_t *changed);
/* The same step, threaded over contiguous point ranges and vectorised over the centroids (kmeans_fast.c);
 * falls back to orc_kmeans_step outside its limits.  orc_set_lloyd_threads(t > 1) makes orc_kmeans (mode L, and
 * with it the codecs) use it: only tests/golden/make_fullsize_digests.py and tests/test_oracle_fast.py do. */
int orc_kmeans_step_fast(int kind, const int32_t *pts, const uint32_t *weight, uint64_t n, uint32_t K,
                         const int32_t *centroids, uint32_t *labels,
                         uint64_t *sums, uint64_t *wsum, uint64_t *members, uint64_t *changed, int threads);
int  orc_kmeans_fast_ok(int kind, const int32_t *pts, uint64_t n, uint32_t K, const int32_t *centroids);
void orc_set_lloyd_threads(int t);
int  orc_get_lloyd_threads(void);
/* Point::mean per cluster (clusterc.rs:83-113, 216-247) + empty-cluster reseed (deviation D2). */
int orc_kmeans_finalize(int kind, const int32_t *pts, uint64_t n, uint32_t K, uint64_t seed,
                        uint64_t iter, const uint64_t *sums, const uint64_t *wsum,
                        const uint64_t *members, int32_t *centroids, uint64_t *n_reseeded);
/* kmeans.rs:61-78 init_assignment: label of point index i. */
uint32_t orc_kmeans_init_label(uint64_t i, uint64_t n, uint32_t K);
/* index of the point stolen for empty cluster c at iteration iter (deviation D2). */
uint64_t orc_kmeans_reseed_index(uint64_t seed, uint64_t iter, uint32_t c, uint64_t n);
/* geom.rs:8-24 / clusterc.rs:206-213 / kmeans.rs:451-459 */
double orc_pt_dist(int kind, const int32_t *a, const int32_t *b);

/* ---- src/hilbert.rs (scan frozen by this build: PARITY UNPINNED, see header) ---- */
/* hilbert.rs:40-43 iter(w,h): xy[2*i], xy[2*i+1] for i in 0..w*h */
int orc_hilbert_iter(uint32_t w, uint32_t h, uint32_t *xy);
/* random access: position d of the same scan */
void orc_hilbert_d2xy(uint32_t w, uint32_t h, uint64_t d, uint32_t *x, uint32_t *y);
/* hilbert.rs:10-12 linearize: rgb gathered in scan order */
int orc_hilbert_linearize(const uint8_t *rgb, uint32_t w, uint32_t h, uint8_t *out);
/* hilbertc.rs:449-477 DiffStream over a linear rgb stream; out = packed ORC_SYM_SIGNED keys */
int orc_delta_diff(const uint8_t *rgb_lin, uint64_t n, uint32_t *syms);
/* hilbertc.rs:482-509 FromDiff (fails like the unwrap at 505-506 if a channel leaves 0..255) */
int orc_delta_undiff(const uint32_t *syms, uint64_t n, uint8_t *rgb_lin);

/* ---- codecs (src/codec/ hufc.rs, clusterc.rs, hilbertc.rs) ---- */
int orc_hufman_encode(const uint8_t *rgb, uint32_t w, uint32_t h, orc_buf *out);         /* hufc.rs:12-17  */
int orc_hufman_decode(const uint8_t *bytes, size_t n, uint8_t *rgb, size_t cap,
                      uint32_t *w, uint32_t *h);                                          /* hufc.rs:19-40  */
int orc_delta_encode(const uint8_t *rgb, uint32_t w, uint32_t h, orc_buf *out);          /* hilbertc.rs:405-415 */
int orc_delta_decode(const uint8_t *bytes, size_t n, uint8_t *rgb, size_t cap,
                     uint32_t *w, uint32_t *h);                                           /* hilbertc.rs:417-431 */
int orc_cluster_colors_encode(const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t K,
                              int mode, uint64_t seed, orc_buf *out, orc_km_stats *st);   /* clusterc.rs:18-53 */
int orc_voronoi_encode(const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t K,
                       int mode, uint64_t seed, orc_buf *out, orc_km_stats *st);          /* clusterc.rs:148-166 */
int orc_voronoi_decode(const uint8_t *bytes, size_t n, uint8_t *rgb, size_t cap,
                       uint32_t *w, uint32_t *h);                                         /* clusterc.rs:168-189 */
/* bench.rs:95-104 compute_error (MSE) */
double orc_mse(const uint8_t *a, const uint8_t *b, uint64_t npx);

/* convenience wrappers for ctypes: encode into caller memory */
/* Hilbert{RLE(0.0)}: exact run-length coding along the scan (hilbertc.rs:12-98, 100-196, 304-337) */
int orc_hilbert_rle_encode(const uint8_t *rgb, uint32_t w, uint32_t h, orc_buf *out);
int orc_hilbert_rle_decode(const uint8_t *bytes, size_t nb, uint8_t *rgb, size_t cap, uint32_t *w, uint32_t *h);
int orc_encode(const char *codec, int mode, uint64_t seed, const uint8_t *rgb, uint32_t w,
               uint32_t h, uint8_t *out, uint64_t cap, uint64_t *len, orc_km_stats *st);
int orc_decode(const char *codec, const uint8_t *bytes, uint64_t n, uint8_t *rgb, uint64_t cap,
               uint32_t *w, uint32_t *h);

#ifdef __cplusplus
}
#endif
#endif
