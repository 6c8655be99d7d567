/*
 * cniic_hip.h -- C ABI of the MI355X-native (gfx950) implementation of cniic's per-pixel
 * compression hot path.  This is the drop-in boundary: the entry points are the cut points a
 * Rust `impl Codec` in the reference would bind over FFI (see INTEGRATION.md for the stub).
 * The reference has no FFI of its own (it is a single Rust crate), so each entry point cites the
 * reference function it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - Every function returns int32_t: 0 = OK, negative = error (table below).  Nothing throws or
 *     aborts across the ABI.  cniic_last_error(ctx) returns a message for the last failure.
 *   - The caller owns every buffer.  Unless stated otherwise a data pointer may be HOST memory
 *     or DEVICE (HBM) memory of the context's GPU; the library detects which
 *     (hipPointerGetAttributes) and stages host buffers over PCIe.  Scalar out-params (`uint64_t
 *     *n_unique`, stats structs, ...) are always host memory.
 *   - A cniic_ctx owns one HIP stream plus scratch HBM.  Calls on one ctx are serialised by an
 *     internal mutex; use one ctx per calling thread for concurrency (the reference calls
 *     encode/decode from rayon workers, src/bench.rs:24-28).  No process-global mutable state.
 *   - Functions return after their results are complete (stream synchronised) unless the name
 *     ends in _async.
 *   - STREAM ORDER OF DEVICE BUFFERS.  A context enqueues on ITS stream only (the one given to
 *     cniic_ctx_create, or its own).  A DEVICE buffer handed to any call must be complete with
 *     respect to that stream when the call is made: either the caller produced it on the same
 *     stream, or the caller has synchronised the producing stream (hipStreamSynchronize /
 *     hipDeviceSynchronize / an event the context's stream waits on) first.  The library does
 *     NOT wait for other streams: an image or byte stream still being filled by another stream
 *     is read half-written (round 3: a `delta` decode answered "colour out of range" once in
 *     40 000 fuzz cases because the test filled the stream's buffer on torch's stream and
 *     decoded on a private one; tests/test_stream_order.py replays that input).  Likewise a
 *     device OUTPUT buffer is complete when the call returns, for every stream.  Host buffers
 *     need nothing: they are staged by copies on the context's stream.
 *   - Images are RGB8, row-major, interleaved (image::DynamicImage::to_rgb, row-major pixels()).
 *   - The library fails (CNIIC_ERR_HIP) when no gfx950 device is usable; there is no CPU fallback.
 */
#ifndef CNIIC_HIP_H
#define CNIIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNIIC_OK                   0
#define CNIIC_ERR_BAD_ARG         -1
#define CNIIC_ERR_TOO_FEW_POINTS  -2  /* src/kmeans.rs:67-68   assert!(points_per_cluster > 0)     */
#define CNIIC_ERR_FEW_ACTIVE      -3  /* src/kmeans.rs:41-57   "Not enough active clusters"        */
#define CNIIC_ERR_HIP             -4  /* HIP runtime / no device                                    */
#define CNIIC_ERR_RCCL            -5
#define CNIIC_ERR_DECODE          -6  /* Option::None from a decode path                            */
#define CNIIC_ERR_NOMEM           -7
#define CNIIC_ERR_CAPACITY        -8  /* caller's output buffer too small (needed size is returned) */
#define CNIIC_ERR_UNSUPPORTED     -9

typedef struct cniic_ctx cniic_ctx;

/* ------------------------------------------------------------------ context */
/* device: HIP device ordinal.  stream: an existing hipStream_t to enqueue on (e.g. the caller's
 * torch stream), or NULL to let the context create its own non-blocking stream. */
int32_t     cniic_ctx_create(int32_t device, void *stream, cniic_ctx **out);
void        cniic_ctx_destroy(cniic_ctx *ctx);
const char *cniic_last_error(const cniic_ctx *ctx);
int32_t     cniic_version(void);
/* 1 for libcniic_hip_testing.so (built with -DCNIIC_TESTING: the test-suite's CNIIC_TEST_* / CNIIC_DBG_* / route-forcing environment
 * knobs are compiled in), 0 for the release library libcniic_hip.so, which reads only the option fallbacks documented below. */
int32_t     cniic_is_testing_build(void);
int32_t     cniic_sync(cniic_ctx *ctx);
/* optional helpers so callers without a HIP binding can keep images resident in HBM */
int32_t     cniic_dev_alloc(cniic_ctx *ctx, uint64_t bytes, void **dptr);
int32_t     cniic_dev_free(cniic_ctx *ctx, void *dptr);
int32_t     cniic_memcpy(cniic_ctx *ctx, void *dst, const void *src, uint64_t bytes);
/* Route switches and thresholds a host can set per context (a Rust caller has no other way: the CNIIC_* environment variables
 * named beside them are read per call ONLY while an option is unset, and exist for the test-suite).  value semantics per option;
 * cniic_ctx_get_opt returns the effective value (set, or the environment's, or the default). */
#define CNIIC_OPT_SP_MIN_PIXELS      1  /* cluster-colors: images of at least this many pixels take the pixel partition by colour        */
                                        /* super-cell (k_points.hip), smaller ones the dense 2^24 table.  Default 2^20. CNIIC_SP_MIN_PIXELS */
#define CNIIC_OPT_HUF_GPU_CODES_MIN  2  /* huf::encode_all: alphabets of at least this many symbols sort their leaves and derive codes  */
                                        /* and decoder on the GPU (the host only merges).  Default 32768.  CNIIC_HUF_GPU_CODES_MIN        */
#define CNIIC_OPT_GPU_DECODE_MIN     3  /* decode: streams of at least this many symbols use the parallel decoder, shorter ones the host  */
                                        /* walk.  Default 16384.  CNIIC_GPU_DECODE_MIN                                                      */
#define CNIIC_OPT_DELTA_ROUTE        4  /* delta: 0 = 16-bit symbol stream where the image allows (default), 32 = always the 32-bit route. */
                                        /* CNIIC_DELTA_ROUTE                                                                                */
#define CNIIC_OPT_STAGE_TIMERS       5  /* 1: HIP-event timers around the stages of every call (they synchronise; read with              */
                                        /* cniic_last_kernel_time).  Default 0.  CNIIC_KERNEL_TIMERS                                       */
#define CNIIC_OPT_FRAME_TREES_HOST   6  /* cniic_cc_finish_frames: 1 = the frames' Huffman trees on host threads instead of the GPU.       */
                                        /* Default 0.  CNIIC_FRAME_TREES_HOST                                                               */
#define CNIIC_OPT_BATCH_STREAMS      7  /* cniic_codec_encode_batch: images in flight at once (worker streams).  Default 8.               */
#define CNIIC_OPT_KM_MAX_BLOCKS      8  /* cluster-colors: cap on the K-means assign kernel's grid (a multiple of 3; 0 = none, 768 blocks  */
                                        /* on large inputs).  A smaller grid lets the launches of several contexts share the machine:     */
                                        /* cniic_codec_encode_batch gives its workers 384 unless this is set.  CNIIC_KM_MAX_BLOCKS          */
#define CNIIC_OPT_KM_LOOP             9  /* cluster-colors (K <= 256, one GPU): 0 = the K-means loop as ONE persistent launch with the       */
                                        /* points resident in LDS (default; falls back by itself when its grid cannot be resident),     */
                                        /* 1 = one launch per iteration.  CNIIC_KM_LOOP                                                    */
#define CNIIC_OPT_COUNT              10
int32_t     cniic_ctx_set_opt(cniic_ctx *ctx, int32_t opt, uint64_t value);
int32_t     cniic_ctx_unset_opt(cniic_ctx *ctx, int32_t opt);
int32_t     cniic_ctx_get_opt(cniic_ctx *ctx, int32_t opt, uint64_t *value);
/* dominant-kernel timing of the most recent call on this ctx, measured with HIP events on the
 * ctx stream: *ms = summed duration, *launches = number of launches of that kernel.
 * Names: "kmeans_rgbw_persist" (the colour K-means as one persistent launch; "kmeans_rgbw_persist_iters": the same duration, launches =
 * its iterations), "kmeans_rgbw_assign" (... as one launch per iteration), "kmeans_xyrgb_iter", "hist_rgb", "remap_rgb", "huff_pack", "hilbert_delta", "undiff_scatter", "hd_pass0" /
 * "hd_check" / "hd_write", and the five consecutive stages of a `delta` encode, which together are the whole call: "delta_gather",
 * "delta_hist", "delta_tree" (compaction, the leaves' sort, the host's merge, the codes), "huff_pack", "delta_finish". */
int32_t     cniic_last_kernel_time(cniic_ctx *ctx, const char *which, double *ms, uint64_t *launches);

/* ------------------------------------------------------------------ H1: utils::count_freqs */
/* Symbol kinds fix the 32-bit key packing and the wire size of a symbol. */
#define CNIIC_SYM_RGB    1  /* Rgb<u8>: key = r<<16|g<<8|b; 11 bytes on the wire (src/ser.rs:210-214)   */
#define CNIIC_SYM_SIGNED 2  /* SignedColor([i16;3]) (src/codec/hilbertc.rs:513-516):                    */
                            /* key = (dr+255)<<18|(dg+255)<<9|(db+255); 6 bytes (src/ser.rs:188-195)    */

/* utils::count_freqs over the pixels of an image (src/utils.rs:4-16; call sites src/huf.rs:30,
 * src/codec/clusterc.rs:21).  Output: distinct colours as packed keys in ASCENDING key order and
 * their occurrence counts.  keys/counts may be NULL to query *n_unique only. */
int32_t cniic_hist_rgb24(cniic_ctx *ctx, const uint8_t *rgb, uint64_t npx,
                         uint32_t *keys, uint64_t *counts, uint64_t cap, uint64_t *n_unique);
/* count_freqs over a stream of packed symbol keys of the given kind. */
int32_t cniic_hist_syms(cniic_ctx *ctx, int32_t sym_kind, const uint32_t *syms, uint64_t n,
                        uint32_t *keys, uint64_t *counts, uint64_t cap, uint64_t *n_unique);

/* ------------------------------------------------------------------ K-means: kmeans::cluster */
typedef struct {
    uint64_t seed;       /* seeds the deterministic empty-cluster reseed (replaces thread_rng,   */
                         /* src/kmeans.rs:123-133); 0 = library default                            */
    uint64_t max_iters;  /* 0 = run until no point moves (the reference has no cap, kmeans.rs:26) */
    uint32_t flags;      /* CNIIC_KM_* */
    uint32_t reserved;
} cniic_kmeans_opts;
#define CNIIC_KM_BRUTE_FORCE 1u  /* disable bound-based pruning (debug / A-B measurement) */
#define CNIIC_KM_PROFILE     2u  /* HIP-event pair around every assign launch -> cniic_last_kernel_time("kmeans_rgbw_assign") */
#define CNIIC_KM_NO_SKIP     4u  /* never use the skip schedule of the pruned assign (A-B measurement) */

typedef struct {
    uint64_t iterations;     /* src/kmeans.rs:33 "#iterations"                         */
    uint64_t moved_last;     /* points that changed cluster in the last iteration      */
    uint64_t empty_reseeds;  /* src/kmeans.rs:117-134 occurrences                      */
    uint64_t active;         /* clusters with >= 1 member (src/kmeans.rs:49-52)        */
    uint64_t pair_evals;     /* point-centroid distance evaluations (mirrors the       */
                             /* "tested neighbours" counters of src/kmeans.rs:401-413) */
} cniic_kmeans_stats;

typedef struct {  /* ColorPos, src/codec/clusterc.rs:200-204 */
    uint32_t x, y;
    uint8_t  rgb[3];
    uint8_t  pad;
} cniic_colorpos;

/* kmeans::cluster::<ColorCount> (src/kmeans.rs:21-39 with src/codec/clusterc.rs:68-114):
 * U distinct colours (packed keys, any order; the order IS the point order of the reference's
 * Vec<ColorCount>) with pixel-count weights.  Out: K centroid colours (K x 3 bytes, r,g,b),
 * final cluster of every input colour (labels[U]), members[K] = colours per cluster. */
int32_t cniic_kmeans_rgbw(cniic_ctx *ctx, const uint32_t *keys, const uint32_t *weight, uint64_t U,
                          uint32_t K, const cniic_kmeans_opts *opts,
                          uint8_t *centroids, uint32_t *labels, uint64_t *members,
                          cniic_kmeans_stats *stats);
/* kmeans::cluster::<ColorPos> over all pixels of an image, row-major point order
 * (src/codec/clusterc.rs:150-153).  labels (N u32) and members may be NULL. */
int32_t cniic_kmeans_xyrgb(cniic_ctx *ctx, const uint8_t *rgb, uint32_t w, uint32_t h,
                           uint32_t K, const cniic_kmeans_opts *opts,
                           cniic_colorpos *centroids, uint32_t *labels, uint64_t *members,
                           cniic_kmeans_stats *stats);

/* One assign step from given centroids and labels (src/kmeans.rs:330-416 + the sums consumed by
 * Point::mean): labels updated in place; sums[K x D] (D = 3 / 5), wsum[K] (sum of weights, or
 * member count), members[K], *changed.  Centroids are not updated.  For parity tests and for
 * callers that own the reduction (multi-GPU). */
int32_t cniic_kmeans_step_rgbw(cniic_ctx *ctx, const uint32_t *keys, const uint32_t *weight,
                               uint64_t U, uint32_t K, const uint8_t *centroids, uint32_t *labels,
                               uint64_t *sums, uint64_t *wsum, uint64_t *members, uint64_t *changed);
int32_t cniic_kmeans_step_xyrgb(cniic_ctx *ctx, const uint8_t *rgb, uint32_t w, uint32_t h,
                                uint32_t K, const cniic_colorpos *centroids, uint32_t *labels,
                                uint64_t *sums, uint64_t *wsum, uint64_t *members, uint64_t *changed);

/* Sharded K-means session (pixels / colours sharded over GPUs, one process per GPU): the library
 * owns assign + partial sums + centroid update on its shard; the CALLER all-reduces the partials
 * buffer (RCCL sum over int64 words) between cniic_km_assign and cniic_km_update.
 * All ranks pass the full point list (U colours); rank r of n passes shard = r, nshards = n and
 * works on its share of the colour-space cells. */
typedef struct cniic_km cniic_km;
int32_t cniic_km_create_rgbw(cniic_ctx *ctx, const uint32_t *keys, const uint32_t *weight, uint64_t U,
                             uint32_t shard, uint32_t nshards, uint32_t K, const cniic_kmeans_opts *opts,
                             void *partials_dev /* device buffer of cniic_km_partial_words(K,3) u64, or NULL */,
                             cniic_km **out);
uint64_t cniic_km_partial_words(uint32_t K, uint32_t D);   /* K*D sums + K wsum + K members + moved + evals */
int32_t cniic_km_partials(cniic_km *km, void **dev_ptr);
/* After create the partials buffer holds this shard's sums of the INITIAL assignment
 * (kmeans.rs:61-78); all-reduce it, then call cniic_km_begin once. */
int32_t cniic_km_begin(cniic_km *km);
/* Each iteration: cniic_km_assign (async; partials <- signed deltas of the points that moved),
 * all-reduce the partials, cniic_km_update. */
int32_t cniic_km_assign(cniic_km *km);                      /* async on the ctx stream */
int32_t cniic_km_update(cniic_km *km, uint64_t *changed);   /* syncs; *changed = global moved count */
/* The labels live on the device in the library's internal (cell-major) point order, all U of
 * them, of which this shard owns [lo,hi): all-gather that range before asking for the result. */
int32_t cniic_km_labels_internal(cniic_km *km, void **dev_ptr, uint64_t *elem_bytes);
/* labels: all U points, the caller's (canonical) order. */
int32_t cniic_km_result(cniic_km *km, uint8_t *centroids, uint32_t *labels, uint64_t *members,
                        cniic_kmeans_stats *stats);
/* average duration (ms) of the assign kernel alone over `reps` back-to-back launches, measured
 * with HIP events on the ctx stream (bench.py's roofline figure) */
int32_t cniic_km_time_assign(cniic_km *km, int32_t reps, double *ms_per_launch);
void    cniic_km_destroy(cniic_km *km);

/* ------------------------------------------------------------------ cluster-colors over several GPUs */
/* One process per GPU, every rank holds its own image(s); the ranks build ONE palette for the
 * union of their pixels (north_star config 4) and each encodes its own image with it.  The caller
 * owns the collectives (RCCL through torch.distributed, or ncclAllReduce directly):
 *
 *   cniic_hist_rgb24_dense(img)            -> local  u32[2^24] colour counts (device)
 *   all-reduce(sum) a COPY of it           -> global counts
 *   cniic_cc_create(global, K, rank, n)    -> distinct colours, K-means state; the global table
 *                                             is overwritten (key -> rank + 1)
 *   repeat: cniic_cc_assign; all-reduce(sum) the partials buffer (int64 words); cniic_cc_update
 *           until *changed == 0  (or update asynchronously and cniic_cc_poll every few iterations)
 *   cniic_cc_export_labels; all-reduce(sum) that buffer; cniic_cc_import_labels
 *   cniic_cc_finish(img, local counts)     -> this rank's Hufman stream (clusterc.rs:31-52)
 *
 * Integer sums make the palette bit-identical for any number of ranks. */
typedef struct cniic_cc cniic_cc;
int32_t  cniic_hist_rgb24_dense(cniic_ctx *ctx, const uint8_t *rgb, uint64_t npx, uint32_t *table_dev);
int32_t  cniic_cc_create(cniic_ctx *ctx, uint32_t *table_dev, uint32_t K, const cniic_kmeans_opts *opts,
                         uint32_t shard, uint32_t nshards,
                         void *partials_dev /* device, cniic_km_partial_words(K,3) u64, or NULL */, cniic_cc **out);
uint64_t cniic_cc_unique(cniic_cc *cc);        /* distinct colours U */
uint32_t cniic_cc_label_bytes(cniic_cc *cc);   /* 1 (K <= 256) or 2: element size of the label buffers */
int32_t  cniic_cc_partials(cniic_cc *cc, void **dev_ptr);
int32_t  cniic_cc_assign(cniic_cc *cc);                        /* async on the ctx stream */
int32_t  cniic_cc_update(cniic_cc *cc, uint64_t *changed);     /* syncs; changed == NULL: asynchronous */
/* iterations completed so far and whether an iteration has moved nothing (syncs).  Iterations issued
 * after convergence are no-ops on the device, so callers may poll only every few iterations. */
int32_t  cniic_cc_poll(cniic_cc *cc, uint64_t *iterations, uint32_t *done);
/* The same without a GPU stall: returns the state as of the PREVIOUS call (*valid = 0 on the first call) and
 * enqueues the copy the next call will read.  Call it after every batch of iterations; every rank sees the
 * same sequence of answers, so all ranks stop after the same batch. */
int32_t  cniic_cc_poll_lagged(cniic_cc *cc, uint64_t *iterations, uint32_t *done, uint32_t *valid);
int32_t  cniic_cc_export_labels(cniic_cc *cc, void *dst_dev);  /* U labels, zero outside this shard */
int32_t  cniic_cc_import_labels(cniic_cc *cc, const void *src_dev);
int32_t  cniic_cc_finish(cniic_cc *cc, const uint8_t *rgb, uint32_t w, uint32_t h,
                         const uint32_t *local_table_dev /* this image's own counts, or NULL = single image */,
                         uint8_t *out, uint64_t cap, uint64_t *len, cniic_kmeans_stats *stats);
/* A BATCH of equally sized frames, contiguous in memory, coded with the session's one palette (north_star: "pixels of an image
 * batch shard across the GPUs ... all-reduce of the K partial centroid sums"; the harness's many-images loop, bench.rs:24-35,
 * with a shared palette -- an extension, the reference has one palette per image).  Open the session on ALL the pixels
 * (cniic_cc_image_begin(frames, F * w * h), or the dense-table calls), run the loop, then this instead of cniic_cc_finish:
 * frame f's Hufman stream (clusterc.rs:31-52 applied to frame f: dims, its own tree, its payload) is written at
 * out + f * stride (stride: a multiple of 4, at least the longest stream rounded up to 4) and its length to lens[f]. */
int32_t  cniic_cc_finish_frames(cniic_cc *cc, const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t frames,
                                uint8_t *out, uint64_t stride, uint64_t *lens, cniic_kmeans_stats *stats);
void     cniic_cc_destroy(cniic_cc *cc);

/* The same shared palette with every rank holding ONLY ITS OWN image's colours (per-rank work and memory do not
 * grow with the number of ranks, no label exchange).  The reference's point list -- the ascending list of the
 * distinct colours of all the pixels -- is then known to every rank as a bitmap:
 *   cniic_hist_rgb24_dense(img)         -> this image's counts, u32[2^24]
 *   cniic_occupancy_pack(counts, occ)   -> one nibble per colour (u32[2^21]), 1 where the colour occurs
 *   all-reduce(sum) occ                 -> non-zero where ANY rank has the colour (<= 15 ranks: nibbles cannot carry)
 *   cniic_cc_create_local(counts, occ)  -> this rank's points, initialised by their position in the list of all
 *                                          occupied colours (init_assignment / init_centroids, kmeans.rs:61-108; the
 *                                          empty-cluster reseed picks from the same list); counts is overwritten
 *   the loop as above (cniic_cc_run, or assign / all-reduce / update); then cniic_cc_finish(img, NULL)
 * A colour that occurs on several ranks is several points with one position: every copy takes the same decisions and
 * the integer sums are those of the single merged point, so centroids, iteration count and palette are bit-identical to
 * clustering the union (only the moved / member COUNTS, used for their zero-ness alone, see each copy). */
int32_t  cniic_occupancy_pack(cniic_ctx *ctx, const uint32_t *table_dev, uint32_t *occ_dev);
int32_t  cniic_cc_create_local(cniic_ctx *ctx, uint32_t *table_dev, const uint32_t *occ_dev, uint32_t K, const cniic_kmeans_opts *opts,
                               void *partials_dev /* u64[5K+2] the caller all-reduces, or NULL with cniic_cc_run */, cniic_cc **out);
/* The same without the dense table, for large images (16-byte aligned device image): the pixels are partitioned by colour
 * super-cell once (what ClusterColors::encode of a single image does here above 2^20 pixels), which gives this image's
 * colours, their occupancy and, after the K-means, every pixel's label without a random read.
 *   cniic_cc_image_begin(img)           -> session holding the partition (no K-means state yet)
 *   cniic_cc_image_occupancy(cc, occ)   -> the nibbles of this image's colours, u32[2^21]; all-reduce(sum) as above
 *   cniic_cc_image_create(cc, occ, K)   -> the K-means state, as cniic_cc_create_local
 *   the loop, then cniic_cc_finish(cc, img, NULL) with the same image */
int32_t  cniic_cc_image_begin(cniic_ctx *ctx, const uint8_t *rgb_dev, uint64_t npx, cniic_cc **out);
int32_t  cniic_cc_image_occupancy(cniic_cc *cc, uint32_t *occ_dev);
int32_t  cniic_cc_image_create(cniic_cc *cc, const uint32_t *occ_dev, uint32_t K, const cniic_kmeans_opts *opts,
                               void *partials_dev /* as cniic_cc_create_local */);

/* ---- RCCL on the context's own stream (ncclAllReduce of the K partial sums between assign and
 * update, no host round trip).  librccl is bound at run time; without it these return CNIIC_ERR_UNSUPPORTED
 * and the caller all-reduces the buffers itself (cniic_cc_assign / cniic_cc_update above).
 *   rank 0: cniic_comm_unique_id -> broadcast the 128 bytes by any means -> every rank: cniic_comm_create
 *   cniic_comm_all_reduce : in-place unsigned sum of a device buffer (elements of 1, 4 or 8 bytes)
 *   cniic_cc_run          : the whole `while changed_assignment` loop (kmeans.rs:26-32) of a cc session, with
 *                           the all-reduce in-stream when comm != NULL; identical on every rank */
typedef struct cniic_comm cniic_comm;
int32_t  cniic_comm_unique_id(uint8_t id[128]);
int32_t  cniic_comm_create(cniic_ctx *ctx, const uint8_t id[128], uint32_t rank, uint32_t nranks, cniic_comm **out);
void     cniic_comm_destroy(cniic_comm *comm);
int32_t  cniic_comm_all_reduce(cniic_comm *comm, void *buf_dev, uint64_t count, int32_t elem_bytes);
/* The same communicator over the caller's own transport (MPI, sockets, gloo ...) where RCCL is not wanted: every
 * all-reduce drains the stream, hands `count` elements of `elem_bytes` bytes to fn in HOST memory and expects the
 * in-place unsigned sum over all ranks there when fn returns 0.  Slower (a host round trip per iteration), same
 * results; cniic_cc_run / cniic_comm_all_reduce take it like the RCCL one.
 * Failures: a rank that fails inside cniic_cc_run (a launch error, a failed collective) aborts its communicator before
 * it returns, so that its peers leave their next all-reduce with CNIIC_ERR_RCCL instead of waiting for it for ever --
 * RCCL: ncclCommAbort here, ncclCommGetAsyncError polled by the peers while they wait for a batch; host transport: fn is
 * called once with (buf_host = NULL, count = 0, elem_bytes = -1) and should tear the caller's transport down (a peer
 * whose fn then fails returns non-zero, which ends that peer's loop the same way).  The communicator is unusable
 * afterwards (every call returns CNIIC_ERR_RCCL): destroy it.
 * A peer that dies WITHOUT aborting (killed process, lost node) is not always reported by RCCL's asynchronous error state
 * (intra-node P2P / SHM transports), so the wait for a batch of launches that contains collectives has a deadline of its own:
 * cniic_comm_set_timeout (default 120 000 ms, or CNIIC_COLLECTIVE_TIMEOUT_MS at creation; 0 = wait for ever).  When it expires
 * cniic_cc_run aborts the communicator and returns CNIIC_ERR_RCCL.  With a host transport every all-reduce is a blocking call
 * of fn: there the transport's own timeout bounds the wait (gloo: the process group's `timeout`). */
typedef int32_t (*cniic_host_sum_fn)(void *user, void *buf_host, uint64_t count, int32_t elem_bytes);
int32_t  cniic_comm_create_host(cniic_ctx *ctx, uint32_t rank, uint32_t nranks, cniic_host_sum_fn fn, void *user, cniic_comm **out);
int32_t  cniic_comm_set_timeout(cniic_comm *comm, uint64_t milliseconds);
/* The same communicator as a ONE-SHOT exchange over mailboxes (k_mailbox.hip), for buffers where RCCL's ring is all latency
 * (the K partial sums: 10 KiB an iteration): every rank owns a mailbox in fine-grained HBM that its peers map through HIP
 * IPC; an all-reduce is one kernel per rank that writes the buffer into a slot of EVERY peer's mailbox over the direct xGMI
 * links, raises a flag behind it, waits for the flags of its own mailbox and adds the slots in rank order (unsigned integer
 * sums: bit-identical to RCCL's result).  At most 16 ranks; buffers larger than max_bytes (0: 1 MiB) go in pieces.
 *   every rank: cniic_comm_create_mailbox -> all-gather the 64-byte handles by any means, in rank order ->
 *   every rank: cniic_comm_connect_mailbox -> cniic_comm_all_reduce / cniic_cc_run / cniic_comm_set_timeout as above
 * (one process per GPU; ranks inside one process find each other's mailboxes without IPC, and need streams that do not share a
 * hardware queue -- a process has four -- since each rank's kernel waits for the kernels of the others).  A wait is bounded INSIDE the
 * kernel by the communicator's timeout (0 or more than 600 000 ms: 600 000 ms), an abort by a peer ends it at once; both
 * surface as CNIIC_ERR_RCCL from cniic_cc_run or the next cniic_comm_all_reduce.  RCCL stays the default transport: this one
 * has only been run between processes that share ONE GPU (tests/test_mailbox.py), not yet across xGMI. */
#define CNIIC_MAILBOX_HANDLE_BYTES 64
int32_t  cniic_comm_create_mailbox(cniic_ctx *ctx, uint32_t rank, uint32_t nranks, uint64_t max_bytes,
                                   uint8_t handle[CNIIC_MAILBOX_HANDLE_BYTES], cniic_comm **out);
int32_t  cniic_comm_connect_mailbox(cniic_comm *comm, const uint8_t *handles /* nranks x 64 bytes, rank order */);
int32_t  cniic_cc_run(cniic_cc *cc, cniic_comm *comm /* NULL: one rank */, cniic_kmeans_stats *stats);

/* ------------------------------------------------------------------ cluster-colors remap */
/* src/codec/clusterc.rs:31-47: every pixel's colour -> the centroid colour of its cluster.
 * keys[U] ascending (as returned by cniic_hist_rgb24), labels[U], centroids[K x 3]. */
int32_t cniic_remap_rgb(cniic_ctx *ctx, const uint8_t *rgb, uint64_t npx, const uint32_t *keys,
                        const uint32_t *labels, uint64_t U, const uint8_t *centroids, uint32_t K,
                        uint8_t *out_rgb);

/* ------------------------------------------------------------------ Hilbert scan + delta */
/* hilbert::iter(w,h) (src/hilbert.rs:40-43): xy[2*d], xy[2*d+1] for d in 0..w*h.
 * The scan is the one frozen by this build (see DESIGN.md "Hilbert scan: parity unpinned"). */
int32_t cniic_hilbert_xy(cniic_ctx *ctx, uint32_t w, uint32_t h, uint32_t *xy);
/* The reference's scan is the un-vendored crate zhang_hilbert 0.1.1 (src/hilbert.rs:40-43, Cargo.toml:15), which this build
 * cannot reproduce (parity unpinned).  A host that HAS the crate injects its order: xy = w * h pairs (x, y), entry d =
 * where ArbHilbertScan32::new([w, h]) is at step d (host or device memory).  From then on every scan-dependent path of this
 * context -- cniic_hilbert_*, `delta`, `hilbert(rle)`, encode and decode -- follows that order for images of exactly w x h
 * (per-position kernels instead of the tile kernels of the built-in 2^n scan), and its streams are the reference's.  The order
 * must visit every pixel exactly once (CNIIC_ERR_BAD_ARG otherwise).  xy == NULL: back to the built-in scan. */
int32_t cniic_ctx_set_scan(cniic_ctx *ctx, uint32_t w, uint32_t h, const uint32_t *xy);
/* hilbert::linearize (src/hilbert.rs:10-12): pixels gathered in scan order. */
int32_t cniic_hilbert_linearize(cniic_ctx *ctx, const uint8_t *rgb, uint32_t w, uint32_t h, uint8_t *out_rgb);
/* DiffStream over the Hilbert-ordered pixels (src/codec/hilbertc.rs:449-477): N packed
 * CNIIC_SYM_SIGNED keys. */
int32_t cniic_hilbert_delta(cniic_ctx *ctx, const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t *syms);
/* Fused gather + delta + count_freqs (the first pass of huf::encode_all inside Delta::encode,
 * src/codec/hilbertc.rs:405-415 -> src/huf.rs:30).  syms may be NULL. */
int32_t cniic_hilbert_delta_hist(cniic_ctx *ctx, const uint8_t *rgb, uint32_t w, uint32_t h,
                                 uint32_t *keys, uint64_t *counts, uint64_t cap, uint64_t *n_unique,
                                 uint32_t *syms);

/* ------------------------------------------------------------------ H2: huf::encode_all */
/* src/huf.rs:22-43: histogram -> tree -> serialised decoder -> MSB-first bit-packed payload.
 * Returns CNIIC_ERR_CAPACITY with *len = needed bytes when cap is too small. */
int32_t cniic_huf_encode_all(cniic_ctx *ctx, int32_t sym_kind, const uint32_t *syms, uint64_t n,
                             uint8_t *out, uint64_t cap, uint64_t *len);
/* size of that stream as a pure function of the histogram. */
int32_t cniic_huf_size(int32_t sym_kind, const uint64_t *counts, uint64_t n, uint64_t *nbytes);

/* ------------------------------------------------------------------ Codec trait (src/codec.rs:14-19) */
/* expr is the reference's --codec= expression: "hufman", "cluster-colors(256)" / "ccol(256)",
 * "voronoi(2048)", "delta", "hilbert(rle)" = "hilbert(rle(0))" (src/codec.rs:41-59, FromStr impls of each
 * codec; hilbertc.rs:341-397 for the last one, whose name() is "hilbert-rle"; rle(d != 0) and zip are not built). */
int32_t cniic_codec_parse(const char *expr, int32_t *kind, uint32_t *arg);
int32_t cniic_codec_name(const char *expr, char *buf, uint64_t cap);   /* Codec::name()        */
int32_t cniic_codec_is_lossless(const char *expr);                     /* 1 / 0 / negative err */
/* Codec::encode: appends nothing, writes the whole stream to out[0..*len). */
int32_t cniic_codec_encode(cniic_ctx *ctx, const char *expr, const uint8_t *rgb, uint32_t w, uint32_t h,
                           uint8_t *out, uint64_t cap, uint64_t *len, cniic_kmeans_stats *stats);
/* same, with explicit K-means options (seed, iteration cap) */
int32_t cniic_codec_encode_opts(cniic_ctx *ctx, const char *expr, const cniic_kmeans_opts *opts, const uint8_t *rgb,
                                uint32_t w, uint32_t h, uint8_t *out, uint64_t cap, uint64_t *len,
                                cniic_kmeans_stats *stats);
/* The harness's many-images loop (src/bench.rs:24-35: `paths.into_par_iter()`, one Codec::encode per rayon worker) as ONE call:
 * `frames` images of w x h, contiguous in memory (image f at rgb + f * w * h * 3), each encoded on its own exactly as
 * cniic_codec_encode would -- its own histogram, its own palette, its own stream (byte for byte; tests) -- written at
 * out + f * stride with its length in lens[f].  The images are dealt to CNIIC_OPT_BATCH_STREAMS worker contexts of this context
 * (own HIP streams, own scratch; created on first use, destroyed with the context), so that the dependent launches of one
 * image's K-means fill the gaps between another's.  rcs (may be NULL): per-image status; the call returns the first failure.
 * stats (may be NULL): `frames` entries. */
int32_t cniic_codec_encode_batch(cniic_ctx *ctx, const char *expr, const cniic_kmeans_opts *opts, const uint8_t *rgb, uint32_t w, uint32_t h,
                                 uint32_t frames, uint8_t *out, uint64_t stride, uint64_t *lens, int32_t *rcs, cniic_kmeans_stats *stats);
/* Codec::decode: CNIIC_ERR_DECODE where the reference returns None / panics. */
int32_t cniic_codec_decode(cniic_ctx *ctx, const char *expr, const uint8_t *bytes, uint64_t n,
                           uint8_t *rgb, uint64_t cap, uint32_t *w, uint32_t *h);
/* bench::compute_error (src/bench.rs:95-104): MSE between two RGB8 images. */
int32_t cniic_mse(cniic_ctx *ctx, const uint8_t *a, const uint8_t *b, uint64_t npx, double *mse);

/* ------------------------------------------------------------------ synthetic inputs (bench/tests) */
#define CNIIC_SYNTH_UNIFORM 0  /* "U": splitmix64 byte stream                               */
#define CNIIC_SYNTH_PHOTO   1  /* "P": bilinear 64-px lattice + noise, photo-like statistics */
int32_t cniic_synth_image(cniic_ctx *ctx, int32_t kind, uint64_t seed, uint32_t w, uint32_t h, uint8_t *rgb);

#ifdef __cplusplus
}
#endif
#endif
