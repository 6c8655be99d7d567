// capi.cpp -- the extern "C" surface declared in include/cniic_hip.h.
// Every entry point: lock the context, bind host-or-device buffers, run the HIP path, report an
// error code.  Nothing here computes on the CPU what the reference computes per pixel.
#include <algorithm>
#include <cstring>
#include <memory>

#include "codec.hpp"
#include <atomic>
#include <thread>

#include "common.hpp"
#include "huff_host.hpp"

using namespace cniic;

struct cniic_ctx : public Ctx {};

// routes every DevBuf allocated during the call through the context's caching pool
struct PoolScope {
    DevPool *prev;
    explicit PoolScope(DevPool *p) : prev(current_pool()) { current_pool() = p; }
    ~PoolScope() { current_pool() = prev; }
};

struct cniic_km {
    Ctx *c = nullptr;
    KmRgbwState *st = nullptr;
    In<uint32_t> keys, weight;
    uint64_t lo = 0, hi = 0, U = 0;
    uint32_t K = 0;
};

#define LOCK(ctx)                          \
    if (!(ctx)) return CNIIC_ERR_BAD_ARG;  \
    std::lock_guard<std::mutex> _lk((ctx)->mu); \
    (ctx)->err.clear();                    \
    PoolScope _ps(&(ctx)->pool);           \
    do { hipError_t _e = hipSetDevice((ctx)->device); if (_e != hipSuccess) return (ctx)->fail(CNIIC_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(_e)); } while (0)

// results computed into host vectors -> caller buffer (host or device)
static int to_caller(Ctx *c, void *dst, const void *src_host, uint64_t bytes) {
    if (!bytes || !dst) return CNIIC_OK;
    if (is_device_ptr(dst)) CNIIC_HIP_TRY(c, hipMemcpy(dst, src_host, bytes, hipMemcpyHostToDevice));
    else memcpy(dst, src_host, bytes);
    return CNIIC_OK;
}
// small caller arrays (host or device) -> host vector
static int from_caller(Ctx *c, void *dst_host, const void *src, uint64_t bytes) {
    if (!bytes) return CNIIC_OK;
    if (is_device_ptr(src)) CNIIC_HIP_TRY(c, hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost));
    else memcpy(dst_host, src, bytes);
    return CNIIC_OK;
}

extern "C" {

int32_t cniic_version(void) { return 100; }
int32_t cniic_is_testing_build(void) {
#ifdef CNIIC_TESTING
    return 1;
#else
    return 0;
#endif
}

int32_t cniic_ctx_create(int32_t device, void *stream, cniic_ctx **out) {
    if (!out) return CNIIC_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CNIIC_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return CNIIC_ERR_HIP;
    auto *c = new cniic_ctx();
    c->device = device;
    if (stream) { c->stream = reinterpret_cast<hipStream_t>(stream); c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return CNIIC_ERR_HIP; }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { delete c; return CNIIC_ERR_HIP; }
    *out = c;
    return CNIIC_OK;
}

void cniic_ctx_destroy(cniic_ctx *c) {
    if (!c) return;
    for (void *w : c->batch_workers) cniic_ctx_destroy(static_cast<cniic_ctx *>(w));
    c->batch_workers.clear();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->dense.release();
    c->scan_xy.release();
    c->scan_leaves.reset();   // (device tables of the scan of large rectangles)
    c->pool.trim();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto e : c->poll_ev) if (e) (void)hipEventDestroy(e);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_ps) (void)hipHostFree(c->pinned_ps);
    if (c->pinned_res) (void)hipHostFree(c->pinned_res);
    if (c->pinned_huf) (void)hipHostFree(c->pinned_huf);
    if (c->res_ev) (void)hipEventDestroy(c->res_ev);
    if (c->pinned_u) (void)hipHostFree(c->pinned_u);
    if (c->u_ev) (void)hipEventDestroy(c->u_ev);
    if (c->huf_ev) (void)hipEventDestroy(c->huf_ev);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *cniic_last_error(const cniic_ctx *c) { return c ? c->err.c_str() : "null context"; }

int32_t cniic_sync(cniic_ctx *c) {
    LOCK(c);
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

int32_t cniic_dev_alloc(cniic_ctx *c, uint64_t bytes, void **dptr) {
    LOCK(c);
    if (!dptr) return c->fail(CNIIC_ERR_BAD_ARG, "dev_alloc: null out pointer");
    CNIIC_HIP_TRY(c, hipMalloc(dptr, bytes ? bytes : 16));
    return CNIIC_OK;
}

int32_t cniic_dev_free(cniic_ctx *c, void *dptr) {
    LOCK(c);
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    CNIIC_HIP_TRY(c, hipFree(dptr));
    return CNIIC_OK;
}

int32_t cniic_memcpy(cniic_ctx *c, void *dst, const void *src, uint64_t bytes) {
    LOCK(c);
    if (!bytes) return CNIIC_OK;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

int32_t cniic_ctx_set_opt(cniic_ctx *c, int32_t opt, uint64_t value) {
    if (!c) return CNIIC_ERR_BAD_ARG;
    LOCK(c);
    if (opt <= 0 || opt >= CNIIC_OPT_COUNT) return c->fail(CNIIC_ERR_BAD_ARG, "ctx_set_opt: unknown option %d", opt);
    c->opt_val[opt] = value;
    c->opt_set |= 1u << opt;
    if (opt == CNIIC_OPT_STAGE_TIMERS) c->timers = value != 0;
    return CNIIC_OK;
}

int32_t cniic_ctx_unset_opt(cniic_ctx *c, int32_t opt) {
    if (!c) return CNIIC_ERR_BAD_ARG;
    LOCK(c);
    if (opt <= 0 || opt >= CNIIC_OPT_COUNT) return c->fail(CNIIC_ERR_BAD_ARG, "ctx_unset_opt: unknown option %d", opt);
    c->opt_set &= ~(1u << opt);
    if (opt == CNIIC_OPT_STAGE_TIMERS) c->timers = getenv("CNIIC_KERNEL_TIMERS") != nullptr;
    return CNIIC_OK;
}

int32_t cniic_ctx_set_scan(cniic_ctx *c, uint32_t w, uint32_t h, const uint32_t *xy) {
    LOCK(c);
    return scan_inject(c, w, h, xy, xy && is_device_ptr(xy));
}

int32_t cniic_ctx_get_opt(cniic_ctx *c, int32_t opt, uint64_t *value) {
    if (!c || !value) return CNIIC_ERR_BAD_ARG;
    LOCK(c);
    static const struct { const char *env; uint64_t dflt; } k[CNIIC_OPT_COUNT] = {
        {nullptr, 0}, {"CNIIC_SP_MIN_PIXELS", 1ull << 20}, {"CNIIC_HUF_GPU_CODES_MIN", 32768}, {"CNIIC_GPU_DECODE_MIN", 1ull << 14},
        {"CNIIC_DELTA_ROUTE", 0}, {nullptr, 0}, {"CNIIC_FRAME_TREES_HOST", 0}, {nullptr, 8}, {"CNIIC_KM_MAX_BLOCKS", 0}, {"CNIIC_KM_LOOP", 0}};
    if (opt <= 0 || opt >= CNIIC_OPT_COUNT) return c->fail(CNIIC_ERR_BAD_ARG, "ctx_get_opt: unknown option %d", opt);
    *value = opt == CNIIC_OPT_STAGE_TIMERS ? (c->timers ? 1 : 0) : c->opt(opt, k[opt].env, k[opt].dflt);
    return CNIIC_OK;
}

int32_t cniic_last_kernel_time(cniic_ctx *c, const char *which, double *ms, uint64_t *launches) {
    if (!c || !which) return CNIIC_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->ktimes.find(which);
    if (it == c->ktimes.end()) { if (ms) *ms = 0; if (launches) *launches = 0; return CNIIC_ERR_BAD_ARG; }
    if (ms) *ms = it->second.ms;
    if (launches) *launches = it->second.launches;
    return CNIIC_OK;
}

// ------------------------------------------------------------------ H1
static int hist_common(Ctx *c, uint32_t bits, uint32_t *table, uint32_t *keys, uint64_t *counts, uint64_t cap, uint64_t *n_unique) {
    CompactPlan plan;
    CNIIC_TRY(hist_compact_count(c, table, bits, &plan));
    if (n_unique) *n_unique = plan.n_unique;
    if (!keys && !counts) return CNIIC_OK;
    if (plan.n_unique > cap)
        return c->fail(CNIIC_ERR_CAPACITY, "histogram has %llu distinct symbols, capacity %llu", (unsigned long long)plan.n_unique,
                       (unsigned long long)cap);
    Out<uint32_t> ko;
    Out<uint64_t> co;
    CNIIC_TRY(ko.bind(c, keys, plan.n_unique));
    CNIIC_TRY(co.bind(c, counts, plan.n_unique));
    CNIIC_TRY(hist_compact_write(c, table, &plan, ko.d, co.d, nullptr));
    CNIIC_TRY(ko.finish(c));
    CNIIC_TRY(co.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

int32_t cniic_hist_rgb24(cniic_ctx *c, const uint8_t *rgb, uint64_t npx, uint32_t *keys, uint64_t *counts, uint64_t cap,
                         uint64_t *n_unique) {
    LOCK(c);
    c->ktimes.clear();
    if (npx && !rgb) return c->fail(CNIIC_ERR_BAD_ARG, "hist_rgb24: null image");
    if (npx >= (1ull << 32)) return c->fail(CNIIC_ERR_BAD_ARG, "hist_rgb24: too many pixels");
    In<uint8_t> in;
    CNIIC_TRY(in.bind(c, rgb, npx * 3));
    uint32_t *table = nullptr;
    CNIIC_TRY(dense_table(c, 24, &table));
    {
        ScopedKernelTimer t(c, "hist_rgb");
        CNIIC_TRY(hist_rgb_dense(c, in.d, npx, table));
        t.stop(1);
    }
    return hist_common(c, 24, table, keys, counts, cap, n_unique);
}

int32_t cniic_hist_syms(cniic_ctx *c, int32_t sym_kind, const uint32_t *syms, uint64_t n, uint32_t *keys, uint64_t *counts,
                        uint64_t cap, uint64_t *n_unique) {
    LOCK(c);
    c->ktimes.clear();
    if (sym_kind != CNIIC_SYM_RGB && sym_kind != CNIIC_SYM_SIGNED) return c->fail(CNIIC_ERR_BAD_ARG, "hist_syms: bad symbol kind");
    if (n && !syms) return c->fail(CNIIC_ERR_BAD_ARG, "hist_syms: null stream");
    if (n >= (1ull << 32)) return c->fail(CNIIC_ERR_BAD_ARG, "hist_syms: too many symbols");
    const uint32_t bits = sym_kind == CNIIC_SYM_RGB ? 24 : 27;
    In<uint32_t> in;
    CNIIC_TRY(in.bind(c, syms, n));
    uint32_t *table = nullptr;
    CNIIC_TRY(dense_table(c, bits, &table));
    CNIIC_TRY(hist_syms_dense(c, in.d, n, table, bits));
    return hist_common(c, bits, table, keys, counts, cap, n_unique);
}

// ------------------------------------------------------------------ K-means
int32_t cniic_kmeans_rgbw(cniic_ctx *c, const uint32_t *keys, const uint32_t *weight, uint64_t U, uint32_t K,
                          const cniic_kmeans_opts *opts, uint8_t *centroids, uint32_t *labels, uint64_t *members,
                          cniic_kmeans_stats *stats) {
    LOCK(c);
    c->ktimes.clear();
    if (!keys || !weight || !centroids) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_rgbw: null argument");
    In<uint32_t> k, w;
    CNIIC_TRY(k.bind(c, keys, U));
    CNIIC_TRY(w.bind(c, weight, U));
    KmRgbwState *km = nullptr;
    CNIIC_TRY(km_rgbw_create(c, k.d, w.d, U, 0, 1, K, opts, nullptr, nullptr, &km));
    std::unique_ptr<KmRgbwState, void (*)(KmRgbwState *)> guard(km, km_rgbw_destroy);
    CNIIC_TRY(km_rgbw_run(km));
    Out<uint32_t> lo;
    CNIIC_TRY(lo.bind(c, labels, U));
    cniic_kmeans_stats st{};
    std::vector<uint8_t> cent(3 * (size_t)K);
    std::vector<uint64_t> mem(K);
    CNIIC_TRY(km_rgbw_result(km, cent.data(), lo.d, mem.data(), nullptr, &st));
    CNIIC_TRY(lo.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    CNIIC_TRY(to_caller(c, centroids, cent.data(), cent.size()));
    CNIIC_TRY(to_caller(c, members, mem.data(), mem.size() * 8));
    if (stats) *stats = st;
    uint64_t min_cc = (uint64_t)(0.99 * (double)K);  // check_enough_active_clusters kmeans.rs:41-57
    if (U < min_cc) min_cc = U;
    if (st.active < min_cc)
        return c->fail(CNIIC_ERR_FEW_ACTIVE, "Not enough active clusters: requested %u, got %llu (min allowed: %llu)", K,
                       (unsigned long long)st.active, (unsigned long long)min_cc);
    return CNIIC_OK;
}

int32_t cniic_kmeans_step_rgbw(cniic_ctx *c, const uint32_t *keys, const uint32_t *weight, uint64_t U, uint32_t K,
                               const uint8_t *centroids, uint32_t *labels, uint64_t *sums, uint64_t *wsum, uint64_t *members,
                               uint64_t *changed) {
    LOCK(c);
    c->ktimes.clear();
    if (!keys || !weight || !centroids || !labels) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_step_rgbw: null argument");
    In<uint32_t> k, w, lin;
    CNIIC_TRY(k.bind(c, keys, U));
    CNIIC_TRY(w.bind(c, weight, U));
    CNIIC_TRY(lin.bind(c, labels, U));
    std::vector<uint8_t> cent(3 * (size_t)K);
    CNIIC_TRY(from_caller(c, cent.data(), centroids, cent.size()));
    KmRgbwState *km = nullptr;
    cniic_kmeans_opts step_opts{0, 0, CNIIC_KM_BRUTE_FORCE, 0};  // explicit centroids + labels: full-sum kernel
    CNIIC_TRY(km_rgbw_create(c, k.d, w.d, U, 0, 1, K, &step_opts, nullptr, nullptr, &km));
    std::unique_ptr<KmRgbwState, void (*)(KmRgbwState *)> guard(km, km_rgbw_destroy);
    CNIIC_TRY(km_rgbw_set_state(km, cent.data(), lin.d));
    CNIIC_TRY(km_rgbw_assign(km));
    std::vector<uint64_t> s(3 * (size_t)K), ws(K), mem(K);
    uint64_t ch = 0;
    CNIIC_TRY(km_rgbw_partials(km, s.data(), ws.data(), mem.data(), &ch));
    Out<uint32_t> lo;
    CNIIC_TRY(lo.bind(c, labels, U));
    CNIIC_TRY(km_rgbw_result(km, nullptr, lo.d, nullptr, nullptr, nullptr));
    CNIIC_TRY(lo.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    CNIIC_TRY(to_caller(c, sums, s.data(), s.size() * 8));
    CNIIC_TRY(to_caller(c, wsum, ws.data(), ws.size() * 8));
    CNIIC_TRY(to_caller(c, members, mem.data(), mem.size() * 8));
    if (changed) *changed = ch;
    return CNIIC_OK;
}

int32_t cniic_kmeans_xyrgb(cniic_ctx *c, const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t K, const cniic_kmeans_opts *opts,
                           cniic_colorpos *centroids, uint32_t *labels, uint64_t *members, cniic_kmeans_stats *stats) {
    LOCK(c);
    c->ktimes.clear();
    if (!rgb || !centroids) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_xyrgb: null argument");
    const uint64_t N = (uint64_t)w * h;
    In<uint8_t> in;
    CNIIC_TRY(in.bind(c, rgb, N * 3));
    Out<uint32_t> lo;
    CNIIC_TRY(lo.bind(c, labels, N));
    std::vector<cniic_colorpos> cent(K ? K : 1);
    std::vector<uint64_t> mem(K ? K : 1);
    cniic_kmeans_stats st{};
    CNIIC_TRY(km_xyrgb_run(c, in.d, w, h, K, opts, cent.data(), lo.d, mem.data(), &st));
    CNIIC_TRY(lo.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    CNIIC_TRY(to_caller(c, centroids, cent.data(), (size_t)K * sizeof(cniic_colorpos)));
    CNIIC_TRY(to_caller(c, members, mem.data(), (size_t)K * 8));
    if (stats) *stats = st;
    uint64_t min_cc = (uint64_t)(0.99 * (double)K);
    if (N < min_cc) min_cc = N;
    if (st.active < min_cc)
        return c->fail(CNIIC_ERR_FEW_ACTIVE, "Not enough active clusters: requested %u, got %llu (min allowed: %llu)", K,
                       (unsigned long long)st.active, (unsigned long long)min_cc);
    return CNIIC_OK;
}

int32_t cniic_kmeans_step_xyrgb(cniic_ctx *c, const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t K,
                                const cniic_colorpos *centroids, uint32_t *labels, uint64_t *sums, uint64_t *wsum,
                                uint64_t *members, uint64_t *changed) {
    LOCK(c);
    c->ktimes.clear();
    if (!rgb || !centroids || !labels) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_step_xyrgb: null argument");
    const uint64_t N = (uint64_t)w * h;
    In<uint8_t> in;
    CNIIC_TRY(in.bind(c, rgb, N * 3));
    // labels are in/out: stage a device copy when the caller's buffer is host memory
    DevBuf lab_own;
    uint32_t *lab_d = labels;
    const bool lab_dev = is_device_ptr(labels);
    if (!lab_dev) {
        CNIIC_HIP_TRY(c, lab_own.alloc(N * 4));
        CNIIC_HIP_TRY(c, hipMemcpyAsync(lab_own.p, labels, N * 4, hipMemcpyHostToDevice, c->stream));
        lab_d = lab_own.as<uint32_t>();
    }
    std::vector<cniic_colorpos> cent(K ? K : 1);
    CNIIC_TRY(from_caller(c, cent.data(), centroids, (size_t)K * sizeof(cniic_colorpos)));
    std::vector<uint64_t> s(5 * (size_t)K + 1), ws(K + 1), mem(K + 1);
    uint64_t ch = 0;
    CNIIC_TRY(km_xyrgb_step(c, in.d, w, h, K, cent.data(), lab_d, s.data(), ws.data(), mem.data(), &ch, nullptr));
    if (!lab_dev) CNIIC_HIP_TRY(c, hipMemcpy(labels, lab_d, N * 4, hipMemcpyDeviceToHost));
    CNIIC_TRY(to_caller(c, sums, s.data(), 5 * (size_t)K * 8));
    CNIIC_TRY(to_caller(c, wsum, ws.data(), (size_t)K * 8));
    CNIIC_TRY(to_caller(c, members, mem.data(), (size_t)K * 8));
    if (changed) *changed = ch;
    return CNIIC_OK;
}

// ------------------------------------------------------------------ sharded session
uint64_t cniic_km_partial_words(uint32_t K, uint32_t D) { return (uint64_t)K * D + 2ull * K + 2; }

int32_t cniic_km_create_rgbw(cniic_ctx *c, const uint32_t *keys, const uint32_t *weight, uint64_t U, uint32_t shard, uint32_t nshards,
                             uint32_t K, const cniic_kmeans_opts *opts, void *partials_dev, cniic_km **out) {
    LOCK(c);
    if (!out || !keys || !weight) return c->fail(CNIIC_ERR_BAD_ARG, "km_create_rgbw: null argument");
    if (partials_dev && !is_device_ptr(partials_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "km_create_rgbw: partials must be device memory");
    auto km = std::make_unique<cniic_km>();
    km->c = c; km->K = K; km->U = U;
    CNIIC_TRY(km->keys.bind(c, keys, U));
    CNIIC_TRY(km->weight.bind(c, weight, U));
    CNIIC_TRY(km_rgbw_create(c, km->keys.d, km->weight.d, U, shard, nshards, K, opts, partials_dev, nullptr, &km->st));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    *out = km.release();
    return CNIIC_OK;
}

int32_t cniic_km_partials(cniic_km *km, void **dev_ptr) {
    if (!km || !dev_ptr) return CNIIC_ERR_BAD_ARG;
    *dev_ptr = km_rgbw_partials_dev(km->st);
    return CNIIC_OK;
}

int32_t cniic_km_begin(cniic_km *km) {
    if (!km) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(km->c);
    LOCK(c);
    return km_rgbw_fold_initial(km->st);
}

int32_t cniic_km_labels_internal(cniic_km *km, void **dev_ptr, uint64_t *elem_bytes) {
    if (!km || !dev_ptr) return CNIIC_ERR_BAD_ARG;
    *dev_ptr = km_rgbw_labels_internal(km->st, elem_bytes);
    return CNIIC_OK;
}

int32_t cniic_km_assign(cniic_km *km) {
    if (!km) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(km->c);
    LOCK(c);
    return km_rgbw_assign(km->st);
}

int32_t cniic_km_update(cniic_km *km, uint64_t *changed) {
    if (!km) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(km->c);
    LOCK(c);
    CNIIC_TRY(km_rgbw_update(km->st));
    uint64_t ch = 0;
    CNIIC_TRY(km_rgbw_poll_changed(km->st, &ch));
    if (changed) *changed = ch;
    return CNIIC_OK;
}

int32_t cniic_km_result(cniic_km *km, uint8_t *centroids, uint32_t *labels_slice, uint64_t *members, cniic_kmeans_stats *stats) {
    if (!km) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(km->c);
    LOCK(c);
    Out<uint32_t> lo;
    CNIIC_TRY(lo.bind(c, labels_slice, km->U));
    std::vector<uint8_t> cent(3 * (size_t)km->K);
    std::vector<uint64_t> mem(km->K);
    cniic_kmeans_stats st{};
    CNIIC_TRY(km_rgbw_result(km->st, cent.data(), lo.d, mem.data(), nullptr, &st));
    CNIIC_TRY(lo.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    CNIIC_TRY(to_caller(c, centroids, cent.data(), cent.size()));
    CNIIC_TRY(to_caller(c, members, mem.data(), mem.size() * 8));
    if (stats) *stats = st;
    return CNIIC_OK;
}

int32_t cniic_km_time_assign(cniic_km *km, int32_t reps, double *ms_per_launch) {
    if (!km || !ms_per_launch || reps <= 0) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(km->c);
    LOCK(c);
    return km_rgbw_time_assign(km->st, reps, ms_per_launch);
}

void cniic_km_destroy(cniic_km *km) {
    if (!km) return;
    {
        std::lock_guard<std::mutex> lk(km->c->mu);
        (void)hipSetDevice(km->c->device);
        (void)hipStreamSynchronize(km->c->stream);
        PoolScope ps(&km->c->pool);
        km_rgbw_destroy(km->st);
        km->keys.own.release();
        km->weight.own.release();
    }
    delete km;
}

// ------------------------------------------------------------------ sharded cluster-colors session
struct cniic_cc {
    Ctx *c = nullptr;
    CcSession *s = nullptr;
};

int32_t cniic_hist_rgb24_dense(cniic_ctx *c, const uint8_t *rgb, uint64_t npx, uint32_t *table_dev) {
    LOCK(c);
    c->ktimes.clear();
    if (!table_dev || !is_device_ptr(table_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "hist_rgb24_dense: table must be device memory (u32[2^24])");
    if (npx && !rgb) return c->fail(CNIIC_ERR_BAD_ARG, "hist_rgb24_dense: null image");
    if (npx >= (1ull << 32)) return c->fail(CNIIC_ERR_BAD_ARG, "hist_rgb24_dense: too many pixels");
    In<uint8_t> in;
    CNIIC_TRY(in.bind(c, rgb, npx * 3));
    CNIIC_HIP_TRY(c, hipMemsetAsync(table_dev, 0, (1ull << 24) * 4, c->stream));
    ScopedKernelTimer t(c, "hist_rgb");
    CNIIC_TRY(hist_rgb_dense(c, in.d, npx, table_dev));
    t.stop(1);
    return CNIIC_OK;
}

int32_t cniic_cc_create(cniic_ctx *c, uint32_t *table_dev, uint32_t K, const cniic_kmeans_opts *opts, uint32_t shard,
                        uint32_t nshards, void *partials_dev, cniic_cc **out) {
    LOCK(c);
    if (!out || !table_dev || !is_device_ptr(table_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "cc_create: table must be device memory");
    if (partials_dev && !is_device_ptr(partials_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "cc_create: partials must be device memory");
    CcSession *s = nullptr;
    CNIIC_TRY(cc_prepare(c, table_dev, K, opts, shard, nshards, partials_dev, &s));
    auto *cc = new cniic_cc();
    cc->c = c; cc->s = s;
    *out = cc;
    return CNIIC_OK;
}

uint64_t cniic_cc_unique(cniic_cc *cc) { return cc && cc->s ? cc->s->U : 0; }
uint32_t cniic_cc_label_bytes(cniic_cc *cc) { return cc && cc->s && cc->s->km && km_rgbw_is_wide(cc->s->km) ? 2 : 1; }

int32_t cniic_cc_assign(cniic_cc *cc) {
    if (!cc) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->km) return c->fail(CNIIC_ERR_BAD_ARG, "the session has no K-means state yet (cniic_cc_image_create comes first)");
    return km_rgbw_assign(cc->s->km);
}

int32_t cniic_cc_update(cniic_cc *cc, uint64_t *changed) {
    if (!cc) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->km) return c->fail(CNIIC_ERR_BAD_ARG, "the session has no K-means state yet (cniic_cc_image_create comes first)");
    CNIIC_TRY(km_rgbw_update(cc->s->km));
    if (!changed) return CNIIC_OK;  // asynchronous: the caller polls later with cniic_cc_poll
    uint64_t ch = 0;
    CNIIC_TRY(km_rgbw_poll_changed(cc->s->km, &ch));
    *changed = ch;
    return CNIIC_OK;
}

int32_t cniic_cc_poll(cniic_cc *cc, uint64_t *iterations, uint32_t *done) {
    if (!cc) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->km) return c->fail(CNIIC_ERR_BAD_ARG, "the session has no K-means state yet (cniic_cc_image_create comes first)");
    cniic_kmeans_stats st{};
    uint32_t d = 0;
    CNIIC_TRY(km_rgbw_poll(cc->s->km, &st, &d));
    if (iterations) *iterations = st.iterations;
    if (done) *done = d;
    return CNIIC_OK;
}

int32_t cniic_occupancy_pack(cniic_ctx *c, const uint32_t *table_dev, uint32_t *occ_dev) {
    LOCK(c);
    if (!table_dev || !occ_dev || !is_device_ptr(table_dev) || !is_device_ptr(occ_dev))
        return c->fail(CNIIC_ERR_BAD_ARG, "occupancy_pack: device buffers u32[2^24] and u32[2^21]");
    return occupancy_pack(c, table_dev, occ_dev);
}

int32_t cniic_cc_create_local(cniic_ctx *c, uint32_t *table_dev, const uint32_t *occ_dev, uint32_t K, const cniic_kmeans_opts *opts,
                              void *partials_dev, cniic_cc **out) {
    LOCK(c);
    if (!table_dev || !occ_dev || !out || !is_device_ptr(table_dev) || !is_device_ptr(occ_dev))
        return c->fail(CNIIC_ERR_BAD_ARG, "cc_create_local: device table, device occupancy and an out pointer are needed");
    if (partials_dev && !is_device_ptr(partials_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "cc_create_local: partials must be device memory");
    CcSession *s = nullptr;
    CNIIC_TRY(cc_prepare(c, table_dev, K, opts, 0, 1, partials_dev, &s, occ_dev));
    *out = new cniic_cc{c, s};
    return CNIIC_OK;
}

int32_t cniic_cc_image_begin(cniic_ctx *c, const uint8_t *rgb_dev, uint64_t npx, cniic_cc **out) {
    LOCK(c);
    if (!rgb_dev || !out || npx == 0 || !is_device_ptr(rgb_dev) || (reinterpret_cast<uintptr_t>(rgb_dev) & 15))
        return c->fail(CNIIC_ERR_BAD_ARG, "cc_image_begin: a non-empty, 16-byte aligned device image and an out pointer are needed");
    if (npx >= (1ull << 32)) return c->fail(CNIIC_ERR_BAD_ARG, "cc_image_begin: too many pixels");
    c->ktimes.clear();
    CcSession *s = nullptr;
    CNIIC_TRY(cc_image_begin(c, rgb_dev, npx, &s));
    *out = new cniic_cc{c, s};
    return CNIIC_OK;
}

int32_t cniic_cc_image_occupancy(cniic_cc *cc, uint32_t *occ_dev) {
    if (!cc) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->sp_mode || !occ_dev || !is_device_ptr(occ_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "cc_image_occupancy: image session and device buffer needed");
    return sp_occupancy(c, &cc->s->sp, occ_dev);
}

int32_t cniic_cc_image_create(cniic_cc *cc, const uint32_t *occ_dev, uint32_t K, const cniic_kmeans_opts *opts, void *partials_dev) {
    if (!cc) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!occ_dev || !is_device_ptr(occ_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "cc_image_create: device occupancy needed");
    if (partials_dev && !is_device_ptr(partials_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "cc_image_create: partials must be device memory");
    return cc_image_create(cc->s, occ_dev, K, opts, partials_dev);
}

int32_t cniic_cc_poll_lagged(cniic_cc *cc, uint64_t *iterations, uint32_t *done, uint32_t *valid) {
    if (!cc || !done || !valid) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->km) return c->fail(CNIIC_ERR_BAD_ARG, "the session has no K-means state yet (cniic_cc_image_create comes first)");
    cniic_kmeans_stats st{};
    CNIIC_TRY(km_rgbw_poll_lagged(cc->s->km, &st, done, valid));
    if (iterations) *iterations = st.iterations;
    return CNIIC_OK;
}

// ---- RCCL communicator on the context's stream
struct cniic_comm { Comm *m = nullptr; };

int32_t cniic_comm_unique_id(uint8_t id[128]) {
    if (!id) return CNIIC_ERR_BAD_ARG;
    return comm_unique_id(id);
}

int32_t cniic_comm_create(cniic_ctx *c, const uint8_t id[128], uint32_t rank, uint32_t nranks, cniic_comm **out) {
    if (!c || !id || !out) return CNIIC_ERR_BAD_ARG;
    LOCK(c);
    Comm *m = nullptr;
    CNIIC_TRY(comm_create(c, id, rank, nranks, &m));
    *out = new cniic_comm{m};
    return CNIIC_OK;
}

int32_t cniic_comm_create_host(cniic_ctx *c, uint32_t rank, uint32_t nranks, cniic_host_sum_fn fn, void *user, cniic_comm **out) {
    if (!c || !out) return CNIIC_ERR_BAD_ARG;
    LOCK(c);
    Comm *m = nullptr;
    CNIIC_TRY(comm_create_host(c, rank, nranks, fn, user, &m));
    *out = new cniic_comm{m};
    return CNIIC_OK;
}

int32_t cniic_comm_create_mailbox(cniic_ctx *c, uint32_t rank, uint32_t nranks, uint64_t max_bytes, uint8_t handle[64], cniic_comm **out) {
    if (!c || !handle || !out) return CNIIC_ERR_BAD_ARG;
    LOCK(c);
    Comm *m = nullptr;
    CNIIC_TRY(comm_create_mailbox(c, rank, nranks, max_bytes, handle, &m));
    *out = new cniic_comm{m};
    return CNIIC_OK;
}

int32_t cniic_comm_connect_mailbox(cniic_comm *cm, const uint8_t *handles) {
    if (!cm || !cm->m || !handles) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(comm_ctx(cm->m));
    LOCK(c);
    return comm_connect_mailbox(cm->m, handles);
}

void cniic_comm_destroy(cniic_comm *cm) {
    if (!cm) return;
    if (cm->m) {
        Ctx *c = comm_ctx(cm->m);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        comm_destroy(cm->m);
    }
    delete cm;
}

int32_t cniic_comm_all_reduce(cniic_comm *cm, void *buf_dev, uint64_t count, int32_t elem_bytes) {
    if (!cm || !cm->m || !buf_dev) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(comm_ctx(cm->m));
    LOCK(c);
    if (!is_device_ptr(buf_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "comm_all_reduce: buffer must be device memory");
    const int kind = elem_bytes == 1 ? 0 : elem_bytes == 4 ? 1 : elem_bytes == 8 ? 2 : -1;
    if (kind < 0) return c->fail(CNIIC_ERR_BAD_ARG, "comm_all_reduce: elements of 1, 4 or 8 bytes (unsigned sum)");
    return comm_all_reduce(cm->m, buf_dev, count, kind);
}

int32_t cniic_comm_set_timeout(cniic_comm *cm, uint64_t milliseconds) {
    if (!cm || !cm->m) return CNIIC_ERR_BAD_ARG;
    comm_set_timeout_ms(cm->m, milliseconds);
    return CNIIC_OK;
}

int32_t cniic_cc_run(cniic_cc *cc, cniic_comm *cm, cniic_kmeans_stats *stats) {
    if (!cc) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->km) return c->fail(CNIIC_ERR_BAD_ARG, "the session has no K-means state yet (cniic_cc_image_create comes first)");
    if (cm && cm->m && comm_ctx(cm->m) != cc->c) return c->fail(CNIIC_ERR_BAD_ARG, "cc_run: communicator of another context");
    host_trace().mark("cc_run: enter");
    CNIIC_TRY(km_rgbw_run(cc->s->km, cm ? cm->m : nullptr));
    host_trace().mark("cc_run: the loop");
    if (stats && !km_rgbw_run_stats(cc->s->km, stats)) {  // (the loop's own last look at the state; no wait for the launches past convergence)
        uint32_t d = 0;
        CNIIC_TRY(km_rgbw_poll(cc->s->km, stats, &d));
    }
    return CNIIC_OK;
}

int32_t cniic_cc_partials(cniic_cc *cc, void **dev_ptr) {
    if (!cc || !dev_ptr || !cc->s->km) return CNIIC_ERR_BAD_ARG;
    *dev_ptr = km_rgbw_partials_dev(cc->s->km);
    return CNIIC_OK;
}

int32_t cniic_cc_export_labels(cniic_cc *cc, void *dst_dev) {
    if (!cc || !dst_dev) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->km) return c->fail(CNIIC_ERR_BAD_ARG, "the session has no K-means state yet (cniic_cc_image_create comes first)");
    return km_rgbw_export_labels(cc->s->km, dst_dev);
}

int32_t cniic_cc_import_labels(cniic_cc *cc, const void *src_dev) {
    if (!cc || !src_dev) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->km) return c->fail(CNIIC_ERR_BAD_ARG, "the session has no K-means state yet (cniic_cc_image_create comes first)");
    return km_rgbw_import_labels(cc->s->km, src_dev);
}

int32_t cniic_cc_finish(cniic_cc *cc, const uint8_t *rgb, uint32_t w, uint32_t h, const uint32_t *local_table_dev, uint8_t *out,
                        uint64_t cap, uint64_t *len, cniic_kmeans_stats *stats) {
    if (!cc) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->km) return c->fail(CNIIC_ERR_BAD_ARG, "the session has no K-means state yet (cniic_cc_image_create comes first)");
    if (!rgb || !out || !len) return c->fail(CNIIC_ERR_BAD_ARG, "cc_finish: null argument");
    if (local_table_dev && !is_device_ptr(local_table_dev)) return c->fail(CNIIC_ERR_BAD_ARG, "cc_finish: local table must be device memory");
    In<uint8_t> in;
    CNIIC_TRY(in.bind(c, rgb, (uint64_t)w * h * 3));
    return cc_finish(cc->s, in.d, w, h, local_table_dev, out, cap, len, stats);
}

int32_t cniic_cc_finish_frames(cniic_cc *cc, const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t frames, uint8_t *out, uint64_t stride,
                               uint64_t *lens, cniic_kmeans_stats *stats) {
    if (!cc) return CNIIC_ERR_BAD_ARG;
    cniic_ctx *c = static_cast<cniic_ctx *>(cc->c);
    LOCK(c);
    if (!cc->s->km) return c->fail(CNIIC_ERR_BAD_ARG, "the session has no K-means state yet (cniic_cc_image_create comes first)");
    if (!rgb || !out || !lens || !frames) return c->fail(CNIIC_ERR_BAD_ARG, "cc_finish_frames: null argument");
    In<uint8_t> in;
    CNIIC_TRY(in.bind(c, rgb, (uint64_t)w * h * 3 * frames));
    return cc_finish_frames(cc->s, in.d, w, h, frames, out, stride, lens, stats);
}

void cniic_cc_destroy(cniic_cc *cc) {
    if (!cc) return;
    {
        std::lock_guard<std::mutex> lk(cc->c->mu);
        (void)hipSetDevice(cc->c->device);
        (void)hipStreamSynchronize(cc->c->stream);
        PoolScope ps(&cc->c->pool);
        delete cc->s;
    }
    delete cc;
}

// ------------------------------------------------------------------ remap
int32_t cniic_remap_rgb(cniic_ctx *c, const uint8_t *rgb, uint64_t npx, const uint32_t *keys, const uint32_t *labels, uint64_t U,
                        const uint8_t *centroids, uint32_t K, uint8_t *out_rgb) {
    LOCK(c);
    c->ktimes.clear();
    if (!npx) return CNIIC_OK;
    if (!rgb || !keys || !labels || !centroids || !out_rgb) return c->fail(CNIIC_ERR_BAD_ARG, "remap_rgb: null argument");
    In<uint8_t> in;
    In<uint32_t> k, l;
    CNIIC_TRY(in.bind(c, rgb, npx * 3));
    CNIIC_TRY(k.bind(c, keys, U));
    CNIIC_TRY(l.bind(c, labels, U));
    std::vector<uint8_t> cent(3 * (size_t)K);
    CNIIC_TRY(from_caller(c, cent.data(), centroids, cent.size()));
    std::vector<uint32_t> ck(K);
    for (uint32_t i = 0; i < K; i++) ck[i] = ((uint32_t)cent[3 * i] << 16) | ((uint32_t)cent[3 * i + 1] << 8) | cent[3 * i + 2];
    DevBuf ck_d, lut_d;
    CNIIC_HIP_TRY(c, ck_d.alloc((uint64_t)K * 4));
    CNIIC_HIP_TRY(c, lut_d.alloc(U * 4));
    CNIIC_HIP_TRY(c, hipMemcpyAsync(ck_d.p, ck.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
    uint32_t *table = nullptr;
    CNIIC_TRY(dense_table(c, 24, &table));
    CNIIC_TRY(rank_from_keys(c, k.d, U, table));
    CNIIC_TRY(label_lut(c, l.d, U, ck_d.as<uint32_t>(), lut_d.as<uint32_t>()));
    Out<uint8_t> o;
    CNIIC_TRY(o.bind(c, out_rgb, npx * 3));
    {
        ScopedKernelTimer t(c, "remap_rgb");
        CNIIC_TRY(remap_rgb(c, in.d, npx, table, lut_d.as<uint32_t>(), o.d));
        t.stop(1);
    }
    CNIIC_TRY(o.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

// ------------------------------------------------------------------ Hilbert + delta
int32_t cniic_hilbert_xy(cniic_ctx *c, uint32_t w, uint32_t h, uint32_t *xy) {
    LOCK(c);
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    if (!xy) return c->fail(CNIIC_ERR_BAD_ARG, "hilbert_xy: null output");
    Out<uint32_t> o;
    CNIIC_TRY(o.bind(c, xy, 2 * n));
    CNIIC_TRY(hilbert_xy(c, w, h, o.d));
    CNIIC_TRY(o.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

int32_t cniic_hilbert_linearize(cniic_ctx *c, const uint8_t *rgb, uint32_t w, uint32_t h, uint8_t *out_rgb) {
    LOCK(c);
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    if (!rgb || !out_rgb) return c->fail(CNIIC_ERR_BAD_ARG, "hilbert_linearize: null argument");
    In<uint8_t> in;
    Out<uint8_t> o;
    CNIIC_TRY(in.bind(c, rgb, 3 * n));
    CNIIC_TRY(o.bind(c, out_rgb, 3 * n));
    CNIIC_TRY(hilbert_linearize(c, in.d, w, h, o.d));
    CNIIC_TRY(o.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

int32_t cniic_hilbert_delta(cniic_ctx *c, const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t *syms) {
    LOCK(c);
    c->ktimes.clear();
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    if (!rgb || !syms) return c->fail(CNIIC_ERR_BAD_ARG, "hilbert_delta: null argument");
    In<uint8_t> in;
    Out<uint32_t> o;
    CNIIC_TRY(in.bind(c, rgb, 3 * n));
    CNIIC_TRY(o.bind(c, syms, n));
    CNIIC_TRY(hilbert_delta(c, in.d, w, h, o.d, nullptr));
    CNIIC_TRY(o.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

int32_t cniic_hilbert_delta_hist(cniic_ctx *c, const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t *keys, uint64_t *counts,
                                 uint64_t cap, uint64_t *n_unique, uint32_t *syms) {
    LOCK(c);
    c->ktimes.clear();
    const uint64_t n = (uint64_t)w * h;
    if (n_unique) *n_unique = 0;
    if (!n) return CNIIC_OK;
    if (!rgb) return c->fail(CNIIC_ERR_BAD_ARG, "hilbert_delta_hist: null image");
    In<uint8_t> in;
    Out<uint32_t> so;
    CNIIC_TRY(in.bind(c, rgb, 3 * n));
    CNIIC_TRY(so.bind(c, syms, n));
    uint32_t *table = nullptr;
    CNIIC_TRY(dense_table(c, 27, &table));
    CNIIC_TRY(hilbert_delta(c, in.d, w, h, so.d, table));
    CNIIC_TRY(so.finish(c));
    return hist_common(c, 27, table, keys, counts, cap, n_unique);
}

// ------------------------------------------------------------------ H2
int32_t cniic_huf_encode_all(cniic_ctx *c, int32_t sym_kind, const uint32_t *syms, uint64_t n, uint8_t *out, uint64_t cap,
                             uint64_t *len) {
    LOCK(c);
    c->ktimes.clear();
    if (sym_kind != CNIIC_SYM_RGB && sym_kind != CNIIC_SYM_SIGNED) return c->fail(CNIIC_ERR_BAD_ARG, "huf_encode_all: bad symbol kind");
    if (!syms || !len) return c->fail(CNIIC_ERR_BAD_ARG, "huf_encode_all: null argument");
    if (n >= (1ull << 32)) return c->fail(CNIIC_ERR_BAD_ARG, "huf_encode_all: too many symbols");
    In<uint32_t> in;
    CNIIC_TRY(in.bind(c, syms, n));
    uint32_t *table = nullptr;
    CNIIC_TRY(dense_table(c, sym_kind == CNIIC_SYM_RGB ? 24 : 27, &table));
    std::vector<uint8_t> header;
    return huf_encode_all_dev(c, sym_kind, nullptr, const_cast<uint32_t *>(in.d), false, n, table, false, header, out, cap, len);
}

int32_t cniic_huf_size(int32_t sym_kind, const uint64_t *counts, uint64_t n, uint64_t *nbytes) {
    if (!counts || !nbytes || n == 0 || huff_symbol_size(sym_kind) < 0) return CNIIC_ERR_BAD_ARG;
    HuffTree t;
    std::vector<uint8_t> len;
    std::vector<uint64_t> code;
    if (!huff_build_tree(counts, n, t) || !huff_codes(t, len, code)) return CNIIC_ERR_BAD_ARG;
    *nbytes = huff_stream_size(sym_kind, counts, len.data(), n);
    return CNIIC_OK;
}

// ------------------------------------------------------------------ Codec trait
int32_t cniic_codec_parse(const char *expr, int32_t *kind, uint32_t *arg) {
    CodecDesc d;
    if (!parse_codec(expr, &d)) return CNIIC_ERR_BAD_ARG;
    if (kind) *kind = d.kind;
    if (arg) *arg = d.arg;
    return CNIIC_OK;
}

int32_t cniic_codec_name(const char *expr, char *buf, uint64_t cap) {
    CodecDesc d;
    if (!parse_codec(expr, &d) || !buf) return CNIIC_ERR_BAD_ARG;
    std::string s = codec_name(d);
    if (s.size() + 1 > cap) return CNIIC_ERR_CAPACITY;
    memcpy(buf, s.c_str(), s.size() + 1);
    return CNIIC_OK;
}

int32_t cniic_codec_is_lossless(const char *expr) {
    CodecDesc d;
    if (!parse_codec(expr, &d)) return CNIIC_ERR_BAD_ARG;
    return codec_is_lossless(d) ? 1 : 0;
}

int32_t cniic_codec_encode(cniic_ctx *c, const char *expr, const uint8_t *rgb, uint32_t w, uint32_t h, uint8_t *out, uint64_t cap,
                           uint64_t *len, cniic_kmeans_stats *stats) {
    LOCK(c);
    c->ktimes.clear();
    CodecDesc d;
    if (!parse_codec(expr, &d)) return c->fail(CNIIC_ERR_BAD_ARG, "Malformed codec argument: %s", expr ? expr : "(null)");
    if (!len || (!rgb && (uint64_t)w * h) || !out) return c->fail(CNIIC_ERR_BAD_ARG, "codec_encode: null argument");
    In<uint8_t> in;
    CNIIC_TRY(in.bind(c, rgb, (uint64_t)w * h * 3));
    return codec_encode(c, d, in.d, w, h, nullptr, out, cap, len, stats);
}

int32_t cniic_codec_encode_opts(cniic_ctx *c, const char *expr, const cniic_kmeans_opts *opts, const uint8_t *rgb, uint32_t w,
                                uint32_t h, uint8_t *out, uint64_t cap, uint64_t *len, cniic_kmeans_stats *stats) {
    LOCK(c);
    c->ktimes.clear();
    CodecDesc d;
    if (!parse_codec(expr, &d)) return c->fail(CNIIC_ERR_BAD_ARG, "Malformed codec argument: %s", expr ? expr : "(null)");
    if (!len || (!rgb && (uint64_t)w * h) || !out) return c->fail(CNIIC_ERR_BAD_ARG, "codec_encode: null argument");
    In<uint8_t> in;
    CNIIC_TRY(in.bind(c, rgb, (uint64_t)w * h * 3));
    return codec_encode(c, d, in.d, w, h, opts, out, cap, len, stats);
}

int32_t cniic_codec_encode_batch(cniic_ctx *c, const char *expr, const cniic_kmeans_opts *opts, const uint8_t *rgb, uint32_t w, uint32_t h,
                                 uint32_t frames, uint8_t *out, uint64_t stride, uint64_t *lens, int32_t *rcs, cniic_kmeans_stats *stats) {
    LOCK(c);
    c->ktimes.clear();
    CodecDesc d;
    if (!parse_codec(expr, &d)) return c->fail(CNIIC_ERR_BAD_ARG, "Malformed codec argument: %s", expr ? expr : "(null)");
    if (!frames) return CNIIC_OK;
    if (!rgb || !out || !lens) return c->fail(CNIIC_ERR_BAD_ARG, "codec_encode_batch: null argument");
    const uint64_t img_bytes = (uint64_t)w * h * 3;
    const uint32_t S = (uint32_t)std::min<uint64_t>(frames, std::max<uint64_t>(1, c->opt(CNIIC_OPT_BATCH_STREAMS, nullptr, 8)));
    while (c->batch_workers.size() < S) {
        cniic_ctx *wk = nullptr;
        const int32_t rc = cniic_ctx_create(c->device, nullptr, &wk);
        if (rc != CNIIC_OK) return c->fail(rc, "codec_encode_batch: cannot create worker context %zu", c->batch_workers.size());
        c->batch_workers.push_back(wk);
    }
    for (uint32_t i = 0; i < S; i++) {  // the workers take this context's route switches
        cniic_ctx *wk = static_cast<cniic_ctx *>(c->batch_workers[i]);
        memcpy(wk->opt_val, c->opt_val, sizeof c->opt_val);
        wk->opt_set = c->opt_set;
        // several images in flight: half-size K-means grids, so that two images' launches are resident together (measured on 64 frames
        // 1920 x 1080 with 8 workers: 768 blocks 0.885 ms per frame, 384: 0.729, 192: 0.80, 96: 1.17)
        if (S > 1 && !((c->opt_set >> CNIIC_OPT_KM_MAX_BLOCKS) & 1u) && !getenv("CNIIC_KM_MAX_BLOCKS")) { wk->opt_val[CNIIC_OPT_KM_MAX_BLOCKS] = 384; wk->opt_set |= 1u << CNIIC_OPT_KM_MAX_BLOCKS; }
        wk->ps_div = S;                        // ... and the persistent K-means launch an S-th of the CUs, so that S of them are resident side by side
        wk->scan_xy.release();                 // ... and its injected scan, as a view of this context's table
        wk->scan_w = wk->scan_h = 0;
        if (c->scan_xy.p) { wk->scan_xy.view(c->scan_xy.p, c->scan_xy.bytes); wk->scan_w = c->scan_w; wk->scan_h = c->scan_h; }
    }
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));   // whatever produced the images on this context's stream is done
    std::atomic<uint32_t> next{0};
    std::vector<int32_t> status(frames, CNIIC_OK);
    auto run = [&](uint32_t i) {
        cniic_ctx *wk = static_cast<cniic_ctx *>(c->batch_workers[i]);
        for (;;) {
            const uint32_t f = next.fetch_add(1);
            if (f >= frames) break;
            cniic_kmeans_stats st{};
            status[f] = cniic_codec_encode_opts(wk, expr, opts, rgb + (uint64_t)f * img_bytes, w, h, out + (uint64_t)f * stride, stride, &lens[f], &st);
            if (stats) stats[f] = st;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t i = 1; i < S; i++) th.emplace_back(run, i);
    run(0);
    for (auto &t : th) t.join();
    int32_t first = CNIIC_OK;
    for (uint32_t f = 0; f < frames; f++) {
        if (rcs) rcs[f] = status[f];
        if (status[f] != CNIIC_OK && first == CNIIC_OK) first = status[f];
    }
    if (first != CNIIC_OK) {
        for (uint32_t i = 0; i < S; i++) {
            cniic_ctx *wk = static_cast<cniic_ctx *>(c->batch_workers[i]);
            if (!wk->err.empty()) { c->err = wk->err; break; }
        }
    }
    return first;
}

int32_t cniic_codec_decode(cniic_ctx *c, const char *expr, const uint8_t *bytes, uint64_t n, uint8_t *rgb, uint64_t cap, uint32_t *w,
                           uint32_t *h) {
    LOCK(c);
    c->ktimes.clear();
    CodecDesc d;
    if (!parse_codec(expr, &d)) return c->fail(CNIIC_ERR_BAD_ARG, "Malformed codec argument: %s", expr ? expr : "(null)");
    if (!bytes || !w || !h) return c->fail(CNIIC_ERR_BAD_ARG, "codec_decode: null argument");
    return codec_decode(c, d, bytes, n, rgb, cap, w, h);  // (the stream may be in host memory or in HBM)
}

int32_t cniic_mse(cniic_ctx *c, const uint8_t *a, const uint8_t *b, uint64_t npx, double *mse) {
    LOCK(c);
    if (!mse || ((!a || !b) && npx)) return c->fail(CNIIC_ERR_BAD_ARG, "mse: null argument");
    In<uint8_t> ia, ib;
    CNIIC_TRY(ia.bind(c, a, npx * 3));
    CNIIC_TRY(ib.bind(c, b, npx * 3));
    return mse_rgb(c, ia.d, ib.d, npx, mse);
}

int32_t cniic_synth_image(cniic_ctx *c, int32_t kind, uint64_t seed, uint32_t w, uint32_t h, uint8_t *rgb) {
    LOCK(c);
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    if (!rgb) return c->fail(CNIIC_ERR_BAD_ARG, "synth_image: null output");
    Out<uint8_t> o;
    CNIIC_TRY(o.bind(c, rgb, 3 * n));
    CNIIC_TRY(synth_image(c, kind, seed, w, h, o.d));
    CNIIC_TRY(o.finish(c));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

}  // extern "C"
