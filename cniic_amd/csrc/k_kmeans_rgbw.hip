// k_kmeans_rgbw.hip -- kmeans::cluster::<ColorCount> on gfx950
// (reference: src/kmeans.rs:21-143, 330-416 with Point = ColorCount, src/codec/clusterc.rs:68-114,
//  distance src/geom.rs:8-24).
//
// Points are the image's DISTINCT colours with pixel-count weights (clusterc.rs:21-28), one
// packed u32 key + one u32 weight + one label byte each: 4 + 4 + 1 read + 1 written = 10 B per
// colour per iteration ("dedup form").
//
// Assign is exact Lloyd under the reference's rules (stay unless another centroid is STRICTLY
// closer, kmeans.rs:350-378; lowest id among equidistant minima).  Because every comparison is
// between distances from the SAME point, |p|^2 cancels and the kernel maximises
//     g_k = 2 p.c_k - |c_k|^2          (p.c_k = one v_dot4_u32_u8)
// packed with the cluster id into one u32 so that arg-max + lowest-id tie-break is a single
// v_max_u32:  key_k = ((g_k + BIAS) << IDBITS) | (IDMASK - k)  =  (dot << (IDBITS+1)) + const_k.
// Three VALU instructions per (colour, centroid): dot4, lshl_add, max.  Centroid constants are
// wave-uniform and come in through scalar loads.  MFMA is deliberately unused: the inner
// dimension is 3 and the arg-max dominates.
//
// Centroid update is exact u64 integer arithmetic (clusterc.rs:92-105): per-block LDS
// accumulators (ds_add_u64) -> per-block slab in HBM -> parallel slab reduction -> K-thread
// finalize (truncating division, empty-cluster reseed).  Integer sums make the result
// independent of block count, launch order and GPU count.
#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

constexpr uint32_t kBias = 1u << 18;  // > max |c|^2 = 195075
constexpr int kPPT = 8;               // colours per thread per sweep
constexpr int kAssignThreads = 256;
constexpr uint32_t kMaxBlocks = 512;

struct KmRgbwState {
    Ctx *c = nullptr;
    uint64_t U = 0, lo = 0, hi = 0, seed = 0, max_iters = 0;
    uint32_t K = 0, Kpad = 0, idbits = 8, nblocks = 1;
    bool wide = false;  // u16 labels
    const uint32_t *keys = nullptr, *weight = nullptr;  // device, full list [0,U)
    DevBuf labels, cconst, slabs, partials_own, dstate, cent, members_last, wsum_last;
    uint64_t *partials = nullptr;  // device: 5K+1 words
    bool sharded = false;
};

__device__ __forceinline__ uint32_t dot4u8(uint32_t a, uint32_t b, uint32_t acc) {
#if __has_builtin(__builtin_amdgcn_udot4)
    return __builtin_amdgcn_udot4(a, b, acc, false);
#else
    return acc + (a & 255) * (b & 255) + ((a >> 8) & 255) * ((b >> 8) & 255) + ((a >> 16) & 255) * ((b >> 16) & 255) +
           (a >> 24) * (b >> 24);
#endif
}

// (packed centroid key, const term) for cluster k
__device__ __forceinline__ uint2 make_cconst(uint32_t ckey, uint32_t k, uint32_t idbits) {
    uint32_t h = dot4u8(ckey, ckey, 0);
    uint32_t idmask = (1u << idbits) - 1;
    return make_uint2(ckey, ((kBias - h) << idbits) | (idmask - k));
}

template <typename LabelT>
__global__ void k_rgbw_init(const uint32_t *__restrict__ keys, uint64_t U, uint64_t lo, uint64_t hi, uint32_t K,
                            uint32_t Kpad, uint32_t idbits, LabelT *__restrict__ labels,
                            uint2 *__restrict__ cconst, uint32_t *__restrict__ cent) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = lo + tid; i < hi; i += stride) labels[i - lo] = (LabelT)init_label(i, U, K);  // kmeans.rs:61-78
    if (tid < Kpad) {
        uint32_t k = (uint32_t)tid;
        if (k < K) {
            // init_centroids (kmeans.rs:101-108): first element of chunk k
            uint64_t ppc = U / K;
            uint64_t first = (k < K - 1) ? U - ((uint64_t)k + 1) * ppc : 0;
            uint32_t ck = keys[first];
            cent[k] = ck;
            cconst[k] = make_cconst(ck, k, idbits);
        } else {
            cconst[k] = make_uint2(0u, 0u);  // padding: key 0 never wins
        }
    }
}

// ---------------------------------------------------------------- assign + partial sums
// slab row layout (u64 words): [3k+d] sum of channel d * weight, [3K+k] sum of weights,
// [4K+k] member count, [5K] moved count.
template <typename LabelT, int IDBITS>
__global__ __launch_bounds__(kAssignThreads) void k_rgbw_assign(
    const uint32_t *__restrict__ keys, const uint32_t *__restrict__ weight, uint64_t lo, uint64_t hi,
    uint32_t K, uint32_t Kpad, const uint2 *__restrict__ cconst, LabelT *__restrict__ labels,
    uint64_t *__restrict__ slabs, const KmDevState *__restrict__ st) {
    extern __shared__ __align__(16) unsigned long long lds[];  // [5K] accumulators, then uint2[K] table
    if (st->done) return;
    unsigned long long *acc = lds;
    uint2 *tab = reinterpret_cast<uint2 *>(lds + 5 * (size_t)K);
    for (uint32_t i = threadIdx.x; i < 5 * K; i += kAssignThreads) acc[i] = 0ull;
    for (uint32_t i = threadIdx.x; i < K; i += kAssignThreads) tab[i] = cconst[i];
    __syncthreads();

    constexpr uint32_t IDMASK = (1u << IDBITS) - 1;
    const uint64_t n = hi - lo;
    const uint64_t sweep = (uint64_t)gridDim.x * kAssignThreads * kPPT;
    uint32_t moved = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * kAssignThreads * kPPT; base < n; base += sweep) {
        uint32_t p[kPPT], best[kPPT];
        bool valid[kPPT];
#pragma unroll
        for (int j = 0; j < kPPT; j++) {
            uint64_t i = base + (uint64_t)j * kAssignThreads + threadIdx.x;
            valid[j] = i < n;
            p[j] = valid[j] ? keys[lo + i] : 0u;
            best[j] = 0u;
        }
        // brute-force sweep over the (padded) centroid table; k is wave-uniform -> scalar loads
#pragma unroll 4
        for (uint32_t k = 0; k < Kpad; k++) {
            const uint2 cc = cconst[k];
#pragma unroll
            for (int j = 0; j < kPPT; j++) {
                uint32_t key = (dot4u8(p[j], cc.x, 0) << (IDBITS + 1)) + cc.y;
                best[j] = max(best[j], key);
            }
        }
#pragma unroll
        for (int j = 0; j < kPPT; j++) {
            if (!valid[j]) continue;
            uint64_t i = base + (uint64_t)j * kAssignThreads + threadIdx.x;
            uint32_t cur = labels[i];
            uint2 cc = tab[cur];
            uint32_t kcur = (dot4u8(p[j], cc.x, 0) << (IDBITS + 1)) + cc.y;
            uint32_t nl = cur;
            if ((best[j] >> IDBITS) > (kcur >> IDBITS)) {  // strictly closer (kmeans.rs:375)
                nl = IDMASK - (best[j] & IDMASK);
                labels[i] = (LabelT)nl;
                moved++;
            }
            uint64_t w = weight[lo + i];
            uint32_t r = (p[j] >> 16) & 255, g = (p[j] >> 8) & 255, b = p[j] & 255;
            atomicAdd(&acc[3 * nl + 0], (unsigned long long)(r * w));  // clusterc.rs:92-98
            atomicAdd(&acc[3 * nl + 1], (unsigned long long)(g * w));
            atomicAdd(&acc[3 * nl + 2], (unsigned long long)(b * w));
            atomicAdd(&acc[3 * K + nl], (unsigned long long)w);
            atomicAdd(&acc[4 * K + nl], 1ull);
        }
    }
    moved = block_reduce_sum<kAssignThreads>(moved);
    __syncthreads();
    uint64_t *row = slabs + (size_t)blockIdx.x * (5 * (size_t)K + 1);
    for (uint32_t i = threadIdx.x; i < 5 * K; i += kAssignThreads) row[i] = acc[i];
    if (threadIdx.x == 0) row[5 * (size_t)K] = moved;
}

// ---------------------------------------------------------------- slab reduction
// grid (ceil(W/64), R): each block sums a stripe of rows for 64 columns, then one atomic per column.
__global__ __launch_bounds__(256) void k_slab_reduce(const uint64_t *__restrict__ slabs, uint32_t nrows, uint32_t W,
                                                     uint64_t *__restrict__ partials,
                                                     const KmDevState *__restrict__ st) {
    if (st->done) return;
    __shared__ unsigned long long sh[4][64];
    const uint32_t col = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t rl = threadIdx.x >> 6;  // 0..3
    unsigned long long s = 0;
    if (col < W)
        for (uint32_t r = blockIdx.y * 4 + rl; r < nrows; r += gridDim.y * 4) s += slabs[(size_t)r * W + col];
    sh[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < W) {
        s = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        if (s) atomicAdd(reinterpret_cast<unsigned long long *>(partials) + col, s);
    }
}

// ---------------------------------------------------------------- centroid update
// Point::mean for ColorCount (clusterc.rs:83-113) + empty-cluster reseed (kmeans.rs:110-137).
__global__ __launch_bounds__(256) void k_rgbw_update(uint64_t *__restrict__ partials, const uint32_t *__restrict__ keys,
                                                     uint64_t U, uint32_t K, uint32_t idbits, uint64_t seed,
                                                     uint64_t max_iters, uint2 *__restrict__ cconst,
                                                     uint32_t *__restrict__ cent, uint64_t *__restrict__ members_out,
                                                     uint64_t *__restrict__ wsum_out,
                                                     KmDevState *__restrict__ st) {
    if (st->done) return;
    __shared__ uint32_t s_reseed, s_active;
    if (threadIdx.x == 0) { s_reseed = 0; s_active = 0; }
    __syncthreads();
    const uint64_t iter = st->iter;
    for (uint32_t k = threadIdx.x; k < K; k += blockDim.x) {
        uint64_t members = partials[4 * (size_t)K + k];
        members_out[k] = members;
        wsum_out[k] = partials[3 * (size_t)K + k];
        uint32_t ck;
        if (members == 0) {
            ck = keys[reseed_index(seed, iter, k, U)];  // fake_clone of the stolen point
            atomicAdd(&s_reseed, 1u);
        } else {
            uint64_t w = partials[3 * (size_t)K + k];
            uint32_t r = (uint32_t)(partials[3 * (size_t)k + 0] / w) & 255;
            uint32_t g = (uint32_t)(partials[3 * (size_t)k + 1] / w) & 255;
            uint32_t b = (uint32_t)(partials[3 * (size_t)k + 2] / w) & 255;
            ck = (r << 16) | (g << 8) | b;
            atomicAdd(&s_active, 1u);
        }
        cent[k] = ck;
        cconst[k] = make_cconst(ck, k, idbits);
    }
    __syncthreads();
    const uint64_t changed = partials[5 * (size_t)K];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 5 * K + 1; i += blockDim.x) partials[i] = 0;  // ready for the next reduce
    if (threadIdx.x == 0) {
        st->changed_ring[iter % kHistRing] = changed;
        st->moved_last = changed;
        st->reseeds += s_reseed;
        st->active = s_active;
        st->iter = iter + 1;
        if (changed == 0 || (max_iters && iter + 1 >= max_iters)) st->done = 1;
    }
}

template <typename LabelT>
__global__ void k_widen_labels(const LabelT *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
template <typename LabelT>
__global__ void k_narrow_labels(const uint32_t *__restrict__ in, LabelT *__restrict__ out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (LabelT)in[i];
}
__global__ void k_set_cconst(const uint32_t *__restrict__ cent, uint32_t K, uint32_t Kpad, uint32_t idbits,
                             uint2 *__restrict__ cconst) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) cconst[k] = make_cconst(cent[k], k, idbits);
    else if (k < Kpad) cconst[k] = make_uint2(0u, 0u);
}

// =========================================================================== host side
static inline uint32_t grid_1d(uint64_t n, uint32_t cap = 2048) {
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, 256), 1), cap);
}

int km_rgbw_create(Ctx *c, const uint32_t *keys_d, const uint32_t *weight_d, uint64_t U, uint64_t lo,
                   uint64_t hi, uint32_t K, const cniic_kmeans_opts *opts, void *partials_dev,
                   KmRgbwState **out) {
    if (K == 0 || U == 0 || lo > hi || hi > U) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_rgbw: bad sizes");
    if (U / K == 0) return c->fail(CNIIC_ERR_TOO_FEW_POINTS, "kmeans: %llu points for %u clusters (src/kmeans.rs:68)",
                                   (unsigned long long)U, K);
    if (K > 2048) return c->fail(CNIIC_ERR_UNSUPPORTED, "kmeans_rgbw: K=%u > 2048 not supported", K);
    if (U >= (1ull << 32)) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_rgbw: too many points");
    auto *s = new KmRgbwState();
    s->c = c; s->U = U; s->lo = lo; s->hi = hi; s->K = K;
    s->Kpad = (K + 3) & ~3u;
    s->wide = K > 256;
    s->idbits = s->wide ? 12 : 8;
    s->seed = (opts && opts->seed) ? opts->seed : kDefaultSeed;
    s->max_iters = opts ? opts->max_iters : 0;
    s->keys = keys_d; s->weight = weight_d;
    s->sharded = !(lo == 0 && hi == U);
    const uint64_t n = hi - lo;
    s->nblocks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(n, (uint64_t)kAssignThreads * kPPT), 1), kMaxBlocks);
    const uint64_t W = 5 * (uint64_t)K + 1;
    hipError_t e = hipSuccess;
    if ((e = s->labels.alloc(std::max<uint64_t>(n, 1) * (s->wide ? 2 : 1))) != hipSuccess ||
        (e = s->cconst.alloc((uint64_t)s->Kpad * 8)) != hipSuccess ||
        (e = s->cent.alloc((uint64_t)K * 4)) != hipSuccess ||
        (e = s->members_last.alloc((uint64_t)K * 8)) != hipSuccess ||
        (e = s->wsum_last.alloc((uint64_t)K * 8)) != hipSuccess ||
        (e = s->slabs.alloc((uint64_t)s->nblocks * W * 8)) != hipSuccess ||
        (e = s->dstate.alloc(sizeof(KmDevState))) != hipSuccess) {
        delete s;
        return c->fail(CNIIC_ERR_HIP, "kmeans_rgbw: hipMalloc failed: %s", hipGetErrorString(e));
    }
    if (partials_dev) s->partials = reinterpret_cast<uint64_t *>(partials_dev);
    else {
        if ((e = s->partials_own.alloc(W * 8)) != hipSuccess) { delete s; return c->fail(CNIIC_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
        s->partials = s->partials_own.as<uint64_t>();
    }
    (void)hipMemsetAsync(s->partials, 0, W * 8, c->stream);
    (void)hipMemsetAsync(s->dstate.p, 0, sizeof(KmDevState), c->stream);
    uint32_t g = grid_1d(std::max<uint64_t>(n, s->Kpad));
    if (s->wide)
        hipLaunchKernelGGL(k_rgbw_init<uint16_t>, dim3(g), dim3(256), 0, c->stream, keys_d, U, lo, hi, K, s->Kpad,
                           s->idbits, s->labels.as<uint16_t>(), s->cconst.as<uint2>(), s->cent.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_rgbw_init<uint8_t>, dim3(g), dim3(256), 0, c->stream, keys_d, U, lo, hi, K, s->Kpad,
                           s->idbits, s->labels.as<uint8_t>(), s->cconst.as<uint2>(), s->cent.as<uint32_t>());
    if ((e = hipGetLastError()) != hipSuccess) { delete s; return c->fail(CNIIC_ERR_HIP, "init launch: %s", hipGetErrorString(e)); }
    *out = s;
    return CNIIC_OK;
}

void km_rgbw_destroy(KmRgbwState *s) { delete s; }

int km_rgbw_set_state(KmRgbwState *s, const uint8_t *centroids_h, const uint32_t *labels_d_u32) {
    Ctx *c = s->c;
    std::vector<uint32_t> ck(s->K);
    for (uint32_t k = 0; k < s->K; k++)
        ck[k] = ((uint32_t)centroids_h[3 * k] << 16) | ((uint32_t)centroids_h[3 * k + 1] << 8) | centroids_h[3 * k + 2];
    CNIIC_HIP_TRY(c, hipMemcpyAsync(s->cent.p, ck.data(), (size_t)s->K * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_set_cconst, dim3(ceil_div(s->Kpad, 256)), dim3(256), 0, c->stream, s->cent.as<uint32_t>(), s->K,
                       s->Kpad, s->idbits, s->cconst.as<uint2>());
    const uint64_t n = s->hi - s->lo;
    if (n) {
        if (s->wide)
            hipLaunchKernelGGL(k_narrow_labels<uint16_t>, dim3(grid_1d(n)), dim3(256), 0, c->stream, labels_d_u32, s->labels.as<uint16_t>(), n);
        else
            hipLaunchKernelGGL(k_narrow_labels<uint8_t>, dim3(grid_1d(n)), dim3(256), 0, c->stream, labels_d_u32, s->labels.as<uint8_t>(), n);
    }
    CNIIC_HIP_TRY(c, hipGetLastError());
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));  // ck is a stack-lifetime source
    return CNIIC_OK;
}

static int launch_assign(KmRgbwState *s) {
    Ctx *c = s->c;
    const size_t lds = (size_t)s->K * (5 * 8 + 8);
    const KmDevState *st = s->dstate.as<KmDevState>();
    if (s->wide)
        hipLaunchKernelGGL((k_rgbw_assign<uint16_t, 12>), dim3(s->nblocks), dim3(kAssignThreads), lds, c->stream, s->keys,
                           s->weight, s->lo, s->hi, s->K, s->Kpad, s->cconst.as<uint2>(), s->labels.as<uint16_t>(),
                           s->slabs.as<uint64_t>(), st);
    else
        hipLaunchKernelGGL((k_rgbw_assign<uint8_t, 8>), dim3(s->nblocks), dim3(kAssignThreads), lds, c->stream, s->keys,
                           s->weight, s->lo, s->hi, s->K, s->Kpad, s->cconst.as<uint2>(), s->labels.as<uint8_t>(),
                           s->slabs.as<uint64_t>(), st);
    return CNIIC_OK;
}

int km_rgbw_assign(KmRgbwState *s) {
    Ctx *c = s->c;
    launch_assign(s);
    const uint32_t W = 5 * s->K + 1;
    const uint32_t ry = std::max(1u, std::min(16u, s->nblocks / 4));
    hipLaunchKernelGGL(k_slab_reduce, dim3(ceil_div(W, 64), ry), dim3(256), 0, c->stream, s->slabs.as<uint64_t>(),
                       s->nblocks, W, s->partials, s->dstate.as<KmDevState>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int km_rgbw_update(KmRgbwState *s) {
    Ctx *c = s->c;
    hipLaunchKernelGGL(k_rgbw_update, dim3(1), dim3(256), 0, c->stream, s->partials, s->keys, s->U, s->K, s->idbits,
                       s->seed, s->max_iters, s->cconst.as<uint2>(), s->cent.as<uint32_t>(), s->members_last.as<uint64_t>(), s->wsum_last.as<uint64_t>(),
                       s->dstate.as<KmDevState>());
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

static int read_state(KmRgbwState *s, KmDevState *h) {
    Ctx *c = s->c;
    CNIIC_HIP_TRY(c, hipMemcpyAsync(h, s->dstate.p, sizeof(KmDevState), hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CNIIC_OK;
}

int km_rgbw_poll_changed(KmRgbwState *s, uint64_t *changed) {
    KmDevState h;
    CNIIC_TRY(read_state(s, &h));
    *changed = h.moved_last;
    return CNIIC_OK;
}

// Whole loop on one GPU: iterations are enqueued in batches with no host round trip inside a
// batch; kernels of iterations past convergence exit on the device-side `done` flag, so the
// result is exactly that of the reference's `while changed_assignment` loop (kmeans.rs:26-32).
int km_rgbw_run(KmRgbwState *s) {
    Ctx *c = s->c;
    const int batch = 8;
    KmDevState h;
    ScopedKernelTimer timer(c, "kmeans_rgbw_iter");
    for (;;) {
        for (int b = 0; b < batch; b++) {
            CNIIC_TRY(km_rgbw_assign(s));
            CNIIC_TRY(km_rgbw_update(s));
        }
        CNIIC_TRY(read_state(s, &h));
        if (h.done) break;
    }
    timer.stop(h.iter);
    return CNIIC_OK;
}

// Average duration of the assign kernel alone (HIP events on the ctx stream around `reps`
// back-to-back launches on the current state).  The slab outputs are overwritten; labels may move
// towards the fixed point of the current centroids (idempotent afterwards).
int km_rgbw_time_assign(KmRgbwState *s, int reps, double *ms_per_launch) {
    Ctx *c = s->c;
    launch_assign(s);  // warm-up
    CNIIC_HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; i++) launch_assign(s);
    CNIIC_HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    CNIIC_HIP_TRY(c, hipEventSynchronize(c->ev1));
    CNIIC_HIP_TRY(c, hipGetLastError());
    float ms = 0.f;
    CNIIC_HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_per_launch = (double)ms / reps;
    return CNIIC_OK;
}

int km_rgbw_partials(KmRgbwState *s, uint64_t *sums_h, uint64_t *wsum_h, uint64_t *members_h, uint64_t *changed_h) {
    Ctx *c = s->c;
    const size_t K = s->K;
    std::vector<uint64_t> p(5 * K + 1);
    CNIIC_HIP_TRY(c, hipMemcpyAsync(p.data(), s->partials, p.size() * 8, hipMemcpyDeviceToHost, c->stream));
    CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (sums_h) memcpy(sums_h, p.data(), 3 * K * 8);
    if (wsum_h) memcpy(wsum_h, p.data() + 3 * K, K * 8);
    if (members_h) memcpy(members_h, p.data() + 4 * K, K * 8);
    if (changed_h) *changed_h = p[5 * K];
    return CNIIC_OK;
}

void *km_rgbw_partials_dev(KmRgbwState *s) { return s->partials; }
const uint8_t *km_rgbw_labels8_dev(KmRgbwState *s) { return s->wide ? nullptr : s->labels.as<uint8_t>(); }
const uint16_t *km_rgbw_labels16_dev(KmRgbwState *s) { return s->wide ? s->labels.as<uint16_t>() : nullptr; }

int km_rgbw_result(KmRgbwState *s, uint8_t *centroids_h, uint32_t *labels_d_u32, uint64_t *members_h,
                   uint64_t *wsum_h, cniic_kmeans_stats *stats) {
    Ctx *c = s->c;
    const uint64_t n = s->hi - s->lo;
    if (labels_d_u32 && n) {
        if (s->wide)
            hipLaunchKernelGGL(k_widen_labels<uint16_t>, dim3(grid_1d(n)), dim3(256), 0, c->stream, s->labels.as<uint16_t>(), labels_d_u32, n);
        else
            hipLaunchKernelGGL(k_widen_labels<uint8_t>, dim3(grid_1d(n)), dim3(256), 0, c->stream, s->labels.as<uint8_t>(), labels_d_u32, n);
        CNIIC_HIP_TRY(c, hipGetLastError());
    }
    std::vector<uint32_t> ck(s->K);
    CNIIC_HIP_TRY(c, hipMemcpyAsync(ck.data(), s->cent.p, (size_t)s->K * 4, hipMemcpyDeviceToHost, c->stream));
    KmDevState h;
    CNIIC_TRY(read_state(s, &h));
    if (centroids_h)
        for (uint32_t k = 0; k < s->K; k++) {
            centroids_h[3 * k] = (uint8_t)(ck[k] >> 16); centroids_h[3 * k + 1] = (uint8_t)(ck[k] >> 8); centroids_h[3 * k + 2] = (uint8_t)ck[k];
        }
    if (members_h) {  // global member counts of the last completed iteration (after any all-reduce)
        CNIIC_HIP_TRY(c, hipMemcpy(members_h, s->members_last.p, (size_t)s->K * 8, hipMemcpyDeviceToHost));
    }
    if (wsum_h) CNIIC_HIP_TRY(c, hipMemcpy(wsum_h, s->wsum_last.p, (size_t)s->K * 8, hipMemcpyDeviceToHost));
    if (stats) {
        stats->iterations = h.iter;
        stats->moved_last = h.moved_last;
        stats->empty_reseeds = h.reseeds;
        stats->active = h.active;
        stats->pair_evals = h.iter * n * (uint64_t)s->K;
    }
    return CNIIC_OK;
}

}  // namespace cniic
