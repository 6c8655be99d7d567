// k_kmeans_wide.hip -- kmeans::cluster beyond the sizes the tuned kernels are laid out for: exact, simple, slow.
// (reference: src/kmeans.rs:21-143, 330-416; the reference takes ANY cluster count -- src/codec/clusterc.rs:116-141, 274-297 parse `\d+`
//  into a usize, src/kmeans.rs:67-68 only asks for len >= K -- and any image size.)
//
// The tuned kernels keep per-block sums and tables in LDS and coordinates in 24-bit multiplies: `cluster-colors` stopped at K = 2048,
// `voronoi` at K = 2048 and sides of 16384 ("a drop-in that returns an error where the reference returns a result").
// Here instead:
//   * k_rgbw_assign_big   ColorCount, K up to 65535 (u16 labels): every colour against every centroid, scores 2 p.c - |c|^2 in 32 bits,
//                         stay unless STRICTLY closer (kmeans.rs:375), lowest id among equals; signed deltas of the movers (full sums at
//                         iteration 0) straight into the K sums in memory, the lanes of a wave that share a cluster adding together.
//                         The state, the update kernel and the loop are k_kmeans_rgbw.hip's (launch_assign picks this kernel for K > 2048).
//   * km_xyrgb_run_wide   ColorPos, any K and any sides with w * h <= 2^32: squared distances in 64 bits (|dx|, |dy| < 2^32 squared would
//                         not fit: the reference's i64 arithmetic, src/geom.rs), u32 labels, FULL sums every iteration (no running sums
//                         to keep exact across label widths), a grid-wide update kernel with exact 64-bit division.
// Both are checked bit for bit by tests/test_wide_limits.py (K = 4096 and 5000 on small images, a 1 x 20000 strip, a 17000-wide sliver).
#include <vector>

#include "kmeans_rgbw.hpp"

namespace cniic {

// ---------------------------------------------------------------- the lanes of a wave that hold the same cluster add together
// v[0..n) of every calling lane into sums[at(i, label)]: round by round, the label of the first lane still to do, everybody who
// shares it summed (DPP), one lane adds.  Pixels of a row / colours of a cell mostly share their cluster: one or two rounds.
template <int NV, typename At>
__device__ __forceinline__ void wave_add_by_label(unsigned long long *sums, uint32_t label, const long long (&v)[NV], bool active, At at) {
    const int lane = threadIdx.x & 63;
    bool todo = active;
#pragma unroll 1
    for (;;) {
        const unsigned long long act = __ballot(todo);
        if (!act) return;
        const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)label, __builtin_ctzll(act));
        const bool same = todo && label == l;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const unsigned long long t = wave_reduce_sum64(same ? (unsigned long long)v[i] : 0ull);
            if (lane == 0 && t) atomicAdd(&sums[at(i, l)], t);
        }
        todo = todo && !same;
    }
}

// ---------------------------------------------------------------- ColorCount, K up to 65535
// partials layout (k_kmeans_rgbw.hip): [3k + d] sum of channel d x count, [3K + k] sum of counts, [4K + k] members, [5K] moved, [5K + 1] pair evaluations
__global__ __launch_bounds__(256) void k_rgbw_assign_big(const uint32_t *__restrict__ ckeys, const uint32_t *__restrict__ cweight, uint64_t U, uint32_t K,
                                                         const uint32_t *__restrict__ cent, uint16_t *__restrict__ labels, unsigned long long *__restrict__ partials,
                                                         const KmDevState *__restrict__ st) {
    if (st->done) return;
    const bool first = st->iter == 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t moved = 0;
    const uint64_t rounds = (U + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; r++) {   // (every lane runs every round: the wave reductions want all of them)
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool in = i < U;
        const uint32_t p = in ? ckeys[i] : 0u, cur = in ? (uint32_t)labels[i] : 0u;
        const int32_t pr = (int32_t)((p >> 16) & 255), pg = (int32_t)((p >> 8) & 255), pb = (int32_t)(p & 255);
        int32_t best = INT32_MIN, bestk = 0, scur = 0;
        for (uint32_t k = 0; k < K; k++) {   // (k is uniform: the centroid comes through the scalar cache)
            const uint32_t c = cent[k];
            const int32_t cr = (int32_t)((c >> 16) & 255), cg = (int32_t)((c >> 8) & 255), cb = (int32_t)(c & 255);
            const int32_t s = 2 * (pr * cr + pg * cg + pb * cb) - (cr * cr + cg * cg + cb * cb);   // |p|^2 cancels: larger = nearer
            if (s > best) { best = s; bestk = (int32_t)k; }   // strict: the lowest id among equals
            if (k == cur) scur = s;
        }
        const bool mv = in && best > scur;   // stay unless strictly closer (kmeans.rs:375)
        const uint32_t nl = mv ? (uint32_t)bestk : cur;
        if (mv) { labels[i] = (uint16_t)nl; moved++; }
        const long long w = in ? (long long)cweight[i] : 0;
        const long long v[5] = {pr * w, pg * w, pb * w, w, 1};
        auto at = [&](int q, uint32_t l) -> size_t { return q < 3 ? 3 * (size_t)l + q : q == 3 ? 3 * (size_t)K + l : 4 * (size_t)K + l; };
        if (first) wave_add_by_label<5>(partials, nl, v, in, at);
        else {
            wave_add_by_label<5>(partials, nl, v, mv, at);
            const long long nv[5] = {-v[0], -v[1], -v[2], -v[3], -1};
            wave_add_by_label<5>(partials, cur, nv, mv, at);
        }
    }
    moved = wave_reduce_sum(moved);
    if ((threadIdx.x & 63) == 0 && moved) atomicAdd(&partials[5 * (size_t)K], (unsigned long long)moved);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&partials[5 * (size_t)K + 1], (unsigned long long)U * K);
}

void launch_rgbw_assign_big(Ctx *c, const uint32_t *ckeys, const uint32_t *cweight, uint64_t U, uint32_t K, const uint32_t *cent, uint16_t *labels,
                            unsigned long long *partials, const KmDevState *st) {
    const uint32_t g = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(U, 256), 1), 4096);
    hipLaunchKernelGGL(k_rgbw_assign_big, dim3(g), dim3(256), 0, c->stream, ckeys, cweight, U, K, cent, labels, partials, st);
}

// ---------------------------------------------------------------- ColorPos, any K, any sides
struct WideCent { int32_t x, y; uint32_t col, pad; };

__global__ void k_xyw_init(const uint8_t *__restrict__ rgb, uint32_t w, uint64_t N, uint32_t K, uint32_t *__restrict__ labels, WideCent *__restrict__ cent) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = tid; i < N; i += stride) labels[i] = init_label(i, N, K);  // kmeans.rs:61-78
    for (uint64_t k = tid; k < K; k += stride) {
        const uint64_t ppc = N / K;
        const uint64_t first = (k < (uint64_t)K - 1) ? N - (k + 1) * ppc : 0;  // init_centroids kmeans.rs:101-108
        cent[k] = WideCent{(int32_t)(first % w), (int32_t)(first / w), rgb_key(rgb + 3 * first), 0u};
    }
}

// sums layout: [5k + d] d = x, y, r, g, b; [5K + k] members; [6K] moved; [6K + 1] pair evaluations
__global__ __launch_bounds__(256) void k_xyw_assign(const uint8_t *__restrict__ rgb, uint32_t w, uint64_t N, uint32_t K, const WideCent *__restrict__ cent,
                                                    uint32_t *__restrict__ labels, unsigned long long *__restrict__ sums) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t moved = 0;
    const uint64_t rounds = (N + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; r++) {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool in = i < N;
        const uint64_t ii = in ? i : 0;
        const long long x = (long long)(ii % w), y = (long long)(ii / w);
        const uint32_t p = rgb_key(rgb + 3 * ii), cur = labels[ii];
        const long long pr = (p >> 16) & 255, pg = (p >> 8) & 255, pb = p & 255;
        unsigned long long best = ~0ull, dcur = 0;
        uint32_t bestk = 0;
        for (uint32_t k = 0; k < K; k++) {
            const WideCent c = cent[k];
            const long long dx = x - c.x, dy = y - c.y, dr = pr - (long long)((c.col >> 16) & 255), dg = pg - (long long)((c.col >> 8) & 255), db = pb - (long long)(c.col & 255);
            const unsigned long long d = (unsigned long long)(dx * dx + dy * dy + dr * dr + dg * dg + db * db);   // (int64 squared distances, as the reference's)
            if (d < best) { best = d; bestk = k; }   // strict: the lowest id among equals
            if (k == cur) dcur = d;
        }
        const bool mv = in && best < dcur;   // stay unless strictly closer (kmeans.rs:375)
        const uint32_t nl = mv ? bestk : cur;
        if (mv) { labels[ii] = nl; moved++; }
        const long long v[6] = {x, y, pr, pg, pb, 1};
        wave_add_by_label<6>(sums, nl, v, in, [&](int q, uint32_t l) -> size_t { return q < 5 ? 5 * (size_t)l + q : 5 * (size_t)K + l; });
    }
    moved = wave_reduce_sum(moved);
    if ((threadIdx.x & 63) == 0 && moved) atomicAdd(&sums[6 * (size_t)K], (unsigned long long)moved);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&sums[6 * (size_t)K + 1], (unsigned long long)N * K);
}

// Point::mean for ColorPos (clusterc.rs:215-248: unweighted sums, truncating division) + empty-cluster reseed (kmeans.rs:110-137)
__global__ void k_xyw_update(unsigned long long *__restrict__ sums, const uint8_t *__restrict__ rgb, uint32_t w, uint64_t N, uint32_t K, uint64_t seed, uint64_t iter,
                             WideCent *__restrict__ cent, uint64_t *__restrict__ members_out, unsigned long long *__restrict__ counters /* reseeds, active */) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += stride) {
        unsigned long long v[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { const size_t at = i < 5 ? 5 * (size_t)k + i : 5 * (size_t)K + k; v[i] = sums[at]; sums[at] = 0ull; }
        members_out[k] = v[5];
        if (v[5] == 0) {
            const uint64_t idx = reseed_index(seed, iter, (uint32_t)k, N);  // fake_clone of the stolen pixel
            cent[k] = WideCent{(int32_t)(idx % w), (int32_t)(idx / w), rgb_key(rgb + 3 * idx), 0u};
            atomicAdd(&counters[0], 1ull);
        } else {
            const uint32_t r = (uint32_t)(v[2] / v[5]) & 255, g = (uint32_t)(v[3] / v[5]) & 255, b = (uint32_t)(v[4] / v[5]) & 255;
            cent[k] = WideCent{(int32_t)(v[0] / v[5]), (int32_t)(v[1] / v[5]), (r << 16) | (g << 8) | b, 0u};
            atomicAdd(&counters[1], 1ull);
        }
    }
}

int km_xyrgb_run_wide(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t K, const cniic_kmeans_opts *opts, cniic_colorpos *centroids_h,
                      uint32_t *labels_d_u32, uint64_t *members_h, cniic_kmeans_stats *stats) {
    const uint64_t N = (uint64_t)w * h;
    if (K == 0 || N == 0) return c->fail(CNIIC_ERR_BAD_ARG, "kmeans_xyrgb: empty problem");
    if (N / K == 0) return c->fail(CNIIC_ERR_TOO_FEW_POINTS, "kmeans: %llu points for %u clusters (src/kmeans.rs:68)", (unsigned long long)N, K);
    if (N > (1ull << 32)) return c->fail(CNIIC_ERR_UNSUPPORTED, "kmeans_xyrgb: more than 2^32 pixels");
    const uint64_t seed = (opts && opts->seed) ? opts->seed : kDefaultSeed, max_iters = opts ? opts->max_iters : 0;
    DevBuf labels_own, cent, sums, members, counters;
    uint32_t *labels = labels_d_u32;
    if (!labels) { CNIIC_HIP_TRY(c, labels_own.alloc(N * 4)); labels = labels_own.as<uint32_t>(); }
    const uint64_t W = 6 * (uint64_t)K + 2;
    CNIIC_HIP_TRY(c, cent.alloc((uint64_t)K * sizeof(WideCent)));
    CNIIC_HIP_TRY(c, sums.alloc(W * 8));
    CNIIC_HIP_TRY(c, members.alloc((uint64_t)K * 8));
    CNIIC_HIP_TRY(c, counters.alloc(16));
    CNIIC_HIP_TRY(c, hipMemsetAsync(sums.p, 0, W * 8, c->stream));
    const uint32_t g = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(N, 256), 1), 4096), gk = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(K, 256), 1), 4096);
    hipLaunchKernelGGL(k_xyw_init, dim3(std::max(g, gk)), dim3(256), 0, c->stream, rgb_d, w, N, K, labels, cent.as<WideCent>());
    uint64_t iter = 0, reseeds = 0, active = 0, changed = 0, evals = 0;
    for (;;) {   // kmeans.rs:26-32: assign, update, until an assign step moves nobody
        hipLaunchKernelGGL(k_xyw_assign, dim3(g), dim3(256), 0, c->stream, rgb_d, w, N, K, (const WideCent *)cent.as<WideCent>(), labels, sums.as<unsigned long long>());
        CNIIC_HIP_TRY(c, hipGetLastError());
        unsigned long long tail[2];
        CNIIC_HIP_TRY(c, hipMemcpyAsync(tail, sums.as<unsigned long long>() + 6 * (size_t)K, 16, hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipMemsetAsync(counters.p, 0, 16, c->stream));
        hipLaunchKernelGGL(k_xyw_update, dim3(gk), dim3(256), 0, c->stream, sums.as<unsigned long long>(), rgb_d, w, N, K, seed, iter, cent.as<WideCent>(), members.as<uint64_t>(),
                           counters.as<unsigned long long>());
        CNIIC_HIP_TRY(c, hipMemsetAsync(sums.as<unsigned long long>() + 6 * (size_t)K, 0, 16, c->stream));
        unsigned long long cn[2];
        CNIIC_HIP_TRY(c, hipMemcpyAsync(cn, counters.p, 16, hipMemcpyDeviceToHost, c->stream));
        CNIIC_HIP_TRY(c, hipStreamSynchronize(c->stream));
        iter++;
        changed = tail[0]; evals += tail[1]; reseeds += cn[0]; active = cn[1];
        if (changed == 0 || (max_iters && iter >= max_iters)) break;
    }
    std::vector<WideCent> ch(K);
    CNIIC_HIP_TRY(c, hipMemcpy(ch.data(), cent.p, (size_t)K * sizeof(WideCent), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < K; k++) {
        centroids_h[k].x = (uint32_t)ch[k].x; centroids_h[k].y = (uint32_t)ch[k].y;
        centroids_h[k].rgb[0] = (uint8_t)(ch[k].col >> 16); centroids_h[k].rgb[1] = (uint8_t)(ch[k].col >> 8); centroids_h[k].rgb[2] = (uint8_t)ch[k].col;
        centroids_h[k].pad = 0;
    }
    if (members_h) CNIIC_HIP_TRY(c, hipMemcpy(members_h, members.p, (size_t)K * 8, hipMemcpyDeviceToHost));
    if (stats) { stats->iterations = iter; stats->moved_last = changed; stats->empty_reseeds = reseeds; stats->active = active; stats->pair_evals = evals; }
    return CNIIC_OK;
}

}  // namespace cniic
