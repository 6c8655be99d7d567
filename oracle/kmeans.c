/*
 * kmeans.c -- oracle (test infrastructure only): K-means engine of the reference.
 * Restates src/kmeans.rs:21-440 generically over three point kinds:
 *   ORC_PT_TOY2  (i32,i32) of the reference's own test module   kmeans.rs:451-477
 *   ORC_PT_RGBW  ColorCount  (colour + pixel-count weight)      clusterc.rs:68-114, geom.rs:8-24
 *   ORC_PT_XYRGB ColorPos    (x, y, colour)                     clusterc.rs:200-248
 *
 * Mode R follows the reference line by line: chunked init (61-78,101-108), triangle-inequality
 * pruned assign over per-cluster neighbour lists that are dynamically truncated (150-323,
 * 330-416), integer means (Point::mean), f64 sqrt distances.
 * Mode L is exact Lloyd with the SAME init, tie and mean rules, on integer squared distances:
 *   a point stays in its cluster when no centroid is strictly closer (kmeans.rs:375 strict '<',
 *   search starts from the current centroid, 350-351); otherwise it moves to the nearest
 *   centroid, lowest cluster id among equidistant minima.
 * Mode L is what the HIP path must match bit for bit; mode R is the CPU baseline and the
 * statistical reference (bytes/px, MSE).
 */
#include "cniic_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_pt_dim(int kind) {
    switch (kind) {
    case ORC_PT_TOY2: return 2;
    case ORC_PT_RGBW: return 3;
    case ORC_PT_XYRGB: return 5;
    }
    return -1;
}

/* geom.rs:8-24: per-channel i32 diff, squared, as f64, summed, sqrt */
static inline double rgb_dist(const int32_t *a, const int32_t *b) {
    double s = 0.0;
    for (int i = 0; i < 3; i++) {
        int32_t d = a[i] - b[i];
        s += (double)(d * d);
    }
    return sqrt(s);
}

double orc_pt_dist(int kind, const int32_t *a, const int32_t *b) {
    switch (kind) {
    case ORC_PT_TOY2: { /* kmeans.rs:451-459 */
        double dx = (double)(a[0] - b[0]);
        double dy = (double)(a[1] - b[1]);
        return sqrt(dx * dx + dy * dy);
    }
    case ORC_PT_RGBW: /* clusterc.rs:74-79: weights ignored */
        return rgb_dist(a, b);
    case ORC_PT_XYRGB: { /* clusterc.rs:206-213: wrapping u32 sub, pow(2), then colour term as sqrt(..).powi(2) */
        uint32_t ux = (uint32_t)a[0] - (uint32_t)b[0];
        uint32_t uy = (uint32_t)a[1] - (uint32_t)b[1];
        double d = (double)(uint32_t)(ux * ux);
        d += (double)(uint32_t)(uy * uy);
        double c = rgb_dist(a + 2, b + 2);
        d += c * c;
        return sqrt(d);
    }
    }
    return NAN;
}

static inline int64_t pt_dist2(int kind, int D, const int32_t *a, const int32_t *b) {
    (void)kind;
    int64_t s = 0;
    for (int i = 0; i < D; i++) {
        int64_t d = (int64_t)a[i] - (int64_t)b[i];
        s += d * d;
    }
    return s;
}

/* kmeans.rs:61-78 init_assignment, expressed per point index */
uint32_t orc_kmeans_init_label(uint64_t i, uint64_t n, uint32_t K) {
    uint64_t ppc = n / K;
    /* cluster c < K-1 owns [n-(c+1)*ppc, n-c*ppc); cluster K-1 owns [0, n-(K-1)*ppc) */
    uint64_t from_end = n - 1 - i;
    uint64_t c = from_end / ppc;
    if (c > (uint64_t)K - 1) c = (uint64_t)K - 1;
    return (uint32_t)c;
}

static inline uint64_t splitmix_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* deviation D2: deterministic stand-in for rand::thread_rng (kmeans.rs:123-133) */
uint64_t orc_kmeans_reseed_index(uint64_t seed, uint64_t iter, uint32_t c, uint64_t n) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (iter * 65536ULL + (uint64_t)c + 1ULL);
    return splitmix_mix(z) % n;
}

/* Point::mean from accumulated sums (clusterc.rs:83-113 weighted / 216-247 unweighted /
 * kmeans.rs:461-477 toy) + empty-cluster handling of kmeans.rs:110-137 */
int orc_kmeans_finalize(int kind, const int32_t *pts, uint64_t n, uint32_t K, uint64_t seed,
                        uint64_t iter, const uint64_t *sums, const uint64_t *wsum,
                        const uint64_t *members, int32_t *centroids, uint64_t *n_reseeded) {
    int D = orc_pt_dim(kind);
    if (D < 0) return ORC_ERR_BAD_ARG;
    uint64_t res = 0;
    for (uint32_t c = 0; c < K; c++) {
        if (members[c] == 0) { /* T::mean -> None */
            uint64_t idx = orc_kmeans_reseed_index(seed, iter, c, n);
            memcpy(centroids + (size_t)c * D, pts + (size_t)idx * D, D * sizeof(int32_t)); /* fake_clone */
            res++;
            continue;
        }
        for (int d = 0; d < D; d++) {
            if (kind == ORC_PT_TOY2) {
                int64_t s = (int64_t)sums[(size_t)c * D + d];
                centroids[(size_t)c * D + d] = (int32_t)(s / (int64_t)members[c]);
            } else if (kind == ORC_PT_RGBW) {
                /* len==1 -> clone (clusterc.rs:87-90) equals c*w/w */
                centroids[(size_t)c * D + d] = (int32_t)(uint8_t)(sums[(size_t)c * D + d] / wsum[c]);
            } else {
                uint64_t v = sums[(size_t)c * D + d] / members[c];
                centroids[(size_t)c * D + d] = d < 2 ? (int32_t)(uint32_t)v : (int32_t)(uint8_t)v;
            }
        }
    }
    if (n_reseeded) *n_reseeded = res;
    return ORC_OK;
}

static void accumulate(int kind, int D, const int32_t *p, uint32_t w, uint64_t *sums, uint64_t *wsum,
                       uint64_t *members) {
    if (kind == ORC_PT_RGBW) {
        for (int d = 0; d < D; d++) sums[d] += (uint64_t)p[d] * (uint64_t)w; /* clusterc.rs:92-98 */
        *wsum += w;
    } else if (kind == ORC_PT_TOY2) {
        for (int d = 0; d < D; d++) sums[d] = (uint64_t)((int64_t)sums[d] + (int64_t)p[d]);
        *wsum += 1;
    } else {
        for (int d = 0; d < D; d++) sums[d] += (uint64_t)(uint32_t)p[d]; /* clusterc.rs:221-228 */
        *wsum += 1;
    }
    *members += 1;
}

/* ---- exact Lloyd step (mode L) ---- */
int orc_kmeans_step(int kind, const int32_t *pts, const uint32_t *weight, uint64_t n, uint32_t K,
                    const int32_t *centroids, uint32_t *labels,
                    uint64_t *sums, uint64_t *wsum, uint64_t *members, uint64_t *changed) {
    int D = orc_pt_dim(kind);
    if (D < 0 || K == 0) return ORC_ERR_BAD_ARG;
    if (kind == ORC_PT_RGBW && !weight) return ORC_ERR_BAD_ARG;
    memset(sums, 0, (size_t)K * D * sizeof(uint64_t));
    memset(wsum, 0, (size_t)K * sizeof(uint64_t));
    memset(members, 0, (size_t)K * sizeof(uint64_t));
    uint64_t ch = 0;
    for (uint64_t i = 0; i < n; i++) {
        const int32_t *p = pts + (size_t)i * D;
        uint32_t cur = labels[i];
        if (cur >= K) return ORC_ERR_BAD_ARG;
        int64_t dcur = pt_dist2(kind, D, centroids + (size_t)cur * D, p);
        int64_t best = dcur;
        uint32_t bk = cur;
        if (dcur != 0) {
            int64_t m = INT64_MAX;
            uint32_t mk = 0;
            for (uint32_t k = 0; k < K; k++) {
                int64_t dk = pt_dist2(kind, D, centroids + (size_t)k * D, p);
                if (dk < m) { m = dk; mk = k; } /* lowest id among minima */
            }
            if (m < dcur) { best = m; bk = mk; } /* strict: ties stay (kmeans.rs:375) */
        }
        (void)best;
        if (bk != cur) ch++;
        labels[i] = bk;
        accumulate(kind, D, p, weight ? weight[i] : 1, sums + (size_t)bk * D, wsum + bk, members + bk);
    }
    *changed = ch;
    return ORC_OK;
}

/* ---- neighbour lists (mode R), kmeans.rs:150-323 ---- */
typedef struct {
    uint32_t *id;
    double   *dist;
    uint32_t  len;
    uint32_t  watermark;
} neigh_t;

static void neigh_reset(neigh_t *nb, uint32_t src, uint32_t K) { /* kmeans.rs:157-170 */
    uint32_t l = 0;
    for (uint32_t d = 0; d < K; d++)
        if (d != src) { nb->id[l] = d; nb->dist[l] = 0.0; l++; }
    nb->len = l;
    nb->watermark = l;
}

/* stable merge sort by dist ascending (deviation D3 for sort_unstable_by, kmeans.rs:181-186) */
static void neigh_sort(neigh_t *nb, uint32_t *tid, double *tdist) {
    uint32_t n = nb->len;
    /* fast path: already sorted */
    int sorted = 1;
    for (uint32_t i = 1; i < n; i++)
        if (nb->dist[i] < nb->dist[i - 1]) { sorted = 0; break; }
    if (sorted) return;
    uint32_t *a_id = nb->id, *b_id = tid;
    double *a_d = nb->dist, *b_d = tdist;
    for (uint32_t width = 1; width < n; width *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * width) {
            uint32_t mid = lo + width < n ? lo + width : n;
            uint32_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            uint32_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (a_d[j] < a_d[i]) { b_id[k] = a_id[j]; b_d[k++] = a_d[j++]; }
                else { b_id[k] = a_id[i]; b_d[k++] = a_d[i++]; }
            }
            while (i < mid) { b_id[k] = a_id[i]; b_d[k++] = a_d[i++]; }
            while (j < hi) { b_id[k] = a_id[j]; b_d[k++] = a_d[j++]; }
        }
        uint32_t *t1 = a_id; a_id = b_id; b_id = t1;
        double *t2 = a_d; a_d = b_d; b_d = t2;
    }
    if (a_id != nb->id) {
        memcpy(nb->id, a_id, n * sizeof(uint32_t));
        memcpy(nb->dist, a_d, n * sizeof(double));
    }
}

/* kmeans.rs:189-248; returns 1 if shrunk */
static int neigh_resize(neigh_t *nb, uint32_t src, uint32_t K) {
    uint32_t upper = K - 1;
    uint32_t lower = (uint32_t)sqrtf((float)K);
    uint64_t dyn = 2ULL * nb->watermark;
    uint64_t nn = dyn > lower ? dyn : lower;
    if (nn > upper) nn = upper;
    uint32_t cur = nb->len;
    if (nn * 3 / 4 <= cur) {
        if (nn < cur) nb->len = (uint32_t)nn; /* truncate */
        nb->watermark = 0;
        return 1;
    }
    neigh_reset(nb, src, K);
    return 0;
}

static uint64_t compute_neighbours(int kind, int D, const int32_t *cent, neigh_t *nbs, uint32_t K,
                                   uint32_t *tid, double *tdist) { /* kmeans.rs:299-318 */
    uint64_t grow = 0;
    for (uint32_t c = 0; c < K; c++) {
        neigh_t *nb = &nbs[c];
        int shrunk = 0;
        while (!shrunk) {
            for (uint32_t i = 0; i < nb->len; i++) /* update_distances kmeans.rs:172-179 */
                nb->dist[i] = orc_pt_dist(kind, cent + (size_t)c * D, cent + (size_t)nb->id[i] * D);
            neigh_sort(nb, tid, tdist);
            shrunk = neigh_resize(nb, c, K);
            if (!shrunk) grow++;
        }
    }
    return grow;
}

static inline double neigh_radius(const neigh_t *nb) { /* kmeans.rs:257-259 */
    return nb->len ? nb->dist[0] / 2.0 : INFINITY;
}

int orc_kmeans(int kind, int mode, const int32_t *pts, const uint32_t *weight, uint64_t n,
               uint32_t K, uint64_t seed, uint64_t max_iters,
               int32_t *centroids, uint32_t *labels, uint64_t *members, double *radii,
               orc_km_stats *stats) {
    int D = orc_pt_dim(kind);
    if (D < 0 || K == 0 || n >= 0xffffffffULL) return ORC_ERR_BAD_ARG;
    if (kind == ORC_PT_RGBW && !weight) return ORC_ERR_BAD_ARG;
    if (n / K == 0) return ORC_ERR_TOO_FEW_POINTS; /* kmeans.rs:67-68 */
    orc_km_stats st;
    memset(&st, 0, sizeof st);

    uint64_t *sums = (uint64_t *)calloc((size_t)K * D, sizeof(uint64_t));
    uint64_t *wsum = (uint64_t *)calloc(K, sizeof(uint64_t));
    int rc = ORC_OK;
    if (!sums || !wsum) { free(sums); free(wsum); return ORC_ERR_NOMEM; }

    /* init: kmeans.rs:80-90 */
    for (uint64_t i = 0; i < n; i++) labels[i] = orc_kmeans_init_label(i, n, K);
    {   /* init_centroids kmeans.rs:101-108: first element of each chunk */
        uint64_t ppc = n / K;
        for (uint32_t c = 0; c < K; c++) {
            uint64_t first = (c < K - 1) ? n - ((uint64_t)c + 1) * ppc : 0;
            memcpy(centroids + (size_t)c * D, pts + (size_t)first * D, D * sizeof(int32_t));
        }
    }

    if (mode == ORC_KM_MODE_L) {
        uint64_t changed = 1;
        while (changed) { /* kmeans.rs:26-32 */
            /* orc_set_lloyd_threads(t > 1): the same step, threaded and vectorised (kmeans_fast.c) */
            if (orc_get_lloyd_threads() > 1)
                rc = orc_kmeans_step_fast(kind, pts, weight, n, K, centroids, labels, sums, wsum, members, &changed,
                                          orc_get_lloyd_threads());
            else
                rc = orc_kmeans_step(kind, pts, weight, n, K, centroids, labels, sums, wsum, members, &changed);
            if (rc) break;
            st.dist_evals += n * (uint64_t)K;
            uint64_t res = 0;
            orc_kmeans_finalize(kind, pts, n, K, seed, st.iterations, sums, wsum, members, centroids, &res);
            st.empty_reseeds += res;
            st.moved_last = changed;
            st.iterations++;
            if (max_iters && st.iterations >= max_iters) break;
        }
    } else {
        /* processing order = concatenation of the per-cluster Vec<T> (kmeans.rs:343-346) */
        uint32_t *order = (uint32_t *)malloc(n * sizeof(uint32_t));
        uint32_t *order2 = (uint32_t *)malloc(n * sizeof(uint32_t));
        uint32_t *newlab = (uint32_t *)malloc(n * sizeof(uint32_t));
        uint64_t *start = (uint64_t *)calloc((size_t)K + 2, sizeof(uint64_t));
        uint64_t *start2 = (uint64_t *)calloc((size_t)K + 2, sizeof(uint64_t));
        neigh_t *nbs = (neigh_t *)calloc(K, sizeof(neigh_t));
        uint32_t *tid = (uint32_t *)malloc((size_t)K * sizeof(uint32_t));
        double *tdist = (double *)malloc((size_t)K * sizeof(double));
        uint32_t *pool_id = (uint32_t *)malloc((size_t)K * K * sizeof(uint32_t));
        double *pool_d = (double *)malloc((size_t)K * K * sizeof(double));
        if (!order || !order2 || !newlab || !start || !start2 || !nbs || !tid || !tdist || !pool_id || !pool_d) {
            rc = ORC_ERR_NOMEM;
            goto r_done;
        }
        /* cluster c (c<K-1) holds points [n-(c+1)ppc, n-c*ppc) in input order; cluster K-1 the rest */
        {
            uint64_t ppc = n / K, p = 0;
            for (uint32_t c = 0; c < K; c++) {
                uint64_t lo = (c < K - 1) ? n - ((uint64_t)c + 1) * ppc : 0;
                uint64_t hi = (c < K - 1) ? n - (uint64_t)c * ppc : n - (uint64_t)(K - 1) * ppc;
                start[c] = p;
                for (uint64_t i = lo; i < hi; i++) order[p++] = (uint32_t)i;
            }
            start[K] = p;
        }
        for (uint32_t c = 0; c < K; c++) { /* init_neighbours kmeans.rs:288-295 */
            nbs[c].id = pool_id + (size_t)c * K;
            nbs[c].dist = pool_d + (size_t)c * K;
            neigh_reset(&nbs[c], c, K);
        }
        compute_neighbours(kind, D, centroids, nbs, K, tid, tdist);

        int changed = 1;
        while (changed) { /* kmeans.rs:26-32 */
            /* ---- assign_points kmeans.rs:330-416 ---- */
            changed = 0;
            uint64_t moved = 0;
            for (uint32_t cci = 0; cci < K; cci++) {
                const int32_t *cc = centroids + (size_t)cci * D;
                neigh_t *nb = &nbs[cci];
                double radius = neigh_radius(nb);
                for (uint64_t p = start[cci]; p < start[cci + 1]; p++) {
                    const int32_t *x = pts + (size_t)order[p] * D;
                    double min_dist = orc_pt_dist(kind, cc, x); /* :350 */
                    st.dist_evals++;
                    uint32_t closest = cci;
                    if (min_dist <= radius) { /* :355 */
                        st.obvious_stay++;
                    } else {
                        double cutoff = 2.0 * min_dist; /* :360 */
                        uint32_t pos = 0;
                        for (;;) { /* iter_and_record :364 */
                            if (pos >= nb->len) break; /* next() -> None; next_pos = len+1 */
                            uint32_t tsi = nb->id[pos];
                            double c2c = nb->dist[pos];
                            pos++;
                            st.tested_neighbours++;
                            if (c2c > cutoff) { st.neighbour_cutoff++; pos--; break; } /* :367-370 */
                            double td = orc_pt_dist(kind, centroids + (size_t)tsi * D, x);
                            st.dist_evals++;
                            if (td < min_dist) { min_dist = td; closest = tsi; } /* :375-378 */
                        }
                        /* Drop for IterAndRecord (:277-281): watermark = next_pos - 1.
                         * break at element p -> next_pos = p+1 -> p ; exhausted -> len */
                        nb->watermark = pos;
                    }
                    newlab[p] = closest;
                    if (closest != cci) { changed = 1; moved++; }
                }
            }
            st.moved_last = moved;
            /* push order == stable counting sort of the processing order by new label */
            memset(start2, 0, ((size_t)K + 2) * sizeof(uint64_t));
            for (uint64_t p = 0; p < n; p++) start2[newlab[p] + 1]++;
            for (uint32_t c = 0; c < K; c++) start2[c + 1] += start2[c];
            {
                uint64_t *cursor = (uint64_t *)malloc((size_t)K * sizeof(uint64_t));
                if (!cursor) { rc = ORC_ERR_NOMEM; goto r_done; }
                memcpy(cursor, start2, (size_t)K * sizeof(uint64_t));
                for (uint64_t p = 0; p < n; p++) order2[cursor[newlab[p]]++] = order[p];
                free(cursor);
            }
            { uint32_t *t = order; order = order2; order2 = t; }
            { uint64_t *t = start; start = start2; start2 = t; }

            /* ---- update_centroids kmeans.rs:139-143 ---- */
            memset(sums, 0, (size_t)K * D * sizeof(uint64_t));
            memset(wsum, 0, (size_t)K * sizeof(uint64_t));
            memset(members, 0, (size_t)K * sizeof(uint64_t));
            for (uint32_t c = 0; c < K; c++)
                for (uint64_t p = start[c]; p < start[c + 1]; p++) {
                    uint32_t idx = order[p];
                    accumulate(kind, D, pts + (size_t)idx * D, weight ? weight[idx] : 1,
                               sums + (size_t)c * D, wsum + c, members + c);
                }
            uint64_t res = 0;
            orc_kmeans_finalize(kind, pts, n, K, seed, st.iterations, sums, wsum, members, centroids, &res);
            st.empty_reseeds += res;
            /* ---- update_neighbours kmeans.rs:320-323 ---- */
            compute_neighbours(kind, D, centroids, nbs, K, tid, tdist);
            st.iterations++;
            if (max_iters && st.iterations >= max_iters) break;
        }
        for (uint32_t c = 0; c < K; c++) {
            members[c] = start[c + 1] - start[c];
            for (uint64_t p = start[c]; p < start[c + 1]; p++) labels[order[p]] = c;
            if (radii) radii[c] = neigh_radius(&nbs[c]);
        }
    r_done:
        free(order); free(order2); free(newlab); free(start); free(start2); free(nbs);
        free(tid); free(tdist); free(pool_id); free(pool_d);
    }
    if (mode == ORC_KM_MODE_L && radii)
        for (uint32_t c = 0; c < K; c++) radii[c] = NAN;
    free(sums); free(wsum);
    if (stats) *stats = st;
    if (rc) return rc;

    /* check_enough_active_clusters kmeans.rs:41-57 */
    uint64_t min_cc = (uint64_t)(0.99 * (double)K);
    if (n < min_cc) min_cc = n;
    uint64_t active = 0;
    for (uint32_t c = 0; c < K; c++) active += members[c] > 0;
    if (active < min_cc) return ORC_ERR_FEW_ACTIVE;
    return ORC_OK;
}
