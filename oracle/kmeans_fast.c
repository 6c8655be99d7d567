/*
 * kmeans_fast.c -- oracle (test infrastructure only): the SAME exact-Lloyd step as orc_kmeans_step
 * (kmeans.c; restates src/kmeans.rs:330-416 with full neighbour lists + clusterc.rs:74-79,206-213 as
 * integer squared distances), arranged so that a CPU can do it at the BASELINE sizes:
 *   - the points are split into contiguous ranges, one POSIX thread each; every thread keeps its own
 *     u64 sums / weights / member counts, added up afterwards (unsigned integer adds: any order gives
 *     the same words);
 *   - per point the K distances are computed in 32-bit arithmetic over structure-of-arrays centroids
 *     (the compiler vectorises the loop) and the arg-min is a min over (distance << 12 | id) keys,
 *     i.e. the nearest centroid, lowest id among equidistant ones -- orc_kmeans_step's rule.
 * Used by orc_kmeans (mode L) only after orc_set_lloyd_threads(t > 1); the default path stays the plain
 * loop.  tests/test_oracle_fast.py holds this file to the plain step bit for bit (labels, sums, counts,
 * changed) on inputs full of ties.  It exists for tests/golden/make_fullsize_digests.py: mode L at
 * 4096^2 with K = 2048 is 6e12 distance evaluations.
 *
 * Limits (else the plain step runs): RGBW / XYRGB points, K <= 4096, coordinates < 16384 (so that a
 * squared distance stays below 2^31: 2 * 16383^2 + 3 * 255^2).
 */
#include "cniic_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static int g_lloyd_threads = 0;

void orc_set_lloyd_threads(int t) { g_lloyd_threads = t < 0 ? 0 : (t > 64 ? 64 : t); }
int orc_get_lloyd_threads(void) { return g_lloyd_threads; }

typedef struct {
    int kind, D;
    const int32_t *pts;
    const uint32_t *weight;
    uint64_t lo, hi;
    uint32_t K, Kpad;
    const int32_t *soa;     /* [D][Kpad], padded with copies of the last centroid */
    const int32_t *cent;    /* [K][D] as given */
    uint32_t *labels;
    uint64_t *sums, *wsum, *members; /* this thread's own */
    uint64_t changed;
    int rc;
} job_t;

static void *run_job(void *arg) {
    job_t *j = (job_t *)arg;
    const int D = j->D;
    const uint32_t K = j->K, Kpad = j->Kpad;
    const int32_t *c0 = j->soa, *c1 = c0 + Kpad, *c2 = c1 + Kpad;
    const int32_t *c3 = c2 + Kpad, *c4 = c3 + Kpad; /* D == 5 only */
    for (uint64_t i = j->lo; i < j->hi; i++) {
        const int32_t *p = j->pts + (size_t)i * D;
        uint32_t cur = j->labels[i];
        if (cur >= K) { j->rc = ORC_ERR_BAD_ARG; return NULL; }
        int64_t dcur = 0;
        for (int d = 0; d < D; d++) {
            int64_t t = (int64_t)j->cent[(size_t)cur * D + d] - (int64_t)p[d];
            dcur += t * t;
        }
        uint32_t bk = cur;
        if (dcur != 0) {
            int64_t m = INT64_MAX;
            if (D == 3) {
                const int32_t p0 = p[0], p1 = p[1], p2 = p[2];
                for (uint32_t k = 0; k < Kpad; k++) {
                    int32_t a = c0[k] - p0, b = c1[k] - p1, c = c2[k] - p2;
                    int32_t dd = a * a + b * b + c * c;
                    int64_t key = ((int64_t)dd << 12) | (int64_t)k;
                    m = key < m ? key : m;
                }
            } else {
                const int32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4];
                for (uint32_t k = 0; k < Kpad; k++) {
                    int32_t a = c0[k] - p0, b = c1[k] - p1, c = c2[k] - p2, e = c3[k] - p3, f = c4[k] - p4;
                    int32_t dd = a * a + b * b + c * c + e * e + f * f;
                    int64_t key = ((int64_t)dd << 12) | (int64_t)k;
                    m = key < m ? key : m;
                }
            }
            int64_t md = m >> 12;
            if (md < dcur) bk = (uint32_t)(m & 4095); /* strict: ties stay (kmeans.rs:375) */
        }
        if (bk != cur) j->changed++;
        j->labels[i] = bk;
        uint64_t *s = j->sums + (size_t)bk * D;
        if (j->kind == ORC_PT_RGBW) {
            uint64_t w = j->weight[i];
            for (int d = 0; d < D; d++) s[d] += (uint64_t)p[d] * w; /* clusterc.rs:92-98 */
            j->wsum[bk] += w;
        } else {
            for (int d = 0; d < D; d++) s[d] += (uint64_t)(uint32_t)p[d]; /* clusterc.rs:221-228 */
            j->wsum[bk] += 1;
        }
        j->members[bk] += 1;
    }
    return NULL;
}

/* 1 if the fast step may stand in for orc_kmeans_step on this input (n * D compares: nothing beside n * K) */
int orc_kmeans_fast_ok(int kind, const int32_t *pts, uint64_t n, uint32_t K, const int32_t *centroids) {
    if (kind != ORC_PT_RGBW && kind != ORC_PT_XYRGB) return 0;
    if (K == 0 || K > 4096) return 0;
    int D = orc_pt_dim(kind);
    for (size_t i = 0; i < (size_t)K * D; i++)
        if (centroids[i] < 0 || centroids[i] >= 16384) return 0;
    for (size_t i = 0; i < (size_t)n * D; i++)
        if (pts[i] < 0 || pts[i] >= 16384) return 0;
    return 1;
}

int orc_kmeans_step_fast(int kind, const int32_t *pts, const uint32_t *weight, uint64_t n, uint32_t K,
                         const int32_t *centroids, uint32_t *labels,
                         uint64_t *sums, uint64_t *wsum, uint64_t *members, uint64_t *changed, int threads) {
    int D = orc_pt_dim(kind);
    if (D < 0 || K == 0) return ORC_ERR_BAD_ARG;
    if (kind == ORC_PT_RGBW && !weight) return ORC_ERR_BAD_ARG;
    if (!orc_kmeans_fast_ok(kind, pts, n, K, centroids))
        return orc_kmeans_step(kind, pts, weight, n, K, centroids, labels, sums, wsum, members, changed);
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n) threads = n ? (int)n : 1;
    uint32_t Kpad = (K + 15u) & ~15u;
    int32_t *soa = (int32_t *)malloc((size_t)D * Kpad * sizeof(int32_t));
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    size_t per = (size_t)K * D + 2 * (size_t)K;
    uint64_t *acc = (uint64_t *)calloc((size_t)threads * per, sizeof(uint64_t));
    if (!soa || !jobs || !tid || !acc) { free(soa); free(jobs); free(tid); free(acc); return ORC_ERR_NOMEM; }
    for (int d = 0; d < D; d++)
        for (uint32_t k = 0; k < Kpad; k++)
            /* padding: copies of the last centroid under ids above every real one -- the same distance
             * with a larger key, so a padded entry is never the minimum */
            soa[(size_t)d * Kpad + k] = centroids[(size_t)(k < K ? k : K - 1) * D + d];
    for (int t = 0; t < threads; t++) {
        job_t *j = &jobs[t];
        j->kind = kind; j->D = D; j->pts = pts; j->weight = weight;
        j->lo = n * (uint64_t)t / (uint64_t)threads;
        j->hi = n * (uint64_t)(t + 1) / (uint64_t)threads;
        j->K = K; j->Kpad = Kpad; j->soa = soa; j->cent = centroids; j->labels = labels;
        j->sums = acc + (size_t)t * per;
        j->wsum = j->sums + (size_t)K * D;
        j->members = j->wsum + K;
    }
    int started = 0;
    for (int t = 1; t < threads; t++) {
        if (pthread_create(&tid[t], NULL, run_job, &jobs[t]) != 0) break;
        started = t;
    }
    run_job(&jobs[0]);
    for (int t = started + 1; t < threads; t++) run_job(&jobs[t]); /* a thread that could not be made: done here */
    for (int t = 1; t <= started; t++) pthread_join(tid[t], NULL);
    memset(sums, 0, (size_t)K * D * sizeof(uint64_t));
    memset(wsum, 0, (size_t)K * sizeof(uint64_t));
    memset(members, 0, (size_t)K * sizeof(uint64_t));
    uint64_t ch = 0;
    int rc = ORC_OK;
    for (int t = 0; t < threads; t++) {
        job_t *j = &jobs[t];
        if (j->rc) rc = j->rc;
        ch += j->changed;
        for (size_t x = 0; x < (size_t)K * D; x++) sums[x] += j->sums[x];
        for (uint32_t k = 0; k < K; k++) { wsum[k] += j->wsum[k]; members[k] += j->members[k]; }
    }
    *changed = ch;
    free(soa); free(jobs); free(tid); free(acc);
    return rc;
}
