/*
 * codecs.c -- oracle (test infrastructure only): the four codecs on the hot path.
 * Restates src/codec/hufc.rs, src/codec/clusterc.rs, src/codec/hilbertc.rs:397-582 and
 * src/bench.rs:95-104 (MSE).  Images are RGB8, row-major interleaved (image::DynamicImage
 * to_rgb(), row-major pixels()).
 */
#include "cniic_oracle.h"
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t rgb_key(const uint8_t *p) {
    return ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
}

static uint32_t *rgb_to_syms(const uint8_t *rgb, uint64_t n) {
    uint32_t *s = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!s) return NULL;
    for (uint64_t i = 0; i < n; i++) s[i] = rgb_key(rgb + 3 * i);
    return s;
}

/* hufc.rs:12-17 */
int orc_hufman_encode(const uint8_t *rgb, uint32_t w, uint32_t h, orc_buf *out) {
    uint64_t n = (uint64_t)w * h;
    int rc = orc_ser_u32(out, w);            /* (u32,u32).serialize ser.rs:146-151 */
    if (!rc) rc = orc_ser_u32(out, h);
    if (rc) return rc;
    uint32_t *s = rgb_to_syms(rgb, n);
    if (!s) return ORC_ERR_NOMEM;
    rc = orc_huf_encode_all(ORC_SYM_RGB, s, n, out);
    free(s);
    return rc;
}

/* hufc.rs:19-40 (create_image_buffer_standard codec.rs:22-26) */
int orc_hufman_decode(const uint8_t *bytes, size_t nb, uint8_t *rgb, size_t cap, uint32_t *w, uint32_t *h) {
    orc_rd r = { bytes, nb, 0 };
    if (orc_de_u32(&r, w) || orc_de_u32(&r, h)) return ORC_ERR_DECODE;
    uint64_t n = (uint64_t)*w * *h;
    if (n * 3 > cap) return ORC_ERR_CAPACITY;
    uint32_t *s = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!s) return ORC_ERR_NOMEM;
    int rc = orc_huf_decode_all(ORC_SYM_RGB, &r, s, n);
    if (!rc)
        for (uint64_t i = 0; i < n; i++) {
            rgb[3 * i] = (uint8_t)(s[i] >> 16); rgb[3 * i + 1] = (uint8_t)(s[i] >> 8); rgb[3 * i + 2] = (uint8_t)s[i];
        }
    free(s);
    return rc;
}

/* hilbertc.rs:405-415 */
int orc_delta_encode(const uint8_t *rgb, uint32_t w, uint32_t h, orc_buf *out) {
    uint64_t n = (uint64_t)w * h;
    int rc = orc_ser_u32(out, w);
    if (!rc) rc = orc_ser_u32(out, h);
    if (rc) return rc;
    uint8_t *lin = (uint8_t *)malloc((n ? n : 1) * 3);
    uint32_t *s = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!lin || !s) { free(lin); free(s); return ORC_ERR_NOMEM; }
    rc = orc_hilbert_linearize(rgb, w, h, lin);       /* hilbertc.rs:410 */
    if (!rc) rc = orc_delta_diff(lin, n, s);           /* hilbertc.rs:411 */
    if (!rc) rc = orc_huf_encode_all(ORC_SYM_SIGNED, s, n, out); /* hilbertc.rs:414 */
    free(lin); free(s);
    return rc;
}

/* hilbertc.rs:417-431 */
int orc_delta_decode(const uint8_t *bytes, size_t nb, uint8_t *rgb, size_t cap, uint32_t *w, uint32_t *h) {
    orc_rd r = { bytes, nb, 0 };
    if (orc_de_u32(&r, w) || orc_de_u32(&r, h)) return ORC_ERR_DECODE;
    uint64_t n = (uint64_t)*w * *h;
    if (n * 3 > cap) return ORC_ERR_CAPACITY;
    uint32_t *s = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint8_t *lin = (uint8_t *)malloc((n ? n : 1) * 3);
    uint32_t *xy = (uint32_t *)malloc((n ? n : 1) * 2 * sizeof(uint32_t));
    int rc = (!s || !lin || !xy) ? ORC_ERR_NOMEM : ORC_OK;
    if (!rc) rc = orc_huf_decode_all(ORC_SYM_SIGNED, &r, s, n);
    if (!rc) rc = orc_delta_undiff(s, n, lin);
    if (!rc) rc = orc_hilbert_iter(*w, *h, xy);
    if (!rc)
        for (uint64_t i = 0; i < n; i++)
            memcpy(rgb + ((uint64_t)xy[2 * i + 1] * *w + xy[2 * i]) * 3, lin + 3 * i, 3);
    free(s); free(lin); free(xy);
    return rc;
}

/* clusterc.rs:18-53 */
int orc_cluster_colors_encode(const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t K, int mode,
                              uint64_t seed, orc_buf *out, orc_km_stats *st) {
    uint64_t n = (uint64_t)w * h;
    if (n == 0) return ORC_ERR_TOO_FEW_POINTS;
    uint32_t *syms = rgb_to_syms(rgb, n);
    uint32_t *keys = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint64_t *counts = (uint64_t *)malloc(n * sizeof(uint64_t));
    int rc = (!syms || !keys || !counts) ? ORC_ERR_NOMEM : ORC_OK;
    uint64_t U = 0;
    if (!rc) rc = orc_count_freqs(syms, n, keys, counts, n, &U);   /* clusterc.rs:21 */
    int32_t *pts = NULL, *cent = NULL;
    uint32_t *wt = NULL, *labels = NULL;
    uint64_t *members = NULL;
    uint8_t *reduced = NULL;
    if (!rc) {
        pts = (int32_t *)malloc(U * 3 * sizeof(int32_t));
        wt = (uint32_t *)malloc(U * sizeof(uint32_t));
        labels = (uint32_t *)malloc(U * sizeof(uint32_t));
        cent = (int32_t *)malloc((size_t)K * 3 * sizeof(int32_t));
        members = (uint64_t *)malloc((size_t)K * sizeof(uint64_t));
        reduced = (uint8_t *)malloc(n * 3);
        if (!pts || !wt || !labels || !cent || !members || !reduced) rc = ORC_ERR_NOMEM;
    }
    if (!rc) {
        for (uint64_t i = 0; i < U; i++) { /* clusterc.rs:22-24, order = ascending key (D1) */
            pts[3 * i] = (keys[i] >> 16) & 255; pts[3 * i + 1] = (keys[i] >> 8) & 255; pts[3 * i + 2] = keys[i] & 255;
            wt[i] = (uint32_t)counts[i];   /* count as u32 */
        }
        rc = orc_kmeans(ORC_PT_RGBW, mode, pts, wt, U, K, seed, 0, cent, labels, members, NULL, st); /* :28 */
    }
    if (!rc) {
        /* clusterc.rs:31-47: colour -> centroid colour lookup, applied to every pixel */
        for (uint64_t i = 0; i < n; i++) {
            uint32_t k = syms[i];
            uint64_t lo = 0, hi = U;
            while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (keys[mid] < k) lo = mid + 1; else hi = mid; }
            const int32_t *c = cent + (size_t)labels[lo] * 3;
            reduced[3 * i] = (uint8_t)c[0]; reduced[3 * i + 1] = (uint8_t)c[1]; reduced[3 * i + 2] = (uint8_t)c[2];
        }
        rc = orc_hufman_encode(reduced, w, h, out);  /* clusterc.rs:52 */
    }
    free(syms); free(keys); free(counts); free(pts); free(wt); free(labels); free(cent); free(members); free(reduced);
    return rc;
}

/* clusterc.rs:148-166 */
int orc_voronoi_encode(const uint8_t *rgb, uint32_t w, uint32_t h, uint32_t K, int mode,
                       uint64_t seed, orc_buf *out, orc_km_stats *st) {
    uint64_t n = (uint64_t)w * h;
    if (n == 0) return ORC_ERR_TOO_FEW_POINTS;
    int32_t *pts = (int32_t *)malloc(n * 5 * sizeof(int32_t));
    uint32_t *labels = (uint32_t *)malloc(n * sizeof(uint32_t));
    int32_t *cent = (int32_t *)malloc((size_t)K * 5 * sizeof(int32_t));
    uint64_t *members = (uint64_t *)malloc((size_t)K * sizeof(uint64_t));
    int rc = (!pts || !labels || !cent || !members) ? ORC_ERR_NOMEM : ORC_OK;
    if (!rc) {
        for (uint32_t y = 0; y < h; y++)      /* img.pixels(): row-major, clusterc.rs:150-152 */
            for (uint32_t x = 0; x < w; x++) {
                uint64_t i = (uint64_t)y * w + x;
                pts[5 * i] = (int32_t)x; pts[5 * i + 1] = (int32_t)y;
                pts[5 * i + 2] = rgb[3 * i]; pts[5 * i + 3] = rgb[3 * i + 1]; pts[5 * i + 4] = rgb[3 * i + 2];
            }
        rc = orc_kmeans(ORC_PT_XYRGB, mode, pts, NULL, n, K, seed, 0, cent, labels, members, NULL, st);
    }
    if (!rc) rc = orc_ser_u32(out, w);         /* :156-158 */
    if (!rc) rc = orc_ser_u32(out, h);
    if (!rc) rc = orc_ser_u64(out, K);         /* :161 usize */
    for (uint32_t c = 0; c < K && !rc; c++) {  /* :162-164, ColorPos::serialize :250-257 */
        const int32_t *p = cent + (size_t)c * 5;
        uint8_t col[3] = { (uint8_t)p[2], (uint8_t)p[3], (uint8_t)p[4] };
        rc = orc_ser_u32(out, (uint32_t)p[0]);
        if (!rc) rc = orc_ser_u32(out, (uint32_t)p[1]);
        if (!rc) rc = orc_ser_rgb(out, col);
    }
    free(pts); free(labels); free(cent); free(members);
    return rc;
}

/* clusterc.rs:168-189 */
int orc_voronoi_decode(const uint8_t *bytes, size_t nb, uint8_t *rgb, size_t cap, uint32_t *w, uint32_t *h) {
    orc_rd r = { bytes, nb, 0 };
    if (orc_de_u32(&r, w) || orc_de_u32(&r, h)) return ORC_ERR_DECODE;
    uint64_t K;
    if (orc_de_u64(&r, &K)) return ORC_ERR_DECODE;
    if (K > (nb - r.pos) / 19) return ORC_ERR_DECODE; /* stream would run dry */
    uint32_t *cx = (uint32_t *)malloc((K ? K : 1) * sizeof(uint32_t));
    uint32_t *cy = (uint32_t *)malloc((K ? K : 1) * sizeof(uint32_t));
    uint8_t *cc = (uint8_t *)malloc((K ? K : 1) * 3);
    int rc = (!cx || !cy || !cc) ? ORC_ERR_NOMEM : ORC_OK;
    for (uint64_t c = 0; c < K && !rc; c++) {
        if (orc_de_u32(&r, &cx[c]) || orc_de_u32(&r, &cy[c]) || orc_de_rgb(&r, cc + 3 * c)) rc = ORC_ERR_DECODE;
    }
    uint64_t n = (uint64_t)*w * *h;
    if (!rc && n * 3 > cap) rc = ORC_ERR_CAPACITY;
    if (!rc && n > 0 && K == 0) rc = ORC_ERR_DECODE; /* min_by_key on empty -> unwrap panic :184 */
    if (!rc)
        for (uint32_t y = 0; y < *h; y++)
            for (uint32_t x = 0; x < *w; x++) {
                uint32_t best = 0, bk = 0;
                for (uint64_t c = 0; c < K; c++) {
                    uint32_t dx = cx[c] - x, dy = cy[c] - y;       /* wrapping u32, :183 */
                    uint32_t key = dx * dx + dy * dy;
                    if (c == 0 || key < best) { best = key; bk = (uint32_t)c; } /* first minimum */
                }
                memcpy(rgb + ((uint64_t)y * *w + x) * 3, cc + 3 * (size_t)bk, 3);
            }
    free(cx); free(cy); free(cc);
    return rc;
}

/* bench.rs:95-104: sum of dist(px,py).powi(2) in f64, divided by w*h */
/* ---- Hilbert { compress: RLE(0.0) }  (hilbertc.rs:12-98): exact run-length coding along the scan ----
 * rle_exact = AbstractRle with the Exact criteria (hilbertc.rs:100-196): a run takes the first colour and
 * every following colour equal to it, up to RepCount::MAX = 255 elements (the element that would be the
 * 256th starts the next run, :129-137); each run is serialised as count:u8 then the colour (:34-37). */
int orc_hilbert_rle_encode(const uint8_t *rgb, uint32_t w, uint32_t h, orc_buf *out) {
    uint64_t n = (uint64_t)w * h;
    int rc = orc_ser_u32(out, w);                      /* img.dimensions().serialize :27 */
    if (!rc) rc = orc_ser_u32(out, h);
    if (rc || n == 0) return rc;
    uint8_t *lin = (uint8_t *)malloc(3 * n);
    if (!lin) return ORC_ERR_NOMEM;
    rc = orc_hilbert_linearize(rgb, w, h, lin);        /* hilbert::linearize(img) :29 */
    uint64_t i = 0;
    while (!rc && i < n) {
        const uint8_t *first = lin + 3 * i;            /* next_val :116-119, start_sequence :123 */
        uint32_t count = 1;
        i++;
        while (i < n) {                                /* :125 */
            if (memcmp(lin + 3 * i, first, 3) != 0) break;   /* not accepted: saved as self.last :139-142 */
            count++;
            i++;
            if (count == 255) break;                   /* :128-137 */
        }
        rc = orc_ser_u8(out, (uint8_t)count);          /* :35 */
        if (!rc) rc = orc_ser_rgb(out, first);         /* :36 */
    }
    free(lin);
    return rc;
}

/* RleDecoder (hilbertc.rs:304-337) zipped with hilbert::iter (:58-61): pixels the stream does not reach stay
 * zero (ImageBuffer::new); a zero count trips `assert!(self.count > 0)` and a cut colour the `unwrap()`,
 * both reported as ORC_ERR_DECODE. */
int orc_hilbert_rle_decode(const uint8_t *bytes, size_t nb, uint8_t *rgb, size_t cap, uint32_t *w, uint32_t *h) {
    orc_rd r = { bytes, nb, 0 };
    if (orc_de_u32(&r, w) || orc_de_u32(&r, h)) return ORC_ERR_DECODE;
    uint64_t n = (uint64_t)*w * *h;
    if (3 * n > cap) return ORC_ERR_CAPACITY;
    memset(rgb, 0, 3 * n);
    if (n == 0) return 0;
    uint32_t *xy = (uint32_t *)malloc(8 * n);
    if (!xy) return ORC_ERR_NOMEM;
    int rc = orc_hilbert_iter(*w, *h, xy);
    uint64_t i = 0;
    while (!rc && i < n) {
        uint8_t count, col[3];
        if (orc_de_u8(&r, &count)) break;              /* stream ended: RepCount::deserialize(..)? */
        if (count == 0 || orc_de_rgb(&r, col)) { rc = ORC_ERR_DECODE; break; }
        for (uint32_t k = 0; k < count && i < n; k++, i++)
            memcpy(rgb + 3 * ((uint64_t)xy[2 * i + 1] * *w + xy[2 * i]), col, 3);
    }
    free(xy);
    return rc;
}

double orc_mse(const uint8_t *a, const uint8_t *b, uint64_t npx) {
    double tot = 0.0;
    for (uint64_t i = 0; i < npx; i++) {
        double s = 0.0;
        for (int c = 0; c < 3; c++) {
            int32_t d = (int32_t)a[3 * i + c] - (int32_t)b[3 * i + c];
            s += (double)(d * d);
        }
        double dist = sqrt(s);
        tot += dist * dist;
    }
    return tot / (double)npx;
}

/* ---- codec expression parsing (FromStr impls) ---- */
static int parse_fun_u32(const char *s, const char *const *prefixes, uint32_t *arg) {
    /* unanchored search, like Regex::captures (clusterc.rs:125-127, 281-283) */
    for (const char *p = s; *p; p++)
        for (int i = 0; prefixes[i]; i++) {
            size_t l = strlen(prefixes[i]);
            if (strncmp(p, prefixes[i], l) == 0 && p[l] == '(') {
                const char *q = p + l + 1;
                if (!isdigit((unsigned char)*q)) continue;
                unsigned long long v = 0;
                while (isdigit((unsigned char)*q)) { v = v * 10 + (unsigned)(*q - '0'); q++; if (v > 0xffffffffULL) return 0; }
                if (*q != ')') continue;
                *arg = (uint32_t)v;
                return 1;
            }
        }
    return 0;
}

enum { CODEC_NONE, CODEC_CLUSTER, CODEC_VORONOI, CODEC_DELTA, CODEC_HUFMAN, CODEC_HILBERT_RLE };

/* Hilbert::from_str (hilbertc.rs:341-397): fun_call with name ^[Hh]ilbert$ and one argument, `rle` or `rle(<f64>)`;
 * only the exact method (d == 0.0) is on the path; rle(d != 0) and zip are not (sequential running average / zip-dict) */
static int parse_hilbert_rle(const char *s) {
    if (strncmp(s, "hilbert(", 8) != 0 && strncmp(s, "Hilbert(", 8) != 0) return 0;
    const char *a = s + 8;
    size_t l = strlen(a);
    if (l < 1 || a[l - 1] != ')') return 0;
    char arg[64];
    if (l - 1 >= sizeof arg) return 0;
    memcpy(arg, a, l - 1);
    arg[l - 1] = 0;
    if (strcmp(arg, "rle") == 0) return 1;
    if (strncmp(arg, "rle(", 4) == 0 && arg[strlen(arg) - 1] == ')') {
        char num[64];
        size_t m = strlen(arg) - 5;
        if (m == 0 || m >= sizeof num) return 0;
        memcpy(num, arg + 4, m);
        num[m] = 0;
        char *end = NULL;
        double d = strtod(num, &end);
        return end && *end == 0 && d == 0.0;
    }
    return 0;
}

static int parse_codec(const char *s, uint32_t *K) {
    /* order of alternatives: codec.rs:120-127 */
    static const char *const cc[] = { "cluster-colors", "cluster-col", "clustercolors", "clustercol",
                                      "c-colors", "c-col", "ccolors", "ccol", NULL };
    static const char *const vo[] = { "voronoi", NULL };
    if (parse_fun_u32(s, cc, K)) return CODEC_CLUSTER;       /* c(?:luster)?-?col(?:ors)?\((\d+)\) */
    if (parse_fun_u32(s, vo, K)) return CODEC_VORONOI;       /* voronoi\((\d+)\) */
    if (strcmp(s, "delta") == 0) return CODEC_DELTA;         /* hilbertc.rs:578-581 ^delta$ */
    if (parse_hilbert_rle(s)) return CODEC_HILBERT_RLE;      /* after Delta, before Hufman: codec.rs:120-127 */
    size_t l = strlen(s);
    if (l == 6) {                                            /* hufc.rs:54-59 eq_ignore_ascii_case */
        char t[7];
        for (int i = 0; i < 6; i++) t[i] = (char)tolower((unsigned char)s[i]);
        t[6] = 0;
        if (strcmp(t, "hufman") == 0) return CODEC_HUFMAN;
    }
    return CODEC_NONE;
}

int orc_encode(const char *codec, int mode, uint64_t seed, const uint8_t *rgb, uint32_t w, uint32_t h,
               uint8_t *out, uint64_t cap, uint64_t *len, orc_km_stats *st) {
    uint32_t K = 0;
    int c = parse_codec(codec, &K);
    orc_buf b;
    orc_buf_init(&b);
    int rc;
    if (st) memset(st, 0, sizeof *st);
    switch (c) {
    case CODEC_HUFMAN: rc = orc_hufman_encode(rgb, w, h, &b); break;
    case CODEC_DELTA: rc = orc_delta_encode(rgb, w, h, &b); break;
    case CODEC_HILBERT_RLE: rc = orc_hilbert_rle_encode(rgb, w, h, &b); break;
    case CODEC_CLUSTER: rc = orc_cluster_colors_encode(rgb, w, h, K, mode, seed, &b, st); break;
    case CODEC_VORONOI: rc = orc_voronoi_encode(rgb, w, h, K, mode, seed, &b, st); break;
    default: rc = ORC_ERR_BAD_ARG;
    }
    *len = b.len;
    if (!rc) {
        if (b.len > cap) rc = ORC_ERR_CAPACITY;
        else memcpy(out, b.data, b.len);
    }
    orc_buf_free(&b);
    return rc;
}

int orc_decode(const char *codec, const uint8_t *bytes, uint64_t n, uint8_t *rgb, uint64_t cap,
               uint32_t *w, uint32_t *h) {
    uint32_t K = 0;
    switch (parse_codec(codec, &K)) {
    case CODEC_HUFMAN: return orc_hufman_decode(bytes, n, rgb, cap, w, h);
    case CODEC_CLUSTER: return orc_hufman_decode(bytes, n, rgb, cap, w, h); /* clusterc.rs:55-57 */
    case CODEC_DELTA: return orc_delta_decode(bytes, n, rgb, cap, w, h);
    case CODEC_HILBERT_RLE: return orc_hilbert_rle_decode(bytes, n, rgb, cap, w, h);
    case CODEC_VORONOI: return orc_voronoi_decode(bytes, n, rgb, cap, w, h);
    }
    return ORC_ERR_BAD_ARG;
}
