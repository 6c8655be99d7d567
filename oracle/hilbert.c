/*
 * hilbert.c -- oracle (test infrastructure only): Hilbert-order scan + neighbour delta.
 *
 * PARITY UNPINNED for the scan itself.  The reference (src/hilbert.rs:40-43) obtains the scan
 * from the third-party crate zhang_hilbert = "0.1.1" (Cargo.toml:15; `ArbHilbertScan32`, a
 * pseudo-Hilbert scan for arbitrary rectangles after Zhang, Kamata & Ueshige, IEICE Trans.
 * Fundamentals E90-A, 2007).  The crate is not vendored, there is no Cargo.lock, its source is
 * not in the build image, and the reference has no test vector at that boundary
 * (hilbert.rs / hilbertc.rs contain zero tests).  The block-size and look-up tables of the
 * published algorithm cannot be restated from the paper's abstract alone, so this build FREEZES
 * ITS OWN scan with the same contract (visit every cell of a w x h rectangle exactly once,
 * consecutive cells adjacent, same sequence for encode and decode):
 *
 *   the generalised Hilbert curve ("gilbert", J. Cerveny 2018, a published recursive
 *   construction): split the rectangle into three sub-rectangles (or two when it is long),
 *   preferring even side lengths, recursing with rotated axes.  For 2^n x 2^n squares it is
 *   exactly the classic Hilbert curve that starts at (0,0), ends at (2^n-1,0) and whose first
 *   step is along x -- the traversal drawn in the reference's README.md:93-99.
 *
 * What IS pinned: the delta rule (hilbertc.rs:445-477, README.md:166-173 worked example).
 */
#include "cniic_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int64_t sgn64(int64_t v) { return (v > 0) - (v < 0); }
static inline int64_t abs64(int64_t v) { return v < 0 ? -v : v; }
static inline int64_t floordiv2(int64_t v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }

typedef struct {
    uint32_t *xy;
    uint64_t  pos;
} gen_t;

static void generate2d(gen_t *g, int64_t x, int64_t y, int64_t ax, int64_t ay, int64_t bx, int64_t by) {
    int64_t w = abs64(ax + ay), h = abs64(bx + by);
    int64_t dax = sgn64(ax), day = sgn64(ay), dbx = sgn64(bx), dby = sgn64(by);
    if (h == 1) {
        for (int64_t i = 0; i < w; i++) {
            g->xy[2 * g->pos] = (uint32_t)x; g->xy[2 * g->pos + 1] = (uint32_t)y; g->pos++;
            x += dax; y += day;
        }
        return;
    }
    if (w == 1) {
        for (int64_t i = 0; i < h; i++) {
            g->xy[2 * g->pos] = (uint32_t)x; g->xy[2 * g->pos + 1] = (uint32_t)y; g->pos++;
            x += dbx; y += dby;
        }
        return;
    }
    int64_t ax2 = floordiv2(ax), ay2 = floordiv2(ay), bx2 = floordiv2(bx), by2 = floordiv2(by);
    int64_t w2 = abs64(ax2 + ay2), h2 = abs64(bx2 + by2);
    if (2 * w > 3 * h) {
        if ((w2 & 1) && w > 2) { ax2 += dax; ay2 += day; }
        generate2d(g, x, y, ax2, ay2, bx, by);
        generate2d(g, x + ax2, y + ay2, ax - ax2, ay - ay2, bx, by);
    } else {
        if ((h2 & 1) && h > 2) { bx2 += dbx; by2 += dby; }
        generate2d(g, x, y, bx2, by2, ax2, ay2);
        generate2d(g, x + bx2, y + by2, ax, ay, bx - bx2, by - by2);
        generate2d(g, x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby),
                   -bx2, -by2, -(ax - ax2), -(ay - ay2));
    }
}

/* hilbert.rs:40-43 iter(xdim, ydim) */
int orc_hilbert_iter(uint32_t w, uint32_t h, uint32_t *xy) {
    if (w == 0 || h == 0) return ORC_OK; /* empty scan */
    gen_t g = { xy, 0 };
    if (w >= h) generate2d(&g, 0, 0, w, 0, 0, h);
    else generate2d(&g, 0, 0, 0, h, w, 0);
    return g.pos == (uint64_t)w * h ? ORC_OK : ORC_ERR_BAD_ARG;
}

/* random access into the same scan (descends the recursion instead of unrolling it) */
void orc_hilbert_d2xy(uint32_t w0, uint32_t h0, uint64_t d0, uint32_t *xo, uint32_t *yo) {
    int64_t x = 0, y = 0, ax, ay, bx, by, d = (int64_t)d0;
    if (w0 >= h0) { ax = w0; ay = 0; bx = 0; by = h0; }
    else { ax = 0; ay = h0; bx = w0; by = 0; }
    for (;;) {
        int64_t w = abs64(ax + ay), h = abs64(bx + by);
        int64_t dax = sgn64(ax), day = sgn64(ay), dbx = sgn64(bx), dby = sgn64(by);
        if (h == 1) { x += dax * d; y += day * d; break; }
        if (w == 1) { x += dbx * d; y += dby * d; break; }
        int64_t ax2 = floordiv2(ax), ay2 = floordiv2(ay), bx2 = floordiv2(bx), by2 = floordiv2(by);
        int64_t w2 = abs64(ax2 + ay2), h2 = abs64(bx2 + by2);
        if (2 * w > 3 * h) {
            if ((w2 & 1) && w > 2) { ax2 += dax; ay2 += day; }
            int64_t n1 = abs64(ax2 + ay2) * h;
            if (d < n1) { ax = ax2; ay = ay2; }
            else { d -= n1; x += ax2; y += ay2; ax -= ax2; ay -= ay2; }
        } else {
            if ((h2 & 1) && h > 2) { bx2 += dbx; by2 += dby; }
            int64_t hh = abs64(bx2 + by2);
            int64_t n1 = hh * w2;
            int64_t n2 = w * (h - hh);
            if (d < n1) {
                ax = bx2; ay = by2; bx = ax2; by = ay2;
            } else if (d < n1 + n2) {
                d -= n1; x += bx2; y += by2; bx -= bx2; by -= by2;
            } else {
                d -= n1 + n2;
                x += (ax - dax) + (bx2 - dbx);
                y += (ay - day) + (by2 - dby);
                int64_t nbx = -(ax - ax2), nby = -(ay - ay2);
                ax = -bx2; ay = -by2; bx = nbx; by = nby;
            }
        }
    }
    *xo = (uint32_t)x;
    *yo = (uint32_t)y;
}

/* hilbert.rs:10-12,34-38 linearize: map the scan through get_pixel(x,y) */
int orc_hilbert_linearize(const uint8_t *rgb, uint32_t w, uint32_t h, uint8_t *out) {
    uint64_t n = (uint64_t)w * h;
    if (n == 0) return ORC_OK;
    uint32_t *xy = (uint32_t *)malloc(n * 2 * sizeof(uint32_t));
    if (!xy) return ORC_ERR_NOMEM;
    int rc = orc_hilbert_iter(w, h, xy);
    if (!rc)
        for (uint64_t i = 0; i < n; i++) {
            const uint8_t *px = rgb + ((uint64_t)xy[2 * i + 1] * w + xy[2 * i]) * 3;
            out[3 * i] = px[0]; out[3 * i + 1] = px[1]; out[3 * i + 2] = px[2];
        }
    free(xy);
    return rc;
}

/* hilbertc.rs:449-477 DiffStream; START = [0;3] (hilbertc.rs:445) */
int orc_delta_diff(const uint8_t *rgb_lin, uint64_t n, uint32_t *syms) {
    int16_t last[3] = { 0, 0, 0 };
    for (uint64_t i = 0; i < n; i++) {
        uint32_t key = 0;
        for (int c = 0; c < 3; c++) {
            int16_t cur = (int16_t)rgb_lin[3 * i + c];      /* From<Rgb<u8>> hilbertc.rs:518-525 */
            int16_t diff = (int16_t)(cur - last[c]);        /* Sub hilbertc.rs:539-548 */
            last[c] = cur;
            key = (key << 9) | (uint32_t)(diff + 255);
        }
        syms[i] = key;
    }
    return ORC_OK;
}

/* hilbertc.rs:482-509 FromDiff; try_into u8 failure = the unwrap() panic at 505-506 */
int orc_delta_undiff(const uint32_t *syms, uint64_t n, uint8_t *rgb_lin) {
    int16_t last[3] = { 0, 0, 0 };
    for (uint64_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            int16_t diff = (int16_t)((int)((syms[i] >> (18 - 9 * c)) & 511) - 255);
            int16_t v = (int16_t)(last[c] + diff);          /* Add hilbertc.rs:550-559 */
            if (v < 0 || v > 255) return ORC_ERR_DECODE;
            last[c] = v;
            rgb_lin[3 * i + c] = (uint8_t)v;
        }
    }
    return ORC_OK;
}
