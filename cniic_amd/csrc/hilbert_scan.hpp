// hilbert_scan.hpp -- the scan d -> (x, y) as device code shared by k_hilbert.hip and k_delta.hip
// (reference: src/hilbert.rs:34-43; the curve's definition is frozen in DESIGN.md, "Hilbert scan: parity unpinned").
#pragma once
#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

// ---------------------------------------------------------------- generic rectangles
// the walk inside the rectangle spanned by a = (ax, ay) and b = (bx, by) from its origin: position d -> the offset added to (x, y)
__host__ __device__ inline int32_t gl_abs(int32_t v) { return v < 0 ? -v : v; }
__host__ __device__ inline int32_t gl_sgn(int32_t v) { return (v > 0) - (v < 0); }
__host__ __device__ inline void gilbert_walk(int32_t ax, int32_t ay, int32_t bx, int32_t by, int64_t d, int32_t &x, int32_t &y) {
    for (;;) {
        const int32_t w = gl_abs(ax + ay), h = gl_abs(bx + by);
        const int32_t dax = gl_sgn(ax), day = gl_sgn(ay), dbx = gl_sgn(bx), dby = gl_sgn(by);
        if (h == 1) { x += dax * (int32_t)d; y += day * (int32_t)d; break; }
        if (w == 1) { x += dbx * (int32_t)d; y += dby * (int32_t)d; break; }
        int32_t ax2 = ax >> 1, ay2 = ay >> 1, bx2 = bx >> 1, by2 = by >> 1;   // (arithmetic shift = floor)
        const int32_t w2 = gl_abs(ax2 + ay2), h2 = gl_abs(bx2 + by2);
        if (2 * (int64_t)w > 3 * (int64_t)h) {  // long rectangle: two halves
            if ((w2 & 1) && w > 2) { ax2 += dax; ay2 += day; }
            const int64_t n1 = (int64_t)gl_abs(ax2 + ay2) * h;
            if (d < n1) { ax = ax2; ay = ay2; }
            else { d -= n1; x += ax2; y += ay2; ax -= ax2; ay -= ay2; }
        } else {  // up, across, down
            if ((h2 & 1) && h > 2) { bx2 += dbx; by2 += dby; }
            const int32_t hh = gl_abs(bx2 + by2);
            const int64_t n1 = (int64_t)hh * w2;
            const int64_t n2 = (int64_t)w * (h - hh);
            if (d < n1) {
                ax = bx2; ay = by2; bx = ax2; by = ay2;
            } else if (d < n1 + n2) {
                d -= n1; x += bx2; y += by2; bx -= bx2; by -= by2;
            } else {
                d -= n1 + n2;
                x += (ax - dax) + (bx2 - dbx);
                y += (ay - day) + (by2 - dby);
                const int32_t nbx = -(ax - ax2), nby = -(ay - ay2);
                ax = -bx2; ay = -by2; bx = nbx; by = nby;
            }
        }
    }
}

// position d of the scan of a w x h rectangle (d < w*h < 2^32, sides < 2^31)
__host__ __device__ inline void gilbert_d2xy(uint32_t w0, uint32_t h0, uint64_t d0, uint32_t &xo, uint32_t &yo) {
    int32_t x = 0, y = 0;
    if (w0 >= h0) gilbert_walk((int32_t)w0, 0, 0, (int32_t)h0, (int64_t)d0, x, y);
    else gilbert_walk(0, (int32_t)h0, (int32_t)w0, 0, (int64_t)d0, x, y);
    xo = (uint32_t)x;
    yo = (uint32_t)y;
}

// ---------------------------------------------------------------- large rectangles: the walk's recursion cut into leaves (round 3)
// Thirteen levels of that recursion per position -- 64-bit products, branches, ~1500 instructions -- made every kernel that follows
// the scan on a rectangle that is no 2^n square ALU-bound at 37 ps a pixel (k_delta_gather_any at 4000 x 3000: 0.44 ms; the tile
// kernel of a 2^n square: 1.5 ps a pixel).  The recursion's tree depends on (w, h) alone: its upper levels are walked ONCE per image
// size on the host down to sub-rectangles ("leaves") of at most a few thousand positions -- a leaf = its first position, its origin
// and the pair of vectors that span it -- and the walk inside a leaf depends on those vectors alone, of which an image has a few
// dozen different ones: one table of (dx, dy) per such class, filled by a kernel.  Position d -> idx[d >> shift] names the leaf of
// the span's first position, a step or two forward finds d's, the class table gives the offset: five loads that neighbours share.
constexpr uint32_t kScanLeavesBit = 0x80000000u;   // in a kernel's `order` argument: `lut` points to a ScanLeavesDev
struct ScanLeavesDev {
    const uint32_t *idx;   // [(n >> shift) + 1]: the leaf that holds position j << shift
    const uint32_t *d0;    // [nleaf + 1]: first position of every leaf, then n
    const int4 *rec;       // [nleaf]: x, y of the leaf's origin, z = first entry of its class in lut
    const uint32_t *lut;   // per class: (int16 dx) | (int16 dy) << 16 for every position of a leaf of that class
    uint32_t shift, nleaf;
};

// ---------------------------------------------------------------- 2^n squares: state machine (tables built in k_hilbert.hip)
struct HilbertLut {
    uint16_t l4[4 * 256];  // x:4 | y:4 << 4 | state << 8, four levels per look-up
    uint8_t  l1[4 * 4];    // x:1 | y:1 << 1 | state << 2, one level
};

__device__ __forceinline__ void pow2_d2xy(const uint16_t *l4, const uint8_t *l1, uint32_t order, uint32_t d, uint32_t &xo, uint32_t &yo) {
    uint32_t st = 0, x = 0, y = 0, rem = order;
    while (rem >= 4) {
        const uint32_t e = l4[st * 256 + ((d >> (2 * (rem - 4))) & 255)];
        x = (x << 4) | (e & 15); y = (y << 4) | ((e >> 4) & 15); st = e >> 8; rem -= 4;
    }
    while (rem >= 1) {
        const uint32_t e = l1[st * 4 + ((d >> (2 * (rem - 1))) & 3)];
        x = (x << 1) | (e & 1); y = (y << 1) | ((e >> 1) & 1); st = e >> 2; rem -= 1;
    }
    xo = x; yo = y;
}

// scan position -> pixel; order > 0 selects the table-driven path (w == h == 1 << order)
// ... or an order the caller injected (cniic_ctx_set_scan): position d -> custom[d] = (x, y)
// a thread that walks consecutive positions keeps the leaf it is in (leaves mode; the other modes have nothing to keep)
struct ScanCursor { uint32_t first = 1, end = 0, base = 0; int32_t ox = 0, oy = 0; };
struct Scan {
    uint32_t w, h, order;
    const uint16_t *l4;
    const uint8_t *l1;
    const uint2 *custom;
    ScanLeavesDev lf;      // lf.d0 != null: the leaves of this image size
    __device__ __forceinline__ void xy(uint64_t d, uint32_t &x, uint32_t &y) const {
        if (order) pow2_d2xy(l4, l1, order, (uint32_t)d, x, y);
        else if (lf.d0) {
            const uint32_t dd = (uint32_t)d;
            uint32_t k = lf.idx[dd >> lf.shift];
            while (lf.d0[k + 1] <= dd) k++;
            const int4 r = lf.rec[k];
            const uint32_t e = lf.lut[(uint32_t)r.z + (dd - lf.d0[k])];
            x = (uint32_t)(r.x + (int32_t)(int16_t)(e & 0xffffu));
            y = (uint32_t)(r.y + (int32_t)(int16_t)(e >> 16));
        }
        else if (custom) { const uint2 v = custom[d]; x = v.x; y = v.y; }
        else gilbert_d2xy(w, h, d, x, y);
    }
    __device__ __forceinline__ void xy_seq(ScanCursor &cu, uint64_t d, uint32_t &x, uint32_t &y) const {
        if (order || !lf.d0) { xy(d, x, y); return; }
        const uint32_t dd = (uint32_t)d;
        if (dd < cu.first || dd >= cu.end) {
            uint32_t k = lf.idx[dd >> lf.shift];
            while (lf.d0[k + 1] <= dd) k++;
            const int4 r = lf.rec[k];
            cu.first = lf.d0[k]; cu.end = lf.d0[k + 1]; cu.base = (uint32_t)r.z; cu.ox = r.x; cu.oy = r.y;
        }
        const uint32_t e = lf.lut[cu.base + (dd - cu.first)];
        x = (uint32_t)(cu.ox + (int32_t)(int16_t)(e & 0xffffu));
        y = (uint32_t)(cu.oy + (int32_t)(int16_t)(e >> 16));
    }
};

// `lut`: the state-machine tables when order > 0; with order == 0 it is either null (the generalised curve is computed) or the
// injected table of positions (scan_select below hands the kernels one or the other)
__device__ __forceinline__ Scan load_scan(uint32_t w, uint32_t h, uint32_t order, const HilbertLut *lut, uint16_t *s_l4, uint8_t *s_l1) {
    if (order && !(order & kScanLeavesBit)) {
        for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) s_l4[i] = lut->l4[i];
        if (threadIdx.x < 16) s_l1[threadIdx.x] = lut->l1[threadIdx.x];
        __syncthreads();
    }
    if (order & kScanLeavesBit) return Scan{w, h, 0u, s_l4, s_l1, nullptr, *reinterpret_cast<const ScanLeavesDev *>(lut)};
    return Scan{w, h, order, s_l4, s_l1, order ? nullptr : reinterpret_cast<const uint2 *>(lut), ScanLeavesDev{}};
}

// pixel idx as r | g << 8 | b << 16 (bits 24..31 unspecified) with one load; the buffer's last pixel by bytes
__device__ __forceinline__ uint32_t px_le24(const uint8_t *__restrict__ rgb, uint64_t idx, uint64_t n) {
    if (idx + 1 < n) {
        uint32_t v;
        __builtin_memcpy(&v, rgb + 3 * idx, 4);
        return v;
    }
    const uint8_t *p = rgb + 3 * idx;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

// the three bytes of a pixel at an address of any alignment: a 16-bit store and a byte instead of three bytes
__device__ __forceinline__ void store_px3(uint8_t *o, uint32_t v) {
    if (reinterpret_cast<uintptr_t>(o) & 1) { o[0] = (uint8_t)v; *reinterpret_cast<uint16_t *>(o + 1) = (uint16_t)(v >> 8); }
    else { *reinterpret_cast<uint16_t *>(o) = (uint16_t)v; o[2] = (uint8_t)(v >> 16); }
}

// the context's copy of the tables (built and self-checked once per process); order of a 2^n square (n >= 1), else 0
int hilbert_lut(Ctx *c, const HilbertLut **lut_d);
uint32_t pow2_order(uint32_t w, uint32_t h);
// what the scan kernels of a w x h image are launched with: (order, tables) of the built-in scan, or (0, the injected positions)
// when the context holds an order for exactly these dimensions (cniic_ctx_set_scan); the tile kernels want sel.order >= 6
// korder: what a per-position kernel is handed as `order` -- sel.order, or kScanLeavesBit with arg = the image size's leaves (large
// rectangles that are no 2^n square and have no injected scan)
struct ScanSel { uint32_t order; const HilbertLut *arg; uint32_t korder; };
int scan_select(Ctx *c, uint32_t w, uint32_t h, ScanSel *sel);

}  // namespace cniic
