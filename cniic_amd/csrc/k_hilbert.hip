// k_hilbert.hip -- Hilbert-order traversal + neighbour delta on gfx950
// (reference: src/hilbert.rs:34-43 iter/linearize, src/codec/hilbertc.rs:445-477 DiffStream,
//  hilbertc.rs:417-431 Delta::decode scatter).
//
// The reference walks an iterator (zhang_hilbert::ArbHilbertScan32) one cell at a time.  Here the
// scan is a pure function d -> (x, y): every thread descends the recursion of the generalised
// Hilbert curve for its own indices, so the index map, the pixel gather, the delta and the
// histogram are all data-parallel.  The scan definition is the one frozen in DESIGN.md ("Hilbert
// scan: parity unpinned"): identical to the classic Hilbert curve on 2^n squares.
#include "common.hpp"
#include "device_utils.hpp"

namespace cniic {

__device__ __forceinline__ int32_t sgn32(int32_t v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ int32_t floordiv2(int32_t v) { return v >> 1; }  // arithmetic shift = floor

// position d of the scan of a w x h rectangle (d < w*h < 2^32, sides < 2^31)
__device__ __forceinline__ void gilbert_d2xy(uint32_t w0, uint32_t h0, uint64_t d0, uint32_t &xo, uint32_t &yo) {
    int32_t x = 0, y = 0, ax, ay, bx, by;
    int64_t d = (int64_t)d0;
    if (w0 >= h0) { ax = (int32_t)w0; ay = 0; bx = 0; by = (int32_t)h0; }
    else { ax = 0; ay = (int32_t)h0; bx = (int32_t)w0; by = 0; }
    for (;;) {
        const int32_t w = abs(ax + ay), h = abs(bx + by);
        const int32_t dax = sgn32(ax), day = sgn32(ay), dbx = sgn32(bx), dby = sgn32(by);
        if (h == 1) { x += dax * (int32_t)d; y += day * (int32_t)d; break; }
        if (w == 1) { x += dbx * (int32_t)d; y += dby * (int32_t)d; break; }
        int32_t ax2 = floordiv2(ax), ay2 = floordiv2(ay), bx2 = floordiv2(bx), by2 = floordiv2(by);
        const int32_t w2 = abs(ax2 + ay2), h2 = abs(bx2 + by2);
        if (2 * (int64_t)w > 3 * (int64_t)h) {  // long rectangle: two halves
            if ((w2 & 1) && w > 2) { ax2 += dax; ay2 += day; }
            const int64_t n1 = (int64_t)abs(ax2 + ay2) * h;
            if (d < n1) { ax = ax2; ay = ay2; }
            else { d -= n1; x += ax2; y += ay2; ax -= ax2; ay -= ay2; }
        } else {  // up, across, down
            if ((h2 & 1) && h > 2) { bx2 += dbx; by2 += dby; }
            const int32_t hh = abs(bx2 + by2);
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
    xo = (uint32_t)x;
    yo = (uint32_t)y;
}

// hilbert::iter (hilbert.rs:40-43)
__global__ __launch_bounds__(256) void k_hilbert_xy(uint32_t w, uint32_t h, uint32_t *__restrict__ xy) {
    const uint64_t n = (uint64_t)w * h;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += stride) {
        uint32_t x, y;
        gilbert_d2xy(w, h, d, x, y);
        reinterpret_cast<uint2 *>(xy)[d] = make_uint2(x, y);
    }
}

// hilbert::linearize (hilbert.rs:10-12, 34-38): out[d] = pixel(scan(d)); SCATTER = inverse
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_hilbert_move(const uint8_t *__restrict__ src, uint32_t w, uint32_t h,
                                                      uint8_t *__restrict__ dst) {
    const uint64_t n = (uint64_t)w * h;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += stride) {
        uint32_t x, y;
        gilbert_d2xy(w, h, d, x, y);
        const uint64_t p = (uint64_t)y * w + x;
        const uint8_t *s = src + 3 * (SCATTER ? d : p);
        uint8_t *o = dst + 3 * (SCATTER ? p : d);
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
    }
}

// DiffStream (hilbertc.rs:449-477) over the Hilbert-ordered pixels, START = [0;3] (hilbertc.rs:445).
// Each thread owns 4 consecutive scan positions (one extra d2xy for the predecessor of the first).
// The packed SignedColor key goes to syms (16-B store per thread) and/or into the dense histogram.
constexpr int kDeltaRun = 4;
__global__ __launch_bounds__(256) void k_hilbert_delta(const uint8_t *__restrict__ rgb, uint32_t w, uint32_t h,
                                                       uint32_t *__restrict__ syms, uint32_t *__restrict__ table) {
    const uint64_t n = (uint64_t)w * h;
    const uint64_t nruns = (n + kDeltaRun - 1) / kDeltaRun;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t run = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; run < nruns; run += stride) {
        const uint64_t d0 = run * kDeltaRun;
        int32_t pr = 0, pg = 0, pb = 0;
        if (d0 > 0) {
            uint32_t x, y;
            gilbert_d2xy(w, h, d0 - 1, x, y);
            const uint8_t *p = rgb + 3 * ((uint64_t)y * w + x);
            pr = p[0]; pg = p[1]; pb = p[2];
        }
        uint32_t key[kDeltaRun];
#pragma unroll
        for (int i = 0; i < kDeltaRun; i++) {
            key[i] = 0;
            if (d0 + i < n) {
                uint32_t x, y;
                gilbert_d2xy(w, h, d0 + i, x, y);
                const uint8_t *p = rgb + 3 * ((uint64_t)y * w + x);
                const int32_t r = p[0], g = p[1], b = p[2];
                key[i] = ((uint32_t)(r - pr + 255) << 18) | ((uint32_t)(g - pg + 255) << 9) | (uint32_t)(b - pb + 255);
                pr = r; pg = g; pb = b;
                if (table) atomicAdd(&table[key[i]], 1u);
            }
        }
        if (syms) {
            if (d0 + kDeltaRun <= n && (reinterpret_cast<uintptr_t>(syms) & 15) == 0)
                reinterpret_cast<uint4 *>(syms)[run] = make_uint4(key[0], key[1], key[2], key[3]);
            else
                for (int i = 0; i < kDeltaRun && d0 + i < n; i++) syms[d0 + i] = key[i];
        }
    }
}

static inline uint32_t hgrid(uint64_t items) {
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(items, 256), 1), 256 * 16);
}

static int check_dims(Ctx *c, uint32_t w, uint32_t h) {
    if (w >= (1u << 30) || h >= (1u << 30) || (uint64_t)w * h >= (1ull << 32))
        return c->fail(CNIIC_ERR_BAD_ARG, "hilbert: image %ux%u too large", w, h);
    return CNIIC_OK;
}

int hilbert_xy(Ctx *c, uint32_t w, uint32_t h, uint32_t *xy_d) {
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    hipLaunchKernelGGL(k_hilbert_xy, dim3(hgrid(n)), dim3(256), 0, c->stream, w, h, xy_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int hilbert_linearize(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint8_t *out_d) {
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    hipLaunchKernelGGL(k_hilbert_move<false>, dim3(hgrid(n)), dim3(256), 0, c->stream, rgb_d, w, h, out_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int hilbert_scatter(Ctx *c, const uint8_t *lin_d, uint32_t w, uint32_t h, uint8_t *rgb_out_d) {
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    hipLaunchKernelGGL(k_hilbert_move<true>, dim3(hgrid(n)), dim3(256), 0, c->stream, lin_d, w, h, rgb_out_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    return CNIIC_OK;
}

int hilbert_delta(Ctx *c, const uint8_t *rgb_d, uint32_t w, uint32_t h, uint32_t *syms_d, uint32_t *table_d) {
    CNIIC_TRY(check_dims(c, w, h));
    const uint64_t n = (uint64_t)w * h;
    if (!n) return CNIIC_OK;
    ScopedKernelTimer timer(c, "hilbert_delta");
    hipLaunchKernelGGL(k_hilbert_delta, dim3(hgrid(ceil_div(n, kDeltaRun))), dim3(256), 0, c->stream, rgb_d, w, h, syms_d, table_d);
    CNIIC_HIP_TRY(c, hipGetLastError());
    timer.stop(1);
    return CNIIC_OK;
}

}  // namespace cniic
