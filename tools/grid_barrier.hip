// grid_barrier.hip -- what does a device-side grid barrier cost on this chip, for the grid the K-means assign kernel uses
// (768 blocks x 512 threads, three blocks per CU, all resident)?  VERDICT r1 item 3 proposes running the skip-schedule tail as
// one persistent launch; that pays only if a barrier is cheaper than the kernel boundary it replaces.  Measured here:
//   flat      one counter, one generation word: every block adds 1, the last one bumps the generation, the rest spin on it
//   xcd       the same in two levels: a counter per XCD (hardware XCC_ID), the last block of an XCD reports to the top counter
//   launches  the same grid as back-to-back launches of a kernel that does nothing (the boundary)
// Each barrier variant also runs with a 10 KiB all-block read of data written before the barrier (what the folded-in centroid
// update needs: every block reads the K partial sums that all blocks added to before the barrier).
//   hipcc --offload-arch=gfx950 -O3 -o tools/grid_barrier tools/grid_barrier.hip && tools/grid_barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#ifdef NOFENCE
#define FENCE do {} while (0)  // (what crosses the barrier is added and read with atomics only: nothing to write back)
#else
#define FENCE __threadfence()
#endif
#ifndef SLEEP
#define SLEEP 1
#endif
constexpr int kSleep = SLEEP;  // x 64 clocks between polls
// (the polls and the arrival are relaxed atomics at agent scope; the __threadfence() before and after carries the ordering --
// an acquire on every poll invalidates the cache every time and the barrier takes 75 us)
constexpr unsigned int kMaxSpins = 1u << 18;  // every wait gives up after ~10 ms: a grid that is not resident ends wrong, not never

// Round 4: every counter and every generation word on a 128-byte line of its own.  The first version kept all of them in
// ONE line -- every arrival and every poll of every level hit the same line, and its "xcd" figure (21 us) said more about that line than
// about barriers (MI355X_MICROARCH.md, barrier-xcd: 4.1 / 5.9 / 9.7 us at 256 / 512 / 1024 workgroups).
struct alignas(128) Line { unsigned int v; unsigned int pad[31]; };
struct Bar {
    Line count, gen;          // flat
    Line xcount[8], xgen[8];  // per XCD
    Line xblocks[8];          // blocks resident on each XCD (counted in the first round)
    Line top, topgen;
};

__device__ __forceinline__ unsigned int xcc_id() {
    unsigned int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__device__ __forceinline__ void barrier_flat(Bar *b, unsigned int nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        FENCE;
        const unsigned int g = __hip_atomic_load(&b->gen.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&b->count.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(&b->count.v, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->gen.v, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (unsigned int spins = 0; __hip_atomic_load(&b->gen.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g && spins < kMaxSpins; spins++) __builtin_amdgcn_s_sleep(kSleep);
        }
        FENCE;
    }
    __syncthreads();
}

__device__ __forceinline__ void barrier_xcd(Bar *b, unsigned int x, unsigned int nx_blocks, unsigned int nxcd) {
    __syncthreads();
    if (threadIdx.x == 0) {
        FENCE;
        const unsigned int g = __hip_atomic_load(&b->xgen[x].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&b->xcount[x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nx_blocks - 1) {
            __hip_atomic_store(&b->xcount[x].v, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int tg = __hip_atomic_load(&b->topgen.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&b->top.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nxcd - 1) {
                __hip_atomic_store(&b->top.v, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&b->topgen.v, tg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (unsigned int spins = 0; __hip_atomic_load(&b->topgen.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tg && spins < kMaxSpins; spins++) __builtin_amdgcn_s_sleep(kSleep);
            }
            __hip_atomic_store(&b->xgen[x].v, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (unsigned int spins = 0; __hip_atomic_load(&b->xgen[x].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g && spins < kMaxSpins; spins++) __builtin_amdgcn_s_sleep(kSleep);
        }
        FENCE;
    }
    __syncthreads();
}

// mode 0 flat, 1 per XCD; with_data: between barriers every block adds to 1280 words and reads them all back after the barrier
__global__ __launch_bounds__(512) void k_loop(Bar *b, int mode, int rounds, int with_data, unsigned long long *data, unsigned long long *sink) {
    const unsigned int x = xcc_id();
    __shared__ unsigned int s_nx, s_nxcd;
    if (mode == 1) {  // census: how many blocks does my XCD hold, how many XCDs are in use?
        if (threadIdx.x == 0) atomicAdd(&b->xblocks[x].v, 1u);
        barrier_flat(b, gridDim.x);
        if (threadIdx.x == 0) {
            unsigned int n = 0;
            for (int i = 0; i < 8; i++) n += __hip_atomic_load(&b->xblocks[i].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            s_nx = __hip_atomic_load(&b->xblocks[x].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_nxcd = n;
        }
        __syncthreads();
    }
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; r++) {
        if (with_data)
            for (unsigned int i = threadIdx.x; i < 1280; i += blockDim.x)
                if (((i + blockIdx.x + r) & 15) == 0) atomicAdd(&data[(r & 1) * 1280 + i], 1ull);
        if (mode == 0) barrier_flat(b, gridDim.x);
        else barrier_xcd(b, x, s_nx, s_nxcd);
        if (with_data) {
            for (unsigned int i = threadIdx.x; i < 1280; i += blockDim.x)
                acc += __hip_atomic_load(&data[(r & 1) * 1280 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

__global__ __launch_bounds__(512) void k_nothing(unsigned long long *sink) {
    if (sink[1] == 0x1234567ull) sink[0] = 1;
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 768, rounds = 200;
    Bar *b;
    unsigned long long *data, *sink;
    CHECK(hipMalloc(&b, sizeof(Bar)));
    CHECK(hipMalloc(&data, 2 * 1280 * 8));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(sink, 0, 64));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_loop, 512, 0));
    printf("grid %d blocks x 512 threads; %d blocks of k_loop fit a CU\n", blocks, occ);
    if (occ * 256 < blocks) { printf("grid not resident at once: barrier would deadlock\n"); return 1; }
    for (int mode = 0; mode < 2; mode++)
        for (int wd = 0; wd < 2; wd++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipMemsetAsync(b, 0, sizeof(Bar), st));
                CHECK(hipMemsetAsync(data, 0, 2 * 1280 * 8, st));
                CHECK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(k_loop, dim3(blocks), dim3(512), 0, st, b, mode, rounds, wd, data, sink);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%-5s barrier%s: %.2f us per round (%d rounds, best of 5)\n", mode ? "xcd" : "flat", wd ? " + 10 KiB of sums added before / read after" : "", best * 1e3 / rounds, rounds);
        }
    {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0, st));
            for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_nothing, dim3(blocks), dim3(512), 0, st, sink);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("launches: %.2f us per launch of a kernel that does nothing (%d back to back, best of 5)\n", best * 1e3 / rounds, rounds);
    }
    return 0;
}
