// persist_probe.hip -- the synchronisation skeleton of the persistent colour K-means (k_kmeans_persist.hip), alone and CHECKED:
// G blocks of 1024 threads, one per CU (their LDS request leaves no room for a second); per round every block adds to a
// triple-buffered array of 5K+2 u64 sums with agent-scope atomics, crosses an XCD-hierarchical barrier made of relaxed agent-scope
// atomics only (no fence: what crosses it was added and is read with atomics / sc1 loads), reads ALL the sums back and compares
// every word with the closed-form total; block 0 clears the buffer of the round after next with write-through stores.  Some blocks
// dawdle before they arrive (uneven load: what hides a stale hand-off on an idle chip, MI355X_MICROARCH.md "Test every hand-off").
//   read modes: 0 = 8-byte sc1 loads   1 = returning atomic add of 0   2 = two 4-byte sc1 loads
//   hipcc --offload-arch=gfx950 -O3 -o tools/persist_probe tools/persist_probe.hip && tools/persist_probe [blocks] [rounds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned W = 5 * 256 + 2;   // words of one buffer (K = 256)
constexpr unsigned WP = 1284;         // ... padded to a multiple of 4
struct alignas(128) Line { unsigned int v; unsigned int pad[31]; };
struct Bar {
    Line xcount[8], xgen[8], xblocks[8];
    Line top, topgen, count, gen, abort_;
};

__device__ __forceinline__ unsigned int xcc_id() {
    unsigned int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned long long now100() { return wall_clock64(); }   // 100 MHz

constexpr unsigned long long kTimeoutTicks = 200ull * 1000 * 100;   // 200 ms at 100 MHz: a grid that is not resident ends wrong, not never

// thread 0 only.  false: timed out or somebody aborted
__device__ __forceinline__ bool spin_until_changed(unsigned int *word, unsigned int old, Bar *b) {
    const unsigned long long t0 = now100();
    unsigned int spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == old) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0) {
            if (__hip_atomic_load(&b->abort_.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (now100() - t0 > kTimeoutTicks) { __hip_atomic_store(&b->abort_.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
    }
    return true;
}

__device__ __forceinline__ bool barrier_flat(Bar *b, unsigned int nblocks) {
    bool ok = true;
    const unsigned int g = __hip_atomic_load(&b->gen.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(&b->count.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
        __hip_atomic_store(&b->count.v, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&b->gen.v, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else ok = spin_until_changed(&b->gen.v, g, b);
    return ok;
}
__device__ __forceinline__ bool barrier_xcd(Bar *b, unsigned int x, unsigned int nx_blocks, unsigned int nxcd) {
    bool ok = true;
    const unsigned int g = __hip_atomic_load(&b->xgen[x].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(&b->xcount[x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nx_blocks - 1) {
        __hip_atomic_store(&b->xcount[x].v, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int tg = __hip_atomic_load(&b->topgen.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&b->top.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nxcd - 1) {
            __hip_atomic_store(&b->top.v, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->topgen.v, tg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else ok = spin_until_changed(&b->topgen.v, tg, b);
        __hip_atomic_store(&b->xgen[x].v, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else ok = spin_until_changed(&b->xgen[x].v, g, b);
    return ok;
}

__device__ __forceinline__ unsigned long long val_of(unsigned r, unsigned i) { return (unsigned long long)(r + 1) * (i + 1) + ((unsigned long long)(r * 977u + i) << 33); }

// flags: 1 adds, 2 block 0 clears, 4 reads; dens: one block in 2^dens adds to a word
__global__ __launch_bounds__(1024) void k_probe(Bar *b, unsigned long long *sums, int rounds, int mode, int dawdle, unsigned long long *errs, unsigned long long *tstamp, int flags, int dens) {
    extern __shared__ unsigned int lds[];
    __shared__ unsigned int s_nx, s_nxcd, s_ok;
    const unsigned int x = xcc_id(), G = gridDim.x, tid = threadIdx.x;
    lds[tid] = tid;   // (the request is what matters: one block per CU)
    if (tid == 0) {
        atomicAdd(&b->xblocks[x].v, 1u);
        bool ok = barrier_flat(b, G);
        unsigned int n = 0;
        for (int i = 0; i < 8; i++) n += __hip_atomic_load(&b->xblocks[i].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        s_nx = __hip_atomic_load(&b->xblocks[x].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_nxcd = n;
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) { if (tid == 0) atomicAdd(&errs[1], 1ull); return; }
    unsigned long long bad = 0, ph[4] = {0, 0, 0, 0}, tq = now100();
    const unsigned dm = (1u << dens) - 1u;
    for (int r = 0; r < rounds; r++) {
        unsigned long long *cur = sums + (size_t)(r % 3) * WP, *nxt = sums + (size_t)((r + 1) % 3) * WP;
        if (blockIdx.x == 0 && tid == 0 && tstamp) tstamp[r] = now100();
        // block 0 clears the buffer of round r + 1 (read last in round r - 2, added to again after barrier r + 1): write-through stores
        if (blockIdx.x == 0 && (flags & 2))
            for (unsigned i = tid; i < W; i += 1024) __hip_atomic_store(&nxt[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the round's additions: a quarter of the blocks per word
        if (flags & 1)
            for (unsigned i = tid; i < W; i += 1024)
                if (((blockIdx.x + i + r) & dm) == 0) atomicAdd(&cur[i], val_of(r, i));
        if (dawdle && ((blockIdx.x * 2654435761u + r * 40503u) >> 29) == 0) {   // an eighth of the blocks arrive late, a different eighth every round
            const unsigned long long t0 = now100();
            while (now100() - t0 < (unsigned long long)dawdle) __builtin_amdgcn_s_sleep(8);
        }
        { const unsigned long long n_ = now100(); ph[0] += n_ - tq; tq = n_; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's atomics and stores are out before its block arrives
        __syncthreads();
        { const unsigned long long n_ = now100(); ph[1] += n_ - tq; tq = n_; }
        if (tid == 0) s_ok = barrier_xcd(b, x, s_nx, s_nxcd);
        __syncthreads();
        { const unsigned long long n_ = now100(); ph[2] += n_ - tq; tq = n_; }
        if (!s_ok) { if (tid == 0) atomicAdd(&errs[1], 1ull); return; }
        const unsigned long long cnt = (flags & 1) ? G >> dens : 0;
        if (!(flags & 4)) {
        } else if (mode == 0) {
            for (unsigned i = tid; i < W; i += 1024) {
                const unsigned long long v = __hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad += v != cnt * val_of(r, i);
            }
        } else if (mode == 1) {
            for (unsigned i = tid; i < W; i += 1024) {
                const unsigned long long v = __hip_atomic_fetch_add(&cur[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad += v != cnt * val_of(r, i);
            }
        } else {
            const unsigned int *c32 = reinterpret_cast<const unsigned int *>(cur);
            for (unsigned i = tid; i < 2 * W; i += 1024) {
                const unsigned int v = __hip_atomic_load(&c32[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long e = cnt * val_of(r, i >> 1);
                bad += v != (unsigned int)((i & 1) ? e >> 32 : e);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        { const unsigned long long n_ = now100(); ph[3] += n_ - tq; tq = n_; }
    }
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 77)) for (int i = 0; i < 4; i++) errs[2 + (blockIdx.x ? 4 : 0) + i] = ph[i];
    if (bad) atomicAdd(&errs[0], bad);
    if (blockIdx.x == 0 && tid == 0 && tstamp) tstamp[rounds] = now100();
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, rounds = argc > 2 ? atoi(argv[2]) : 400;
    if (blocks % 4) { printf("blocks must be a multiple of 4\n"); return 1; }
    Bar *b;
    unsigned long long *sums, *errs, *ts;
    CHECK(hipMalloc(&b, sizeof(Bar)));
    CHECK(hipMalloc(&sums, 3 * WP * 8));
    CHECK(hipMalloc(&errs, 128));
    CHECK(hipMalloc(&ts, (rounds + 1) * 8));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const unsigned lds = 120 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_probe, 1024, lds));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("grid %d blocks x 1024 threads, %u B of LDS each; %d fit a CU, %d CUs\n", blocks, lds, occ, prop.multiProcessorCount);
    if (occ * prop.multiProcessorCount < blocks) { printf("grid not resident at once\n"); return 1; }
    const char *names[3] = {"8-byte sc1 loads", "atomic add of 0", "4-byte sc1 loads"};
    struct Cfg { int flags, dens, mode, dawdle; const char *what; };
    const Cfg cfgs[] = {
        {0, 2, 0, 0, "barrier only"}, {1, 2, 0, 0, "adds 1/4"}, {1, 4, 0, 0, "adds 1/16"}, {1, 0, 0, 0, "adds by every block"}, {2, 2, 0, 0, "clears only"},
        {4, 2, 0, 0, "reads only, 8-byte"}, {4, 2, 1, 0, "reads only, atomic"}, {4, 2, 2, 0, "reads only, 4-byte"},
        {7, 2, 0, 0, "all, 1/4, 8-byte"}, {7, 2, 1, 0, "all, 1/4, atomic"}, {7, 0, 0, 0, "all, every block, 8-byte"}, {7, 0, 1, 0, "all, every block, atomic"},
        {7, 0, 0, 500, "all, every block, 8-byte, dawdle 5 us"}, {7, 0, 1, 500, "all, every block, atomic, dawdle 5 us"}, {7, 0, 2, 500, "all, every block, 4-byte, dawdle 5 us"},
    };
    for (const Cfg &cf : cfgs) {
        float best = 1e9f;
        unsigned long long h[16] = {0};
        unsigned long long bad_total = 0, aborted = 0;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipMemsetAsync(b, 0, sizeof(Bar), st));
            CHECK(hipMemsetAsync(sums, 0, 3 * WP * 8, st));
            CHECK(hipMemsetAsync(errs, 0, 128, st));
            CHECK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(1024), lds, st, b, sums, rounds, cf.mode, cf.dawdle, errs, ts, cf.flags, cf.dens);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CHECK(hipMemcpy(h, errs, 128, hipMemcpyDeviceToHost));
            bad_total += h[0]; aborted += h[1];
        }
        printf("%-40s %-18s %.2f us per round; wrong words %llu, aborted %llu | block 0: issue %.2f drain %.2f barrier %.2f reads %.2f | block 77: %.2f %.2f %.2f %.2f\n", cf.what, names[cf.mode], best * 1e3 / rounds, bad_total, aborted,
               h[2] / 100.0 / rounds, h[3] / 100.0 / rounds, h[4] / 100.0 / rounds, h[5] / 100.0 / rounds, h[6] / 100.0 / rounds, h[7] / 100.0 / rounds, h[8] / 100.0 / rounds, h[9] / 100.0 / rounds);
    }
    return 0;
}
