// Micro-probe (development aid, not part of the library): the rate at which one workgroup per CU can stream operand
// tiles into LDS with global_load_lds, as a function of WHERE the bytes are served from.  Every workgroup walks a ring of
// 64 KiB "K-tiles" (512 rows x 128 B, the operand bytes of one K-step of the 256x256 GEMM tile) laid out back to back;
// the ring's size per XCD decides the level that serves it:
//     <= 4 MiB per XCD  : that XCD's L2 (all 32 CUs of the XCD walk the same ring, offset by their position)
//     <= 256 MiB total  : Infinity Cache
//     larger            : HBM
// Reconciles round 1's "11 TB/s L2->LDS ceiling" (tools/micro/fetch_probe.hip: its footprint was 268 MB read by
// workgroups spread over all eight XCDs, i.e. Infinity-Cache / HBM traffic, not L2 hits) with the guide's L2 figures.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2_fetch_probe.hip -o gpurun_out/l2_fetch_probe && gpurun_out/l2_fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

constexpr int TILE = 64 * 1024;          // bytes per K-tile: 64 wave-instructions of 1 KiB, 8 per wave

// NBUF LDS buffers, PD = NBUF-1 tiles in flight; the wait leaves the youngest PD-1 tiles outstanding
template <int NBUF>
__global__ __launch_bounds__(512) void walk(const char *__restrict__ base, size_t ring_bytes_per_xcd, int tiles_per_wg,
                                            int xcd_private, int *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;       // blocks with equal id mod 8 share an XCD
    const size_t ring_tiles = ring_bytes_per_xcd / TILE;
    const char *ring = base + (xcd_private ? (size_t)xcd * ring_bytes_per_xcd : 0);
    auto stage = [&](int t) {
        if (t >= tiles_per_wg) return;
        // CU `slot` of the XCD starts `slot` tiles into the ring: neighbours touch neighbouring tiles at the same time, like
        // the CUs of an XCD that work on tiles sharing A panels / W tiles
        const char *src = ring + ((size_t)(slot * 7 + t) % ring_tiles) * TILE;
        char *dst = smem + (t % NBUF) * TILE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int inst = wave * 8 + i;
            glds16(src + inst * 1024 + lane * 16, dst + inst * 1024);
        }
    };
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p) stage(p);
    for (int t = 0; t < tiles_per_wg; ++t) {
        stage(t + NBUF - 1);
        if (t + NBUF - 1 < tiles_per_wg) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * 8) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (threadIdx.x == 0 && smem[5] == 77) sink[0] = 1;
}

int main()
{
    const size_t max_bytes = (size_t)2 << 30;
    char *A; int *sink;
    hipMalloc(&A, max_bytes); hipMalloc(&sink, 4);
    hipMemset(A, 1, max_bytes);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int tiles_per_wg = 256;         // 16 MiB streamed per workgroup
    auto run = [&](auto kern, int nbuf, size_t ring_per_xcd, int priv, int wgs, const char *what) {
        const int lds = nbuf * TILE;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, A, ring_per_xcd, tiles_per_wg, priv, sink);
        hipEventRecord(s);
        const int it = 5;
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, A, ring_per_xcd, tiles_per_wg, priv, sink);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        const double us = ms * 1e3 / it, tot = (double)wgs * tiles_per_wg * TILE;
        printf("%-58s %d bufs: %8.1f us  %6.2f TB/s  %6.1f B/ns/CU\n", what, nbuf, us, tot / us / 1e6, tot / us / 1e3 / (wgs < 256 ? wgs : 256));
    };
    struct Case { size_t ring; int priv; const char *what; } cases[] = {
        {(size_t)512 << 10, 1, "ring 0.5 MiB per XCD (L2 hits)"},
        {(size_t)2 << 20, 1, "ring 2 MiB per XCD (L2 hits)"},
        {(size_t)3 << 20, 1, "ring 3 MiB per XCD (L2, near capacity)"},
        {(size_t)8 << 20, 1, "ring 8 MiB per XCD (64 MiB total: Infinity Cache)"},
        {(size_t)24 << 20, 1, "ring 24 MiB per XCD (192 MiB total: Infinity Cache)"},
        {(size_t)2 << 20, 0, "ring 2 MiB shared by all XCDs (L2 hits, 8 copies)"},
        {(size_t)16 << 20, 0, "ring 16 MiB shared by all XCDs (L2 miss, Infinity Cache)"},
        {(size_t)128 << 20, 1, "ring 128 MiB per XCD (1 GiB total: HBM)"},
    };
    for (const auto &c : cases) {
        run(walk<2>, 2, c.ring, c.priv, 256, c.what);
    }
    // one CU alone (no contention): the per-CU ingest ceiling
    run(walk<2>, 2, (size_t)512 << 10, 1, 1, "ONE workgroup alone, ring 0.5 MiB (L2 hits)");
    run(walk<2>, 2, (size_t)512 << 10, 1, 8, "one workgroup per XCD, ring 0.5 MiB (L2 hits)");
    run(walk<2>, 2, (size_t)512 << 10, 1, 64, "8 workgroups per XCD, ring 0.5 MiB (L2 hits)");
    return 0;
}
