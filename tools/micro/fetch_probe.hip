// Micro-probe (development aid, not part of the library): how fast can a CU stream operand tiles L2 -> LDS with
// global_load_lds when a row contributes 128 B (BK=64) vs 64 B (BK=32) per K-tile?  One workgroup per CU walks K.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_probe.hip -o gpurun_out/fetch_probe && gpurun_out/fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// rows x K bf16 matrix, row-major.  Each workgroup (THREADS) streams `rows_per_wg` rows over the whole K in K-tiles of
// BKB bytes per row; per K-tile it issues rows_per_wg*BKB/16/THREADS glds per thread, double-buffered in LDS.
// deeper pipeline: NBUF buffers, wait only for the oldest tile (per_wave glds per tile per wave must equal PW)
template <int BKB, int THREADS, int NBUF, int PW>
__global__ __launch_bounds__(THREADS) void probe_deep(const char *__restrict__ A, int rows_per_wg, int Kbytes, int wgs_per_panel,
                                                      int *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = BKB / 16, RPI = 64 / LPR;
    const int panel = blockIdx.x / wgs_per_panel;
    const char *base = A + (size_t)panel * rows_per_wg * Kbytes;
    const int nkt = Kbytes / BKB;
    const int stage_bytes = rows_per_wg * BKB;
    auto stage = [&](int kt) {
        if (kt >= nkt) return;
        char *dst = smem + (kt % NBUF) * stage_bytes;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int inst = wave * PW + i;
            const int row = inst * RPI + lane / LPR;
            glds16(base + (size_t)row * Kbytes + (size_t)kt * BKB + (lane % LPR) * 16, dst + inst * 1024);
        }
    };
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p) stage(p);
    for (int kt = 0; kt < nkt; ++kt) {
        stage(kt + NBUF - 1);
        if (kt + NBUF - 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (threadIdx.x == 0 && smem[5] == 77) sink[0] = 1;
}

template <int BKB, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const char *__restrict__ A, int rows_per_wg, int Kbytes, int wgs_per_panel,
                                                 int *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = BKB / 16;                 // lanes per row
    constexpr int RPI = 64 / LPR;                 // rows per glds instruction
    const int panel = blockIdx.x / wgs_per_panel; // several workgroups re-read the same rows (L2 reuse, like N-tiles)
    const char *base = A + (size_t)panel * rows_per_wg * Kbytes;
    const int insts = rows_per_wg / RPI;          // glds per K-tile per workgroup
    const int per_wave = insts / (THREADS / 64);
    const int nkt = Kbytes / BKB;
    const int stage_bytes = rows_per_wg * BKB;
    for (int kt = 0; kt < nkt; ++kt) {
        char *dst = smem + (kt & 1) * stage_bytes;
        for (int i = 0; i < per_wave; ++i) {
            const int inst = wave * per_wave + i;
            const int row = inst * RPI + lane / LPR;
            glds16(base + (size_t)row * Kbytes + (size_t)kt * BKB + (lane % LPR) * 16, dst + inst * 1024);
        }
        if (kt >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");   // simple: drain every tile
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && smem[5] == 77) sink[0] = 1;
}

int main()
{
    const int rows_per_wg = 384, K = 768 * 2 /*bytes per row chunk*/;   // operand rows of a 256x128 tile; K=768 bf16
    const int Kbytes = 768 * 2;
    const int wgs = 256 * 2 * 8, wgs_per_panel = 12;
    const int max_rows = 512;                                            // largest rows-per-workgroup any variant uses
    const size_t bytes = (size_t)(wgs / wgs_per_panel + 2) * max_rows * Kbytes;
    char *A; int *sink;
    hipMalloc(&A, bytes); hipMalloc(&sink, 4);
    hipMemset(A, 1, bytes);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    auto run = [&](auto kern, int threads, int bkb, const char *name) {
        const int lds = 2 * rows_per_wg * bkb;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, 0, A, rows_per_wg, Kbytes, wgs_per_panel, sink);
        hipEventRecord(s);
        const int it = 20;
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, 0, A, rows_per_wg, Kbytes, wgs_per_panel, sink);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        const double us = ms * 1e3 / it, tot = (double)wgs * rows_per_wg * Kbytes;
        printf("%-28s lds %3d KB/WG: %8.1f us  %7.2f TB/s  %6.1f B/ns/CU\n", name, lds / 1024, us, tot / us / 1e6, tot / us / 1e3 / 256);
    };
    auto run_deep = [&](auto kern, int threads, int bkb, int nbuf, int rows, const char *name) {
        if (rows > max_rows) { printf("skip %s\n", name); return; }
        const int lds = nbuf * rows * bkb;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, 0, A, rows, Kbytes, wgs_per_panel, sink);
        hipEventRecord(s);
        const int it = 20;
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, 0, A, rows, Kbytes, wgs_per_panel, sink);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        const double us = ms * 1e3 / it, tot = (double)wgs * rows * Kbytes;
        printf("%-40s lds %3d KB/WG: %8.1f us  %7.2f TB/s  %6.1f B/ns/CU\n", name, lds / 1024, us, tot / us / 1e6, tot / us / 1e3 / 256);
    };
    // 512 rows (A 256 + W 256) x 128 B = 64 KB per K-tile: 512 rows / 8 rows per inst = 64 inst / 8 waves = 8 per wave
    run_deep(probe_deep<128, 512, 2, 8>, 512, 128, 2, 512, "256^2 tile: BK=64 2 bufs 512 thr");
    // 384 rows x 128 B, 4 waves: 48 inst / 4 = 12 per wave; 3 bufs = 144 KB (1 WG/CU), 2 bufs = 96 KB
    run_deep(probe_deep<128, 256, 2, 12>, 256, 128, 2, 384, "256x128 tile: BK=64 2 bufs 256 thr");
    // 384 rows x 64 B: 24 inst / 4 waves = 6 per wave; 3 bufs = 72 KB (2 WGs/CU)
    run_deep(probe_deep<64, 256, 3, 6>, 256, 64, 3, 384, "256x128 tile: BK=32 3 bufs 256 thr");
    run_deep(probe_deep<64, 256, 2, 6>, 256, 64, 2, 384, "256x128 tile: BK=32 2 bufs 256 thr");
    run(probe<128, 256>, 256, 128, "BK=64 (128 B/row) 256 thr");
    run(probe<64, 256>, 256, 64, "BK=32 (64 B/row) 256 thr");
    run(probe<128, 512>, 512, 128, "BK=64 (128 B/row) 512 thr");
    run(probe<64, 512>, 512, 64, "BK=32 (64 B/row) 512 thr");
    return 0;
}
