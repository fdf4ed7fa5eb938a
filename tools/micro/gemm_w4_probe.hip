// Micro-probe (development aid, not part of the library): does a 256x256x64 GEMM tile run by FOUR waves of 128x128 (one
// wave per SIMD, 256 accumulator registers each, 0.25 KiB of LDS fragment reads per MFMA) hold a higher clock on random
// operands than the library's EIGHT waves of 128x64 (two per SIMD, 0.375 KiB per MFMA)?  Main loop only: both sides run
// without an epilogue (the library through its -DMMR_GEMM_NOEPI diagnostic build), on shapes whose tile count is a multiple
// of 256 so that neither pays tile quantisation, alternating in one process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I<csrc> tools/micro/gemm_w4_probe.hip -o gpurun_out/gemm_w4_probe -ldl
//   gpurun_out/gemm_w4_probe tools/_ab/lib_noepi.so
#include "mmr_common.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using namespace mmr;

constexpr int BK = 64;
constexpr int IMG = 256 * BK * 2;        // one operand image: 256 rows x 64 bf16 = 32 KiB
constexpr int STAGE = 2 * IMG;           // A image, W image
constexpr int LDS_BYTES = 2 * STAGE;     // two stages: 128 KiB

__device__ __forceinline__ int tile_off(int row, int c) { return (row >> 3) * 1024 + (row & 7) * 128 + ((c ^ (row & 7)) << 4); }

template <bool STORE>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const bf16_t *__restrict__ A, const bf16_t *__restrict__ W, int M, int N,
                                                      int K, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int gn = N / 256;
    // XCD-contiguous order, column-major inside groups of 4 row panels (the library's order)
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, qd = nblk >> 3, rm = nblk & 7, xcd = bid & 7;
        bid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    }
    const int panels = M / 256, pg = 4, per_group = pg * gn;
    const int grp = bid / per_group, rem = bid - grp * per_group, rows_here = min(pg, panels - grp * pg);
    const int m0 = (grp * pg + rem % rows_here) * 256, n0 = (rem / rows_here) * 256;

    const int rr = lane >> 3, sc = (lane & 7) ^ rr;
    const uint32_t lane_off = (uint32_t)((rr * K + sc * 8) * 2);
    const int nkt = K / BK;
    auto stage = [&](int kt, int buf) {
        if (kt >= nkt) return;
        char *dst = smem + buf * STAGE;
        const char *ba = reinterpret_cast<const char *>(A) + ((size_t)(m0 + wave * 64) * K + (size_t)kt * BK) * 2;
        const char *bw = reinterpret_cast<const char *>(W) + ((size_t)(n0 + wave * 64) * K + (size_t)kt * BK) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(ba + (size_t)(j * 8) * K * 2 + lane_off, dst + (wave * 8 + j) * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(bw + (size_t)(j * 8) * K * 2 + lane_off, dst + IMG + (wave * 8 + j) * 1024);
    };

    f32x4 acc[8][8];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;

    bf16x8 af[2][8], wf[2][8];        // [ks][mi / ni]
    auto read_frags = [&](int buf, int ks) {
        const char *ta = smem + buf * STAGE, *tw = ta + IMG;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            af[ks][i] = *reinterpret_cast<const bf16x8 *>(ta + tile_off(wr * 128 + i * 16 + fr, ks * 4 + fg));
            wf[ks][i] = *reinterpret_cast<const bf16x8 *>(tw + tile_off(wc * 128 + i * 16 + fr, ks * 4 + fg));
        }
    };
    // MFMAs of one 32-deep k-step, rows [ni0, ni1) of the 8 x 8 accumulator grid
    auto mma = [&](int ks, int ni0, int ni1) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
            if (ni >= ni0 && ni < ni1)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][ni], af[ks][mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    stage(0, 0);
    stage(1, 1);
    if (nkt > 1) wait_vmcnt<16>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0);
    // Each half of a K-tile: the first 8 MFMAs go out BEFORE the next fragments are requested -- hipcc's own wait in front of
    // the first use of a fragment can only be lgkmcnt(0) (the counter has 4 bits), and placed there it finds nothing pending
    // (the fragments were requested 56 MFMAs ago) instead of waiting for the 32 reads issued just before it.
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        mma(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1, 8);
        __builtin_amdgcn_sched_barrier(0);
        // K-tile kt+1 (staged one tile ago) has landed for everyone; nobody reads stage `buf` any more
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage(kt + 2, buf);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nkt) read_frags(buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1, 8);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (STORE) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = m0 + wr * 128 + mi * 16 + fr, n = n0 + wc * 128 + ni * 16 + fg * 4;
                *reinterpret_cast<float4 *>(out + (size_t)m * N + n) =
                    make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
            }
    } else {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 123.456f) out[0] = keep;
    }
}

__global__ void naive_kernel(const bf16_t *A, const bf16_t *W, int M, int N, int K, float *out)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += bf16_to_f32(A[(size_t)m * K + k]) * bf16_to_f32(W[(size_t)n * K + k]);
    out[(size_t)m * N + n] = s;
}

__global__ void fill_kernel(bf16_t *p, size_t n, uint32_t seed, float scale)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float u = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.f;
    p[i] = f32_to_bf16(u * scale);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int (*debug_gemm_fn)(int, const void *, const void *, int, int, int, const void *, void *, void *);

int main(int argc, char **argv)
{
    debug_gemm_fn lib_gemm = nullptr;
    if (argc > 1) {
        void *h = dlopen(argv[1], RTLD_NOW);
        if (!h) { printf("dlopen %s: %s\n", argv[1], dlerror()); return 1; }
        lib_gemm = (debug_gemm_fn)dlsym(h, "mmr_debug_gemm");
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_w4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_w4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    // ---- correctness on a small shape
    {
        const int M = 512, N = 512, K = 256;
        bf16_t *A, *W; float *o1, *o2;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2));
        CK(hipMalloc(&o1, (size_t)M * N * 4)); CK(hipMalloc(&o2, (size_t)M * N * 4));
        fill_kernel<<<(M * K + 255) / 256, 256>>>(A, (size_t)M * K, 1, 1.f);
        fill_kernel<<<(N * K + 255) / 256, 256>>>(W, (size_t)N * K, 2, 1.f);
        gemm_w4_kernel<true><<<(M / 256) * (N / 256), 256, LDS_BYTES>>>(A, W, M, N, K, o1);
        naive_kernel<<<dim3((N + 255) / 256, M), 256>>>(A, W, M, N, K, o2);
        CK(hipDeviceSynchronize());
        std::vector<float> h1((size_t)M * N), h2((size_t)M * N);
        CK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost));
        double md = 0;
        for (size_t i = 0; i < h1.size(); ++i) { double d = fabs((double)h1[i] - h2[i]); if (d > md) md = d; }
        printf("check 512x512x256: max |w4 - naive| = %.3e (values ~ +-%.1f)\n", md, sqrt(256.0 / 9));
        hipFree(A); hipFree(W); hipFree(o1); hipFree(o2);
        if (!(md < 1e-2)) { printf("MISMATCH\n"); return 1; }
    }
    struct Shape { const char *name; int M, N, K; } shapes[] = {
        {"256 tiles, K=768", 16384, 1024, 768}, {"256 tiles, K=3072", 16384, 1024, 3072},
        {"512 tiles, K=768", 32768, 1024, 768}, {"512 tiles, K=1024", 32768, 1024, 1024}};
    for (const Shape &s : shapes) {
        bf16_t *A, *W, *ob; float *o, *bias;
        CK(hipMalloc(&A, (size_t)s.M * s.K * 2)); CK(hipMalloc(&W, (size_t)s.N * s.K * 2));
        CK(hipMalloc(&o, (size_t)s.M * s.N * 4)); CK(hipMalloc(&ob, (size_t)s.M * s.N * 2)); CK(hipMalloc(&bias, s.N * 4));
        CK(hipMemset(bias, 0, s.N * 4));
        fill_kernel<<<(unsigned)(((size_t)s.M * s.K + 255) / 256), 256>>>(A, (size_t)s.M * s.K, 3, 0.5f);
        fill_kernel<<<(unsigned)(((size_t)s.N * s.K + 255) / 256), 256>>>(W, (size_t)s.N * s.K, 4, 0.05f);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int tiles = (s.M / 256) * (s.N / 256), iters = 200;
        const double flop = 2.0 * s.M * s.N * s.K;
        for (int rep = 0; rep < 3; ++rep) {
            float ms_new = 0, ms_lib = 0;
            for (int i = 0; i < 20; ++i) gemm_w4_kernel<false><<<tiles, 256, LDS_BYTES>>>(A, W, s.M, s.N, s.K, o);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) gemm_w4_kernel<false><<<tiles, 256, LDS_BYTES>>>(A, W, s.M, s.N, s.K, o);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_new, e0, e1));
            if (lib_gemm) {
                for (int i = 0; i < 20; ++i) lib_gemm(0, A, W, s.M, s.N, s.K, bias, ob, nullptr);
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) lib_gemm(0, A, W, s.M, s.N, s.K, bias, ob, nullptr);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_lib, e0, e1));
            }
            printf("%-18s %dx%dx%d: 4 waves x 128x128 %7.2f us %7.1f TFLOP/s | library main loop (8 waves x 128x64) %7.2f us %7.1f TFLOP/s\n",
                   s.name, s.M, s.N, s.K, ms_new / iters * 1e3, flop / (ms_new / iters * 1e-3) / 1e12, ms_lib / iters * 1e3,
                   ms_lib > 0 ? flop / (ms_lib / iters * 1e-3) / 1e12 : 0.0);
        }
        hipFree(A); hipFree(W); hipFree(o); hipFree(ob); hipFree(bias);
    }
    return 0;
}
