// Shared host/device helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mmr.h"

namespace mmr {

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; NaN stays NaN (a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// two floats -> one dword of two bf16 (lo in bits 0..15): the vector cast lowers to ONE v_cvt_pk_bf16_f32; converting
// the halves separately and OR-ing them costs three VALU instructions per pair
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
    typedef float f32x2_v __attribute__((ext_vector_type(2)));
    const f32x2_v f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_v));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Fixed-order fp64 butterfly: identical in every lane, and identical to oracle/search_ref.c.
__device__ __forceinline__ double wave_sum_f64_butterfly(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// sum over each aligned group of 16 lanes (a DPP "row"), result in every lane of the group:
// quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- plain VALU, no LDS round trip
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

// Per-launch extras of the LayerNorm-folding GEMM epilogues (gemm.hip).  Row statistics travel as
// LNFOLD_NP partial (sum, sum of squares) slots per row: a producer whose row is covered by S column slabs
// (one per wave column) fills slot `slab` and zeroes slots slab+S, slab+2S, ...; the consumer adds all
// LNFOLD_NP slots in a fixed order, so results do not depend on workgroup scheduling or on the tile width
// the producer happened to run with.
constexpr int LNFOLD_NP = 16;
struct GemmAux {
    const float *colsum;     // LNFOLD: c[n] = sum_k W'[n,k] of the gamma-folded weight
    const float2 *stats_in;  // LNFOLD: [M][LNFOLD_NP] partial (sum, sumsq) of the LayerNorm input rows
    float2 *stats_out;       // RESID_STATS: [M][LNFOLD_NP] partials of the new residual rows
    bf16_t *xout;            // RESID_STATS: bf16 copy of the new residual rows
    float inv_d, eps;        // 1 / width, LayerNorm epsilon
    // PATCH form of the 256-row kernel (patch-embed conv as GEMM, patch 32): A is not a matrix in memory but gathered
    // from bf16 NCHW pixels: row = (image b, patch gy, gx), k = (channel, ky, kx)
    const bf16_t *pix;       // [B,3,S,S]
    int pix_size, pix_grid;  // S, G = S / 32
    int patch_rows;          // B * G * G: rows past it (tile padding) re-read the last patch
};

__device__ __forceinline__ void emit_row_partial(float2 *stats, size_t row, int slab, int nslabs, float s, float ss)
{
    float2 *p = stats + row * LNFOLD_NP;
    p[slab] = make_float2(s, ss);
    for (int j = slab + nslabs; j < LNFOLD_NP; j += nslabs) p[j] = make_float2(0.f, 0.f);
}

// LNFOLD prologue, two threads per tile row (blockDim = 2 * tile rows): thread t adds slots t&1, (t&1)+2, ... of
// row t>>1 in index order, the pair is combined with one DPP swap, and (mean, rstd) lands in LDS for the
// epilogue.  The loads are issued before the first K-tile is staged, so their latency hides under it.
struct LnfoldLoads { float2 v[LNFOLD_NP / 2]; };
__device__ __forceinline__ void lnfold_issue(const GemmAux &aux, int m0, LnfoldLoads &ld)
{
    const int t = threadIdx.x;
    const float2 *p = aux.stats_in + (size_t)(m0 + (t >> 1)) * LNFOLD_NP + (t & 1);
#pragma unroll
    for (int i = 0; i < LNFOLD_NP / 2; ++i) ld.v[i] = p[2 * i];
}
__device__ __forceinline__ void lnfold_finish(const GemmAux &aux, const LnfoldLoads &ld, float2 *lds_stats)
{
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < LNFOLD_NP / 2; ++i) { s += ld.v[i].x; ss += ld.v[i].y; }
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xf, 0xf, false));
    ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0xB1, 0xf, 0xf, false));
    const float mean = s * aux.inv_d;
    const float rstd = rsqrtf(fmaxf(ss * aux.inv_d - mean * mean, 0.f) + aux.eps);
    if (!(threadIdx.x & 1)) lds_stats[threadIdx.x >> 1] = make_float2(mean, rstd);
}

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------ host side
void set_error(const char *fmt, ...);

// Per-kernel "attributes already set on this device" flag.  hipFuncSetAttribute applies to the current device only,
// so the flag is kept per device (a process normally drives one GPU, but nothing here assumes it).
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;   // unknown: just set it again
        const unsigned long long bit = 1ull << dev;
        return (mask.fetch_or(bit, std::memory_order_relaxed) & bit) == 0;
    }
};

// Optional launch profiler (api_common.cpp): a ProfScope at a launch site records a HIP event pair
// on the launch stream when mmr_prof_enable(1, n) is in force, and costs one branch otherwise.
struct ProfScope {
    ProfScope(int cls, hipStream_t st);
    ~ProfScope();
    long slot_;
    hipStream_t st_;
};

#define MMR_CHECK_ARG(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            mmr::set_error(__VA_ARGS__);      \
            return MMR_EINVAL;                \
        }                                     \
    } while (0)

#define MMR_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            mmr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MMR_EIO;                                                              \
        }                                                                                \
    } while (0)

#define MMR_CHECK_LAUNCH() MMR_CHECK_HIP(hipGetLastError())

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace mmr
