// Shared host/device helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
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
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// Fixed-order fp64 butterfly: identical in every lane, and identical to oracle/search_ref.c.
__device__ __forceinline__ double wave_sum_f64_butterfly(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
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
