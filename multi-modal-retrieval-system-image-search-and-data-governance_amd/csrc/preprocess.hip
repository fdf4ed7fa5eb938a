// On-device CLIP preprocessing for gfx950: Pillow-exact bicubic resize (8-bit fixed point, horizontal
// pass then vertical pass, each rounded to uint8), centre crop, /255, (x - mean) / std.
//
// Replaces the PIL/torchvision `preprocess` the reference applies per image on the host
// (reference code/search_image.py:127,155; constants code/custom.py:28).  Byte/integer work, HBM/L2
// bound; the coefficient tables come from the host (preprocess.py mirrors Pillow's
// precompute_coeffs / normalize_coeffs_8bpc), the kernels only do int32 multiply-adds, so the uint8
// result is bit-identical to Pillow's.  Only the crop window and the input rows it needs are computed.
#include "mmr_common.h"

namespace mmr {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Pillow Resample.c

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[r][x][c] for r in [0, rows) (input row row0 + r), x in [0, S)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t *__restrict__ img, int W, int S, int row0,
                                                         int rows, const int2 *__restrict__ hb,
                                                         const int *__restrict__ hc, int hk,
                                                         uint8_t *__restrict__ tmp)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * S) return;
    const int x = id % S, r = id / S;
    const int2 b = hb[x];
    const int *k = hc + (size_t)x * hk;
    const uint8_t *p = img + ((size_t)(row0 + r) * W + b.x) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < b.y; ++t) {
        const int kk = k[t];
        a0 += p[3 * t + 0] * kk;
        a1 += p[3 * t + 1] * kk;
        a2 += p[3 * t + 2] * kk;
    }
    uint8_t *o = tmp + (size_t)id * 3;
    o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
}

// out[c][y][x] = ((u8 / 255) - mean[c]) / std[c], u8 = vertical pass over tmp
template <typename TOUT>
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t *__restrict__ tmp, int S, int row0,
                                                              const int2 *__restrict__ vb, const int *__restrict__ vc,
                                                              int vk, float m0, float m1, float m2, float s0, float s1,
                                                              float s2, TOUT *__restrict__ out, uint8_t *__restrict__ u8)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= S * S) return;
    const int x = id % S, y = id / S;
    const int2 b = vb[y];
    const int *k = vc + (size_t)y * vk;
    const uint8_t *p = tmp + ((size_t)(b.x - row0) * S + x) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < b.y; ++t) {
        const int kk = k[t];
        const uint8_t *q = p + (size_t)t * S * 3;
        a0 += q[0] * kk; a1 += q[1] * kk; a2 += q[2] * kk;
    }
    const int c0 = clip8(a0), c1 = clip8(a1), c2 = clip8(a2);
    if (u8) { u8[(size_t)id * 3 + 0] = (uint8_t)c0; u8[(size_t)id * 3 + 1] = (uint8_t)c1; u8[(size_t)id * 3 + 2] = (uint8_t)c2; }
    // ToTensor then Normalize, each a correctly rounded fp32 op like the CPU reference
    const float f0 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)c0, 255.0f), m0), s0);
    const float f1 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)c1, 255.0f), m1), s1);
    const float f2 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)c2, 255.0f), m2), s2);
    const size_t plane = (size_t)S * S;
    if constexpr (sizeof(TOUT) == 2) {
        ((bf16_t *)out)[id] = f32_to_bf16(f0);
        ((bf16_t *)out)[plane + id] = f32_to_bf16(f1);
        ((bf16_t *)out)[2 * plane + id] = f32_to_bf16(f2);
    } else {
        ((float *)out)[id] = f0;
        ((float *)out)[plane + id] = f1;
        ((float *)out)[2 * plane + id] = f2;
    }
}

// ---- batched form: one launch pair for B images of different sizes, driven by a descriptor table
struct PreDesc {            // mirrors mmr_preprocess_desc in include/mmr.h
    const uint8_t *img;     // uint8 RGB [H,W,3]
    const int32_t *hbounds, *hcoeffs, *vbounds, *vcoeffs;
    uint8_t *tmp;           // uint8 [rows,S,3] scratch of this image
    int32_t H, W, row0, rows, hk, vk;
};

__global__ __launch_bounds__(256) void resample_h_batch_kernel(const PreDesc *__restrict__ desc, int S, int max_rows)
{
    const PreDesc d = desc[blockIdx.y];
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= d.rows * S) return;
    const int x = id % S, r = id / S;
    const int2 b = reinterpret_cast<const int2 *>(d.hbounds)[x];
    const int *k = d.hcoeffs + (size_t)x * d.hk;
    const uint8_t *p = d.img + ((size_t)(d.row0 + r) * d.W + b.x) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < b.y; ++t) {
        const int kk = k[t];
        a0 += p[3 * t + 0] * kk; a1 += p[3 * t + 1] * kk; a2 += p[3 * t + 2] * kk;
    }
    uint8_t *o = d.tmp + (size_t)id * 3;
    o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
}

template <typename TOUT>
__global__ __launch_bounds__(256) void resample_v_norm_batch_kernel(const PreDesc *__restrict__ desc, int S, float m0,
                                                                    float m1, float m2, float s0, float s1, float s2,
                                                                    TOUT *__restrict__ out)
{
    const PreDesc d = desc[blockIdx.y];
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= S * S) return;
    const int x = id % S, y = id / S;
    const int2 b = reinterpret_cast<const int2 *>(d.vbounds)[y];
    const int *k = d.vcoeffs + (size_t)y * d.vk;
    const uint8_t *p = d.tmp + ((size_t)(b.x - d.row0) * S + x) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < b.y; ++t) {
        const int kk = k[t];
        const uint8_t *q = p + (size_t)t * S * 3;
        a0 += q[0] * kk; a1 += q[1] * kk; a2 += q[2] * kk;
    }
    const float f0 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(a0), 255.0f), m0), s0);
    const float f1 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(a1), 255.0f), m1), s1);
    const float f2 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(a2), 255.0f), m2), s2);
    const size_t plane = (size_t)S * S;
    TOUT *o = out + (size_t)blockIdx.y * 3 * plane;
    if constexpr (sizeof(TOUT) == 2) {
        ((bf16_t *)o)[id] = f32_to_bf16(f0); ((bf16_t *)o)[plane + id] = f32_to_bf16(f1); ((bf16_t *)o)[2 * plane + id] = f32_to_bf16(f2);
    } else {
        ((float *)o)[id] = f0; ((float *)o)[plane + id] = f1; ((float *)o)[2 * plane + id] = f2;
    }
}

}  // namespace mmr

using namespace mmr;

extern "C" int mmr_preprocess_batch(const void *desc, int B, int S, int max_rows, float mean0, float mean1, float mean2,
                                    float std0, float std1, float std2, void *out, mmr_dtype out_dtype, void *stream)
{
    MMR_CHECK_ARG(desc && out, "mmr_preprocess_batch: null pointer");
    MMR_CHECK_ARG(B >= 0 && B <= 65535 && S >= 1 && S <= 4096 && max_rows >= 1, "mmr_preprocess_batch: bad size B=%d S=%d max_rows=%d", B, S, max_rows);
    MMR_CHECK_ARG(out_dtype == MMR_F32 || out_dtype == MMR_BF16, "mmr_preprocess_batch: out dtype %d", (int)out_dtype);
    MMR_CHECK_ARG(std0 != 0.f && std1 != 0.f && std2 != 0.f, "mmr_preprocess_batch: zero std");
    MMR_CHECK_ARG(sizeof(PreDesc) == 72, "mmr_preprocess_batch: descriptor layout drifted");
    if (B == 0) return MMR_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(MMR_PROF_ROWWISE, st);
    const PreDesc *d = (const PreDesc *)desc;
    hipLaunchKernelGGL(resample_h_batch_kernel, dim3(((size_t)max_rows * S + 255) / 256, B), dim3(256), 0, st, d, S, max_rows);
    MMR_CHECK_LAUNCH();
    if (out_dtype == MMR_BF16)
        hipLaunchKernelGGL(resample_v_norm_batch_kernel<bf16_t>, dim3((S * S + 255) / 256, B), dim3(256), 0, st, d, S, mean0,
                           mean1, mean2, std0, std1, std2, (bf16_t *)out);
    else
        hipLaunchKernelGGL(resample_v_norm_batch_kernel<float>, dim3((S * S + 255) / 256, B), dim3(256), 0, st, d, S, mean0,
                           mean1, mean2, std0, std1, std2, (float *)out);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

extern "C" int mmr_preprocess_image(const uint8_t *img, int H, int W, int S, int row0, int rows, const int32_t *hbounds,
                                    const int32_t *hcoeffs, int hk, const int32_t *vbounds, const int32_t *vcoeffs,
                                    int vk, float mean0, float mean1, float mean2, float std0, float std1, float std2,
                                    uint8_t *tmp, void *out, mmr_dtype out_dtype, uint8_t *out_u8, void *stream)
{
    MMR_CHECK_ARG(img && hbounds && hcoeffs && vbounds && vcoeffs && tmp && out, "mmr_preprocess_image: null pointer");
    MMR_CHECK_ARG(H >= 1 && W >= 1 && S >= 1 && S <= 4096, "mmr_preprocess_image: bad size H=%d W=%d S=%d", H, W, S);
    MMR_CHECK_ARG(row0 >= 0 && rows >= 1 && row0 + rows <= H, "mmr_preprocess_image: rows [%d,%d) outside the image", row0, row0 + rows);
    MMR_CHECK_ARG(hk >= 1 && vk >= 1, "mmr_preprocess_image: empty coefficient table");
    MMR_CHECK_ARG(out_dtype == MMR_F32 || out_dtype == MMR_BF16, "mmr_preprocess_image: out dtype %d", (int)out_dtype);
    MMR_CHECK_ARG(std0 != 0.f && std1 != 0.f && std2 != 0.f, "mmr_preprocess_image: zero std");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(MMR_PROF_ROWWISE, st);
    hipLaunchKernelGGL(resample_h_kernel, dim3((rows * S + 255) / 256), dim3(256), 0, st, img, W, S, row0, rows,
                       (const int2 *)hbounds, hcoeffs, hk, tmp);
    MMR_CHECK_LAUNCH();
    if (out_dtype == MMR_BF16)
        hipLaunchKernelGGL(resample_v_norm_kernel<bf16_t>, dim3((S * S + 255) / 256), dim3(256), 0, st, tmp, S, row0,
                           (const int2 *)vbounds, vcoeffs, vk, mean0, mean1, mean2, std0, std1, std2, (bf16_t *)out, out_u8);
    else
        hipLaunchKernelGGL(resample_v_norm_kernel<float>, dim3((S * S + 255) / 256), dim3(256), 0, st, tmp, S, row0,
                           (const int2 *)vbounds, vcoeffs, vk, mean0, mean1, mean2, std0, std1, std2, (float *)out, out_u8);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}
