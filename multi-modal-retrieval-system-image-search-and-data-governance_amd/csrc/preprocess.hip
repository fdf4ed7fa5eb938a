// On-device CLIP preprocessing for gfx950: Pillow-exact bicubic resize (8-bit fixed point, horizontal
// pass then vertical pass, each rounded to uint8), centre crop, /255, (x - mean) / std.
//
// Replaces the PIL/torchvision `preprocess` the reference applies per image on the host
// (reference code/search_image.py:127,155; constants code/custom.py:28).  Byte/integer work, HBM/L2
// bound; the coefficient tables come from the host (preprocess.py mirrors Pillow's
// precompute_coeffs / normalize_coeffs_8bpc), the kernels only do int32 multiply-adds, so the uint8
// result is bit-identical to Pillow's.  Only the crop window and the input rows it needs are computed.
#include "mmr_common.h"

#include <stdlib.h>

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

// ---------------------------------------------------------------------------------------------
// Fast forms (round 3).  The kernels above issue one BYTE load per tap and channel -- 30 per output pixel of a VGA image --
// and are bound by the rate the texture-address unit accepts instructions, not by memory: 0.5 ms for 256 VGA images whose
// 236 MB stream in 40 us.  The forms below move the same bytes with 12-byte loads and keep the integer arithmetic, hence the
// results, identical (sums of int32 products are exact in any order; the padded taps have zero coefficients):
//   H pass: a thread still owns one (input row, output column) but walks its tap window four taps = 12 bytes = one
//           dwordx3 load at a time, with the four coefficients as one int4 load -- the coefficient rows are padded with
//           zeros to a multiple of 4 taps on the host (preprocess.py; `hk` is the padded row length);
//   V pass: the flattened (x, channel) row of the intermediate image is contiguous, and the vertical taps weigh every byte
//           of a row alike, so a thread takes 12 consecutive bytes = 4 pixels per tap with one dwordx3 load; one wave per
//           output row, coefficients wave-uniform;
//   /255, -mean, /std: only 256 x 3 different results exist -- each workgroup tabulates them in LDS with the SAME correctly
//           rounded fp32 operations (bit-identical to computing them per pixel) instead of two IEEE divisions per value.
// Windows that would read past the last byte of their image take a byte-wise path (last row, rightmost columns).
// Requires S % 4 == 0 and hk % 4 == 0; otherwise the launchers use the generic kernels above.
// ---------------------------------------------------------------------------------------------
struct u32x3 { uint32_t x, y, z; };

__device__ __forceinline__ u32x3 load12(const uint8_t *p) {          // unaligned 12-byte load (one global_load_dwordx3)
    typedef uint32_t u3 __attribute__((ext_vector_type(3), aligned(1)));
    const u3 v = *reinterpret_cast<const u3 *>(p);
    return u32x3{v.x, v.y, v.z};
}
__device__ __forceinline__ u32x3 load12_bounded(const uint8_t *p, const uint8_t *end) {   // bytes at or past `end` read as 0
    uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 12; ++i)
        if (p + i < end) w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
    return u32x3{w[0], w[1], w[2]};
}
__device__ __forceinline__ int byte_of(uint32_t w, int i) { return (int)((w >> (8 * i)) & 0xffu); }
// byte * coefficient: Pillow's 8-bit coefficients are |k| <= 2^22 (PRECISION_BITS = 22, |weight| <= 1), so both factors fit
// the 24-bit multiplier (v_mad_i32_i24, full rate; a 32-bit v_mul_lo is a quarter-rate instruction and there are 36 of them
// per output pixel).  The host checks the bound when it builds the tables.
__device__ __forceinline__ int bmul(int byte, int k) { return __mul24(byte, k); }

// four taps (12 bytes: t0c0 t0c1 t0c2 t1c0 | t1c1 t1c2 t2c0 t2c1 | t2c2 t3c0 t3c1 t3c2) into the three channel sums
__device__ __forceinline__ void mac4taps(const u32x3 d, const int4 k, int &a0, int &a1, int &a2) {
    a0 += bmul(byte_of(d.x, 0), k.x) + bmul(byte_of(d.x, 3), k.y) + bmul(byte_of(d.y, 2), k.z) + bmul(byte_of(d.z, 1), k.w);
    a1 += bmul(byte_of(d.x, 1), k.x) + bmul(byte_of(d.y, 0), k.y) + bmul(byte_of(d.y, 3), k.z) + bmul(byte_of(d.z, 2), k.w);
    a2 += bmul(byte_of(d.x, 2), k.x) + bmul(byte_of(d.y, 1), k.y) + bmul(byte_of(d.z, 0), k.z) + bmul(byte_of(d.z, 3), k.w);
}

// horizontal pass of one image: tmp[r][x][c], r in [0, rows), x in [0, S); hk4 = hk / 4 tap groups per column
__device__ __forceinline__ void resample_h_fast(const uint8_t *__restrict__ img, int H, int W, int S, int row0, int rows,
                                                const int2 *__restrict__ hb, const int *__restrict__ hc, int hk,
                                                uint8_t *__restrict__ tmp, int id)
{
    if (id >= rows * S) return;
    const int x = id % S, r = id / S;
    const int2 b = hb[x];
    const int4 *k4 = reinterpret_cast<const int4 *>(hc + (size_t)x * hk);
    const uint8_t *p = img + ((size_t)(row0 + r) * W + b.x) * 3;
    const uint8_t *end = img + (size_t)H * W * 3;
    const int groups = (b.y + 3) >> 2;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    if (p + 12 * groups <= end) {
        for (int g = 0; g < groups; ++g) mac4taps(load12(p + 12 * g), k4[g], a0, a1, a2);
    } else {
        for (int g = 0; g < groups; ++g) mac4taps(load12_bounded(p + 12 * g, end), k4[g], a0, a1, a2);
    }
    uint8_t *o = tmp + (size_t)id * 3;
    o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
}

// vertical pass + normalise of one image; block = 4 waves over VR output rows (VR / 4 consecutive ones per wave), lane l the
// pixels 4l .. 4l+3 of a row.  lut: LDS [3][256] of TOUT-rounded ((v/255) - mean[c]) / std[c] -- two IEEE divisions per
// entry, about the cost of one output row per wave, hence VR = 8 for batches (4, 16, 32 measured within 5 % of it) (VR = 4 keeps a single image's few
// workgroups short).  Taps go two at a time (12 byte products + 12 three-operand adds per pair) and the next pair's rows are
// requested before the current pair is multiplied; the coefficient row is wave-uniform (scalar loads).
template <typename TOUT, int VR>
__device__ __forceinline__ void resample_v_fast(const uint8_t *__restrict__ tmp, int S, int row0, const int2 *__restrict__ vb,
                                                const int *__restrict__ vc, int vk, float m0, float m1, float m2, float s0,
                                                float s1, float s2, TOUT *__restrict__ out, uint8_t *__restrict__ u8, int yblk)
{
    static_assert(VR % 4 == 0, "whole rows per wave");
    __shared__ float lut[3][256];
    for (int i = threadIdx.x; i < 768; i += blockDim.x) {
        const int c = i >> 8, v = i & 255;
        const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        // ToTensor then Normalize, each a correctly rounded fp32 op like the CPU reference
        lut[c][v] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.0f), m), sd);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (lane * 4 >= S) return;
    const size_t stride = (size_t)S * 3;
    for (int j = 0; j < VR / 4; ++j) {
        const int y = yblk * VR + wave * (VR / 4) + j;
        if (y >= S) return;
        const int2 b = vb[y];
        const int n = b.y;
        const int *k = vc + (size_t)y * vk;
        const uint8_t *p = tmp + ((size_t)(b.x - row0) * S + lane * 4) * 3;     // tmp carries 16 bytes of slack behind its last row
        int a[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) a[i] = 1 << (PRECISION_BITS - 1);
        if (n > 0) {
            // rows past the window are read as its last row and weighted 0
            u32x3 d0 = load12(p), d1 = load12(p + (size_t)min(1, n - 1) * stride);
            for (int t = 0; t < n; t += 2) {
                const u32x3 e0 = load12(p + (size_t)min(t + 2, n - 1) * stride);
                const u32x3 e1 = load12(p + (size_t)min(t + 3, n - 1) * stride);
                const int k0 = k[t], k1 = t + 1 < n ? k[t + 1] : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] += bmul(byte_of(d0.x, i), k0) + bmul(byte_of(d1.x, i), k1);
                    a[4 + i] += bmul(byte_of(d0.y, i), k0) + bmul(byte_of(d1.y, i), k1);
                    a[8 + i] += bmul(byte_of(d0.z, i), k0) + bmul(byte_of(d1.z, i), k1);
                }
                d0 = e0; d1 = e1;
            }
        }
        int c[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) c[i] = clip8(a[i]);
        if (u8) {
            uint32_t w[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) w[q] = (uint32_t)c[4 * q] | ((uint32_t)c[4 * q + 1] << 8) | ((uint32_t)c[4 * q + 2] << 16) | ((uint32_t)c[4 * q + 3] << 24);
            uint32_t *o8 = reinterpret_cast<uint32_t *>(u8 + ((size_t)y * S + lane * 4) * 3);     // 12-byte aligned: S % 4 == 0
            o8[0] = w[0]; o8[1] = w[1]; o8[2] = w[2];
        }
        const size_t plane = (size_t)S * S, o = (size_t)y * S + lane * 4;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float f0 = lut[ch][c[ch]], f1 = lut[ch][c[3 + ch]], f2 = lut[ch][c[6 + ch]], f3 = lut[ch][c[9 + ch]];
            if constexpr (sizeof(TOUT) == 2) {
                uint2 pk;
                pk.x = pack_bf16x2(f0, f1);
                pk.y = pack_bf16x2(f2, f3);
                *reinterpret_cast<uint2 *>((bf16_t *)out + ch * plane + o) = pk;
            } else {
                *reinterpret_cast<float4 *>((float *)out + ch * plane + o) = make_float4(f0, f1, f2, f3);
            }
        }
    }
}

// Horizontal pass with the input rows staged in LDS.  resample_h_fast above still issues 6 global loads and 3 byte stores
// per output pixel, each touching 64 scattered 12-byte spans: the texture-address unit, not memory, paces it (0.22 ms of
// the 0.27 ms for 256 VGA images).  Here a workgroup copies RB whole input rows into LDS with coalesced 16-byte loads,
// every thread owns ONE output column for all RB rows (its coefficients are loaded once), reads its tap windows from LDS
// (aligned dword reads + v_alignbyte to the window's byte offset) and the RB finished rows of the intermediate image leave
// through LDS as whole 16-byte stores: ~1 global instruction per output pixel instead of 9.  Same integer sums, same bytes.
// LDS: RB * pitch (input rows, pitch = row bytes rounded up to 16, + 32 of slack) + RB * S * 3 (output rows).
template <int RG>
__device__ __forceinline__ void resample_h_lds(const uint8_t *__restrict__ img, int H, int W, int S, int row0, int rows,
                                               const int2 *__restrict__ hb, const int *__restrict__ hc, int hk,
                                               uint8_t *__restrict__ tmp, int rblk, int RB, int pitch, char *lds)
{
    const int r0 = rblk * RB;                                   // first row (of `rows`) of this workgroup
    const int nr = min(RB, rows - r0);
    if (nr <= 0) return;
    const int rowbytes = W * 3;
    const uint8_t *end = img + (size_t)H * rowbytes;
    uint8_t *lin = reinterpret_cast<uint8_t *>(lds);
    uint8_t *lout = lin + (size_t)RB * pitch;
    // ---- stage: 16 bytes per thread per step; the tail of the last image row is read byte-wise (nothing past the image)
    const int chunks = pitch >> 4;
    for (int i = threadIdx.x; i < nr * chunks; i += blockDim.x) {
        const int r = i / chunks, c = i - r * chunks;
        const uint8_t *src = img + (size_t)(row0 + r0 + r) * rowbytes + c * 16;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (c * 16 < rowbytes + 16) {                           // chunks wholly behind the row's slack are left zero
            if (src + 16 <= end) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));
                const u4 t = *reinterpret_cast<const u4 *>(src);
                v = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                for (int j = 0; j < 16; ++j)
                    if (src + j < end) w[j >> 2] |= (uint32_t)src[j] << (8 * (j & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        *reinterpret_cast<uint4 *>(lin + (size_t)r * pitch + c * 16) = v;
    }
    __syncthreads();
    // ---- compute: item = (column x, group of RG rows).  Tap groups run in the OUTER loop and the RG rows inside it, so a
    // column's coefficient group is loaded once per RG rows (and one group ahead of its use) and the inner loop is LDS reads and
    // integer math only; rows past nr compute on stale LDS and are not stored
    const int nrg = (nr + RG - 1) / RG;
    for (int it = threadIdx.x; it < S * nrg; it += blockDim.x) {
        const int rg = it / S, x = it - rg * S, rbase = rg * RG;
        const int2 b = hb[x];
        const int groups = (b.y + 3) >> 2;
        const int4 *k4 = reinterpret_cast<const int4 *>(hc + (size_t)x * hk);
        const int off = b.x * 3, a = off & ~3, sh = off & 3;
        const uint8_t *base = lin + (size_t)rbase * pitch + a;
        int acc[RG][3];
#pragma unroll
        for (int r = 0; r < RG; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 1 << (PRECISION_BITS - 1);
        int4 k = k4[0];
        for (int g = 0; g < groups; ++g) {
            const int4 kn = k4[g + 1 < groups ? g + 1 : g];
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const uint32_t *w = reinterpret_cast<const uint32_t *>(base + (size_t)r * pitch) + 3 * g;
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
                u32x3 d;                                         // the 12 window bytes of this tap group, realigned
                d.x = __builtin_amdgcn_alignbyte(w1, w0, sh);
                d.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
                d.z = __builtin_amdgcn_alignbyte(w3, w2, sh);
                mac4taps(d, k, acc[r][0], acc[r][1], acc[r][2]);
            }
            k = kn;
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            if (rbase + r < nr) {
                uint8_t *o = lout + ((size_t)(rbase + r) * S + x) * 3;
                o[0] = (uint8_t)clip8(acc[r][0]); o[1] = (uint8_t)clip8(acc[r][1]); o[2] = (uint8_t)clip8(acc[r][2]);
            }
        }
    }
    __syncthreads();
    // ---- finished rows: contiguous in tmp ([rows][S][3]); 16 bytes per thread per step, byte-wise tail
    const int total = nr * S * 3;
    uint8_t *dst = tmp + (size_t)r0 * S * 3;
    const int head = (int)((16 - ((uintptr_t)dst & 15)) & 15);          // bytes up to the first 16-byte boundary of dst
    for (int i = threadIdx.x; i < head && i < total; i += blockDim.x) dst[i] = lout[i];
    const int body = total > head ? (total - head) >> 4 : 0;
    if (head == 0) {                                            // the usual case (S * 3 and the scratch are multiples of 16)
        for (int i = threadIdx.x; i < body; i += blockDim.x)
            *reinterpret_cast<uint4 *>(dst + i * 16) = *reinterpret_cast<const uint4 *>(lout + i * 16);
    } else {
        for (int i = threadIdx.x; i < body; i += blockDim.x) {
            const uint8_t *sp = lout + head + i * 16;
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (uint32_t)sp[4 * j] | ((uint32_t)sp[4 * j + 1] << 8) | ((uint32_t)sp[4 * j + 2] << 16) | ((uint32_t)sp[4 * j + 3] << 24);
            *reinterpret_cast<uint4 *>(dst + head + i * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    for (int i = head + body * 16 + threadIdx.x; i < total; i += blockDim.x) dst[i] = lout[i];
}

template <int RG>
__global__ __launch_bounds__(512) void resample_h_lds_kernel(const uint8_t *__restrict__ img, int H, int W, int S, int row0,
                                                             int rows, const int2 *__restrict__ hb, const int *__restrict__ hc,
                                                             int hk, uint8_t *__restrict__ tmp, int RB, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) char pre_lds[];
    resample_h_lds<RG>(img, H, W, S, row0, rows, hb, hc, hk, tmp, blockIdx.x, RB, pitch, pre_lds);
}

// batched: RB / pitch are sized on the host for the WIDEST image of the batch (max_w); an image whose tables are not in
// the padded layout takes the byte-wise path inside the same launch (thread id -> (row, column) of this row block)
template <int RG>
__global__ __launch_bounds__(512) void resample_h_lds_batch_kernel(const PreDesc *__restrict__ desc, int S, int RB, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) char pre_lds[];
    const PreDesc d = desc[blockIdx.y];
    if ((d.hk & 3) == 0 && ((uintptr_t)d.hcoeffs & 15) == 0 && d.W * 3 + 32 <= pitch) {
        resample_h_lds<RG>(d.img, d.H, d.W, S, d.row0, d.rows, reinterpret_cast<const int2 *>(d.hbounds), d.hcoeffs, d.hk, d.tmp,
                       blockIdx.x, RB, pitch, pre_lds);
        return;
    }
    const int r0 = blockIdx.x * RB, nr = min(RB, d.rows - r0);
    for (int i = threadIdx.x; i < nr * S; i += blockDim.x) {
        const int x = i % S, r = r0 + i / S;
        const int2 b = reinterpret_cast<const int2 *>(d.hbounds)[x];
        const int *k = d.hcoeffs + (size_t)x * d.hk;
        const uint8_t *p = d.img + ((size_t)(d.row0 + r) * d.W + b.x) * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int t = 0; t < b.y; ++t) {
            const int kk = k[t];
            a0 += p[3 * t + 0] * kk; a1 += p[3 * t + 1] * kk; a2 += p[3 * t + 2] * kk;
        }
        uint8_t *o = d.tmp + ((size_t)r * S + x) * 3;
        o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
    }
}

__global__ __launch_bounds__(256) void resample_h_fast_kernel(const uint8_t *__restrict__ img, int H, int W, int S, int row0,
                                                              int rows, const int2 *__restrict__ hb,
                                                              const int *__restrict__ hc, int hk, uint8_t *__restrict__ tmp)
{
    resample_h_fast(img, H, W, S, row0, rows, hb, hc, hk, tmp, blockIdx.x * blockDim.x + threadIdx.x);
}

#ifndef MMR_V_ROWS_BATCH
#define MMR_V_ROWS_BATCH 8
#endif
constexpr int V_ROWS_IMAGE = 4, V_ROWS_BATCH = MMR_V_ROWS_BATCH;
template <typename TOUT>
__global__ __launch_bounds__(256) void resample_v_fast_kernel(const uint8_t *__restrict__ tmp, int S, int row0,
                                                              const int2 *__restrict__ vb, const int *__restrict__ vc, int vk,
                                                              float m0, float m1, float m2, float s0, float s1, float s2,
                                                              TOUT *__restrict__ out, uint8_t *__restrict__ u8)
{
    resample_v_fast<TOUT, V_ROWS_IMAGE>(tmp, S, row0, vb, vc, vk, m0, m1, m2, s0, s1, s2, out, u8, blockIdx.x);
}

// per image: the 12-byte form when its coefficient rows are padded to whole groups of 4 taps and 16-byte aligned (what
// preprocess.py builds), the byte-wise form otherwise -- same results either way
__global__ __launch_bounds__(256) void resample_h_fast_batch_kernel(const PreDesc *__restrict__ desc, int S)
{
    const PreDesc d = desc[blockIdx.y];
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if ((d.hk & 3) == 0 && ((uintptr_t)d.hcoeffs & 15) == 0) {
        resample_h_fast(d.img, d.H, d.W, S, d.row0, d.rows, reinterpret_cast<const int2 *>(d.hbounds), d.hcoeffs, d.hk, d.tmp, id);
        return;
    }
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
__global__ __launch_bounds__(256) void resample_v_fast_batch_kernel(const PreDesc *__restrict__ desc, int S, float m0, float m1,
                                                                    float m2, float s0, float s1, float s2, TOUT *__restrict__ out)
{
    const PreDesc d = desc[blockIdx.y];
    resample_v_fast<TOUT, V_ROWS_BATCH>(d.tmp, S, d.row0, reinterpret_cast<const int2 *>(d.vbounds), d.vcoeffs, d.vk, m0, m1, m2, s0, s1, s2,
                          out + (size_t)blockIdx.y * 3 * S * S, nullptr, blockIdx.x);
}

}  // namespace mmr

using namespace mmr;

// LDS plan of the staged horizontal pass for images up to `width` pixels wide: rows per workgroup (RB), rows per thread
// (RG, a template parameter: RB is a multiple of it), row pitch and workgroup size (one thread per (column, row group),
// whole waves, at most 512); RB = 0 when even one row does not fit the 64 KiB the pass allows itself (two workgroups per CU)
struct HPlan { int RB, RG, pitch, threads; };
static HPlan plan_h_lds(int width, int S)
{
    HPlan p;
    p.pitch = (int)align_up((size_t)width * 3, 16) + 32;
    const int per_row = p.pitch + S * 3;
    p.RB = width > 0 ? (64 * 1024) / per_row : 0;
    static const int rb_cap = getenv("MMR_PRE_RB") ? atoi(getenv("MMR_PRE_RB")) : 8;      // A/B aids
    static const int rg_env = getenv("MMR_PRE_RG") ? atoi(getenv("MMR_PRE_RG")) : 8;
    if (p.RB > rb_cap) p.RB = rb_cap;
    p.RG = (p.RB >= 8 && rg_env >= 8) ? 8 : (p.RB >= 4 && rg_env >= 4 ? 4 : 1);
    p.RB -= p.RB % p.RG;
    const long items = (long)S * (p.RB / (p.RG ? p.RG : 1));
    p.threads = (int)(items >= 512 ? 512 : (items <= 64 ? 64 : align_up((size_t)items, 64)));
    return p;
}

template <int RG>
static int launch_h_lds_batch(const PreDesc *d, int B, int S, int max_rows, const HPlan &p, hipStream_t st)
{
    static DeviceOnce once;
    if (once.first())
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&resample_h_lds_batch_kernel<RG>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL(resample_h_lds_batch_kernel<RG>, dim3((max_rows + p.RB - 1) / p.RB, B), dim3(p.threads),
                       (size_t)p.RB * (p.pitch + S * 3), st, d, S, p.RB, p.pitch);
    return MMR_OK;
}

template <int RG>
static int launch_h_lds(const uint8_t *img, int H, int W, int S, int row0, int rows, const int32_t *hbounds,
                        const int32_t *hcoeffs, int hk, uint8_t *tmp, const HPlan &p, hipStream_t st)
{
    static DeviceOnce once;
    if (once.first())
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&resample_h_lds_kernel<RG>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL(resample_h_lds_kernel<RG>, dim3((rows + p.RB - 1) / p.RB), dim3(p.threads),
                       (size_t)p.RB * (p.pitch + S * 3), st, img, H, W, S, row0, rows, (const int2 *)hbounds, hcoeffs, hk, tmp,
                       p.RB, p.pitch);
    return MMR_OK;
}

extern "C" int mmr_preprocess_batch(const void *desc, int B, int S, int max_rows, float mean0, float mean1, float mean2,
                                    float std0, float std1, float std2, void *out, mmr_dtype out_dtype, void *stream)
{
    return mmr_preprocess_batch_ex(desc, B, S, max_rows, 0, mean0, mean1, mean2, std0, std1, std2, out, out_dtype, stream);
}

extern "C" int mmr_preprocess_batch_ex(const void *desc, int B, int S, int max_rows, int max_width, float mean0, float mean1,
                                       float mean2, float std0, float std1, float std2, void *out, mmr_dtype out_dtype,
                                       void *stream)
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
    const HPlan hp = plan_h_lds(max_width, S);
    if (hp.RB >= 1) {                     // input rows staged in LDS (max_width = the widest image of the batch)
        const int rc = hp.RG == 8 ? launch_h_lds_batch<8>(d, B, S, max_rows, hp, st)
                     : hp.RG == 4 ? launch_h_lds_batch<4>(d, B, S, max_rows, hp, st)
                                  : launch_h_lds_batch<1>(d, B, S, max_rows, hp, st);
        if (rc != MMR_OK) return rc;
    } else {
        hipLaunchKernelGGL(resample_h_fast_batch_kernel, dim3(((size_t)max_rows * S + 255) / 256, B), dim3(256), 0, st, d, S);
    }
    MMR_CHECK_LAUNCH();
    if (S % 4 == 0) {                     // 4 pixels per lane, V_ROWS_BATCH output rows per workgroup
        const dim3 grid((S + V_ROWS_BATCH - 1) / V_ROWS_BATCH, B);
        if (out_dtype == MMR_BF16)
            hipLaunchKernelGGL(resample_v_fast_batch_kernel<bf16_t>, grid, dim3(256), 0, st, d, S, mean0, mean1, mean2, std0, std1,
                               std2, (bf16_t *)out);
        else
            hipLaunchKernelGGL(resample_v_fast_batch_kernel<float>, grid, dim3(256), 0, st, d, S, mean0, mean1, mean2, std0, std1,
                               std2, (float *)out);
    } else if (out_dtype == MMR_BF16)
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
    const HPlan hp = plan_h_lds(W, S);
    if (hk % 4 == 0 && ((uintptr_t)hcoeffs & 15) == 0 && hp.RB >= 1) {
        const int rc = hp.RG == 8 ? launch_h_lds<8>(img, H, W, S, row0, rows, hbounds, hcoeffs, hk, tmp, hp, st)
                     : hp.RG == 4 ? launch_h_lds<4>(img, H, W, S, row0, rows, hbounds, hcoeffs, hk, tmp, hp, st)
                                  : launch_h_lds<1>(img, H, W, S, row0, rows, hbounds, hcoeffs, hk, tmp, hp, st);
        if (rc != MMR_OK) return rc;
    } else if (hk % 4 == 0 && ((uintptr_t)hcoeffs & 15) == 0)
        hipLaunchKernelGGL(resample_h_fast_kernel, dim3((rows * S + 255) / 256), dim3(256), 0, st, img, H, W, S, row0, rows,
                           (const int2 *)hbounds, hcoeffs, hk, tmp);
    else
        hipLaunchKernelGGL(resample_h_kernel, dim3((rows * S + 255) / 256), dim3(256), 0, st, img, W, S, row0, rows,
                           (const int2 *)hbounds, hcoeffs, hk, tmp);
    MMR_CHECK_LAUNCH();
    if (S % 4 == 0) {
        if (out_dtype == MMR_BF16)
            hipLaunchKernelGGL(resample_v_fast_kernel<bf16_t>, dim3((S + V_ROWS_IMAGE - 1) / V_ROWS_IMAGE), dim3(256), 0, st, tmp, S, row0,
                               (const int2 *)vbounds, vcoeffs, vk, mean0, mean1, mean2, std0, std1, std2, (bf16_t *)out, out_u8);
        else
            hipLaunchKernelGGL(resample_v_fast_kernel<float>, dim3((S + V_ROWS_IMAGE - 1) / V_ROWS_IMAGE), dim3(256), 0, st, tmp, S, row0,
                               (const int2 *)vbounds, vcoeffs, vk, mean0, mean1, mean2, std0, std1, std2, (float *)out, out_u8);
        MMR_CHECK_LAUNCH();
        return MMR_OK;
    }
    if (out_dtype == MMR_BF16)
        hipLaunchKernelGGL(resample_v_norm_kernel<bf16_t>, dim3((S * S + 255) / 256), dim3(256), 0, st, tmp, S, row0,
                           (const int2 *)vbounds, vcoeffs, vk, mean0, mean1, mean2, std0, std1, std2, (bf16_t *)out, out_u8);
    else
        hipLaunchKernelGGL(resample_v_norm_kernel<float>, dim3((S * S + 255) / 256), dim3(256), 0, st, tmp, S, row0,
                           (const int2 *)vbounds, vcoeffs, vk, mean0, mean1, mean2, std0, std1, std2, (float *)out, out_u8);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}
