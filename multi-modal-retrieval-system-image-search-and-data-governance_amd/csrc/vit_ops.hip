// Non-GEMM kernels of the CLIP towers for gfx950: patch gather, embeddings (+pre-LN), LayerNorm,
// attention core, pooling LayerNorm, output cast / L2-normalise.
// Arithmetic follows SURVEY.md Appendix A (transformers/models/clip/modeling_clip.py @5.15.0):
//   embeddings :209-217,251-254   pre_layrnorm :642   attention :259-277   post/final LN :559,650
// LayerNorm statistics, softmax and the residual stream are fp32; GEMM inputs are bf16.
#include "mmr_common.h"

#include <math.h>

namespace mmr {

// ---------------------------------------------------------------------------------------------
// patches: pixels[B,3,S,S] -> Ap[B*G*G (padded rows untouched), Kpad] bf16, k = (c, ky, kx), zero pad
// ---------------------------------------------------------------------------------------------
template <typename TIN>
__global__ __launch_bounds__(256) void im2col_kernel(const TIN *__restrict__ px, bf16_t *__restrict__ ap, int B,
                                                     int S, int P, int G, int K, int Kpad)
{
    const int chunks_per_row = Kpad / 8;
    const int64_t total = (int64_t)B * G * G * chunks_per_row;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(id % chunks_per_row);
        const int64_t row = id / chunks_per_row;
        const int gx = (int)(row % G), gy = (int)((row / G) % G), b = (int)(row / ((int64_t)G * G));
        float v[8];
        const int k0 = ch * 8;
        if ((P & 7) == 0 && k0 + 8 <= K) {
            // 8 consecutive kx inside one (c, ky): a contiguous pixel run
            const int c = k0 / (P * P), rem = k0 % (P * P), ky = rem / P, kx = rem % P;
            const TIN *src = px + (((size_t)b * 3 + c) * S + (gy * P + ky)) * S + gx * P + kx;
            if constexpr (sizeof(TIN) == 4) {
                const float4 a = *reinterpret_cast<const float4 *>(src);
                const float4 bq = *reinterpret_cast<const float4 *>(src + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
            } else {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(src);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = bf16_to_f32((bf16_t)a[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                float x = 0.f;
                if (k < K) {
                    const int c = k / (P * P), rem = k % (P * P), ky = rem / P, kx = rem % P;
                    const TIN *src = px + (((size_t)b * 3 + c) * S + (gy * P + ky)) * S + gx * P + kx;
                    if constexpr (sizeof(TIN) == 4) x = *src; else x = bf16_to_f32(*(const bf16_t *)src);
                }
                v[j] = x;
            }
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4 *>(ap + (size_t)row * Kpad + k0) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// row LayerNorm helpers: one wave per row, VPL = d/64 values per lane held in registers
// ---------------------------------------------------------------------------------------------
// wave-wide sum in plain VALU: DPP inside each row of 16 lanes, then gfx950's row swaps across the four rows.
// (__shfl_xor lowers to ds_bpermute: six dependent LDS-crossbar round trips per reduction.)
__device__ __forceinline__ float wave_sum_valu(float v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    v = row16_sum(v);
    const u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a.x) + __uint_as_float(a.y);           // .x/.y + __uint_as_float: see quad_rows_reduce
    const u32x2 c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(c.x) + __uint_as_float(c.y);
}

template <int VPL>
__device__ __forceinline__ void ln_row(float (&x)[VPL], const float *__restrict__ w, const float *__restrict__ b,
                                       int lane, int d, float eps)
{
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) s += x[j];
    const float mean = wave_sum_valu(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) { const float t = x[j] - mean; ss += t * t; }
    const float rstd = rsqrtf(wave_sum_valu(ss) / (float)d + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int col = j * 64 + lane;
        x[j] = (x[j] - mean) * rstd * w[col] + b[col];
    }
}

// LN-fold producers (tower.hip, fold_ln): alongside h emit x = bf16(h) and the row's (sum, sumsq) in the
// partial-sum layout the GEMM epilogue reads (mmr_common.h GemmAux): slot 0 carries the row, the rest are 0.
template <int VPL>
__device__ __forceinline__ void emit_fold_inputs(const float (&x)[VPL], int lane, int64_t row, int d, bf16_t *__restrict__ xb,
                                                 float2 *__restrict__ stats)
{
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        s += x[j];
        ss += x[j] * x[j];
        xb[(size_t)row * d + j * 64 + lane] = f32_to_bf16(x[j]);
    }
    s = wave_sum_valu(s);
    ss = wave_sum_valu(ss);
    if (lane < LNFOLD_NP) stats[(size_t)row * LNFOLD_NP + lane] = lane == 0 ? make_float2(s, ss) : make_float2(0.f, 0.f);
}

// h[b*T + t] = LN_pre( (t == 0 ? cls : pe[b*G*G + t-1]) + pos[t] )        (fp32 residual stream)
template <int VPL>
__global__ __launch_bounds__(256) void embed_vision_kernel(const float *__restrict__ pe, const float *__restrict__ cls,
                                                           const float *__restrict__ pos, const float *__restrict__ lw,
                                                           const float *__restrict__ lb, float *__restrict__ h, int B,
                                                           int T, int d, float eps, bf16_t *__restrict__ xb,
                                                           float2 *__restrict__ stats, int32_t *__restrict__ status)
{
    const int lane = threadIdx.x & 63;
    // the workspace's status word (include/mmr.h): nothing in a vision forward can raise a bit, so the word is zeroed
    // here instead of by a memset launch of its own (one dependent launch less on the batch-1 path)
    if (blockIdx.x == 0 && threadIdx.x == 0) *status = 0;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * T) return;
    const int t = (int)(row % T);
    const int64_t b = row / T;
    const float *src = t == 0 ? cls : pe + (size_t)(b * (T - 1) + t - 1) * d;
    float x[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) x[j] = src[j * 64 + lane] + pos[(size_t)t * d + j * 64 + lane];
    ln_row<VPL>(x, lw, lb, lane, d, eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) h[(size_t)row * d + j * 64 + lane] = x[j];
    if (xb) emit_fold_inputs<VPL>(x, lane, row, d, xb, stats);
}

// h[n*T + t] = tok[ids[n,t]] + pos[t]
template <int VPL>
__global__ __launch_bounds__(256) void embed_text_kernel(const int32_t *__restrict__ ids, const bf16_t *__restrict__ tok,
                                                         const float *__restrict__ pos, float *__restrict__ h, int Nb,
                                                         int T, int d, int vocab, bf16_t *__restrict__ xb,
                                                         float2 *__restrict__ stats, int32_t *__restrict__ status)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)Nb * T) return;
    const int t = (int)(row % T);
    int id = ids[row];
    // out-of-range ids: clamped (memory guard) and reported through the workspace's status word -- the host does
    // not read ids that already live on the GPU (that would be a sync per call)
    if ((id < 0 || id >= vocab) && lane == 0) atomicOr(status, 1);
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    float x[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int col = j * 64 + lane;
        x[j] = bf16_to_f32(tok[(size_t)id * d + col]) + pos[(size_t)t * d + col];
        h[(size_t)row * d + col] = x[j];
    }
    if (xb) emit_fold_inputs<VPL>(x, lane, row, d, xb, stats);
}

// x_bf16[row] = LN(h[row]).  RPW rows per wave, all of their loads issued before the first reduction (several rounds of
// waves over the chip's 8 192 wave slots otherwise pay the load latency once per round).
template <int VPL, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ h, const float *__restrict__ w,
                                                        const float *__restrict__ b, bf16_t *__restrict__ x, int64_t rows,
                                                        int d, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    float v[RPW][VPL];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;      // a ragged last wave re-reads the last row
#pragma unroll
        for (int j = 0; j < VPL; ++j) v[r][j] = h[(size_t)row * d + j * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        ln_row<VPL>(v[r], w, b, lane, d, eps);
        if (row0 + r < rows) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) x[(size_t)(row0 + r) * d + j * 64 + lane] = f32_to_bf16(v[r][j]);
        }
    }
}

// BERT embeddings (transformers/models/bert/modeling_bert.py:53-108): word[ids] + type[0] + pos[t] -> LayerNorm
// -> h (fp32) and x = bf16(h)
template <int VPL>
__global__ __launch_bounds__(256) void embed_bert_kernel(const int32_t *__restrict__ ids, const bf16_t *__restrict__ tok,
                                                         const float *__restrict__ pos, const float *__restrict__ type0,
                                                         const float *__restrict__ lw, const float *__restrict__ lb,
                                                         float *__restrict__ h, bf16_t *__restrict__ x, int Nb, int T,
                                                         int d, int vocab, float eps, const int32_t *__restrict__ types,
                                                         int32_t *__restrict__ status)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)Nb * T) return;
    const int t = (int)(row % T);
    int id = ids[row];
    int ty = types ? types[row] : 0;                       // token_type_ids (segment 0 / 1); absent = all zero
    if ((id < 0 || id >= vocab || ty < 0 || ty > 1) && lane == 0) atomicOr(status, 1);
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    ty = ty < 0 ? 0 : (ty > 1 ? 1 : ty);
    const float *typ = type0 + (size_t)ty * d;
    float v[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int col = j * 64 + lane;
        // same association as HF: (word + type) + position
        v[j] = (bf16_to_f32(tok[(size_t)id * d + col]) + typ[col]) + pos[(size_t)t * d + col];
    }
    ln_row<VPL>(v, lw, lb, lane, d, eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        h[(size_t)row * d + j * 64 + lane] = v[j];
        x[(size_t)row * d + j * 64 + lane] = f32_to_bf16(v[j]);
    }
}

// post-LN blocks (BERT): h = LN(h) in place, x = bf16(h)
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_inplace_kernel(float *__restrict__ h, const float *__restrict__ w,
                                                                const float *__restrict__ b, bf16_t *__restrict__ x,
                                                                int64_t rows, int d, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) v[j] = h[(size_t)row * d + j * 64 + lane];
    ln_row<VPL>(v, w, b, lane, d, eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        h[(size_t)row * d + j * 64 + lane] = v[j];
        x[(size_t)row * d + j * 64 + lane] = f32_to_bf16(v[j]);
    }
}

// xc[n] = x[n*T]  (first token of every sequence, bf16 copy for the BERT pooler)
__global__ __launch_bounds__(256) void gather_first_rows_kernel(const bf16_t *__restrict__ x, bf16_t *__restrict__ xc, int Nb,
                                                                int T, int d)
{
    const int chunks = d / 8;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= Nb * chunks) return;
    const int n = id / chunks, c = id % chunks;
    *reinterpret_cast<uint4 *>(xc + (size_t)n * d + c * 8) = *reinterpret_cast<const uint4 *>(x + (size_t)n * T * d + c * 8);
}

// pooled rows: vision -> token 0 of each image; text -> first position of the largest id (EOT).
// xc_bf16[n] = LN(h[n*T + pick])
template <int VPL>
__global__ __launch_bounds__(256) void pool_ln_kernel(const float *__restrict__ h, const int32_t *__restrict__ ids,
                                                      const float *__restrict__ w, const float *__restrict__ b,
                                                      bf16_t *__restrict__ xc, int Nb, int T, int d, float eps)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= Nb) return;
    int pick = 0;
    if (ids) {
        int best = -2147483647 - 1, bpos = 0;
        for (int t = lane; t < T; t += 64) {
            const int v = ids[(size_t)n * T + t];
            if (v > best) { best = v; bpos = t; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int ov = __shfl_xor(best, off, 64), op = __shfl_xor(bpos, off, 64);
            if (ov > best || (ov == best && op < bpos)) { best = ov; bpos = op; }
        }
        pick = bpos;
    }
    const float *src = h + ((size_t)n * T + pick) * d;
    float v[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) v[j] = src[j * 64 + lane];
    ln_row<VPL>(v, w, b, lane, d, eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) xc[(size_t)n * d + j * 64 + lane] = f32_to_bf16(v[j]);
}

// out[n] = (normalize ? f / ||f|| : f) cast to the output dtype
template <typename TOUT>
__global__ __launch_bounds__(256) void finish_kernel(const float *__restrict__ feat, TOUT *__restrict__ out, int Nb, int E,
                                                     int normalize)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= Nb) return;
    const float *f = feat + (size_t)n * E;
    float inv = 1.f;
    if (normalize) {
        float ss = 0.f;
        for (int j = lane; j < E; j += 64) ss += f[j] * f[j];
        inv = 1.0f / sqrtf(wave_sum(ss));
    }
    for (int j = lane; j < E; j += 64) {
        const float v = f[j] * inv;
        if constexpr (sizeof(TOUT) == 2) ((bf16_t *)out)[(size_t)n * E + j] = f32_to_bf16(v);
        else ((float *)out)[(size_t)n * E + j] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// attention helpers
// ---------------------------------------------------------------------------------------------
// ds_read_b64_tr_b16 (gfx950): per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered column-major:
// lane 4q+p of the group supplies the address of row q, columns 4p..4p+3; lane i receives column i of the 4 rows.
// With V stored row-major [key][dim] this is the "8 keys of one dim per lane" operand of O^T = V^T.P^T for free.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));   // native vector types: asm operands stay in registers
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2_t lds_read_tr16(const char *addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)addr) : "memory");
    return v;
}

// reduce over the 4 lanes that share a query (lane, lane^16, lane^32, lane^48) with gfx950's row swaps (plain VALU;
// __shfl_xor would go through ds_bpermute and the LDS crossbar's latency)
template <typename F>
__device__ __forceinline__ float quad_rows_reduce(float v, F op) {
#ifdef MMR_ATTN_SHFL
    v = op(v, __shfl_xor(v, 16, 64));
    return op(v, __shfl_xor(v, 32, 64));
#endif
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    // NB: read the two results as .x/.y through __uint_as_float.  __builtin_bit_cast(float, a[1]) on the returned
    // ext-vector is miscompiled by ROCm 7.2 clang (element 0 is used for both: the ISA shows v_add v1, v1, v1).
    const u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);   // rows 0<->1, 2<->3
    v = op(__uint_as_float(a.x), __uint_as_float(a.y));
    const u32x2 c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);   // halves
    return op(__uint_as_float(c.x), __uint_as_float(c.y));
}

// ---------------------------------------------------------------------------------------------
// attention core: one workgroup per (image|text, head); head dim 64; T <= 16*NT (NT even)
//   S^T = K.Q^T (keys on the MFMA row axis, so the softmax reduction over keys is mostly in-lane
//   and the bf16 P tile is already the B operand of the P.V product), softmax fp32, O^T = V^T.P^T.
// ---------------------------------------------------------------------------------------------
//   MASKED: a key-padding mask kmask[B,T] (int32, non-zero = attend), HF's `attention_mask` for padded text batches
//   (BERT text tower): masked keys get weight 0.  A sequence whose keys are ALL masked yields zeros (HF would
//   yield the mean of V; no caller produces such a row).
// Round 3: the workgroup is PERSISTENT over (sequence, head) pairs and prefetches.  Round 2 ran one workgroup per pair --
// load K/V/Q, barrier, compute, store -- and measured 17 us per ViT-B/32 layer at batch 256 for 79 MB of traffic (4.6 TB/s
// of data that sits in the Infinity Cache): every workgroup spends half its life waiting for its own loads, and with 5-6
// workgroups per CU there are not enough of them to cover that.  Now a workgroup walks pairs blockIdx.x, + gridDim.x, ...
// and while it computes pair k the NEXT pair's K and V rows are already streaming into the other half of its LDS by
// LDS-DMA (global_load_lds, 16 B per lane, the XOR swizzle applied on the source address so the image stays lane-linear)
// and its Q fragments into registers -- no load of a pair is ever waited for before a pair's worth of compute has passed.
// The grid is sized so every workgroup gets the same number of pairs (3072 pairs -> 1024 workgroups x 3).
// Rows past T are staged from row T-1 instead of zeros: keys >= T are masked to -inf explicitly and their probabilities
// are exact zeros, so finite filler contributes exact zeros to P.V -- the result is bit-identical to the zero-filled form.
// The Q loads are inline asm (as C++ loads hipcc would wait vmcnt(0) at their first use -- LDS-DMA and register loads are
// "mixed events" on one counter -- and drain the prefetch); they are waited for and tied before the loop's back edge.
template <int NT, bool CAUSAL, bool MASKED>
__global__ __launch_bounds__(256, NT <= 4 ? (MASKED ? 4 : 5) : 3) void attention_kernel(
    const bf16_t *__restrict__ qkv, bf16_t *__restrict__ o, int T, int d, float scale, const int32_t *__restrict__ kmask,
    int heads, int npairs)
{
    constexpr int TPAD = NT * 16;
    constexpr int BUF = 2 * TPAD * 128 + (MASKED ? TPAD * 4 : 0);     // K image + V image (+ key mask) of one pair
    constexpr int NQ = NT > 4 ? 2 : 1;                                  // query blocks a wave can own (nqb <= NT, 4 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t ld = (size_t)3 * d;
    const int fr = lane & 15, fg = lane >> 4;
    const int nqb = (T + 15) / 16;

    // K and V rows of `pair` -> buffer `buf`: 8-row blocks (1 KiB), wave w stages blocks w, w + 4, ...
    auto stage = [&](int pair, int buf) {
        const int b = pair / heads, hd = pair - b * heads;
        const bf16_t *base = qkv + (size_t)b * T * ld + hd * 64;
        char *Ks = smem + buf * BUF, *Vs = Ks + TPAD * 128;
#pragma unroll
        for (int blk0 = 0; blk0 < TPAD / 8; blk0 += 4) {
            const int blk = blk0 + wave;
            if (blk < TPAD / 8) {
                const int row = blk * 8 + (lane >> 3);
                const int srow = row < T ? row : T - 1;
                const int c = (lane & 7) ^ (row & 7);              // source chunk that lands in LDS slot (lane & 7)
                glds16(base + (size_t)srow * ld + d + c * 8, Ks + blk * 1024);
                glds16(base + (size_t)srow * ld + 2 * d + c * 8, Vs + blk * 1024);
            }
        }
        if constexpr (MASKED) {
            int32_t *Ms = reinterpret_cast<int32_t *>(smem + buf * BUF + 2 * TPAD * 128);
#pragma unroll
            for (int i0 = 0; i0 < TPAD; i0 += 256) {
                const int i = i0 + wave * 64;                      // wave-uniform 64-key slab
                if (i < TPAD) {
                    const int key = i + lane < T ? i + lane : T - 1;   // keys >= T are masked by index anyway
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kmask + (size_t)b * T + key),
                                                     (__attribute__((address_space(3))) void *)(Ms + i), 4, 0, 0);
                }
            }
        }
    };
    // Q fragments of this wave's query blocks (qb = wave, wave + 4) of `pair`
    auto load_q = [&](int pair, u32x4_t (&q)[NQ][2]) {
        const int b = pair / heads, hd = pair - b * heads;
        const bf16_t *base = qkv + (size_t)b * T * ld + hd * 64;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int qi = (wave + 4 * j) * 16 + fr;
            const bf16_t *src = base + (size_t)(qi < T ? qi : T - 1) * ld + fg * 8;
#pragma unroll
            for (int s = 0; s < 2; ++s)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[j][s]) : "v"(src + s * 32) : "memory");
        }
    };
    auto tie_q = [&](u32x4_t (&q)[NQ][2]) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(q[j][0]), "+v"(q[j][1]));
    };

    int pair = blockIdx.x;
    if (pair >= npairs) return;
    u32x4_t qcur[NQ][2], qnext[NQ][2];
#pragma unroll
    for (int j = 0; j < NQ; ++j) { qnext[j][0] = (u32x4_t){0, 0, 0, 0}; qnext[j][1] = (u32x4_t){0, 0, 0, 0}; }
    load_q(pair, qcur);
    stage(pair, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie_q(qcur);
    __builtin_amdgcn_s_barrier();

    const int tq = fr >> 2, tp = fr & 3, trow = 4 * fg + tq;   // transposed-read address roles (see lds_read_tr16)
    const float c2 = scale * 1.44269504088896341f;            // softmax in the exp2 domain: exp2(s*c2 - max*c2)
    int buf = 0;
    for (; pair < npairs; pair += gridDim.x, buf ^= 1) {
        const int next = pair + gridDim.x;
        if (next < npairs) {               // the other buffer was last read one iteration ago, before the closing barrier
            load_q(next, qnext);
            stage(next, buf ^ 1);
        }
        const int b = pair / heads, hd = pair - b * heads;
        const char *Ks = smem + buf * BUF, *Vs = Ks + TPAD * 128;
        const int32_t *Ms = reinterpret_cast<const int32_t *>(smem + buf * BUF + 2 * TPAD * 128);   // MASKED only
        int stores = 0;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int qb = wave + 4 * j;
            if (qb >= nqb) break;
            stores += 4;
            const int qi = qb * 16 + fr;  // this lane's query (column of S^T)
            const bf16x8 qf[2] = {__builtin_bit_cast(bf16x8, qcur[j][0]), __builtin_bit_cast(bf16x8, qcur[j][1])};
            // CAUSAL: key tiles past this query block's diagonal tile are masked for all 16 queries: the wave skips their
            // MFMAs, exponentials and P.V steps (they contributed exact zeros, so the result is bit-identical); qb is wave-uniform
            f32x4 sc[NT];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!CAUSAL || jt <= qb) {
                    const int row = jt * 16 + fr;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(Ks + row * 128 + (((s * 4 + fg) ^ (row & 7)) << 4));
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], a, 0, 0, 0);
                    }
                }
                sc[jt] = a;
            }
            // V fragments, issued before the softmax so their latency hides under it:
            // k-slot (fg, jj) of k-step s2 <-> key 16*(2*s2 + (jj>>2)) + 4*fg + (jj&3)
            u32x2_t vraw[NT / 2][4][2];
#pragma unroll
            for (int s2 = 0; s2 < NT / 2; ++s2) {
                if (CAUSAL && 2 * s2 > qb) {          // both key tiles of the pair are past the diagonal: never read, never used
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) { vraw[s2][dt][0] = (u32x2_t){0, 0}; vraw[s2][dt][1] = (u32x2_t){0, 0}; }
                    continue;
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int r0 = 32 * s2 + trow;
                    const int ch = ((dt * 2 + (tp >> 1)) ^ (trow & 7)) << 4;
                    vraw[s2][dt][0] = lds_read_tr16(Vs + r0 * 128 + ch + 8 * (tp & 1));
                    vraw[s2][dt][1] = lds_read_tr16(Vs + (r0 + 16) * 128 + ch + 8 * (tp & 1));
                }
            }
            // sc[jt][r] = S[query qi][key jt*16 + 4*fg + r]
            float mx = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                if (CAUSAL && jt > qb) continue;
                int4 mk = make_int4(1, 1, 1, 1);
                if constexpr (MASKED) mk = *reinterpret_cast<const int4 *>(Ms + jt * 16 + fg * 4);
                const int mkr[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = jt * 16 + fg * 4 + r;
                    if (key >= T || (CAUSAL && key > qi) || (MASKED && mkr[r] == 0)) sc[jt][r] = -INFINITY;
                    mx = fmaxf(mx, sc[jt][r]);
                }
            }
            mx = quad_rows_reduce(mx, [](float p, float q) { return fmaxf(p, q); });
            const float m2 = mx == -INFINITY ? 0.f : mx * c2;      // scale > 0: max commutes with the scaling
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                if (CAUSAL && jt > qb) continue;            // sc[jt] stays 0 = the probability of a masked key
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[jt][r], c2, -m2));
                    sc[jt][r] = p;
                    sum += p;
                }
            }
            sum = quad_rows_reduce(sum, [](float p, float q) { return p + q; });
            const float inv = (MASKED && sum == 0.f) ? 0.f : 1.f / sum;

            // the transposed reads are invisible to the compiler's counters: wait, then re-define the raw registers here so
            // no copy into the MFMA operand tuples can be scheduled before the data has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s2 = 0; s2 < NT / 2; ++s2)
                asm volatile("" : "+v"(vraw[s2][0][0]), "+v"(vraw[s2][0][1]), "+v"(vraw[s2][1][0]), "+v"(vraw[s2][1][1]),
                                  "+v"(vraw[s2][2][0]), "+v"(vraw[s2][2][1]), "+v"(vraw[s2][3][0]), "+v"(vraw[s2][3][1]));
            __builtin_amdgcn_sched_barrier(0);
            f32x4 oacc[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < NT / 2; ++s2) {
                if (CAUSAL && 2 * s2 > qb) continue;
                union { bf16x8 v; uint32_t u[4]; } pf;
                pf.u[0] = pack_bf16x2(sc[2 * s2][0] * inv, sc[2 * s2][1] * inv);
                pf.u[1] = pack_bf16x2(sc[2 * s2][2] * inv, sc[2 * s2][3] * inv);
                pf.u[2] = pack_bf16x2(sc[2 * s2 + 1][0] * inv, sc[2 * s2 + 1][1] * inv);
                pf.u[3] = pack_bf16x2(sc[2 * s2 + 1][2] * inv, sc[2 * s2 + 1][3] * inv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vraw[s2][dt][0], vraw[s2][dt][1], 0, 1, 2, 3));
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf.v, oacc[dt], 0, 0, 0);
                }
            }
            // oacc[dt][r] = O[query qi][dim dt*16 + 4*fg + r]; rows past T: stored nowhere (the 4 store instructions are
            // still ISSUED by the wave, lanes masked -- the counted wait below relies on that)
            bf16_t *dst = o + ((size_t)b * T + (qi < T ? qi : T - 1)) * d + hd * 64 + fg * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 pk;
                pk.x = pack_bf16x2(oacc[dt][0], oacc[dt][1]);
                pk.y = pack_bf16x2(oacc[dt][2], oacc[dt][3]);
                if (qi < T) *reinterpret_cast<uint2 *>(dst + dt * 16) = pk;
            }
        }
        // the next pair's loads (issued a whole pair of compute ago) must have landed; this pair's output stores, the
        // YOUNGEST queue entries, may stay in flight (4 per query block processed; a masked store still occupies a slot
        // unless the whole wave skipped it, hence the conservative 0 for partial blocks is not needed: qb < nqb blocks
        // always have a live lane)
        if (stores == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (stores == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tie_q(qnext);
#pragma unroll
        for (int j = 0; j < NQ; ++j) { qcur[j][0] = qnext[j][0]; qcur[j][1] = qnext[j][1]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// attention core for long sequences (T = 257, 577: ViT-L/14): flash-style streaming.
//   grid (ceil(T/64), heads, B); a workgroup owns 64 queries (one 16-query block per wave) and streams
//   the head's keys/values in blocks of 64 through a double-buffered 32 KiB LDS image, so several
//   workgroups share a CU and the next block's global loads (issued to registers before the block's
//   MFMAs, written to LDS after them) overlap the current block's compute.  Online softmax: running
//   max per query, un-normalised bf16 P, the 64x16 output tile rescaled only when a running max moved.
//   K and V are both stored row-major [key][64 dims] with the XOR chunk swizzle; S^T = K.Q^T reads K rows
//   with ds_read_b128, and O^T = V^T.P^T needs V column-wise (8 keys of one dim per lane), which gfx950's
//   ds_read_b64_tr_b16 delivers straight from the row-major image (4 keys x 16 dims per 16-lane group,
//   conflict-free on this image) -- no transposing 2-byte LDS writes.
//   The softmax runs in the log2 domain on the raw dot products: q.k * (scale*log2e) - m, one FMA and one
//   v_exp_f32 per score; keys past T are masked only in the last block.
// ---------------------------------------------------------------------------------------------
constexpr int AKB = 64;                       // keys per block
constexpr int AIMG = AKB * 128;               // one [64 keys][64 dims] bf16 image
constexpr int ABUF = 2 * AIMG;                // one buffer: K image + V image


// QB = 16-query blocks per wave (1 or 2).  With QB = 2 every K row fragment and every transposed V read feeds two
// MFMAs, which halves the LDS bytes per FLOP -- the limiter at QB = 1, where each wave re-reads the whole 16 KiB
// K/V block for 16 queries -- at the price of 128-query workgroups (more padding when T mod 128 is small).
// MASKED: key-padding mask as in attention_kernel; the block's 64 mask words ride in a small LDS array per buffer.
template <bool CAUSAL, int QB, bool MASKED>
__global__ __launch_bounds__(256) void attention_stream_kernel(const bf16_t *__restrict__ qkv, bf16_t *__restrict__ o,
                                                               int T, int d, float scale, const int32_t *__restrict__ kmask)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * ABUF + (MASKED ? 2 * AKB * 4 : 0)];
    int32_t *Ms = reinterpret_cast<int32_t *>(smem + 2 * ABUF);     // MASKED: [2][AKB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hd = blockIdx.y, b = blockIdx.z;
    const size_t ld = (size_t)3 * d;
    const bf16_t *base = qkv + (size_t)b * T * ld + hd * 64;
    const int fr = lane & 15, fg = lane >> 4;
    const int q0 = (blockIdx.x * 4 + wave) * (16 * QB) + fr;   // this lane's first query; block x adds 16*x
    const int nkb = (T + AKB - 1) / AKB;
    const float c2 = scale * 1.44269504088896341f;          // scores enter exp2 as s*c2 - m

    bf16x8 qf[QB][2];
#pragma unroll
    for (int x = 0; x < QB; ++x)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[x][s] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (q0 + 16 * x < T) qf[x][s] = *reinterpret_cast<const bf16x8 *>(base + (size_t)(q0 + 16 * x) * ld + s * 32 + fg * 8);
        }

    // each thread moves 2 K chunks and 2 V chunks (16 B each) per key block
    uint4 kreg[2], vreg[2];
    int32_t mreg = 0;
    auto gload = [&](int kb) {
        if constexpr (MASKED) {
            const int key = kb * AKB + tid;
            mreg = (tid < AKB && key < T) ? kmask[(size_t)b * T + key] : 0;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256, row = id >> 3, c = id & 7, key = kb * AKB + row;
            kreg[i] = make_uint4(0, 0, 0, 0);
            vreg[i] = make_uint4(0, 0, 0, 0);
            if (key < T) {
                kreg[i] = *reinterpret_cast<const uint4 *>(base + (size_t)key * ld + d + c * 8);
                vreg[i] = *reinterpret_cast<const uint4 *>(base + (size_t)key * ld + 2 * d + c * 8);
            }
        }
    };
    auto lwrite = [&](int buf) {
        char *Ks = smem + buf * ABUF, *Vs = Ks + AIMG;
        if constexpr (MASKED) { if (tid < AKB) Ms[buf * AKB + tid] = mreg; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256, row = id >> 3, c = id & 7;
            const int off = row * 128 + ((c ^ (row & 7)) << 4);
            *reinterpret_cast<uint4 *>(Ks + off) = kreg[i];
            *reinterpret_cast<uint4 *>(Vs + off) = vreg[i];
        }
    };
    // transposed-read address of this lane inside a V image: lane 4q+p of its 16-lane group supplies key r0+q,
    // dims 16*dt + 4p .. 4p+3, for the block of keys r0 = 4*fg (+ a multiple of 16)
    const int tq = fr >> 2, tp = fr & 3;
    const int trow = 4 * fg + tq;                            // key row inside a 16-key group; (row & 7) = trow & 7

    float m[QB], l[QB];                       // running max in the exp2 domain (uniform over a query's 4 lanes), partial sum
    f32x4 oacc[QB][4];
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        m[x] = -INFINITY;
        l[x] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[x][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    gload(0);
    lwrite(0);
    __syncthreads();
    // One key block.  TAIL = the sequence's last block, which may hold as little as ONE valid key (T = 577 = 9 x 64 + 1):
    // only its first njt 16-key tiles are multiplied, exponentiated and fed to P.V (the rest contributed exact zeros).
    auto block = [&](int kb, auto tailc) {
        constexpr bool TAIL = decltype(tailc)::value;
        const int buf = kb & 1;
        if constexpr (!TAIL) gload(kb + 1);    // in flight during this block's MFMAs
        const int njt = TAIL ? (T - kb * AKB + 15) / 16 : 4;       // wave-uniform
        const char *Ks = smem + buf * ABUF, *Vs = Ks + AIMG;

        f32x4 sc[QB][4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            if (TAIL && jt >= njt) {
#pragma unroll
                for (int x = 0; x < QB; ++x) sc[x][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const int row = jt * 16 + fr;
            bf16x8 kf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
                kf[s] = *reinterpret_cast<const bf16x8 *>(Ks + row * 128 + (((s * 4 + fg) ^ (row & 7)) << 4));
#pragma unroll
            for (int x = 0; x < QB; ++x) {
                f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s], qf[x][s], a, 0, 0, 0);
                sc[x][jt] = a;
            }
        }
        // V fragments for the whole block, issued now so their LDS latency hides under the softmax:
        // the MFMA's k slots 8*fg .. 8*fg+7 are keys 32*s2 + 4*fg + {0..3} and 32*s2 + 16 + 4*fg + {0..3}
        u32x2_t vraw[2][4][2];                 // [s2][dt][half]; not to be touched before the wait below
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (TAIL && 2 * s2 >= njt) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { vraw[s2][dt][0] = (u32x2_t){0, 0}; vraw[s2][dt][1] = (u32x2_t){0, 0}; }
                continue;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int r0 = 32 * s2 + trow;
                const int ch = ((dt * 2 + (tp >> 1)) ^ (trow & 7)) << 4;      // (r0 + 16) & 7 == r0 & 7 == trow & 7
                vraw[s2][dt][0] = lds_read_tr16(Vs + r0 * 128 + ch + 8 * (tp & 1));
                vraw[s2][dt][1] = lds_read_tr16(Vs + (r0 + 16) * 128 + ch + 8 * (tp & 1));
            }
        }
        // keys past T (last block only) and, for the causal form, keys after the query
        if (CAUSAL || MASKED || TAIL) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                if (TAIL && jt >= njt) continue;
                int4 mk = make_int4(1, 1, 1, 1);
                if constexpr (MASKED) mk = *reinterpret_cast<const int4 *>(Ms + buf * AKB + jt * 16 + fg * 4);
                const int mkr[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
                for (int x = 0; x < QB; ++x)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kb * AKB + jt * 16 + fg * 4 + r;
                        if (key >= T || (CAUSAL && key > q0 + 16 * x) || (MASKED && mkr[r] == 0)) sc[x][jt][r] = -INFINITY;
                    }
            }
        }
        bool moved = false;
        float alpha[QB];
#pragma unroll
        for (int x = 0; x < QB; ++x) {
            float bm = fmaxf(fmaxf(sc[x][0][0], sc[x][0][1]), fmaxf(sc[x][0][2], sc[x][0][3]));
#pragma unroll
            for (int jt = 1; jt < 4; ++jt)
                if (!TAIL || jt < njt)
                    bm = fmaxf(bm, fmaxf(fmaxf(sc[x][jt][0], sc[x][jt][1]), fmaxf(sc[x][jt][2], sc[x][jt][3])));
            bm = quad_rows_reduce(bm, [](float p, float q) { return fmaxf(p, q); });
            const float mn = fmaxf(m[x], bm * c2);                // scale > 0: max commutes with the scaling
            const float msafe = mn == -INFINITY ? 0.f : mn;       // fully masked so far (causal padding rows)
            moved |= mn != m[x];
            alpha[x] = __builtin_amdgcn_exp2f(m[x] - msafe);
            m[x] = mn;
            float ps = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                if (TAIL && jt >= njt) continue;                  // sc stays 0 = the probability of a key past T
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[x][jt][r], c2, -msafe));
                    sc[x][jt][r] = p;
                    ps += p;
                }
            }
            l[x] = l[x] * alpha[x] + ps;
        }
        if (__any(moved)) {                    // wave-uniform: rescale only when some query's running max moved
#pragma unroll
            for (int x = 0; x < QB; ++x)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[x][dt][r] *= alpha[x];
        }
        // The transposed V reads are invisible to the compiler's counters: wait for them, and re-define the raw
        // registers at this point ("+v") so that no copy into the MFMA operand tuples can be scheduled before the
        // data has landed.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
            asm volatile("" : "+v"(vraw[s2][0][0]), "+v"(vraw[s2][0][1]), "+v"(vraw[s2][1][0]), "+v"(vraw[s2][1][1]),
                              "+v"(vraw[s2][2][0]), "+v"(vraw[s2][2][1]), "+v"(vraw[s2][3][0]), "+v"(vraw[s2][3][1]));
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 vf[2][4];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                vf[s2][dt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vraw[s2][dt][0], vraw[s2][dt][1], 0, 1, 2, 3));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (TAIL && 2 * s2 >= njt) continue;
            union { bf16x8 v; uint32_t u[4]; } pf[QB];
#pragma unroll
            for (int x = 0; x < QB; ++x) {
                pf[x].u[0] = pack_bf16x2(sc[x][2 * s2][0], sc[x][2 * s2][1]);
                pf[x].u[1] = pack_bf16x2(sc[x][2 * s2][2], sc[x][2 * s2][3]);
                pf[x].u[2] = pack_bf16x2(sc[x][2 * s2 + 1][0], sc[x][2 * s2 + 1][1]);
                pf[x].u[3] = pack_bf16x2(sc[x][2 * s2 + 1][2], sc[x][2 * s2 + 1][3]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int x = 0; x < QB; ++x)
                    oacc[x][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[s2][dt], pf[x].v, oacc[x][dt], 0, 0, 0);
        }
        if constexpr (!TAIL) {
            lwrite(buf ^ 1);                   // the other buffer was last read in iteration kb-1
            __syncthreads();
        }
    };
    for (int kb = 0; kb + 1 < nkb; ++kb) block(kb, std::false_type{});
    block(nkb - 1, std::true_type{});
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        const float lx = quad_rows_reduce(l[x], [](float p, float q) { return p + q; });
        const float inv = (MASKED && lx == 0.f) ? 0.f : 1.f / lx;
        const int qi = q0 + 16 * x;
        if (qi < T) {
            bf16_t *dst = o + ((size_t)b * T + qi) * d + hd * 64 + fg * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 pk;
                pk.x = pack_bf16x2(oacc[x][dt][0] * inv, oacc[x][dt][1] * inv);
                pk.y = pack_bf16x2(oacc[x][dt][2] * inv, oacc[x][dt][3] * inv);
                *reinterpret_cast<uint2 *>(dst + dt * 16) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers (internal; argument validation is done by tower.hip)
// ---------------------------------------------------------------------------------------------
#define MMR_VPL_SWITCH(d, ...)                                                     \
    switch ((d) / 64) {                                                            \
        case 2: { constexpr int VPL = 2; __VA_ARGS__; } break;                     \
        case 8: { constexpr int VPL = 8; __VA_ARGS__; } break;                     \
        case 12: { constexpr int VPL = 12; __VA_ARGS__; } break;                   \
        case 16: { constexpr int VPL = 16; __VA_ARGS__; } break;                   \
        default: set_error("width %d unsupported (128, 512, 768, 1024)", (d)); return MMR_ENOTSUP; \
    }

int launch_im2col(const void *px, mmr_dtype dt, bf16_t *ap, int B, int S, int P, int G, int K, int Kpad, hipStream_t st)
{
    ProfScope prof(MMR_PROF_ROWWISE, st);
    const int64_t total = (int64_t)B * G * G * (Kpad / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dt == MMR_F32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)px, ap, B, S, P, G, K, Kpad);
    else hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t *)px, ap, B, S, P, G, K, Kpad);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

int launch_embed_vision(const float *pe, const float *cls, const float *pos, const float *lw, const float *lb, float *h,
                        int B, int T, int d, float eps, bf16_t *xb, float2 *stats, int32_t *status, hipStream_t st)
{
    ProfScope prof(MMR_PROF_ROWWISE, st);
    const dim3 grid((unsigned)(((int64_t)B * T + 3) / 4));
    MMR_VPL_SWITCH(d, hipLaunchKernelGGL(embed_vision_kernel<VPL>, grid, dim3(256), 0, st, pe, cls, pos, lw, lb, h, B, T, d, eps, xb, stats, status));
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

int launch_embed_text(const int32_t *ids, const bf16_t *tok, const float *pos, float *h, int Nb, int T, int d, int vocab,
                      bf16_t *xb, float2 *stats, int32_t *status, hipStream_t st)
{
    ProfScope prof(MMR_PROF_ROWWISE, st);
    const dim3 grid((unsigned)(((int64_t)Nb * T + 3) / 4));
    MMR_VPL_SWITCH(d, hipLaunchKernelGGL(embed_text_kernel<VPL>, grid, dim3(256), 0, st, ids, tok, pos, h, Nb, T, d, vocab, xb, stats, status));
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

int launch_layernorm(const float *h, const float *w, const float *b, bf16_t *x, int64_t rows, int d, float eps,
                     hipStream_t st)
{
    ProfScope prof(MMR_PROF_ROWWISE, st);
    static const int force_rpw = getenv("MMR_LN_RPW") ? atoi(getenv("MMR_LN_RPW")) : 0;      // 1 / 2: A/B aid
    // measured: no change at 12 800 rows (ViT-B/32 batch 256), -6 % at 73 856 rows (ViT-L/14@336 batch 128)
    const bool two = force_rpw ? force_rpw == 2 : rows >= 32768;
    if (two) {
        const dim3 grid((unsigned)((rows + 7) / 8));
        MMR_VPL_SWITCH(d, hipLaunchKernelGGL((layernorm_kernel<VPL, 2>), grid, dim3(256), 0, st, h, w, b, x, rows, d, eps));
    } else {
        const dim3 grid((unsigned)((rows + 3) / 4));
        MMR_VPL_SWITCH(d, hipLaunchKernelGGL((layernorm_kernel<VPL, 1>), grid, dim3(256), 0, st, h, w, b, x, rows, d, eps));
    }
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

int launch_pool_ln(const float *h, const int32_t *ids, const float *w, const float *b, bf16_t *xc, int Nb, int T, int d,
                   float eps, hipStream_t st)
{
    ProfScope prof(MMR_PROF_ROWWISE, st);
    const dim3 grid((unsigned)((Nb + 3) / 4));
    MMR_VPL_SWITCH(d, hipLaunchKernelGGL(pool_ln_kernel<VPL>, grid, dim3(256), 0, st, h, ids, w, b, xc, Nb, T, d, eps));
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

int launch_embed_bert(const int32_t *ids, const bf16_t *tok, const float *pos, const float *type0, const float *lw,
                      const float *lb, float *h, bf16_t *x, int Nb, int T, int d, int vocab, float eps, const int32_t *types,
                      int32_t *status, hipStream_t st)
{
    ProfScope prof(MMR_PROF_ROWWISE, st);
    const dim3 grid((unsigned)(((int64_t)Nb * T + 3) / 4));
    MMR_VPL_SWITCH(d, hipLaunchKernelGGL(embed_bert_kernel<VPL>, grid, dim3(256), 0, st, ids, tok, pos, type0, lw, lb, h, x, Nb, T, d, vocab, eps, types, status));
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

int launch_layernorm_inplace(float *h, const float *w, const float *b, bf16_t *x, int64_t rows, int d, float eps,
                             hipStream_t st)
{
    ProfScope prof(MMR_PROF_ROWWISE, st);
    const dim3 grid((unsigned)((rows + 3) / 4));
    MMR_VPL_SWITCH(d, hipLaunchKernelGGL(layernorm_inplace_kernel<VPL>, grid, dim3(256), 0, st, h, w, b, x, rows, d, eps));
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

int launch_gather_first_rows(const bf16_t *x, bf16_t *xc, int Nb, int T, int d, hipStream_t st)
{
    ProfScope prof(MMR_PROF_ROWWISE, st);
    const int total = Nb * (d / 8);
    hipLaunchKernelGGL(gather_first_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, st, x, xc, Nb, T, d);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

int launch_finish(const float *feat, void *out, mmr_dtype odt, int Nb, int E, int normalize, hipStream_t st)
{
    ProfScope prof(MMR_PROF_ROWWISE, st);
    const dim3 grid((unsigned)((Nb + 3) / 4));
    if (odt == MMR_BF16) hipLaunchKernelGGL(finish_kernel<bf16_t>, grid, dim3(256), 0, st, feat, (bf16_t *)out, Nb, E, normalize);
    else hipLaunchKernelGGL(finish_kernel<float>, grid, dim3(256), 0, st, feat, (float *)out, Nb, E, normalize);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

template <int NT, bool CAUSAL, bool MASKED>
static int launch_attention_t(const bf16_t *qkv, bf16_t *o, int Bn, int T, int heads, int d, const int32_t *kmask, hipStream_t st)
{
    ProfScope prof(MMR_PROF_ATTENTION, st);
    constexpr int TPAD = NT * 16;
    constexpr int lds = 2 * (2 * TPAD * 128 + (MASKED ? TPAD * 4 : 0));    // two pairs' K image + V image (+ key mask)
    static DeviceOnce once;
    static int cus = 256, per_cu = 1;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel<NT, CAUSAL, MASKED>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        int dev = 0, nb = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&attention_kernel<NT, CAUSAL, MASKED>),
                                                         256, lds) == hipSuccess && nb > 0)
            per_cu = nb;
    }
    // persistent grid: every workgroup resident at once and (nearly) the same number of pairs for each
    const int npairs = heads * Bn;
    const int cap = cus * per_cu;
    const int rounds = (npairs + cap - 1) / cap;
    const int grid = (npairs + rounds - 1) / rounds;
    hipLaunchKernelGGL((attention_kernel<NT, CAUSAL, MASKED>), dim3(grid), dim3(256), lds, st, qkv, o, T, d, 0.125f, kmask, heads,
                       npairs);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

// kmask: nullable key-padding mask [Bn,T] int32 (non-zero = attend); only the non-causal (BERT) form takes one
int launch_attention(const bf16_t *qkv, bf16_t *o, int Bn, int T, int heads, int d, int causal, const int32_t *kmask, hipStream_t st)
{
    if (kmask && causal) { set_error("attention: a key mask with the causal form is not built"); return MMR_ENOTSUP; }
    const int nt = (T + 31) / 32 * 2;
    if (nt > 6) {   // long sequences: streaming kernel
        ProfScope prof(MMR_PROF_ATTENTION, st);
        // 32 queries per wave (128 per workgroup) unless that pads the sequence by more than 64-query workgroups would
        static const int force_qb = getenv("MMR_ATTN_QB") ? atoi(getenv("MMR_ATTN_QB")) : 0;   // 1 / 2: A/B aid
        const int pad128 = (T + 127) / 128 * 128, pad64 = (T + 63) / 64 * 64;
        const int qb = force_qb ? force_qb : (pad128 == pad64 ? 2 : 1);
        const dim3 grid(qb == 2 ? pad128 / 128 : pad64 / 64, heads, Bn);
        if (causal) {
            if (qb == 2) hipLaunchKernelGGL((attention_stream_kernel<true, 2, false>), grid, dim3(256), 0, st, qkv, o, T, d, 0.125f, nullptr);
            else hipLaunchKernelGGL((attention_stream_kernel<true, 1, false>), grid, dim3(256), 0, st, qkv, o, T, d, 0.125f, nullptr);
        } else if (kmask) {
            if (qb == 2) hipLaunchKernelGGL((attention_stream_kernel<false, 2, true>), grid, dim3(256), 0, st, qkv, o, T, d, 0.125f, kmask);
            else hipLaunchKernelGGL((attention_stream_kernel<false, 1, true>), grid, dim3(256), 0, st, qkv, o, T, d, 0.125f, kmask);
        } else {
            if (qb == 2) hipLaunchKernelGGL((attention_stream_kernel<false, 2, false>), grid, dim3(256), 0, st, qkv, o, T, d, 0.125f, nullptr);
            else hipLaunchKernelGGL((attention_stream_kernel<false, 1, false>), grid, dim3(256), 0, st, qkv, o, T, d, 0.125f, nullptr);
        }
        MMR_CHECK_LAUNCH();
        return MMR_OK;
    }
    if (kmask) {
        switch (nt) {
            case 2: return launch_attention_t<2, false, true>(qkv, o, Bn, T, heads, d, kmask, st);
            case 4: return launch_attention_t<4, false, true>(qkv, o, Bn, T, heads, d, kmask, st);
            case 6: return launch_attention_t<6, false, true>(qkv, o, Bn, T, heads, d, kmask, st);
        }
    } else if (!causal) {
        switch (nt) {
            case 2: return launch_attention_t<2, false, false>(qkv, o, Bn, T, heads, d, nullptr, st);
            case 4: return launch_attention_t<4, false, false>(qkv, o, Bn, T, heads, d, nullptr, st);
            case 6: return launch_attention_t<6, false, false>(qkv, o, Bn, T, heads, d, nullptr, st);
        }
    } else {
        switch (nt) {
            case 2: return launch_attention_t<2, true, false>(qkv, o, Bn, T, heads, d, nullptr, st);
            case 4: return launch_attention_t<4, true, false>(qkv, o, Bn, T, heads, d, nullptr, st);
            case 6: return launch_attention_t<6, true, false>(qkv, o, Bn, T, heads, d, nullptr, st);
        }
    }
    set_error("attention: %d tokens (%s) has no kernel instance", T, causal ? "causal" : "full");
    return MMR_ENOTSUP;
}

}  // namespace mmr
