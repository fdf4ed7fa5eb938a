// bf16 MFMA GEMM for the encoder's dense layers on gfx950:  C[M,N] = A[M,K] . W[N,K]^T  (+ epilogue)
//
// Both operands are K-contiguous (activations [tokens, in], torch Linear weights [out, in]) -- the
// layout of every projection in the CLIP towers (q/k/v, out_proj, fc1, fc2, patch-embed as GEMM,
// visual/text projection).
//
// Two kernels (this file, top to bottom):
//   gemm_bf16_kernel     128x128x64 tile, 256 threads = 2x2 waves of 64x64, two workgroups per CU: small batches, the
//                        tiny test geometry, the final projection.
//   gemm256_bf16_kernel  256x256x64 (or 256x192x64) tile, 512 threads = 2x4 waves of 128x64 (128x48), one workgroup
//                        per CU, hand-pipelined: the kernel every large shape runs on (header further down).
// Common to both: v_mfma_f32_16x16x32_bf16; operands staged by global_load_lds (16 B/lane) into an XOR-swizzled LDS
// image (swizzle applied on the SOURCE address so the image stays lane-linear); fragments read with conflict-free
// ds_read_b128.  The MFMA is issued with the weight fragment as the "A" operand so each lane ends up with 4
// CONSECUTIVE output columns of one row; the epilogue transposes the wave's sub-tile through LDS and stores whole
// row segments, 16 B per lane.
// Fused epilogues (EPI_*): +bias -> bf16 | +bias, QuickGELU / exact GELU / tanh -> bf16 | +bias, += fp32 residual |
// plain fp32 | +bias -> fp32 | LayerNorm-folded forms.
#include "mmr_common.h"

#include <stdlib.h>

#include <type_traits>

namespace mmr {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;
constexpr int TILE_A_BYTES = BM * BK * 2;  // 16 KiB
constexpr int TILE_B_BYTES = BN * BK * 2;
constexpr int STAGE_BYTES = TILE_A_BYTES + TILE_B_BYTES;

enum { EPI_BIAS_BF16 = 0, EPI_BIAS_GELU_BF16 = 1, EPI_BIAS_RESID_F32 = 2, EPI_STORE_F32 = 3,
       EPI_BIAS_F32 = 4, EPI_BIAS_GELU_ERF_BF16 = 5, EPI_BIAS_TANH_BF16 = 6,
       // LayerNorm folded into the GEMM that consumes it (A = bf16 of the UN-normalised residual, W' = W*gamma):
       //   LN(h) . W^T + b  =  rstd * (h . W'^T - mean * colsum(W')) + (beta . W^T + b)
       EPI_LNFOLD_BF16 = 7, EPI_LNFOLD_GELU_BF16 = 8,
       // residual update that also emits what the NEXT folded GEMM needs: bf16(h_new) and per-row (sum, sumsq)
       EPI_RESID_STATS_F32 = 9, EPI_COUNT = 10 };

constexpr bool epi_lnfold(int e) { return e == EPI_LNFOLD_BF16 || e == EPI_LNFOLD_GELU_BF16; }
constexpr bool epi_bf16(int e) { return e == EPI_BIAS_BF16 || e == EPI_BIAS_GELU_BF16 || e == EPI_BIAS_GELU_ERF_BF16 || e == EPI_BIAS_TANH_BF16 || epi_lnfold(e); }
constexpr bool epi_bias(int e) { return e != EPI_STORE_F32; }
constexpr bool epi_resid(int e) { return e == EPI_BIAS_RESID_F32 || e == EPI_RESID_STATS_F32; }

// activation fused into the bf16 epilogues
template <int EPI>
__device__ __forceinline__ float epi_act(float v) {
    if constexpr (EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_LNFOLD_GELU_BF16) {
        // QuickGELU x * sigmoid(1.702 x) (transformers/activations.py:117-123); hardware exp/rcp (1 ulp)
        // is far inside bf16 rounding.  Same formulation in both tile sizes.
        // exp(-1.702 v) as ONE multiply into exp2: -1.702 * log2(e) folded by hand (the compiler keeps two multiplies)
        return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -2.45546696f));
    } else if constexpr (EPI == EPI_BIAS_GELU_ERF_BF16) {
        // exact-GELU 0.5 x (1 + erf(x / sqrt 2)) (BERT "gelu").  erf by Abramowitz-Stegun 7.1.26 (|error| <=
        // 1.5e-7, far inside bf16 rounding) on the hardware exp/rcp: libm's erff cost 42 us on the fc1 shape.
        const float z = fabsf(v) * 0.70710678118654752f;
        const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
        const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        const float erf_abs = 1.f - poly * __expf(-z * z);
        return 0.5f * v * (1.f + copysignf(erf_abs, v));
    } else if constexpr (EPI == EPI_BIAS_TANH_BF16) {
        return tanhf(v);                                             // BERT pooler
    } else {
        return v;
    }
}

// diagnostic build (-DMMR_GEMM_FOLDROWS): every tile stores into the first 256 output rows, so the stores stay in
// L2 -- separates the epilogue's instruction cost from the HBM write burst (outputs are wrong)
#ifdef MMR_GEMM_FOLDROWS
#define MMR_OUT_ROW(r) ((r) & 255)
#else
#define MMR_OUT_ROW(r) (r)
#endif
// 16-byte epilogue store.  MMR_GEMM_NT_STORE=1 (A/B knob) marks it non-temporal.
#ifndef MMR_GEMM_NT_STORE
#define MMR_GEMM_NT_STORE 0
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <typename V>
__device__ __forceinline__ void epi_store16(void *dst, const V &v) {
    static_assert(sizeof(V) == 16, "16-byte vector");
#if MMR_GEMM_NT_STORE
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), reinterpret_cast<u32x4_t *>(dst));
#else
    *reinterpret_cast<V *>(dst) = v;
#endif
}

// diagnostic build (-DMMR_GEMM_STAMPS): wave 0 of every workgroup records s_memrealtime (100 MHz) at the phase
// boundaries of each tile it runs into a buffer no kernel reads; mmr_debug_gemm_stamps copies it out
// (tools/gemm_phase_times.py).  The shipped build has no stamp instruction.
#ifdef MMR_GEMM_STAMPS
__device__ unsigned long long mmr_stamp_buf[8192 * 4];
__device__ unsigned long long mmr_cycle_buf[8192 * 4];      // s_memtime (shader clock) at the same points: clock = d(cycles)/d(realtime)
#define MMR_STAMP(slot, which)                                                                                  \
    do {                                                                                                        \
        if (threadIdx.x == 0 && (slot) < 8192) {                                                                \
            mmr_stamp_buf[(slot) * 4 + (which)] = __builtin_amdgcn_s_memrealtime();                             \
            mmr_cycle_buf[(slot) * 4 + (which)] = __builtin_amdgcn_s_memtime();                                 \
        }                                                                                                       \
    } while (0)
#else
#define MMR_STAMP(slot, which) do { } while (0)
#endif

// byte offset of 16-B chunk `c` (0..7) of tile row `row` inside a [rows][64] bf16 tile image
__device__ __forceinline__ int tile_off(int row, int c) {
    return (row >> 3) * 1024 + (row & 7) * 128 + ((c ^ (row & 7)) << 4);
}

// NSTG = LDS stages.  2: 64 KiB, two workgroups per CU (many tiles).  4: 128 KiB, one workgroup per CU, three K-tiles
// in flight -- for grids that do not fill the chip anyway (small batches), where each K-iteration is otherwise one
// exposed load latency: fc2 at M=512 ran 48 iterations x 0.8 us.
template <int NSTG>
struct Gemm128Cfg { static constexpr int LDS = NSTG * STAGE_BYTES + BM * 8; };

template <int EPI, int NSTG>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_bf16_kernel(
    const bf16_t *__restrict__ A, const bf16_t *__restrict__ W, int M, int N, int K,
    const float *__restrict__ bias, void *__restrict__ out, GemmAux aux)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware block order: blocks that share an XCD (bid % 8) walk a contiguous range of tiles, so an
    // A row-panel is re-read from that XCD's L2 by the N/BN blocks that need it.
    const int gn = N / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int qd = nblk >> 3, rm = nblk & 7, xcd = bid & 7;
        bid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    }
    const int m0 = (bid / gn) * BM;
    const int n0 = (bid % gn) * BN;

    // ---- staging: each wave issues 4 + 4 global_load_lds per K-tile (1 KiB = 8 rows x 128 B each)
    const int rr = lane >> 3;                 // row inside the 8-row block
    const int sc = (lane & 7) ^ rr;           // source chunk that lands in LDS slot (lane & 7)
    const bf16_t *a_src = A + (size_t)(m0 + wave * 32 + rr) * K + sc * 8;
    const bf16_t *w_src = W + (size_t)(n0 + wave * 32 + rr) * K + sc * 8;
    auto stage = [&](int kt, int buf) {
        char *base = smem + buf * STAGE_BYTES;
        const size_t ko = (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int blk = wave * 4 + i;     // 8-row block index, 16 per tile
            glds16(a_src + (size_t)i * 8 * K + ko, base + blk * 1024);
            glds16(w_src + (size_t)i * 8 * K + ko, base + TILE_A_BYTES + blk * 1024);
        }
    };

    f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int nkt = K / BK;
    float2 *row_stats = reinterpret_cast<float2 *>(smem + NSTG * STAGE_BYTES);   // LNFOLD only: (mean, rstd) of the BM tile rows
    auto prologue = [&]() {
#pragma unroll
        for (int s = 0; s < NSTG - 1; ++s)
            if (s < nkt) stage(s, s);
    };
    if constexpr (epi_lnfold(EPI)) {
        LnfoldLoads ld;
        lnfold_issue(aux, m0, ld);
        prologue();
        lnfold_finish(aux, ld, row_stats);
    } else {
        prologue();
    }
    for (int kt = 0; kt < nkt; ++kt) {
        // K-tile kt has landed once at most `younger` later stages (8 loads per wave each) are still in flight ...
        const int younger = min(NSTG - 2, nkt - 1 - kt);
        if (younger >= 2) wait_vmcnt<16>();
        else if (younger == 1) wait_vmcnt<8>();
        else wait_vmcnt<0>();
        // ... for every wave after the barrier, which also says everyone is done reading K-tile kt-1's buffer:
        __builtin_amdgcn_s_barrier();
        if (kt + NSTG - 1 < nkt) stage(kt + NSTG - 1, (kt + NSTG - 1) % NSTG);   // into that buffer
        const int cur = kt % NSTG;
        const char *ta = smem + cur * STAGE_BYTES;
        const char *tw = ta + TILE_A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8 *>(ta + tile_off(wm * 64 + i * 16 + fr, ks * 4 + fg));
                wf[i] = *reinterpret_cast<const bf16x8 *>(tw + tile_off(wn * 64 + i * 16 + fr, ks * 4 + fg));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    __syncthreads();      // the epilogue reuses the staging buffers

#ifdef MMR_GEMM_NOEPI   // diagnostic build: time prologue + main loop only (outputs are wrong)
    {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 123.456f) ((float *)out)[0] = keep;
        return;
    }
#endif
    // ---- epilogue: in the accumulator layout a lane owns 4 consecutive columns of 16 different rows.  Each
    // wave transposes its 64x64 sub-tile through a private LDS region (the staging buffers are idle after
    // the loop's last __syncthreads) and stores whole 128 B (bf16) / 256 B (fp32) row segments.
    const size_t row_base = (size_t)(m0 + wm * 64);
    const int col_base = n0 + wn * 64;
    if constexpr (epi_bf16(EPI)) {
        char *my = smem + wave * 8192;           // [64 rows][8 chunks of 8 bf16], chunk ^ (row & 7)
        float mean[4], rstd[4];
        if constexpr (epi_lnfold(EPI)) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const float2 st = row_stats[wm * 64 + mi * 16 + fr];
                mean[mi] = st.x; rstd[mi] = st.y;
            }
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bias + col_base + ni * 16 + fg * 4);
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (epi_lnfold(EPI)) c4 = *reinterpret_cast<const float4 *>(aux.colsum + col_base + ni * 16 + fg * 4);
            const int c = ni * 2 + (fg >> 1);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                float a0 = acc[ni][mi][0], a1 = acc[ni][mi][1], a2 = acc[ni][mi][2], a3 = acc[ni][mi][3];
                if constexpr (epi_lnfold(EPI)) {
                    a0 = rstd[mi] * (a0 - mean[mi] * c4.x); a1 = rstd[mi] * (a1 - mean[mi] * c4.y);
                    a2 = rstd[mi] * (a2 - mean[mi] * c4.z); a3 = rstd[mi] * (a3 - mean[mi] * c4.w);
                }
                const float v0 = epi_act<EPI>(a0 + b4.x), v1 = epi_act<EPI>(a1 + b4.y);
                const float v2 = epi_act<EPI>(a2 + b4.z), v3 = epi_act<EPI>(a3 + b4.w);
                const int row = mi * 16 + fr;
                uint2 pk;
                pk.x = pack_bf16x2(v0, v1);
                pk.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2 *>(my + row * 128 + ((c ^ (row & 7)) << 4) + (fg & 1) * 8) = pk;
            }
        }
        const int rc = lane & 7, rr0 = lane >> 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + rr0;
            const uint4 v = *reinterpret_cast<const uint4 *>(my + row * 128 + ((rc ^ (row & 7)) << 4));
            epi_store16((bf16_t *)out + (row_base + row) * N + col_base + rc * 8, v);
        }
    } else {
        char *my = smem + wave * 16384;          // [64 rows][16 chunks of 4 floats], chunk ^ (row & 15)
        const int rc = lane & 15, rr0 = lane >> 4;
        auto row_ptr = [&](int i) { return reinterpret_cast<float4 *>((float *)out + (row_base + i * 4 + rr0) * N + col_base + rc * 4); };
        // residual rows first: 16 independent loads in flight while the tile goes through LDS (see the 256-row kernel)
        float4 hv[16];
        if constexpr (epi_resid(EPI)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) hv[i] = *row_ptr(i);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (epi_bias(EPI)) b4 = *reinterpret_cast<const float4 *>(bias + col_base + ni * 16 + fg * 4);
            const int c = ni * 4 + fg;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const f32x4 a = acc[ni][mi];
                const int row = mi * 16 + fr;
                *reinterpret_cast<float4 *>(my + row * 256 + ((c ^ (row & 15)) << 4)) =
                    make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i * 4 + rr0;
            float4 v = *reinterpret_cast<const float4 *>(my + row * 256 + ((rc ^ (row & 15)) << 4));
            const size_t grow = row_base + row;
            if constexpr (epi_resid(EPI)) { v.x += hv[i].x; v.y += hv[i].y; v.z += hv[i].z; v.w += hv[i].w; }
            epi_store16(row_ptr(i), v);
            if constexpr (EPI == EPI_RESID_STATS_F32) {
                uint2 pk;
                pk.x = pack_bf16x2(v.x, v.y);
                pk.y = pack_bf16x2(v.z, v.w);
                *reinterpret_cast<uint2 *>(aux.xout + grow * N + col_base + rc * 4) = pk;
                const float s1 = row16_sum((v.x + v.y) + (v.z + v.w));
                const float s2 = row16_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
                if (rc == 0) emit_row_partial(aux.stats_out, grow, col_base >> 6, N >> 6, s1, s2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 256 x (64*NIW) x 64 tile, 8 waves (2 along M x 4 along N), each wave a 128 x (16*NIW) sub-tile.
//   NIW = 4: 256x256 tile, 32 accumulators per wave.   NIW = 3: 256x192 tile, 24 accumulators.
// Operand fetch per CU, not MFMA issue, bounds the 128^2 kernel on these shapes (both structures
// settle near the same ~17 B/clk/CU of L2->LDS traffic), so the lever is FLOPs per fetched byte:
// this tile does 2x the work per byte.  The 192-wide variant exists for tile-count quantisation:
// N = 768 at M = 12800 is 150 tiles of 256x256 -- 59 % of the 256 CUs for one round -- but 200 tiles of
// 256x192 (78 %), each 3/4 of the work (launch_gemm_aux picks by rounds x tile width).
// Pipeline: every K-tile is 2 phases of 8*NIW MFMAs (one 64-row half of the wave's sub-tile over K=64);
// each phase reads its fragments and stages two pieces of a FUTURE K-tile by global_load_lds (LOAD
// segment), then runs its MFMA cluster (COMPUTE segment); segments are separated by raw s_barriers and
// the two wave groups (waves 0-3 / 4-7 = the two waves of each SIMD) run staggered by one segment, so
// LDS reads of one hide under the MFMAs of the other.  Pieces are staged in the rotating order
// W0,W1,A0,A1 (A0/A1 = tile rows 0-127 / 128-255; W0 = W rows 0-127, W1 = the remaining 128 or 64 rows)
// so each lands 1.5-3 K-tiles before its first read; the only wait on the load queue is one counted
// vmcnt per K-tile (never 0 inside the loop), two barriers before the first read of the data it retires.
// Tried and dropped (measured slower on MI355X): 4 phases of 16 MFMAs per K-tile (more barrier overhead),
// issuing one of the two stages between the MFMA clusters (-8 %), a 3-deep LDS ring.
// ---------------------------------------------------------------------------------------------
constexpr int BM2 = 256;
constexpr int GEMM2_THREADS = 512;
constexpr int HALF_BYTES = 128 * BK * 2;            // 16 KiB: 128 rows x 64 bf16
constexpr int STAGE2_BYTES = 4 * HALF_BYTES;        // A0 A1 W (up to 256 rows, contiguous image)
constexpr int GEMM2_LDS = 2 * STAGE2_BYTES;         // 128 KiB

// XCD-aware tile order of one region of the tile space (`nblk` workgroups = `panels` row panels x `gn` column tiles of
// `bnt` columns): blocks that share an XCD (same id mod 8) walk a contiguous range; inside it, groups of `pg` row
// panels, column-major inside a group, so the ~32 tiles an XCD runs at once share few A panels AND few W tiles (their
// union has to fit its 4 MiB L2; row-major order with 12 column tiles touches all of W every round and re-fetches it:
// 139 MB vs 24 MB of operands)
__device__ __forceinline__ void decode_tile256(int bid, int nblk, int gn, int panels, int pg, int bnt, int &m0, int &n0)
{
    const int qd = nblk >> 3, rm = nblk & 7, xcd = bid & 7;
    bid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int per_group = pg * gn;
    const int grp = bid / per_group, rem = bid - grp * per_group;
    const int rows_here = min(pg, panels - grp * pg);
    m0 = (grp * pg + rem % rows_here) * BM2;
    n0 = (rem / rows_here) * bnt;
}

// one output tile [m0, m0+256) x [n0, n0 + 64*NIW)
template <int EPI, int NIW, bool PATCH>
__device__ __forceinline__ void gemm256_tile(
    const bf16_t *__restrict__ A, const bf16_t *__restrict__ W, int M, int N, int K,
    const float *__restrict__ bias, void *__restrict__ out, const GemmAux &aux, const int m0, const int n0, char *smem)
{
    static_assert(NIW == 3 || NIW == 4, "tile width 192 or 256");
    constexpr int WCOLS = 16 * NIW;               // columns per wave
    constexpr int W1_LOADS = NIW == 4 ? 2 : 1;    // global_load_lds per wave for piece W1 (128 or 64 rows)
    constexpr int W_LOADS = 2 + W1_LOADS;         // ... for W0 + W1: the queue depth the counted wait leaves
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- staging: a 128-row piece is 16 blocks of 8 rows (1 KiB each); wave w issues blocks 2w, 2w+1.
    // The 64-row W1 of the 192-wide tile is 8 blocks, one per wave.
    const int rr = lane >> 3;
    const int sc = (lane & 7) ^ rr;
    // Staging addresses = wave-uniform base (tile, piece, K-tile: SGPRs) + ONE per-lane 32-bit byte offset that never
    // changes (row rr of the 8-row block, swizzled source chunk sc): the loads take the saddr + voffset form, so no 64-bit
    // per-lane pointers live across the main loop (fewer VGPRs in a kernel that runs against the 256-register budget)
    const uint32_t lane_off = (uint32_t)((rr * K + sc * 8) * 2);
    // PATCH: the four A rows this lane stages per K-tile (half 0/1 x rows +0/+8) as pixel addresses.  K-tile kt covers
    // channel kt/16, patch rows ky = 2*(kt%16) and +1; source chunk sc is kx = 8*(sc&3) .. +7 of row ky + (sc>>2)
    // -- the same 16 contiguous bytes im2col would have copied.
    const bf16_t *pa[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    if constexpr (PATCH) {
        const int S = aux.pix_size, G = aux.pix_grid;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int r = m0 + hh * 128 + wave * 16 + rr + 8 * j;
                r = r < aux.patch_rows ? r : aux.patch_rows - 1;
                const int gx = r % G, gy = (r / G) % G, b = r / (G * G);
                pa[hh][j] = aux.pix + ((size_t)(b * 3) * S + gy * 32) * S + gx * 32 + (sc >> 2) * S + (sc & 3) * 8;
            }
    }
    const int nkt = K / BK;
    // kind: 0 = W0, 1 = W1, 2 = A0, 3 = A1  (the rotating stream order)
    auto stage = [&](int kt, int kind) {
        if (kt >= nkt) return;
        const int half = kind & 1;
        const bool isA = kind >= 2;
        char *dst = smem + (kt & 1) * STAGE2_BYTES + ((isA ? 0 : 2) + half) * HALF_BYTES;
        if (NIW == 3 && kind == 1) {               // the 64-row W1 of the 192-wide tile: 8 blocks, one per wave
            const char *b1 = reinterpret_cast<const char *>(W) + ((size_t)(n0 + 128 + wave * 8) * K + (size_t)kt * BK) * 2;
            glds16(b1 + lane_off, dst + wave * 1024);
            return;
        }
        if (PATCH && isA) {
            const size_t off = (size_t)(kt >> 4) * aux.pix_size * aux.pix_size + (size_t)((kt & 15) * 2) * aux.pix_size;
            glds16(pa[half][0] + off, dst + wave * 2048);
            glds16(pa[half][1] + off, dst + wave * 2048 + 1024);
            return;
        }
        const char *base = reinterpret_cast<const char *>(isA ? A : W) +
                           ((size_t)((isA ? m0 : n0) + half * 128 + wave * 16) * K + (size_t)kt * BK) * 2;
        glds16(base + lane_off, dst + wave * 2048);
        glds16(base + (size_t)16 * K + lane_off, dst + wave * 2048 + 1024);        // 8 rows further
    };

    f32x4 acc[NIW][8];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < NIW; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int w_row0 = wc * WCOLS;                // this wave's rows inside the contiguous W tile image
    // the epilogue's bias values, fetched now so their latency hides under the whole main loop
    float4 bias4[NIW];
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) {
        bias4[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (epi_bias(EPI)) bias4[ni] = *reinterpret_cast<const float4 *>(bias + n0 + wc * WCOLS + ni * 16 + fg * 4);
    }

    MMR_STAMP(blockIdx.x, 0);
    // prologue: stream elements 0..5 = W0,W1,A0,A1 of tile 0 and W0,W1 of tile 1
    float2 *row_stats = reinterpret_cast<float2 *>(smem + GEMM2_LDS);   // LNFOLD only: (mean, rstd) of the BM2 tile rows
    LnfoldLoads ld;
    if constexpr (epi_lnfold(EPI)) lnfold_issue(aux, m0, ld);
    stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3); stage(1, 0); stage(1, 1);
    if (nkt > 1) wait_vmcnt<W_LOADS>(); else wait_vmcnt<0>();
    if constexpr (epi_lnfold(EPI)) lnfold_finish(aux, ld, row_stats);
    __builtin_amdgcn_s_barrier();
    MMR_STAMP(blockIdx.x, 1);
    // Stagger: waves 4-7 (wr == 1, the SIMD partners of waves 0-3) run one segment behind, so on every
    // SIMD one wave is in a LOAD segment (ds_read + global_load_lds) while its partner is in a COMPUTE
    // segment.  Every wave executes the same number of barriers (compensated after the loop).
    if (wr == 1) __builtin_amdgcn_s_barrier();

    bf16x8 af[4][2], wf[NIW][2];   // af[mi][ks];  wf[ni][ks]
#define MMR_LOAD_DONE()  do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
    for (int kt = 0; kt < nkt; ++kt) {
        const char *sb = smem + (kt & 1) * STAGE2_BYTES;
        const char *ta = sb + wr * HALF_BYTES;        // the A half-tile this wave reads
        const char *tw = sb + 2 * HALF_BYTES;

        auto read_w = [&]() {
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wf[ni][ks] = *reinterpret_cast<const bf16x8 *>(tw + tile_off(w_row0 + ni * 16 + fr, ks * 4 + fg));
        };
        auto read_a = [&](int mh) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    af[mi][ks] = *reinterpret_cast<const bf16x8 *>(ta + tile_off(mh * 64 + mi * 16 + fr, ks * 4 + fg));
        };
        auto mma = [&](int mh) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mh * 4 + mi] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][ks], af[mi][ks], acc[ni][mh * 4 + mi], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };

        // phase A: rows 0-63 of the wave's sub-tile: reads W (all of the wave's columns) + A rows 0-63;
        // stages A0, A1 of tile kt+1
        read_w(); read_a(0);
        stage(kt + 1, 2);
        stage(kt + 1, 3);
        MMR_LOAD_DONE();
        mma(0);
        __builtin_amdgcn_s_barrier();
        // phase B: rows 64-127: reads A rows 64-127; stages W0, W1 of tile kt+2 (W of tile kt was last
        // read in phase A); retires tile kt+1's pieces (youngest staged in phase A of this tile)
        read_a(1);
        stage(kt + 2, 0);
        stage(kt + 2, 1);
        if (kt + 2 < nkt) wait_vmcnt<W_LOADS>(); else wait_vmcnt<0>();
        MMR_LOAD_DONE();
        mma(1);
        __builtin_amdgcn_s_barrier();
    }
#undef MMR_LOAD_DONE
    if (wr == 0) __builtin_amdgcn_s_barrier();
    MMR_STAMP(blockIdx.x, 2);

#ifdef MMR_GEMM_NOEPI   // diagnostic build: time prologue + main loop only (outputs are wrong)
    {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < NIW; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 123.456f) ((float *)out)[0] = keep;
        return;
    }
#endif
    // ---- epilogue.  In the accumulator layout a lane owns 4 consecutive columns of 16 different rows,
    // so direct stores touch 16 rows x 32 B per instruction (measured: 14-33 us per GEMM, a third of
    // the kernel).  The staging LDS is idle now (the loop's last barriers retired every read), so each
    // wave transposes its own sub-tile through a PRIVATE 16 KiB region (no barrier needed) and stores
    // whole row segments (128 or 96 B bf16 / 256 or 192 B fp32), 16 B per lane.  The LDS image keeps the
    // 8- / 16-chunk row pitch for both widths, so the XOR swizzle stays inside a row.
    char *my = smem + wave * 16384;
    const size_t row_base = (size_t)(m0 + wr * 128);
    const int col_base = n0 + wc * WCOLS;
    if constexpr (epi_bf16(EPI)) {
        // image: [128 rows][8 chunks of 8 bf16], chunk index XOR (row & 7)
        float mean[8], rstd[8];
        if constexpr (epi_lnfold(EPI)) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const float2 st = row_stats[wr * 128 + mi * 16 + fr];
                mean[mi] = st.x; rstd[mi] = st.y;
            }
        }
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
            const float4 b4 = bias4[ni];
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (epi_lnfold(EPI)) c4 = *reinterpret_cast<const float4 *>(aux.colsum + col_base + ni * 16 + fg * 4);
            const int c = ni * 2 + (fg >> 1);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                float a0 = acc[ni][mi][0], a1 = acc[ni][mi][1], a2 = acc[ni][mi][2], a3 = acc[ni][mi][3];
                if constexpr (epi_lnfold(EPI)) {
                    a0 = rstd[mi] * (a0 - mean[mi] * c4.x); a1 = rstd[mi] * (a1 - mean[mi] * c4.y);
                    a2 = rstd[mi] * (a2 - mean[mi] * c4.z); a3 = rstd[mi] * (a3 - mean[mi] * c4.w);
                }
                const float v0 = epi_act<EPI>(a0 + b4.x), v1 = epi_act<EPI>(a1 + b4.y);
                const float v2 = epi_act<EPI>(a2 + b4.z), v3 = epi_act<EPI>(a3 + b4.w);
                const int row = mi * 16 + fr;
                uint2 pk;
                pk.x = pack_bf16x2(v0, v1);
                pk.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2 *>(my + row * 128 + ((c ^ (row & 7)) << 4) + (fg & 1) * 8) = pk;
            }
        }
        const int rc = lane & 7, rr0 = lane >> 3;
        if (rc < 2 * NIW) {
            uint4 rows16[16];                 // all row reads first (the accumulators are dead): one LDS latency, not 8
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i * 8 + rr0;
                rows16[i] = *reinterpret_cast<const uint4 *>(my + row * 128 + ((rc ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
                epi_store16((bf16_t *)out + MMR_OUT_ROW(row_base + i * 8 + rr0) * N + col_base + rc * 8, rows16[i]);
        }
    } else {
        // fp32: two passes of 64 rows; image [64 rows][16 chunks of 4 floats], chunk index XOR (row & 15).
        // Residual epilogues read h before adding into it.  All 16 row segments of a pass are loaded up front
        // (pass 1's before pass 0's stores: CDNA4 counts stores in vmcnt, in order), so the wave pays one
        // load latency per pass instead of a load -> wait -> add -> store round trip per row.
        const int rc = lane & 15, rr0 = lane >> 4;
        const bool live = rc < 4 * NIW;
        auto row_ptr = [&](int mh, int i) {
            return reinterpret_cast<float4 *>((float *)out + (row_base + mh * 64 + i * 4 + rr0) * N + col_base + rc * 4);
        };
        auto load_resid = [&](int mh, float4 (&hv)[16]) {
            if constexpr (epi_resid(EPI)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) hv[i] = live ? *row_ptr(mh, i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto to_lds = [&](int mh) {
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) {
                const int c = ni * 4 + fg;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const f32x4 a = acc[ni][mh * 4 + mi];
                    const int row = mi * 16 + fr;
                    *reinterpret_cast<float4 *>(my + row * 256 + ((c ^ (row & 15)) << 4)) =
                        make_float4(a[0] + bias4[ni].x, a[1] + bias4[ni].y, a[2] + bias4[ni].z, a[3] + bias4[ni].w);
                }
            }
        };
        auto store_rows = [&](int mh, const float4 (&hv)[16]) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i * 4 + rr0;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const size_t grow = row_base + mh * 64 + row;
                if (live) {
                    v = *reinterpret_cast<const float4 *>(my + row * 256 + ((rc ^ (row & 15)) << 4));
                    if constexpr (epi_resid(EPI)) { v.x += hv[i].x; v.y += hv[i].y; v.z += hv[i].z; v.w += hv[i].w; }
                    epi_store16(row_ptr(mh, i), v);
                }
                if constexpr (EPI == EPI_RESID_STATS_F32) {
                    if (live) {
                        uint2 pk;
                        pk.x = pack_bf16x2(v.x, v.y);
                        pk.y = pack_bf16x2(v.z, v.w);
                        *reinterpret_cast<uint2 *>(aux.xout + grow * N + col_base + rc * 4) = pk;
                    }
                    const float s1 = row16_sum((v.x + v.y) + (v.z + v.w));
                    const float s2 = row16_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
                    if (rc == 0) emit_row_partial(aux.stats_out, grow, col_base / WCOLS, N / WCOLS, s1, s2);
                }
            }
        };
        float4 hv0[16], hv1[16];
        load_resid(0, hv0);
        to_lds(0);
        if constexpr (NIW == 3) load_resid(1, hv1);   // 24 accumulators: room for both passes' rows at once
        store_rows(0, hv0);
        if constexpr (NIW == 4) load_resid(1, hv1);   // 32 accumulators: the second batch has to wait for registers
        to_lds(1);
        store_rows(1, hv1);
    }
#ifdef MMR_GEMM_STAMPS
    MMR_STAMP(blockIdx.x, 3);          // epilogue instructions issued (stores may still be in flight)
#endif
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the 256x256 kernel for launches of MORE tiles than CUs with a bf16 epilogue (qkv, fc1: 450 / 600
// tiles at batch 256).  In-kernel stamps (tools/gemm_phase_times.py) put a tile "round" of the one-workgroup-per-tile kernel
// at ~17 us of main loop plus ~5 us in which the matrix pipes idle: 1.3 us waiting for the first operand pieces, 1.6-2.9 us
// of epilogue, ~2 us between a CU's consecutive workgroups (store drain + dispatch).  Here one workgroup per CU walks a
// static tile list and
//   * issues the NEXT tile's prologue loads -- all eight 16 KiB pieces of K-tiles 0 and 1: the staging buffers are idle once
//     the main loop's last barrier has passed -- BEFORE it runs the current tile's epilogue, which works in a separate
//     4 KiB-per-wave LDS scratch (four passes of 32 rows), so the loads fly during the epilogue;
//   * never waits for its epilogue stores: the VM counter is in order, so the next tile's first counted waits name the 16
//     stores as the YOUNGEST queue entries (s_waitcnt vmcnt(stores + bias loads + newer pieces)) and the first loads issued
//     behind them (K-tile 2's W pieces) are not needed until 1.5 K-tiles later;
//   * evens out the tile count: the last row panels are cut into 256x128 half tiles handed to the workgroups with the
//     fewest full tiles (PersistPlan), so fc1's 600 tiles cost the slowest workgroup 2 full + 1 half instead of 3 full.
// Main loop, staging order, swizzles and accumulation order are those of gemm256_tile<EPI, 4>: results are bit-identical.
// Measured (MI355X, ViT-B/32 batch 256, same box, us): qkv 49.0 -> 46.2, fc1 75.2 (persistent, no halves) / 73.0 (round 2's
// first attempt: half tiles dispatched first, not persistent) -> 72.5, ViT-L/14 fc1 520 -> 510; forward 2.97 -> 2.93 ms.
// ---------------------------------------------------------------------------------------------
constexpr int EPI_SCRATCH = 4096;                       // per wave: [32 rows][8 chunks of 8 bf16]
constexpr int GEMM2P_LDS = GEMM2_LDS + 8 * EPI_SCRATCH; // 160 KiB: the whole LDS of a CU
constexpr int GEMM2P_STORES = 16;                       // epilogue global stores per wave per tile (4 passes x 4)

// Work of one persistent workgroup: full (256x256) tiles b, b + G, b + 2G, ... of the FULL region (row panels ph ..), then
// half (256x128) tiles G-1-b, 2G-1-b, ... of the HALF region (row panels 0 .. ph-1) -- the workgroups with the fewest full
// tiles take half tiles first.  The host picks ph so that the slowest workgroup runs 2.65 tile times instead of 3
// (persist_half_panels): the launch's tile count need not be a multiple of the CU count.
struct PersistPlan {
    int ph;                  // row panels cut into half tiles (0 = none)
    int nfull, nhalf;        // tiles per region
    int pg_full, pg_half;    // tile-order group sizes (decode_tile256)
};

template <int EPI>
__global__ __launch_bounds__(GEMM2_THREADS, 2) void gemm256_persist_kernel(
    const bf16_t *__restrict__ A, const bf16_t *__restrict__ W, int M, int N, int K,
    const float *__restrict__ bias, bf16_t *__restrict__ out, PersistPlan plan)
{
    static_assert(epi_bf16(EPI) && !epi_lnfold(EPI), "persistent form: plain bf16 epilogues");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int rr = lane >> 3;
    const int sc = (lane & 7) ^ rr;
    const int fr = lane & 15, fg = lane >> 4;
    const int gn = N / 256, panels = M / BM2;
    const int nkt = K / BK;
    char *scratch = smem + GEMM2_LDS + wave * EPI_SCRATCH;

    // this workgroup's tile sequence
    const int G = gridDim.x, b = blockIdx.x;
    const int my_full = plan.nfull > b ? (plan.nfull - 1 - b) / G + 1 : 0;
    const int h0 = G - 1 - b;
    const int my_half = plan.nhalf > h0 ? (plan.nhalf - 1 - h0) / G + 1 : 0;
    const int my_tiles = my_full + my_half;
    if (my_tiles == 0) return;

    // staging target: the tile whose operand pieces stage() fetches (it changes to the NEXT tile before the current tile's
    // epilogue).  Addresses = wave-uniform base (tile, K-tile, piece: SGPRs) + ONE per-lane 32-bit byte offset that never
    // changes (row rr of the 8-row block, swizzled source chunk sc): saddr + voffset loads, no 64-bit per-lane pointers
    int m0 = 0, n0 = 0, niw = 4;
    auto set_tile = [&](int i) {           // i-th tile of this workgroup
        if (i < my_full) {
            decode_tile256(b + i * G, plan.nfull, gn, panels - plan.ph, plan.pg_full, 256, m0, n0);
            m0 += plan.ph * BM2;
            niw = 4;
        } else {
            decode_tile256(h0 + (i - my_full) * G, plan.nhalf, 2 * gn, plan.ph, plan.pg_half, 128, m0, n0);
            niw = 2;
        }
    };
    const uint32_t lane_off = (uint32_t)((rr * K + sc * 8) * 2);
    // kind: 0 = W0, 1 = W1, 2 = A0, 3 = A1 (gemm256_tile's rotating stream order); a half tile has no W1 piece
    auto stage = [&](int kt, int kind) {
        if (kt >= nkt || (kind == 1 && niw == 2)) return;
        const int half = kind & 1;
        const bool isA = kind >= 2;
        char *dst = smem + (kt & 1) * STAGE2_BYTES + ((isA ? 0 : 2) + half) * HALF_BYTES;
        const char *base = reinterpret_cast<const char *>(isA ? A : W) +
                           ((size_t)((isA ? m0 : n0) + half * 128 + wave * 16) * K + (size_t)kt * BK) * 2;
        glds16(base + lane_off, dst + wave * 2048);
        glds16(base + (size_t)16 * K + lane_off, dst + wave * 2048 + 1024);       // 8 rows further
    };
    // a tile's prologue: ALL pieces of K-tiles 0 and 1 (the whole 128 KiB of staging is idle between tiles), so the
    // first loads issued after an epilogue are the W pieces of K-tile 2 -- needed 1.5 K-tiles later, by which time the
    // epilogue's stores, which the in-order VM counter makes them wait behind, have long drained
    auto tile_prologue = [&]() {
        stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3); stage(1, 0); stage(1, 1); stage(1, 2); stage(1, 3);
    };
#ifdef MMR_GEMM_STAMPS
    int stamp_slot = blockIdx.x;
#endif

    // one tile of width 64*NIW.  On entry its prologue pieces are in flight; on exit the next tile's are (if any).
    auto run_tile = [&](auto NIWC, const int i) {
        constexpr int NIW = decltype(NIWC)::value;
        constexpr int WCOLS = 16 * NIW, W_LOADS = NIW == 4 ? 4 : 2;     // glds per wave for the W pieces of one K-tile
        constexpr int BIAS_LOADS = NIW, KT1_LOADS = W_LOADS + 4;        // ... and for all pieces of one K-tile
        const bool first = i == 0;
        const int cm0 = m0, cn0 = n0;
        const int w_row0 = wc * WCOLS;
        MMR_STAMP(stamp_slot, 0);
        float4 bias4[NIW];
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) bias4[ni] = *reinterpret_cast<const float4 *>(bias + cn0 + wc * WCOLS + ni * 16 + fg * 4);
        f32x4 acc[NIW][8];
#pragma unroll
        for (int ii = 0; ii < NIW; ++ii)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[ii][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // K-tile 0's pieces have landed once only these YOUNGER entries remain in the in-order queue: the loads of K-tile 1's
        // pieces, (after the first tile) the previous tile's 16 epilogue stores, and the bias loads just issued
        if (nkt > 1) { if (first) wait_vmcnt<KT1_LOADS + BIAS_LOADS>(); else wait_vmcnt<KT1_LOADS + GEMM2P_STORES + BIAS_LOADS>(); }
        else { if (first) wait_vmcnt<BIAS_LOADS>(); else wait_vmcnt<GEMM2P_STORES + BIAS_LOADS>(); }
        __builtin_amdgcn_s_barrier();
        MMR_STAMP(stamp_slot, 1);
        if (wr == 1) __builtin_amdgcn_s_barrier();

        bf16x8 af[4][2], wf[NIW][2];
#define MMR_LOAD_DONE()  do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
        for (int kt = 0; kt < nkt; ++kt) {
            const char *sb = smem + (kt & 1) * STAGE2_BYTES;
            const char *ta = sb + wr * HALF_BYTES;
            const char *tw = sb + 2 * HALF_BYTES;
            auto read_w = [&]() {
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        wf[ni][ks] = *reinterpret_cast<const bf16x8 *>(tw + tile_off(w_row0 + ni * 16 + fr, ks * 4 + fg));
            };
            auto read_a = [&](int mh) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        af[mi][ks] = *reinterpret_cast<const bf16x8 *>(ta + tile_off(mh * 64 + mi * 16 + fr, ks * 4 + fg));
            };
            auto mma = [&](int mh) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi)
                            acc[ni][mh * 4 + mi] =
                                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][ks], af[mi][ks], acc[ni][mh * 4 + mi], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            };
            read_w(); read_a(0);
            if (kt > 0) {                      // K-tile 1's A pieces came with the tile prologue
                stage(kt + 1, 2);
                stage(kt + 1, 3);
            }
            MMR_LOAD_DONE();
            mma(0);
            __builtin_amdgcn_s_barrier();
            read_a(1);
            stage(kt + 2, 0);
            stage(kt + 2, 1);
            if (kt == 0 && nkt > 2) {
                // K-tile 1's pieces are OLDER than the previous tile's stores and this tile's bias loads: leave those (and
                // the W pieces of K-tile 2 just issued) outstanding -- the stores get until K-tile 1's wait to drain
                if (first) wait_vmcnt<W_LOADS + BIAS_LOADS>(); else wait_vmcnt<W_LOADS + BIAS_LOADS + GEMM2P_STORES>();
            } else if (kt + 2 < nkt) wait_vmcnt<W_LOADS>();
            else wait_vmcnt<0>();
            MMR_LOAD_DONE();
            mma(1);
            __builtin_amdgcn_s_barrier();
        }
#undef MMR_LOAD_DONE
        if (wr == 0) __builtin_amdgcn_s_barrier();
        MMR_STAMP(stamp_slot, 2);
        // Every wave is past its last fragment read (they all arrived at the barrier above) and the VM queue is empty (the
        // last K-tile waited vmcnt(0)).  Touch the bias registers here so that the compiler's own wait for those loads sits
        // at this point, where it is free -- not in the epilogue, behind the prologue loads issued next.
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) asm volatile("" : "+v"(bias4[ni].x), "+v"(bias4[ni].y), "+v"(bias4[ni].z), "+v"(bias4[ni].w));

        if (i + 1 < my_tiles) {
            set_tile(i + 1);
            tile_prologue();
        }

        // ---- epilogue of tile (cm0, cn0): four passes of 32 rows through this wave's private 4 KiB scratch.
        // The scratch accesses are inline asm on purpose: for C++ LDS accesses hipcc orders every ds_write behind the
        // pending LDS-DMA of the same __shared__ array with an s_waitcnt vmcnt(0) (checked in the ISA), which would hold
        // the epilogue until the next tile's prologue loads have landed -- the overlap this kernel exists for.  The
        // scratch is private to the wave and the LDS executes one wave's operations in order, so a pass needs no wait
        // between its writes and its reads, nor between its reads and the next pass's writes; only the registers that
        // receive the reads are waited for (lgkmcnt) before the stores use them.  16 store instructions per wave whatever
        // the tile width (a half tile's stores have lanes rc < 4 active): the next tile's counted waits rely on that.
        const size_t row_base = (size_t)(cm0 + wr * 128);
        const int col_base = cn0 + wc * WCOLS;
        const int rc = lane & 7, rr0 = lane >> 3;
        const uint32_t sbase = (uint32_t)(uintptr_t)scratch;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) {
                const float4 b4 = bias4[ni];
                const int c = ni * 2 + (fg >> 1);
#pragma unroll
                for (int mj = 0; mj < 2; ++mj) {
                    const f32x4 a = acc[ni][2 * p + mj];
                    const float v0 = epi_act<EPI>(a[0] + b4.x), v1 = epi_act<EPI>(a[1] + b4.y);
                    const float v2 = epi_act<EPI>(a[2] + b4.z), v3 = epi_act<EPI>(a[3] + b4.w);
                    const int row = mj * 16 + fr;
                    u32x2_t pk;
                    pk.x = pack_bf16x2(v0, v1);
                    pk.y = pack_bf16x2(v2, v3);
                    const uint32_t addr = sbase + row * 128 + ((c ^ (row & 7)) << 4) + (fg & 1) * 8;
                    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(pk) : "memory");
                }
            }
            u32x4_t rows4[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int row = ii * 8 + rr0;
                const uint32_t addr = sbase + row * 128 + ((rc ^ (row & 7)) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(rows4[ii]) : "v"(addr) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rows4[0]), "+v"(rows4[1]), "+v"(rows4[2]), "+v"(rows4[3])::"memory");
            if (rc < 2 * NIW) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    epi_store16(out + MMR_OUT_ROW(row_base + p * 32 + ii * 8 + rr0) * N + col_base + rc * 8, rows4[ii]);
            }
        }
#ifdef MMR_GEMM_STAMPS
        MMR_STAMP(stamp_slot, 3);
        stamp_slot += 256;
#endif
    };

    set_tile(0);
    tile_prologue();
    for (int i = 0; i < my_full; ++i) run_tile(std::integral_constant<int, 4>{}, i);
    for (int i = my_full; i < my_tiles; ++i) run_tile(std::integral_constant<int, 2>{}, i);
}

template <int EPI, int NIW, bool PATCH = false>
__global__ __launch_bounds__(GEMM2_THREADS, 2) void gemm256_bf16_kernel(
    const bf16_t *__restrict__ A, const bf16_t *__restrict__ W, int M, int N, int K,
    const float *__restrict__ bias, void *__restrict__ out, GemmAux aux, int pg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int m0, n0;
    decode_tile256(blockIdx.x, gridDim.x, N / (64 * NIW), M / BM2, pg, 64 * NIW, m0, n0);
    gemm256_tile<EPI, NIW, PATCH>(A, W, M, N, K, bias, out, aux, m0, n0, smem);
}

static int device_cus()
{
    // CUs of the current device (256 on MI355X): the persistent launches run one workgroup per CU and the tile-shape choice
    // counts rounds over the CUs -- taken from the device, not assumed
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        hipDeviceProp_t prop;
        cached = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cached_dev = dev;
    }
    return cached;
}

static int tile_group_panels(int gn)
{
    // panels per tile-order group.  Round 1 sized it so that one XCD's concurrent set (32 CUs) is near-square (6 .. 8
    // panels); measured in situ on the forward, groups of 4 panels are 0.5-1.9 % faster on every tower (ViT-B/32 GEMM time
    // 2.45 -> 2.40 ms, ViT-L/14@336 40.4 -> 40.2 ms; three boxes, alternating runs): the workgroups of an XCD drift apart
    // over a round, and a narrower group keeps the A panels they share closer together in time.
    static const int force_pg = getenv("MMR_GEMM_PG") ? atoi(getenv("MMR_GEMM_PG")) : 0;
    (void)gn;
    return force_pg > 0 ? force_pg : 4;
}

template <int EPI, int NIW, bool PATCH = false>
static int launch_gemm256(const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out,
                          const GemmAux &aux, hipStream_t st)
{
    static DeviceOnce once;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_bf16_kernel<EPI, NIW, PATCH>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, GEMM2_LDS + BM2 * 8));
    }
    const int gn = N / (64 * NIW);
    const int pg = tile_group_panels(gn);
    hipLaunchKernelGGL((gemm256_bf16_kernel<EPI, NIW, PATCH>), dim3((M / BM2) * gn), dim3(GEMM2_THREADS),
                       GEMM2_LDS + BM2 * 8, st, A, W, M, N, K, bias, out, aux, pg);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

// Row panels to cut into half tiles for the persistent launch: the smallest ph whose static assignment (full tiles dealt
// round-robin, then half tiles dealt from the other end) gives the shortest slowest workgroup, a half tile costing
// HALF_COST (0.65, measured) of a full one.
static int persist_half_panels(int panels, int gn, int G)
{
    static const int forced = getenv("MMR_GEMM_HALF_PANELS") ? atoi(getenv("MMR_GEMM_HALF_PANELS")) : -1;   // A/B aid
    if (forced >= 0) return forced < panels ? forced : panels;
    const double hc = 0.65;
    auto makespan = [&](int ph) {
        const int nfull = (panels - ph) * gn, nhalf = ph * 2 * gn;
        double mx = 0.0;
        for (int b = 0; b < G; ++b) {                 // loads repeat with period G: a few representative workgroups would do,
            const int f = nfull > b ? (nfull - 1 - b) / G + 1 : 0;       // but G <= 256 and this runs once per shape
            const int h0 = G - 1 - b;
            const int h = nhalf > h0 ? (nhalf - 1 - h0) / G + 1 : 0;
            const double t = f + hc * h;
            mx = t > mx ? t : mx;
        }
        return mx;
    };
    struct Memo { int panels, gn, G, ph; };
    static thread_local Memo memo[8];
    static thread_local int memo_n = 0;
    for (int i = 0; i < memo_n; ++i)
        if (memo[i].panels == panels && memo[i].gn == gn && memo[i].G == G) return memo[i].ph;
    int best = 0;
    double best_t = makespan(0);
    for (int ph = 1; ph <= panels / 2; ++ph) {
        const double t = makespan(ph);
        if (t < best_t - 1e-9) { best_t = t; best = ph; }
    }
    if (best_t > 0.97 * makespan(0)) best = 0;
    if (memo_n < 8) memo[memo_n++] = Memo{panels, gn, G, best};
    return best;
}

template <int EPI>
static int launch_gemm256_persist(const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out,
                                  hipStream_t st)
{
    static DeviceOnce once;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_persist_kernel<EPI>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, GEMM2P_LDS));
    }
    const int panels = M / BM2, gn = N / 256;
    const int cus = device_cus();
    const int grid = panels * gn < cus ? panels * gn : cus;
    PersistPlan plan{};
    plan.ph = persist_half_panels(panels, gn, grid);
    plan.nfull = (panels - plan.ph) * gn;
    plan.nhalf = plan.ph * 2 * gn;
    plan.pg_full = tile_group_panels(gn);
    plan.pg_half = tile_group_panels(2 * gn);
    hipLaunchKernelGGL((gemm256_persist_kernel<EPI>), dim3(grid), dim3(GEMM2_THREADS), GEMM2P_LDS, st, A, W, M, N, K, bias,
                       (bf16_t *)out, plan);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

template <int EPI, int NSTG>
static int launch_gemm128s(const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out,
                           const GemmAux &aux, hipStream_t st)
{
    constexpr int lds = Gemm128Cfg<NSTG>::LDS;
    static DeviceOnce once;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_bf16_kernel<EPI, NSTG>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, NSTG>), dim3((M / BM) * (N / BN)), dim3(GEMM_THREADS), lds, st, A, W, M, N, K,
                       bias, out, aux);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

template <int EPI>
static int launch_gemm128(const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out,
                          const GemmAux &aux, hipStream_t st)
{
    static const int force = getenv("MMR_GEMM128_STAGES") ? atoi(getenv("MMR_GEMM128_STAGES")) : 0;   // 2 / 4: A/B aid
    const long long tiles = (long long)(M / BM) * (N / BN);
    const bool deep = force ? force == 4 : tiles <= 256;      // one workgroup per CU anyway: spend the LDS on prefetch depth
    return deep ? launch_gemm128s<EPI, 4>(A, W, M, N, K, bias, out, aux, st) : launch_gemm128s<EPI, 2>(A, W, M, N, K, bias, out, aux, st);
}

// ---------------------------------------------------------------------------------------------
// 128 x 32 x 64 tile for few rows (M <= 2048: up to 40 images / 26 texts per call -- the reference's own loops run batch
// 1 and 10).  A 128^2 tile gives such a GEMM only (M/128) x (N/128) workgroups -- 6 .. 24 at M = 128 -- each walking the
// whole K in one-exposed-latency steps (fc2 at M = 128: 37 us); 32-column tiles put 4x as many CUs to work and a 4-deep
// counted-wait pipeline keeps three K-tiles in flight.  ViT-B/32 forward: batch 1 0.94 -> 0.53 ms, batch 8 0.97 -> 0.62 ms
// (0.44 ms with the 64-row tile below).
// Waves split the rows (32 each, 2x2 MFMA tiles); every output element accumulates its K-steps in the same order as
// in the other two kernels, so results stay bit-identical across batch sizes.
// Epilogue: through a per-wave fp32 LDS image, then row-wise (8 lanes x 16 B = one 128-byte fp32 row segment).
// ---------------------------------------------------------------------------------------------
// The loop is paced by what ONE CU can ingest through LDS-DMA (~90 GB/s: 20 KiB per K-tile every 0.23 us; a 7-stage ring
// with six K-tiles in flight measured SLOWER), so when the 128-row grid leaves CUs idle the tile is halved to 64 rows
// (SK_BM = 64: 12 KiB per K-tile and twice the workgroups; batch 1 has 50 token rows in a 128-row tile anyway).
constexpr int SK_BN = 32, SK_STAGES = 4;
template <int SK_BM> struct SkinnyCfg {
    static constexpr int A_BYTES = SK_BM * BK * 2;                     // 16 / 8 KiB
    static constexpr int STAGE_BYTES = A_BYTES + SK_BN * BK * 2;       // + 4 KiB of W
    static constexpr int LDS = SK_STAGES * STAGE_BYTES;                // 80 / 48 KiB
};

template <int EPI, int SK_BM>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_skinny_kernel(
    const bf16_t *__restrict__ A, const bf16_t *__restrict__ W, int M, int N, int K,
    const float *__restrict__ bias, void *__restrict__ out)
{
    static_assert(!epi_lnfold(EPI) && EPI != EPI_RESID_STATS_F32, "folded-LayerNorm epilogues stay on the 128^2 kernel");
    static_assert(SK_BM == 128 || SK_BM == 64, "rows per tile");
    constexpr int WR = SK_BM / 4;                // rows per wave: 32 / 16
    constexpr int MI = WR / 16;                  // 16-row MFMA blocks per wave: 2 / 1
    constexpr int AB = WR / 8;                   // 8-row staging blocks per wave: 4 / 2
    constexpr int SK_STAGE_BYTES = SkinnyCfg<SK_BM>::STAGE_BYTES, A_BYTES = SkinnyCfg<SK_BM>::A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gn = N / SK_BN;
    const int m0 = (blockIdx.x / gn) * SK_BM;
    const int n0 = (blockIdx.x % gn) * SK_BN;

    // staging: A tile = SK_BM / 8 blocks of 8 rows (wave w: blocks AB*w .. AB*w + AB-1), W tile = 4 blocks (wave w: block w)
    const int rr = lane >> 3;
    const int sc = (lane & 7) ^ rr;
    const bf16_t *a_src = A + (size_t)(m0 + wave * WR + rr) * K + sc * 8;
    const bf16_t *w_src = W + (size_t)(n0 + wave * 8 + rr) * K + sc * 8;
    auto stage = [&](int kt, int buf) {
        char *base = smem + buf * SK_STAGE_BYTES;
        const size_t ko = (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < AB; ++i) glds16(a_src + (size_t)i * 8 * K + ko, base + (wave * AB + i) * 1024);
        glds16(w_src + ko, base + A_BYTES + wave * 1024);
    };
    constexpr int LPS = AB + 1;                  // loads per wave per stage

    f32x4 acc[2][MI];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const int nkt = K / BK;
    float4 bias4[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        bias4[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (epi_bias(EPI)) bias4[ni] = *reinterpret_cast<const float4 *>(bias + n0 + ni * 16 + fg * 4);
    }
#pragma unroll
    for (int s = 0; s < SK_STAGES - 1; ++s)
        if (s < nkt) stage(s, s);
    for (int kt = 0; kt < nkt; ++kt) {
        const int younger = min(SK_STAGES - 2, nkt - 1 - kt);
        if (younger >= 2) wait_vmcnt<2 * LPS>();
        else if (younger == 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();            // K-tile kt landed for every wave; K-tile kt-1's buffer is free
        if (kt + SK_STAGES - 1 < nkt) stage(kt + SK_STAGES - 1, (kt + SK_STAGES - 1) % SK_STAGES);
        const char *ta = smem + (kt % SK_STAGES) * SK_STAGE_BYTES;
        const char *tw = ta + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[MI], wf[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(ta + tile_off(wave * WR + i * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const bf16x8 *>(tw + tile_off(i * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    __syncthreads();      // the epilogue reuses the staging buffers

    // per-wave image [WR rows][8 chunks of 4 floats], chunk index XOR (row & 7)
    char *my = smem + wave * 4096;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int c = ni * 4 + fg;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const f32x4 a = acc[ni][mi];
            const int row = mi * 16 + fr;
            *reinterpret_cast<float4 *>(my + row * 128 + ((c ^ (row & 7)) << 4)) =
                make_float4(a[0] + bias4[ni].x, a[1] + bias4[ni].y, a[2] + bias4[ni].z, a[3] + bias4[ni].w);
        }
    }
    const int rc = lane & 7, rr0 = lane >> 3;
    const size_t row_base = (size_t)(m0 + wave * WR);
    float4 hv[AB];
    if constexpr (epi_resid(EPI)) {
#pragma unroll
        for (int i = 0; i < AB; ++i)
            hv[i] = *reinterpret_cast<const float4 *>((const float *)out + (row_base + i * 8 + rr0) * N + n0 + rc * 4);
    }
#pragma unroll
    for (int i = 0; i < AB; ++i) {
        const int row = i * 8 + rr0;
        float4 v = *reinterpret_cast<const float4 *>(my + row * 128 + ((rc ^ (row & 7)) << 4));
        const size_t o = (row_base + row) * N + n0 + rc * 4;
        if constexpr (epi_bf16(EPI)) {
            uint2 pk;
            pk.x = pack_bf16x2(epi_act<EPI>(v.x), epi_act<EPI>(v.y));
            pk.y = pack_bf16x2(epi_act<EPI>(v.z), epi_act<EPI>(v.w));
            *reinterpret_cast<uint2 *>((bf16_t *)out + o) = pk;
        } else {
            if constexpr (epi_resid(EPI)) { v.x += hv[i].x; v.y += hv[i].y; v.z += hv[i].z; v.w += hv[i].w; }
            *reinterpret_cast<float4 *>((float *)out + o) = v;
        }
    }
}

template <int EPI, int SK_BM>
static int launch_gemm_skinny_m(const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out, hipStream_t st)
{
    constexpr int lds = SkinnyCfg<SK_BM>::LDS;
    static DeviceOnce once;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_skinny_kernel<EPI, SK_BM>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    hipLaunchKernelGGL((gemm_skinny_kernel<EPI, SK_BM>), dim3((M / SK_BM) * (N / SK_BN)), dim3(GEMM_THREADS), lds, st, A, W, M, N,
                       K, bias, out);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

template <int EPI>
static int launch_gemm_skinny(const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out, hipStream_t st)
{
    static const int force = getenv("MMR_SKINNY_ROWS") ? atoi(getenv("MMR_SKINNY_ROWS")) : 0;      // 64 / 128: A/B aid
    // 128-row tiles would leave a third of the CUs idle (measured crossover, ViT-B/32: batch ~16)
    const bool half = force ? force == 64 : (long long)(M / BM) * (N / SK_BN) < 160;
    return half ? launch_gemm_skinny_m<EPI, 64>(A, W, M, N, K, bias, out, st) : launch_gemm_skinny_m<EPI, 128>(A, W, M, N, K, bias, out, st);
}

// Hint from the tower forward (mmr_tower_set_shared_chip): this launch shares the chip with other concurrent work (a second
// batch in flight on another stream).  The tile choice below then stops trading efficiency for occupancy: with the chip to
// itself N = 768 at M = 12800 runs as 200 tiles of 256x192 rather than 150 of 256x256 (78 % instead of 59 % of the CUs), but
// with a second forward filling the idle CUs the full tiles win -- 12 % fewer operand bytes per FLOP (measured, two ViT-B/32
// forwards in flight: +2.5-3 % images/s; one in flight: -5 %).  Host-side and per thread: a forward issues all its launches
// from the calling thread.  Same K order per output element either way, so the results do not change.
static thread_local bool g_shared_chip = false;
void gemm_set_shared_chip_hint(bool on) { g_shared_chip = on; }

// host launcher (internal): shapes are validated by the caller in tower.hip
int launch_gemm_aux(int epi, const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out,
                    const GemmAux &aux, hipStream_t st)
{
    if (M % BM || N % BN || K % BK || M <= 0 || N <= 0 || K <= 0) {
        set_error("gemm: M=%d N=%d K=%d must be positive multiples of %d/%d/%d", M, N, K, BM, BN, BK);
        return MMR_EINVAL;
    }
    if (epi < 0 || epi >= EPI_COUNT) { set_error("gemm: unknown epilogue %d", epi); return MMR_EINVAL; }
    if (epi_lnfold(epi) && (!aux.stats_in || !aux.colsum)) { set_error("gemm: LN-fold epilogue needs row stats and column sums"); return MMR_EINVAL; }
    if (epi == EPI_RESID_STATS_F32 && (!aux.stats_out || !aux.xout)) { set_error("gemm: RESID_STATS epilogue needs stats_out and xout"); return MMR_EINVAL; }
    ProfScope prof(MMR_PROF_GEMM, st);
    static const int force = getenv("MMR_GEMM_TILE") ? atoi(getenv("MMR_GEMM_TILE")) : 0;   // 128 / 192 / 256: A/B aid
    // Tile choice.  The 256-row kernels run one workgroup per CU: they need enough tiles to occupy the chip, and
    // among the two widths the cheaper is the one with fewer (rounds over 256 CUs) x (tile width).
    const int cus = device_cus();
    auto cost = [&](int bn) -> long long {
        if (M % BM2 || N % bn) return -1;
        const long long tiles = (long long)(M / BM2) * (N / bn);
        static const int min_tiles = getenv("MMR_GEMM_MINTILES") ? atoi(getenv("MMR_GEMM_MINTILES")) : 128;   // A/B aid
        if (tiles < min_tiles) return -1;
        return (tiles + cus - 1) / cus * bn;
    };
    const long long c256 = cost(256), c192 = cost(192);
    int tile = 128;
    if (c256 > 0) tile = 256;
    if (c192 > 0 && (c256 < 0 || (c192 < c256 && !g_shared_chip))) tile = 192;
    if (force == 128) tile = 128;
    if (force == 256 && M % BM2 == 0 && N % 256 == 0) tile = 256;
    if (force == 192 && M % BM2 == 0 && N % 192 == 0) tile = 192;
    if (epi == EPI_RESID_STATS_F32 && N / (tile == 192 ? 48 : 64) > LNFOLD_NP) {
        set_error("gemm: RESID_STATS row of %d columns needs more than %d partial slots", N, LNFOLD_NP);
        return MMR_EINVAL;
    }
    // very few rows: 32-column tiles (see gemm_skinny_kernel); the folded-LayerNorm epilogues stay on the 128^2 kernel
    static const int skinny_rows = getenv("MMR_GEMM_SKINNY_ROWS") ? atoi(getenv("MMR_GEMM_SKINNY_ROWS")) : 2048;   // measured crossover with the 128^2 kernel (ViT-B/32: batch 40)
    if (tile == 128 && force == 0 && M <= skinny_rows && !epi_lnfold(epi) && epi != EPI_RESID_STATS_F32) {
        switch (epi) {
            case EPI_BIAS_BF16: return launch_gemm_skinny<EPI_BIAS_BF16>(A, W, M, N, K, bias, out, st);
            case EPI_BIAS_GELU_BF16: return launch_gemm_skinny<EPI_BIAS_GELU_BF16>(A, W, M, N, K, bias, out, st);
            case EPI_BIAS_RESID_F32: return launch_gemm_skinny<EPI_BIAS_RESID_F32>(A, W, M, N, K, bias, out, st);
            case EPI_STORE_F32: return launch_gemm_skinny<EPI_STORE_F32>(A, W, M, N, K, bias, out, st);
            case EPI_BIAS_F32: return launch_gemm_skinny<EPI_BIAS_F32>(A, W, M, N, K, bias, out, st);
            case EPI_BIAS_GELU_ERF_BF16: return launch_gemm_skinny<EPI_BIAS_GELU_ERF_BF16>(A, W, M, N, K, bias, out, st);
            case EPI_BIAS_TANH_BF16: return launch_gemm_skinny<EPI_BIAS_TANH_BF16>(A, W, M, N, K, bias, out, st);
        }
    }
    // multi-round launches with a bf16 epilogue (qkv, fc1): persistent workgroups over a full + half tile list
    // (gemm256_persist_kernel); MMR_GEMM_PERSIST=0 falls back to one workgroup per tile (A/B aid)
    static const int persist = getenv("MMR_GEMM_PERSIST") ? atoi(getenv("MMR_GEMM_PERSIST")) : 1;
    // ... unless the chip is shared (mmr_tower_set_shared_chip) and the launch is only a few rounds of long tiles: 256 workgroups
    // pinned for the whole launch keep the other batch's kernels out, and with K >= 768 the next-tile prologue the persistent
    // schedule hides is a small share of a tile (two ViT-B/32 forwards in flight: +1 % images/s, raw-image build +2 %; the text
    // tower's K = 512 tiles and ViT-L/14's 13-18 rounds lose 2.5 % without persistence and keep it)
    static const int shared_nopersist = getenv("MMR_GEMM_SHARED_NOPERSIST") ? atoi(getenv("MMR_GEMM_SHARED_NOPERSIST")) : 1;   // A/B aid
    const long long tiles256 = (long long)(M / BM2) * (N / 256);
    const bool yield_chip = g_shared_chip && shared_nopersist && K >= 768 && tiles256 <= 3LL * cus;
    if (persist && tile == 256 && tiles256 > cus && !yield_chip) {
        switch (epi) {
            case EPI_BIAS_BF16: return launch_gemm256_persist<EPI_BIAS_BF16>(A, W, M, N, K, bias, out, st);
            case EPI_BIAS_GELU_BF16: return launch_gemm256_persist<EPI_BIAS_GELU_BF16>(A, W, M, N, K, bias, out, st);
            case EPI_BIAS_GELU_ERF_BF16: return launch_gemm256_persist<EPI_BIAS_GELU_ERF_BF16>(A, W, M, N, K, bias, out, st);
            default: break;
        }
    }
#define MMR_GEMM_CASE(E)                                                                                   \
    case E: return tile == 256   ? launch_gemm256<E, 4>(A, W, M, N, K, bias, out, aux, st)                 \
                   : tile == 192 ? launch_gemm256<E, 3>(A, W, M, N, K, bias, out, aux, st)                 \
                                 : launch_gemm128<E>(A, W, M, N, K, bias, out, aux, st);
    switch (epi) {
        MMR_GEMM_CASE(EPI_BIAS_BF16)
        MMR_GEMM_CASE(EPI_BIAS_GELU_BF16)
        MMR_GEMM_CASE(EPI_BIAS_RESID_F32)
        MMR_GEMM_CASE(EPI_STORE_F32)
        MMR_GEMM_CASE(EPI_BIAS_F32)
        MMR_GEMM_CASE(EPI_BIAS_GELU_ERF_BF16)
        MMR_GEMM_CASE(EPI_BIAS_TANH_BF16)
        MMR_GEMM_CASE(EPI_LNFOLD_BF16)
        MMR_GEMM_CASE(EPI_LNFOLD_GELU_BF16)
        MMR_GEMM_CASE(EPI_RESID_STATS_F32)
    }
#undef MMR_GEMM_CASE
    return MMR_EINVAL;
}

// Patch-embed conv as a GEMM with the patch gather fused into the A-tile loads (patch 32, bf16 pixels): out[M,N] fp32 =
// patches(pix)[M,3072] . W[N,3072]^T, no im2col pass.  Returns MMR_ENOTSUP when the shape does not fit the 256-row
// kernel (the caller then runs im2col + launch_gemm).
int launch_gemm_patch32(const bf16_t *pix, int B, int S, const bf16_t *W, int M, int N, float *out, hipStream_t st)
{
    const int G = S / 32, K = 3 * 32 * 32;
    if (S % 32 || B < 1 || M % BM2 || M < B * G * G) return MMR_ENOTSUP;
    const long long t256 = N % 256 ? -1 : (long long)(M / BM2) * (N / 256), t192 = N % 192 ? -1 : (long long)(M / BM2) * (N / 192);
    if (t256 < 128 && t192 < 128) return MMR_ENOTSUP;
    GemmAux aux{};
    aux.pix = pix; aux.pix_size = S; aux.pix_grid = G; aux.patch_rows = B * G * G;
    ProfScope prof(MMR_PROF_GEMM, st);
    const long long c256 = t256 >= 128 ? (t256 + 255) / 256 * 256 : -1, c192 = t192 >= 128 ? (t192 + 255) / 256 * 192 : -1;
    if (c192 > 0 && (c256 < 0 || (c192 < c256 && !g_shared_chip))) return launch_gemm256<EPI_STORE_F32, 3, true>(nullptr, W, M, N, K, nullptr, out, aux, st);
    return launch_gemm256<EPI_STORE_F32, 4, true>(nullptr, W, M, N, K, nullptr, out, aux, st);
}

#ifdef MMR_GEMM_STAMPS
}  // namespace mmr
extern "C" int mmr_debug_gemm_stamps(unsigned long long *host_out, int clear)
{
    using namespace mmr;
    if (host_out) MMR_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mmr_stamp_buf), sizeof(unsigned long long) * 8192 * 4));
    if (host_out) MMR_CHECK_HIP(hipMemcpyFromSymbol(host_out + 8192 * 4, HIP_SYMBOL(mmr_cycle_buf), sizeof(unsigned long long) * 8192 * 4));
    if (clear) {
        void *p = nullptr;
        MMR_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(mmr_stamp_buf)));
        MMR_CHECK_HIP(hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 4));
    }
    return MMR_OK;
}
namespace mmr {
#endif

int launch_gemm(int epi, const bf16_t *A, const bf16_t *W, int M, int N, int K, const float *bias, void *out,
                hipStream_t st)
{
    if (epi >= EPI_LNFOLD_BF16) { set_error("gemm: epilogue %d needs launch_gemm_aux", epi); return MMR_EINVAL; }
    return launch_gemm_aux(epi, A, W, M, N, K, bias, out, GemmAux{}, st);
}

}  // namespace mmr
