// Cosine / dot-product top-k over a gallery matrix for gfx950 (MI355X).
//
// Replaces the reference's scoring + ranking expressions
//     similarity = 100. * features @ ref_feature.t()     reference code/search_image.py:107
//     output.topk(k, 1, True, True)                       reference code/utils.py:17
// generalised to Q query rows.  The [Q,N] score matrix is never
// written.
//
// Structure (DESIGN.md "search"):
//   scan_kernel      HBM-bound.  Streams the bf16 gallery once through LDS (global_load_lds,
//                    3-deep ring, counted vmcnt, raw s_barrier), multiplies each 32-row tile with
//                    up to 256 register-resident queries on v_mfma_f32_32x32x16_bf16 and keeps only
//                    the per-(query, tile) maximum ("bucket max") and per-(query, task) maximum.
//   finalize_kernel  one workgroup per query: picks the KS best tasks, then the KS best tiles inside
//                    them, then re-scores those KS*32 rows EXACTLY (fp64, fixed summation order
//                    shared with oracle/search_ref.c) and orders them by (-dot, +row).  It certifies
//                    that no excluded tile could hold a top-k row; uncertified queries are flagged.
//   exh_* kernels    exhaustive exact fp64 path: fp32 galleries, unsupported E / large k, and any
//                    query the certificate rejected.  Launched unconditionally; unflagged queries
//                    exit at once, so there is no host round trip.
// Why this is exact: every row with dot >= (k-th best dot) lives in a tile whose max is >= that
// value; at most KS tiles can have such a max unless more than KS-k rows are within the MFMA
// rounding error of the k-th -- which is exactly what the certificate checks.
#include "mmr_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace mmr {

constexpr int TILE_ROWS = 32;
#ifndef MMR_SCAN_NBUF
#define MMR_SCAN_NBUF 3
#endif
constexpr int SCAN_NBUF = MMR_SCAN_NBUF;
#ifndef MMR_SCAN_CHAINS_FOR
#define MMR_SCAN_CHAINS_FOR(waves) ((waves) == 4 ? 2 : 1)   // independent MFMA accumulation chains per wave
#endif
static_assert(SCAN_NBUF == 3 || SCAN_NBUF == 4, "wait counts below assume a prefetch distance of 2 or 3 tiles");
constexpr int MAX_TPT = 64;                 // tiles per task
constexpr int KS_MAX = 32;                  // candidate tiles kept per query
constexpr int K_MAX = 64;                   // largest k (exhaustive path)
constexpr int FIN_THREADS = 256;

__host__ __device__ static inline bool ranks_before(double sa, int64_t ia, double sb, int64_t ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// ---------------------------------------------------------------------------------------------
// scan
// ---------------------------------------------------------------------------------------------
// E <= 512: 8 waves x 32 queries (2 waves per SIMD, <= 256 VGPRs each).  E = 768 needs 192 VGPRs
// for the resident queries alone, so it runs 4 waves x 32 queries at one wave per SIMD.
template <int E>
struct ScanCfg {
    static constexpr int SCAN_WAVES = E <= 512 ? 8 : 4;
    static constexpr int SCAN_THREADS = SCAN_WAVES * 64;
    static constexpr int QMAX = SCAN_WAVES * 32;      // queries per scan pass
    static constexpr int CH = E / 8;                  // 16-byte chunks per gallery row
    static constexpr int ROWB = E * 2;                // bytes per row
    static constexpr int TILE_BYTES = TILE_ROWS * ROWB;
    static constexpr int LOADS = TILE_ROWS * CH / 64; // glds wave-instructions per tile
    static constexpr int LPW = LOADS / SCAN_WAVES;    // per wave
    static constexpr int KSTEPS = E / 16;
    static_assert(LOADS % SCAN_WAVES == 0, "tile loads must split evenly over the waves");
    static_assert(CH % 16 == 0, "XOR swizzle works on groups of 16 chunks");
};

template <int E>
__global__ __launch_bounds__(ScanCfg<E>::SCAN_THREADS, ScanCfg<E>::SCAN_WAVES / 4) void scan_kernel(
    const bf16_t *__restrict__ q, const bf16_t *__restrict__ gal, int Q, int64_t N, int ntiles, int tpt,
    int qwaves, int qpad, float *__restrict__ bmax, float *__restrict__ tmax)
{
    using C = ScanCfg<E>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int task = blockIdx.x;
    const int t0 = task * tpt;
    const int t1 = min(ntiles, t0 + tpt);
    const bool compute = wave < qwaves;

    // B operand: this wave's 32 queries, resident for the whole task.  Lane (c,h) holds, for
    // k-step s, the 8 elements [16s + 8h, 16s + 8h + 8) of query wave*32 + c.
    bf16x8 bq[C::KSTEPS];
    {
        const int qrow = wave * 32 + c;
        const bool live = compute && qrow < Q;
        const bf16_t *qp = q + (size_t)(live ? qrow : 0) * E + h * 8;
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            bf16x8 v = *reinterpret_cast<const bf16x8 *>(qp + s * 16);
            bq[s] = live ? v : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    // Stage one 32-row tile: LDS slot p (16 B) of the tile holds chunk ((p%CH) ^ row) of row p/CH
    // (XOR on the low 4 bits): the LDS image stays lane-linear for global_load_lds while
    // ds_read_b128 of 32 different rows at one k offset is bank-conflict free.
    auto stage = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < C::LPW; ++i) {
            const int instr = wave * C::LPW + i;
            const int p = instr * 64 + lane;
            const int row = p / C::CH;
            const int pos = p % C::CH;
            const int chunk = (pos & ~15) | ((pos ^ row) & 15);
            int64_t grow = (int64_t)tile * TILE_ROWS + row;
            grow = grow < N ? grow : N - 1;  // clamp: rows past N are masked after the MFMA
            glds16(gal + grow * E + chunk * 8, smem + buf * C::TILE_BYTES + instr * 1024);
        }
    };

    const int rowoff = c * C::ROWB;

    float task_max = -INFINITY;
    float pend = -INFINITY;
    int pend_tile = -1;

    // prefetch distance PD = NBUF - 1 tiles
    constexpr int PD = SCAN_NBUF - 1;
#pragma unroll
    for (int i = 0; i < PD; ++i)
        if (t0 + i < t1) stage(t0 + i, i);
    int cur = 0;

    for (int t = t0; t < t1; ++t) {
        // this wave's loads of tile t have landed (younger tiles may stay in flight) ...
        const int younger = min(PD - 1, t1 - 1 - t);
        if (younger >= 2) wait_vmcnt<2 * C::LPW>();
        else if (younger == 1) wait_vmcnt<C::LPW>();
        else wait_vmcnt<0>();
        // ... and after the barrier so have every other wave's.
        __builtin_amdgcn_s_barrier();

        if (compute && pend_tile >= 0 && h == 0) bmax[(size_t)pend_tile * qpad + wave * 32 + c] = pend;

        int nxt = cur + PD; nxt = nxt >= SCAN_NBUF ? nxt - SCAN_NBUF : nxt;
        if (t + PD < t1) stage(t + PD, nxt);  // overwrites tile t-1's buffer: all waves are past it

        if (compute) {
            const char *tb = smem + cur * C::TILE_BYTES + rowoff;
            // CHAINS = 2 (one wave per SIMD, E = 768): even and odd k-steps accumulate into separate registers, so a
            // wave that has no SIMD partner to alternate with is not held to one dependent MFMA at a time
            constexpr int CHAINS = MMR_SCAN_CHAINS_FOR(C::SCAN_WAVES);
            f32x16 acc, acc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
            // A fragments run PF k-steps ahead of the MFMA that consumes them.  hipcc waits lgkmcnt(0) in
            // front of every second MFMA when it schedules these reads itself (each wait then exposes the
            // LDS latency and the MFMA pipe idles half the time), so the reads are issued as inline asm and
            // retired with COUNTED waits: the fragment consumed at step s was issued PF steps earlier and
            // PF-1 younger reads may stay in flight.  The wait statement names the fragment "+v" so no use
            // of it can be scheduled above the wait (cdna_hip_programming.md section 5.7, form ii).
#ifndef MMR_SCAN_PF
#define MMR_SCAN_PF 4
#endif
            constexpr int PF = MMR_SCAN_PF;
            bf16x8 a[PF];
            auto issue = [&](int s, bf16x8 &dst) {
                const int chunk = 2 * s + h;
                const int pos = (chunk & ~15) | ((chunk ^ c) & 15);
                const uint32_t addr = (uint32_t)(uintptr_t)(tb + pos * 16);   // LDS byte address
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
            };
#pragma unroll
            for (int s = 0; s < PF; ++s) issue(s, a[s]);
#pragma unroll
            for (int s = 0; s < C::KSTEPS; ++s) {
                const int younger = (C::KSTEPS - 1 - s) < (PF - 1) ? (C::KSTEPS - 1 - s) : (PF - 1);
                if (younger == 7) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a[s % PF]));
                else if (younger == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[s % PF]));
                else if (younger == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a[s % PF]));
                else if (younger == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[s % PF]));
                else if (younger == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[s % PF]));
                else if (younger == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[s % PF]));
                else if (younger == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a[s % PF]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[s % PF]));
                const bool second = CHAINS == 2 && (s & 1);
                if (second) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % PF], bq[s], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % PF], bq[s], acc, 0, 0, 0);
                if (s + PF < C::KSTEPS) {
                    // the MFMA above must have READ a[s % PF] before the next load overwrites it: the
                    // empty statement ties the accumulator to this point so the load cannot move above it
                    if (second) asm volatile("" : "+v"(acc2)); else asm volatile("" : "+v"(acc));
                    issue(s + PF, a[s % PF]);
                }
            }
            if (CHAINS == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
            }
            // acc[i] = dot(query c, tile row (i&3) + 8*(i>>2) + 4*h)
            float m = -INFINITY;
            if ((int64_t)(t + 1) * TILE_ROWS <= N) {
#pragma unroll
                for (int i = 0; i < 16; ++i) m = fmaxf(m, acc[i]);
            } else {
                const int64_t base = (int64_t)t * TILE_ROWS + 4 * h;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t r = base + (i & 3) + 8 * (i >> 2);
                    m = fmaxf(m, r < N ? acc[i] : -INFINITY);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            task_max = fmaxf(task_max, m);
            pend = m;
            pend_tile = t;
        }
        cur = cur + 1 >= SCAN_NBUF ? 0 : cur + 1;
    }
    if (compute && h == 0) {
        if (pend_tile >= 0) bmax[(size_t)pend_tile * qpad + wave * 32 + c] = pend;
        tmax[(size_t)task * qpad + wave * 32 + c] = task_max;
    }
}

// ---------------------------------------------------------------------------------------------
// scan for wide rows (E = 768, the ViT-L/14 embedding; BASELINE configs[4]).  32 resident queries of 768 dims are 192
// VGPRs, which forced the 32x32 form above down to one wave per SIMD with the queries parked in AccVGPRs and copied
// back in front of every MFMA (0.33 of the HBM roof).  Here each of 8 waves keeps 16 queries (96 VGPRs) and multiplies
// with v_mfma_f32_16x16x32_bf16: the same 128 queries per pass, but two waves per SIMD (one reads LDS while the other
// issues MFMAs), no register shuffling, and the 16x16 shape's higher sustained clock.  A 32-row tile is two 16-row
// blocks with one accumulator chain each.  Same LDS image, staging ring, counted waits and bmax/tmax outputs as
// scan_kernel, so the finalize kernels do not care which scan ran.
//   B operand: lane (c = lane & 15, g = lane >> 4) holds elements [32s + 8g, +8) of query wave*16 + c for k-step s;
//   A operand: the same 8 elements of tile row 16*rb + c;  D: acc[i] = dot(query c, tile row 16*rb + 4g + i).
// Bank check for the A reads (ds_read_b128, 16-lane groups {0-3,12-15,20-27} ...): a group's lanes read rows
// {0-3,12-15} at chunk 4s and rows {4-11} at chunk 4s+1; slot = (chunk ^ row) & 15 gives 16 distinct slots.
// ---------------------------------------------------------------------------------------------
template <int E>
struct Scan16Cfg {
    static constexpr int SCAN_WAVES = 8;
    static constexpr int SCAN_THREADS = SCAN_WAVES * 64;
    static constexpr int QMAX = SCAN_WAVES * 16;      // 128 queries per scan pass
    static constexpr int CH = E / 8;
    static constexpr int ROWB = E * 2;
    static constexpr int TILE_BYTES = TILE_ROWS * ROWB;
    static constexpr int LOADS = TILE_ROWS * CH / 64;
    static constexpr int LPW = LOADS / SCAN_WAVES;
    static constexpr int KSTEPS = E / 32;
    static_assert(LOADS % SCAN_WAVES == 0 && CH % 16 == 0, "tile geometry");
};

template <int E>
__global__ __launch_bounds__(Scan16Cfg<E>::SCAN_THREADS, 2) void scan16_kernel(
    const bf16_t *__restrict__ q, const bf16_t *__restrict__ gal, int Q, int64_t N, int ntiles, int tpt,
    int qwaves, int qpad, float *__restrict__ bmax, float *__restrict__ tmax)
{
    using C = Scan16Cfg<E>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int task = blockIdx.x;
    const int t0 = task * tpt;
    const int t1 = min(ntiles, t0 + tpt);
    const bool compute = wave < qwaves;

    bf16x8 bq[C::KSTEPS];
    {
        const int qrow = wave * 16 + c;
        const bool live = compute && qrow < Q;
        const bf16_t *qp = q + (size_t)(live ? qrow : 0) * E + g * 8;
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            bf16x8 v = *reinterpret_cast<const bf16x8 *>(qp + s * 32);
            bq[s] = live ? v : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    auto stage = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < C::LPW; ++i) {
            const int instr = wave * C::LPW + i;
            const int p = instr * 64 + lane;
            const int row = p / C::CH;
            const int pos = p % C::CH;
            const int chunk = (pos & ~15) | ((pos ^ row) & 15);
            int64_t grow = (int64_t)tile * TILE_ROWS + row;
            grow = grow < N ? grow : N - 1;
            glds16(gal + grow * E + chunk * 8, smem + buf * C::TILE_BYTES + instr * 1024);
        }
    };
    float task_max = -INFINITY, pend = -INFINITY;
    int pend_tile = -1;
    constexpr int PD = SCAN_NBUF - 1;
#pragma unroll
    for (int i = 0; i < PD; ++i)
        if (t0 + i < t1) stage(t0 + i, i);
    int cur = 0;
    for (int t = t0; t < t1; ++t) {
        const int younger = min(PD - 1, t1 - 1 - t);
        if (younger >= 2) wait_vmcnt<2 * C::LPW>();
        else if (younger == 1) wait_vmcnt<C::LPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (compute && pend_tile >= 0 && g == 0) bmax[(size_t)pend_tile * qpad + wave * 16 + c] = pend;
        int nxt = cur + PD; nxt = nxt >= SCAN_NBUF ? nxt - SCAN_NBUF : nxt;
        if (t + PD < t1) stage(t + PD, nxt);
        if (compute) {
            const char *tb = smem + cur * C::TILE_BYTES;
            // step u = 2*s + rb: k-step s of row block rb; the two row blocks alternate, so consecutive MFMAs belong to
            // different accumulation chains.  Fragment reads run PF steps ahead (inline asm + counted lgkmcnt: see scan_kernel).
            constexpr int NU = 2 * C::KSTEPS;
            constexpr int PF = 6;
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x8 a[PF];
            auto issue = [&](int u, bf16x8 &dst) {
                const int s = u >> 1, rb = u & 1;
                const int row = rb * 16 + c;
                const int chunk = 4 * s + g;
                const int pos = (chunk & ~15) | ((chunk ^ row) & 15);
                const uint32_t addr = (uint32_t)(uintptr_t)(tb + row * C::ROWB + pos * 16);
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
            };
#pragma unroll
            for (int u = 0; u < PF; ++u) issue(u, a[u]);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int younger_r = (NU - 1 - u) < (PF - 1) ? (NU - 1 - u) : (PF - 1);
                if (younger_r == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a[u % PF]));
                else if (younger_r == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[u % PF]));
                else if (younger_r == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[u % PF]));
                else if (younger_r == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[u % PF]));
                else if (younger_r == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a[u % PF]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[u % PF]));
                if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u % PF], bq[u >> 1], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u % PF], bq[u >> 1], acc0, 0, 0, 0);
                if (u + PF < NU) {
                    // the MFMA above must have read a[u % PF] before the next load overwrites it
                    if (u & 1) asm volatile("" : "+v"(acc1)); else asm volatile("" : "+v"(acc0));
                    issue(u + PF, a[u % PF]);
                }
            }
            // acc0[i] = dot(query c, tile row 4g + i); acc1[i]: tile row 16 + 4g + i
            float m = -INFINITY;
            if ((int64_t)(t + 1) * TILE_ROWS <= N) {
                m = fmaxf(fmaxf(fmaxf(acc0[0], acc0[1]), fmaxf(acc0[2], acc0[3])),
                          fmaxf(fmaxf(acc1[0], acc1[1]), fmaxf(acc1[2], acc1[3])));
            } else {
                const int64_t base = (int64_t)t * TILE_ROWS + 4 * g;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m = fmaxf(m, base + i < N ? acc0[i] : -INFINITY);
                    m = fmaxf(m, base + 16 + i < N ? acc1[i] : -INFINITY);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            task_max = fmaxf(task_max, m);
            pend = m;
            pend_tile = t;
        }
        cur = cur + 1 >= SCAN_NBUF ? 0 : cur + 1;
    }
    if (compute && g == 0) {
        if (pend_tile >= 0) bmax[(size_t)pend_tile * qpad + wave * 16 + c] = pend;
        tmax[(size_t)task * qpad + wave * 16 + c] = task_max;
    }
}

// ---------------------------------------------------------------------------------------------
// scan for fp32 galleries (the dtype of the reference's feature caches): same pipeline, 16-row
// tiles, v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, 1/16 of the bf16 MFMA rate -> MFMA-bound).
// Each wave keeps 16 queries resident (E/4 VGPRs); lane (r, g) owns the 16-byte chunk 4S+g of row r
// for "super-step" S and feeds its 4 floats to MFMA steps 4S..4S+3 -- a k-permutation shared by
// both operands, so one conflict-free ds_read_b128 serves four MFMAs.
// ---------------------------------------------------------------------------------------------
constexpr int TILE_ROWS_F32 = 16;

template <int E>
struct ScanF32Cfg {
    static constexpr int SCAN_WAVES = E <= 512 ? 8 : 4;
    static constexpr int SCAN_THREADS = SCAN_WAVES * 64;
    static constexpr int QMAX = SCAN_WAVES * 16;
    static constexpr int CH = E / 4;                       // 16-byte chunks per fp32 row
    static constexpr int ROWB = E * 4;
    static constexpr int TILE_BYTES = TILE_ROWS_F32 * ROWB;
    static constexpr int LOADS = TILE_ROWS_F32 * CH / 64;
    static constexpr int LPW = LOADS / SCAN_WAVES;
    static constexpr int SSTEPS = E / 16;                  // super-steps (4 MFMAs each)
    static_assert(LOADS % SCAN_WAVES == 0 && CH % 16 == 0, "tile geometry");
};

template <int E>
__global__ __launch_bounds__(ScanF32Cfg<E>::SCAN_THREADS, ScanF32Cfg<E>::SCAN_WAVES / 4) void scan_f32_kernel(
    const float *__restrict__ q, const float *__restrict__ gal, int Q, int64_t N, int ntiles, int tpt, int qwaves,
    int qpad, float *__restrict__ bmax, float *__restrict__ tmax)
{
    using C = ScanF32Cfg<E>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int task = blockIdx.x;
    const int t0 = task * tpt;
    const int t1 = min(ntiles, t0 + tpt);
    const bool compute = wave < qwaves;

    float4 bq[C::SSTEPS];     // query (wave*16 + r): floats [16S + 4g, +4)
    {
        const int qrow = wave * 16 + r;
        const bool live = compute && qrow < Q;
        const float *qp = q + (size_t)(live ? qrow : 0) * E + g * 4;
#pragma unroll
        for (int S = 0; S < C::SSTEPS; ++S) {
            const float4 v = *reinterpret_cast<const float4 *>(qp + S * 16);
            bq[S] = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    auto stage = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < C::LPW; ++i) {
            const int instr = wave * C::LPW + i;
            const int p = instr * 64 + lane;
            const int row = p / C::CH;
            const int pos = p % C::CH;
            const int chunk = (pos & ~15) | ((pos ^ row) & 15);
            int64_t grow = (int64_t)tile * TILE_ROWS_F32 + row;
            grow = grow < N ? grow : N - 1;
            glds16(gal + grow * E + chunk * 4, smem + buf * C::TILE_BYTES + instr * 1024);
        }
    };
    const int rowoff = r * C::ROWB;
    float task_max = -INFINITY, pend = -INFINITY;
    int pend_tile = -1;
    constexpr int PD = SCAN_NBUF - 1;
#pragma unroll
    for (int i = 0; i < PD; ++i)
        if (t0 + i < t1) stage(t0 + i, i);
    int cur = 0;
    for (int t = t0; t < t1; ++t) {
        const int younger = min(PD - 1, t1 - 1 - t);
        if (younger >= 2) wait_vmcnt<2 * C::LPW>();
        else if (younger == 1) wait_vmcnt<C::LPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (compute && pend_tile >= 0 && g == 0) bmax[(size_t)pend_tile * qpad + wave * 16 + r] = pend;
        int nxt = cur + PD; nxt = nxt >= SCAN_NBUF ? nxt - SCAN_NBUF : nxt;
        if (t + PD < t1) stage(t + PD, nxt);
        if (compute) {
            const char *tb = smem + cur * C::TILE_BYTES + rowoff;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < C::SSTEPS; ++S) {
                const int chunk = 4 * S + g;
                const int pos = (chunk & ~15) | ((chunk ^ r) & 15);
                const float4 a = *reinterpret_cast<const float4 *>(tb + pos * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[S].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[S].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[S].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[S].w, acc, 0, 0, 0);
            }
            // acc[i] = dot(query r, tile row 4*g + i)
            float m = -INFINITY;
            const int64_t base = (int64_t)t * TILE_ROWS_F32 + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) m = fmaxf(m, base + i < N ? acc[i] : -INFINITY);
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            task_max = fmaxf(task_max, m);
            pend = m;
            pend_tile = t;
        }
        cur = cur + 1 >= SCAN_NBUF ? 0 : cur + 1;
    }
    if (compute && g == 0) {
        if (pend_tile >= 0) bmax[(size_t)pend_tile * qpad + wave * 16 + r] = pend;
        tmax[(size_t)task * qpad + wave * 16 + r] = task_max;
    }
}

// ---------------------------------------------------------------------------------------------
// scan for fp32 galleries at the bf16 MFMA rate: split-bf16.  Every fp32 value x is split into hi = bf16(x) and
// lo = bf16(x - hi) (bf16 keeps 8 significand bits: |x - hi| <= 2^-8 |x|, so x = hi + lo up to 2^-16 |x|) and the dot product
// is accumulated as q_hi.g_hi + q_lo.g_hi + q_hi.g_lo on v_mfma_f32_16x16x32_bf16 -- three MFMAs per 32-deep k-step instead of
// the eight v_mfma_f32_16x16x4_f32 (1/16 of the bf16 rate) scan_f32_kernel needs.  Error of the approximate dot against the
// exact one: the dropped q_lo.g_lo and the two representation residuals, 3 * 2^-16 |q||g| = 4.6e-5 in the worst case (1e-5
// typical), plus fp32 accumulation (a 32-term tree per MFMA, then E/32 chained adds: <= ~30 * 2^-24 = 2e-6): inside the
// certificate's 8e-5 |q||g| margin, and the ranking itself is still done on exact fp64 re-scores.
// Per 16-row tile: the fp32 rows arrive by global_load_lds (ring of NBUF tiles); ALL waves then split the tile ONCE into
// two bf16 images (hi, lo; 24 VALU instructions per 8 values -- done per consuming wave instead, the conversion was 8x
// redundant and the kernel VALU-bound: 0.78 ms for 1M x 512 x 128 queries); the computing waves read their A fragments
// from those images exactly like scan16_kernel (conflict-free ds_read_b128, prefetched with counted lgkmcnt waits).
// Each wave keeps 16 queries resident as hi/lo B fragments.
//   conversion unit (row = id & 15, cg = id >> 4): fp32 chunks 4cg .. 4cg+3 of the row -> bf16 chunks 2cg, 2cg+1 of both
//   images; a 16-lane group works on 16 different rows at one chunk index, so slot = (chunk ^ row) & 15 is a bijection
//   for its reads and its writes.
// ---------------------------------------------------------------------------------------------
template <int E>
struct ScanF32sCfg {
    static constexpr int SCAN_WAVES = E <= 512 ? 8 : 4;    // E = 768: 192 VGPRs of resident queries -> one wave per SIMD
    static constexpr int SCAN_THREADS = SCAN_WAVES * 64;
    static constexpr int QMAX = SCAN_WAVES * 16;
    static constexpr int NBUF = E <= 512 ? 3 : 2;          // fp32 tiles in the ring (E = 768: 48 KiB each)
    static constexpr int CH = E / 4;                       // 16-byte chunks per fp32 row
    static constexpr int ROWB = E * 4;
    static constexpr int TILE_BYTES = TILE_ROWS_F32 * ROWB;
    static constexpr int LOADS = TILE_ROWS_F32 * CH / 64;
    static constexpr int LPW = LOADS / SCAN_WAVES;
    static constexpr int KSTEPS = E / 32;
    static constexpr int CHB = E / 8;                      // 16-byte chunks per bf16 image row
    static constexpr int IMG_BYTES = TILE_ROWS_F32 * E * 2;// one bf16 image (hi or lo)
    static constexpr int UNITS = TILE_ROWS_F32 * (E / 16); // conversion units per tile
    static constexpr int LDS = NBUF * TILE_BYTES + 2 * IMG_BYTES;     // E = 512: 96 + 32 KiB
    static_assert(LOADS % SCAN_WAVES == 0 && CH % 16 == 0 && CHB % 16 == 0, "tile geometry");
};

// 8 fp32 -> hi and lo bf16 fragments (24 VALU instructions)
__device__ __forceinline__ void split_bf16x8(const float4 &a0, const float4 &a1, bf16x8 &hi, bf16x8 &lo)
{
    const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    union { bf16x8 v; uint32_t u[4]; } h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h.u[j] = pack_bf16x2(x[2 * j], x[2 * j + 1]);
        const float r0 = x[2 * j] - __uint_as_float(h.u[j] << 16);
        const float r1 = x[2 * j + 1] - __uint_as_float(h.u[j] & 0xffff0000u);
        l.u[j] = pack_bf16x2(r0, r1);
    }
    hi = h.v;
    lo = l.v;
}

template <int E>
__global__ __launch_bounds__(ScanF32sCfg<E>::SCAN_THREADS, ScanF32sCfg<E>::SCAN_WAVES / 4) void scan_f32s_kernel(
    const float *__restrict__ q, const float *__restrict__ gal, int Q, int64_t N, int ntiles, int tpt, int qwaves,
    int qpad, float *__restrict__ bmax, float *__restrict__ tmax)
{
    using C = ScanF32sCfg<E>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *img = smem + C::NBUF * C::TILE_BYTES;          // [hi | lo] bf16 images of the tile being multiplied
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int task = blockIdx.x;
    const int t0 = task * tpt;
    const int t1 = min(ntiles, t0 + tpt);
    const bool compute = wave < qwaves;

    bf16x8 bqh[C::KSTEPS], bql[C::KSTEPS];     // query (wave*16 + r), elements [32s + 8g, +8), split
    {
        const int qrow = wave * 16 + r;
        const bool live = compute && qrow < Q;
        const float *qp = q + (size_t)(live ? qrow : 0) * E + g * 8;
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            float4 a0 = *reinterpret_cast<const float4 *>(qp + s * 32), a1 = *reinterpret_cast<const float4 *>(qp + s * 32 + 4);
            if (!live) { a0 = make_float4(0.f, 0.f, 0.f, 0.f); a1 = a0; }
            split_bf16x8(a0, a1, bqh[s], bql[s]);
        }
    }
    auto stage = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < C::LPW; ++i) {
            const int instr = wave * C::LPW + i;
            const int p = instr * 64 + lane;
            const int row = p / C::CH;
            const int pos = p % C::CH;
            const int chunk = (pos & ~15) | ((pos ^ row) & 15);
            int64_t grow = (int64_t)tile * TILE_ROWS_F32 + row;
            grow = grow < N ? grow : N - 1;
            glds16(gal + grow * E + chunk * 4, smem + buf * C::TILE_BYTES + instr * 1024);
        }
    };
    float task_max = -INFINITY, pend = -INFINITY;
    int pend_tile = -1;

    // split the fp32 tile in ring slot `slot` into the hi / lo bf16 images `im`, once for the whole workgroup.  The LDS
    // accesses are inline asm: for C++ accesses hipcc orders them behind the LDS-DMA in flight with s_waitcnt vmcnt(0)
    // (seen in the ISA), which would expose one HBM latency per tile.
    auto convert = [&](int slot, int im) {
        typedef float f32x4_raw __attribute__((ext_vector_type(4)));
        const uint32_t tf = (uint32_t)(uintptr_t)(smem + slot * C::TILE_BYTES);
        const uint32_t ih = (uint32_t)(uintptr_t)(img + im * 2 * C::IMG_BYTES), il = ih + C::IMG_BYTES;
#pragma unroll
        for (int u = threadIdx.x; u < C::UNITS; u += C::SCAN_THREADS) {
            const int row = u & 15, cg = u >> 4;
            f32x4_raw x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * cg + j;
                const uint32_t a = tf + row * C::ROWB + (((c & ~15) | ((c ^ row) & 15)) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(x[j]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x8 hi, lo;
                split_bf16x8(make_float4(x[2 * j][0], x[2 * j][1], x[2 * j][2], x[2 * j][3]),
                             make_float4(x[2 * j + 1][0], x[2 * j + 1][1], x[2 * j + 1][2], x[2 * j + 1][3]), hi, lo);
                const int cb = 2 * cg + j;
                const uint32_t off = row * (E * 2) + (((cb & ~15) | ((cb ^ row) & 15)) << 4);
                asm volatile("ds_write_b128 %0, %1" ::"v"(ih + off), "v"(hi) : "memory");
                asm volatile("ds_write_b128 %0, %1" ::"v"(il + off), "v"(lo) : "memory");
            }
        }
    };
    // bucket maximum of tile t from the images `im` (computing waves)
    auto compute_tile = [&](int t, int im) {
        const char *img_hi = img + im * 2 * C::IMG_BYTES, *img_lo = img_hi + C::IMG_BYTES;
        const int rowoff = r * (E * 2);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc3 = (f32x4){0.f, 0.f, 0.f, 0.f};
        // hi / lo A fragments run PF k-steps ahead of the MFMAs through inline-asm reads with counted waits (scan_kernel)
        constexpr int PF = C::KSTEPS < 3 ? C::KSTEPS : 3;     // deeper (6) measured no faster
        bf16x8 fh[PF], fl[PF];
        auto issue = [&](int s, bf16x8 &dh, bf16x8 &dl) {
            const int c = 4 * s + g;
            const int off = rowoff + (((c & ~15) | ((c ^ r) & 15)) << 4);
            const uint32_t ah = (uint32_t)(uintptr_t)(img_hi + off), al = (uint32_t)(uintptr_t)(img_lo + off);
            asm volatile("ds_read_b128 %0, %1" : "=v"(dh) : "v"(ah));
            asm volatile("ds_read_b128 %0, %1" : "=v"(dl) : "v"(al));
        };
#pragma unroll
        for (int s = 0; s < PF && s < C::KSTEPS; ++s) issue(s, fh[s], fl[s]);
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            const int younger_s = (C::KSTEPS - 1 - s) < (PF - 1) ? (C::KSTEPS - 1 - s) : (PF - 1);
            bf16x8 &ah = fh[s % PF], &al = fl[s % PF];
            if (younger_s == 5) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(ah), "+v"(al));
            else if (younger_s == 4) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ah), "+v"(al));
            else if (younger_s == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ah), "+v"(al));
            else if (younger_s == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah), "+v"(al));
            else if (younger_s == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah), "+v"(al));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah), "+v"(al));
            // three accumulation chains (hi.hi, lo.hi, hi.lo): no MFMA waits on the one issued just before it
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bqh[s], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bqh[s], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bql[s], acc3, 0, 0, 0);
            if (s + PF < C::KSTEPS) {
                // the MFMAs above must have READ the fragments before the refill overwrites them
                asm volatile("" : "+v"(acc), "+v"(acc2), "+v"(acc3));
                issue(s + PF, ah, al);
            }
        }
        // acc[i] (+ acc2 + acc3) = dot(query r, tile row 4*g + i)
        float m = -INFINITY;
        const int64_t base = (int64_t)t * TILE_ROWS_F32 + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) m = fmaxf(m, base + i < N ? acc[i] + (acc2[i] + acc3[i]) : -INFINITY);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        task_max = fmaxf(task_max, m);
        pend = m;
        pend_tile = t;
    };
    auto flush_pending = [&]() {
        if (compute && pend_tile >= 0 && g == 0) bmax[(size_t)pend_tile * qpad + wave * 16 + r] = pend;
    };

    {
        // split, barrier, multiply: two barriers per tile.  A second image pair with tile t+1 split while tile t is multiplied
        // (one barrier per tile, also with the two waves of a SIMD taking the two jobs in opposite orders) measured no faster:
        // 0.62 vs 0.60 ms for 1M x 512 x 128 queries -- and needs the whole 160 KiB of LDS at E = 512.
        constexpr int PD = C::NBUF - 1;
#pragma unroll
        for (int i = 0; i < PD; ++i)
            if (t0 + i < t1) stage(t0 + i, i);
        int cur = 0;
        for (int t = t0; t < t1; ++t) {
            const int younger = min(PD - 1, t1 - 1 - t);
            if (younger >= 1) wait_vmcnt<C::LPW>(); else wait_vmcnt<0>();
            // tile t has landed for every wave, and every wave is done reading the bf16 images of tile t-1
            __builtin_amdgcn_s_barrier();
            flush_pending();
            int nxt = cur + PD; nxt = nxt >= C::NBUF ? nxt - C::NBUF : nxt;
            if (t + PD < t1) stage(t + PD, nxt);
            convert(cur, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // the images are complete
            if (compute) compute_tile(t, 0);
            cur = cur + 1 >= C::NBUF ? 0 : cur + 1;
        }
    }
    if (compute && g == 0) {
        if (pend_tile >= 0) bmax[(size_t)pend_tile * qpad + wave * 16 + r] = pend;
        tmax[(size_t)task * qpad + wave * 16 + r] = task_max;
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 galleries that are searched many times (GalleryIndex): the hi / lo split of the gallery is done ONCE, into two bf16
// arrays (together the bytes of the fp32 gallery), and the scan streams those.  scan_f32s_kernel spends its time above the
// HBM stream on the split itself: 24 VALU instructions per 8 values, the images' LDS writes and a second barrier per tile
// (0.60 ms for 1M x 512 x 128 queries against 0.35 ms of HBM time); here a tile arrives by LDS-DMA already as the two
// images scan_f32s builds, there is one barrier per tile, and what is left is the three MFMAs per k-step.  Same split
// function, hence the same MFMA operands, the same bucket maxima and the same candidates; the exact re-score still reads
// the fp32 gallery.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_gallery_kernel(const float *__restrict__ g, int64_t n8, bf16_t *__restrict__ hi,
                                                            bf16_t *__restrict__ lo)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one unit = 8 consecutive values
    if (i >= n8) return;
    const float4 a0 = *reinterpret_cast<const float4 *>(g + i * 8), a1 = *reinterpret_cast<const float4 *>(g + i * 8 + 4);
    bf16x8 h, l;
    split_bf16x8(a0, a1, h, l);
    *reinterpret_cast<bf16x8 *>(hi + i * 8) = h;
    *reinterpret_cast<bf16x8 *>(lo + i * 8) = l;
}

// fp32 queries rounded to bf16 (nearest-even, the gallery split's hi): operands of the first-tier scan.  One wave per query;
// qres[query] = ||q - bf16(q)||_2, rounded up: the query half of that tier's certificate margin.
__global__ __launch_bounds__(256) void queries_to_bf16_kernel(const float *__restrict__ q, int Q, int E, bf16_t *__restrict__ out,
                                                              float *__restrict__ qres)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    const float *p = q + (size_t)row * E;
    float ss = 0.f;
    for (int c = lane; c < E / 8; c += 64) {
        const float4 a0 = *reinterpret_cast<const float4 *>(p + c * 8), a1 = *reinterpret_cast<const float4 *>(p + c * 8 + 4);
        const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        union { bf16x8 v; uint32_t u[4]; } h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h.u[j] = pack_bf16x2(x[2 * j], x[2 * j + 1]);
            const float r0 = x[2 * j] - __uint_as_float(h.u[j] << 16);            // exact: the residual of a rounding
            const float r1 = x[2 * j + 1] - __uint_as_float(h.u[j] & 0xffff0000u);
            ss += r0 * r0 + r1 * r1;
        }
        *reinterpret_cast<bf16x8 *>(out + (size_t)row * E + c * 8) = h.v;
    }
    ss = wave_sum(ss);
    if (lane == 0) qres[row] = sqrtf(ss) * 1.00001f;
}

template <int E>
__global__ __launch_bounds__(ScanF32sCfg<E>::SCAN_THREADS, ScanF32sCfg<E>::SCAN_WAVES / 4) void scan_split_kernel(
    const float *__restrict__ q, const bf16_t *__restrict__ ghi, const bf16_t *__restrict__ glo, int Q, int64_t N, int ntiles,
    int tpt, int qwaves, int qpad, float *__restrict__ bmax, float *__restrict__ tmax, const int32_t *__restrict__ gate)
{
    using C = ScanF32sCfg<E>;
    if (gate) {                              // second tier: nothing to do unless the first tier left one of these queries open
        int open = 0;
        for (int i = threadIdx.x; i < Q; i += blockDim.x) open |= gate[i];
        if (!__syncthreads_or(open)) return;
    }
    constexpr int SNBUF = C::NBUF;          // a fourth ring slot and fragment reads six k-steps ahead both measured no faster
    static_assert(2 * C::IMG_BYTES == C::TILE_BYTES, "a ring slot holds the hi and the lo image of one 16-row tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int task = blockIdx.x;
    const int t0 = task * tpt;
    const int t1 = min(ntiles, t0 + tpt);
    const bool compute = wave < qwaves;

    bf16x8 bqh[C::KSTEPS], bql[C::KSTEPS];     // query (wave*16 + r), elements [32s + 8g, +8), split
    {
        const int qrow = wave * 16 + r;
        const bool live = compute && qrow < Q;
        const float *qp = q + (size_t)(live ? qrow : 0) * E + g * 8;
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            float4 a0 = *reinterpret_cast<const float4 *>(qp + s * 32), a1 = *reinterpret_cast<const float4 *>(qp + s * 32 + 4);
            if (!live) { a0 = make_float4(0.f, 0.f, 0.f, 0.f); a1 = a0; }
            split_bf16x8(a0, a1, bqh[s], bql[s]);
        }
    }
    // one ring slot = [hi image | lo image], each [16 rows][E bf16] with scan_f32s' chunk swizzle, lane-linear for LDS-DMA
    auto stage = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < C::LPW; ++i) {
            const int instr = wave * C::LPW + i;
            const int p = instr * 64 + lane;
            const int im = p / (TILE_ROWS_F32 * C::CHB);          // 0 = hi, 1 = lo (wave-instruction uniform)
            const int pp = p - im * (TILE_ROWS_F32 * C::CHB);
            const int row = pp / C::CHB;
            const int pos = pp % C::CHB;
            const int chunk = (pos & ~15) | ((pos ^ row) & 15);
            int64_t grow = (int64_t)tile * TILE_ROWS_F32 + row;
            grow = grow < N ? grow : N - 1;
            glds16((im ? glo : ghi) + grow * E + chunk * 8, smem + buf * C::TILE_BYTES + instr * 1024);
        }
    };
    float task_max = -INFINITY, pend = -INFINITY;
    int pend_tile = -1;
    auto compute_tile = [&](int t, int slot) {
        const char *img_hi = smem + slot * C::TILE_BYTES, *img_lo = img_hi + C::IMG_BYTES;
        const int rowoff = r * (E * 2);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc3 = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int PF = C::KSTEPS < 3 ? C::KSTEPS : 3;
        bf16x8 fh[PF], fl[PF];
        auto issue = [&](int s, bf16x8 &dh, bf16x8 &dl) {
            const int c = 4 * s + g;
            const int off = rowoff + (((c & ~15) | ((c ^ r) & 15)) << 4);
            const uint32_t ah = (uint32_t)(uintptr_t)(img_hi + off), al = (uint32_t)(uintptr_t)(img_lo + off);
            asm volatile("ds_read_b128 %0, %1" : "=v"(dh) : "v"(ah));
            asm volatile("ds_read_b128 %0, %1" : "=v"(dl) : "v"(al));
        };
#pragma unroll
        for (int s = 0; s < PF && s < C::KSTEPS; ++s) issue(s, fh[s], fl[s]);
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            const int younger_s = (C::KSTEPS - 1 - s) < (PF - 1) ? (C::KSTEPS - 1 - s) : (PF - 1);
            bf16x8 &ah = fh[s % PF], &al = fl[s % PF];
            if (younger_s == 5) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(ah), "+v"(al));
            else if (younger_s == 4) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ah), "+v"(al));
            else if (younger_s == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ah), "+v"(al));
            else if (younger_s == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah), "+v"(al));
            else if (younger_s == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah), "+v"(al));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah), "+v"(al));
            // the same three accumulation chains, in the same order, as scan_f32s_kernel: identical bucket maxima
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bqh[s], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bqh[s], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bql[s], acc3, 0, 0, 0);
            if (s + PF < C::KSTEPS) {
                asm volatile("" : "+v"(acc), "+v"(acc2), "+v"(acc3));
                issue(s + PF, ah, al);
            }
        }
        float m = -INFINITY;
        const int64_t base = (int64_t)t * TILE_ROWS_F32 + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) m = fmaxf(m, base + i < N ? acc[i] + (acc2[i] + acc3[i]) : -INFINITY);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        task_max = fmaxf(task_max, m);
        pend = m;
        pend_tile = t;
    };
    constexpr int PD = SNBUF - 1;
#pragma unroll
    for (int i = 0; i < PD; ++i)
        if (t0 + i < t1) stage(t0 + i, i);
    int cur = 0;
    for (int t = t0; t < t1; ++t) {
        const int younger = min(PD - 1, t1 - 1 - t);
        if (younger >= 2) wait_vmcnt<2 * C::LPW>();
        else if (younger == 1) wait_vmcnt<C::LPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();            // tile t landed for every wave; every wave is done with tile t-1's slot
        if (compute && pend_tile >= 0 && g == 0) bmax[(size_t)pend_tile * qpad + wave * 16 + r] = pend;
        int nxt = cur + PD; nxt = nxt >= SNBUF ? nxt - SNBUF : nxt;
        if (t + PD < t1) stage(t + PD, nxt);
        if (compute) compute_tile(t, cur);
        cur = cur + 1 >= SNBUF ? 0 : cur + 1;
    }
    if (compute && g == 0) {
        if (pend_tile >= 0) bmax[(size_t)pend_tile * qpad + wave * 16 + r] = pend;
        tmax[(size_t)task * qpad + wave * 16 + r] = task_max;
    }
}

// ---------------------------------------------------------------------------------------------
// exact fp64 dot, shared by finalize / exhaustive / similarity
// ---------------------------------------------------------------------------------------------
// A row of E elements is 64 contiguous chunks of PER = E/64 elements.  load_chunk fetches chunk
// `c` (for a full-wave dot, c = lane).
template <int PER>
__device__ __forceinline__ void load_chunk_bf16(const bf16_t *row, int c, float (&out)[PER]) {
    const bf16_t *p = row + c * PER;
    if constexpr (PER % 8 == 0) {
#pragma unroll
        for (int v = 0; v < PER / 8; ++v) {
            bf16x8 x = *reinterpret_cast<const bf16x8 *>(p + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) out[v * 8 + j] = bf16_to_f32((bf16_t)x[j]);
        }
    } else if constexpr (PER % 4 == 0) {
#pragma unroll
        for (int v = 0; v < PER / 4; ++v) {
            uint2 x = *reinterpret_cast<const uint2 *>(p + v * 4);
            out[v * 4 + 0] = __uint_as_float(x.x << 16);
            out[v * 4 + 1] = __uint_as_float(x.x & 0xffff0000u);
            out[v * 4 + 2] = __uint_as_float(x.y << 16);
            out[v * 4 + 3] = __uint_as_float(x.y & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) out[j] = bf16_to_f32(p[j]);
    }
}
template <int PER>
__device__ __forceinline__ void load_chunk_f32(const float *row, int c, float (&out)[PER]) {
    const float *p = row + c * PER;
    if constexpr (PER % 4 == 0) {
#pragma unroll
        for (int v = 0; v < PER / 4; ++v) {
            float4 x = *reinterpret_cast<const float4 *>(p + v * 4);
            out[v * 4 + 0] = x.x; out[v * 4 + 1] = x.y; out[v * 4 + 2] = x.z; out[v * 4 + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) out[j] = p[j];
    }
}
template <typename T, int PER>
__device__ __forceinline__ void load_chunk(const T *row, int c, float (&out)[PER]) {
    if constexpr (sizeof(T) == 2) load_chunk_bf16<PER>((const bf16_t *)row, c, out);
    else load_chunk_f32<PER>((const float *)row, c, out);
}

// One chunk's partial: PER products summed left to right from 0.0 in fp64 (products are exact).
template <int PER>
__device__ __forceinline__ double chunk_partial(const float (&qv)[PER], const float (&gv)[PER]) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) acc += (double)qv[j] * (double)gv[j];
    return acc;
}

// Full-wave form: lane l owns chunk l; the 64 partials meet in an xor-butterfly (32,16,...,1):
// the order oracle/search_ref.c replicates.
template <int PER>
__device__ __forceinline__ double exact_dot(const float (&qv)[PER], const float (&gv)[PER]) {
    return wave_sum_f64_butterfly(chunk_partial<PER>(qv, gv));
}

// Quarter-wave form: 16 lanes own one row; lane m (0..15) owns chunks m, m+16, m+32, m+48.  The
// butterfly's 32- and 16-steps pair exactly those chunks, so they become in-lane adds and only
// the 8,4,2,1 steps cross lanes: bit-identical to exact_dot, four rows per wave pass.
template <typename T, int PER>
struct QuadQuery {
    float v[4][PER];
    __device__ __forceinline__ void load(const T *qrow, int m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_chunk<T, PER>(qrow, m + 16 * i, v[i]);
    }
};
template <typename T, int PER>
struct QuadRow {
    float v[4][PER];
    __device__ __forceinline__ void load(const T *grow, int m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_chunk<T, PER>(grow, m + 16 * i, v[i]);
    }
};
template <typename T, int PER>
__device__ __forceinline__ double quad_dot(const QuadQuery<T, PER> &q, const QuadRow<T, PER> &g) {
    const double p0 = chunk_partial<PER>(q.v[0], g.v[0]);
    const double p1 = chunk_partial<PER>(q.v[1], g.v[1]);
    const double p2 = chunk_partial<PER>(q.v[2], g.v[2]);
    const double p3 = chunk_partial<PER>(q.v[3], g.v[3]);
    double s = (p0 + p2) + (p1 + p3);   // butterfly steps 32 then 16
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
    return s;
}

// ---------------------------------------------------------------------------------------------
// selection: extract the best (value, key) pairs in (-value, +key) order
// ---------------------------------------------------------------------------------------------
#ifndef MMR_SEL_THRESH
#define MMR_SEL_THRESH 1          // -DMMR_SEL_THRESH=0: serial extraction everywhere (A/B build)
#endif
constexpr int32_t KEY_NONE = 0x7fffffff;
constexpr int SEL_R = 16;                       // candidates per lane in the register path
constexpr int SEL_FAST_MAX = SEL_R * FIN_THREADS;  // 4096

template <typename V, typename K>
__device__ __forceinline__ bool before(V sa, K ia, V sb, K ib) { return sa > sb || (sa == sb && ia < ib); }

template <typename V>
struct SelScratch {
    V pv[FIN_THREADS / 64][K_MAX + 1];
    int32_t pk[FIN_THREADS / 64][K_MAX + 1];
    V bv;
    int32_t bk;
};

// ---- branch-free candidate keys.  A candidate (value, key) becomes an unsigned sort key whose MAXIMUM is the
// best candidate in (-value, +key) order: high part = order-preserving bits of the value, low part = ~key.
// Empty slots are all-zero (below every real candidate: real low parts are >= 1 because key < KEY_NONE).
__device__ __forceinline__ uint32_t ord_f32(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ uint64_t ord_f64(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double unord_f64(uint64_t o) {
    return __longlong_as_double((long long)((o >> 63) ? (o & 0x7fffffffffffffffull) : ~o));
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// DPP controls: quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror: four steps that leave the
// maximum of each 16-lane row in all of its lanes (plain VALU, no LDS round trip); rows are then merged
// with two cross-row shuffles.
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

struct Key64 {    // float value
    uint64_t k;
    __device__ __forceinline__ static Key64 make(float v, int32_t key) { return {((uint64_t)ord_f32(v) << 32) | (uint32_t)~(uint32_t)key}; }
    __device__ __forceinline__ static Key64 none() { return {0}; }
    __device__ __forceinline__ bool empty() const { return k == 0; }
    __device__ __forceinline__ void take_max(const Key64 &o) { k = o.k > k ? o.k : k; }
    __device__ __forceinline__ bool same(const Key64 &o) const { return k == o.k; }
    template <int CTRL> __device__ __forceinline__ Key64 via_dpp() const {
        return {((uint64_t)dpp<CTRL>((uint32_t)(k >> 32)) << 32) | dpp<CTRL>((uint32_t)k)};
    }
    __device__ __forceinline__ Key64 via_xor(int off) const { return {(uint64_t)__shfl_xor((unsigned long long)k, off, 64)}; }
    __device__ __forceinline__ float value() const { return unord_f32((uint32_t)(k >> 32)); }
    __device__ __forceinline__ int32_t key() const { return (int32_t)~(uint32_t)k; }
};
struct Key96 {    // double value
    uint64_t hi; uint32_t lo;
    __device__ __forceinline__ static Key96 make(double v, int32_t key) { return {ord_f64(v), (uint32_t)~(uint32_t)key}; }
    __device__ __forceinline__ static Key96 none() { return {0, 0}; }
    __device__ __forceinline__ bool empty() const { return (hi | lo) == 0; }
    __device__ __forceinline__ void take_max(const Key96 &o) {
        const bool t = (o.hi > hi) | ((o.hi == hi) & (o.lo > lo));
        hi = t ? o.hi : hi; lo = t ? o.lo : lo;
    }
    __device__ __forceinline__ bool same(const Key96 &o) const { return (hi == o.hi) & (lo == o.lo); }
    template <int CTRL> __device__ __forceinline__ Key96 via_dpp() const {
        return {((uint64_t)dpp<CTRL>((uint32_t)(hi >> 32)) << 32) | dpp<CTRL>((uint32_t)hi), dpp<CTRL>(lo)};
    }
    __device__ __forceinline__ Key96 via_xor(int off) const {
        return {(uint64_t)__shfl_xor((unsigned long long)hi, off, 64), (uint32_t)__shfl_xor((int)lo, off, 64)};
    }
    __device__ __forceinline__ double value() const { return unord_f64(hi); }
    __device__ __forceinline__ int32_t key() const { return (int32_t)~lo; }
};
template <typename V> struct KeyOf;
template <> struct KeyOf<float> { using type = Key64; };
template <> struct KeyOf<double> { using type = Key96; };

template <typename KT>
__device__ __forceinline__ KT wave_max_key(KT b) {
    b.take_max(b.template via_dpp<DPP_XOR1>());
    b.take_max(b.template via_dpp<DPP_XOR2>());
    b.take_max(b.template via_dpp<DPP_HALF_MIRROR>());
    b.take_max(b.template via_dpp<DPP_MIRROR>());
    b.take_max(b.via_xor(16));
    b.take_max(b.via_xor(32));
    return b;
}

// `rounds` extractions from the R register candidates of each lane of ONE wave; lane 0 records
// them.  Exhausted slots come out as (-inf, KEY_NONE).
template <typename V, int R>
__device__ __forceinline__ void wave_rounds(V (&v)[R], int32_t (&key)[R], int rounds, V *out_v, int32_t *out_k,
                                            int lane) {
    using KT = typename KeyOf<V>::type;
    KT c[R];
#pragma unroll
    for (int j = 0; j < R; ++j) c[j] = key[j] == KEY_NONE ? KT::none() : KT::make(v[j], key[j]);
    for (int r = 0; r < rounds; ++r) {
        KT b = c[0];
#pragma unroll
        for (int j = 1; j < R; ++j) b.take_max(c[j]);
        b = wave_max_key(b);
        if (lane == 0) {
            out_v[r] = b.empty() ? (V)-INFINITY : b.value();
            out_k[r] = b.empty() ? KEY_NONE : b.key();
        }
#pragma unroll
        for (int j = 0; j < R; ++j) c[j] = c[j].same(b) ? KT::none() : c[j];
    }
}

template <typename V, int R, typename F>
__device__ __forceinline__ void wave_select_single(int n, int rounds, F get, V *out_v, int32_t *out_k) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        V v[R];
        int32_t key[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int i = tid + j * 64;
            V tv; int32_t tk;
            const bool ok = get(min(i, n - 1), tv, tk) && i < n && tv == tv;     // see wg_select: loads never sit behind a branch
            v[j] = ok ? tv : (V)-INFINITY; key[j] = ok ? tk : KEY_NONE;
        }
        wave_rounds<V, R>(v, key, rounds, out_v, out_k, tid);
    }
    __syncthreads();
}

// Order-preserving unsigned image of a candidate value (the high part of Key64 / Key96)
template <typename V> struct OrdOf;
template <> struct OrdOf<float> {
    using H = uint32_t;
    static constexpr int BITS = 32;
    __device__ __forceinline__ static H ord(float v) { return ord_f32(v); }
    __device__ __forceinline__ static H lane_value(H v, int l) { return (H)__builtin_amdgcn_readlane((int)v, l); }
    // lanes with v >= c as a wave mask (v_cmp straight into an SGPR pair; __ballot(v >= c) compiles to a select + re-compare)
    __device__ __forceinline__ static uint64_t lanes_ge(H v, H c) { return __builtin_amdgcn_uicmp(v, c, 35 /* ICMP_UGE */); }
};
template <> struct OrdOf<double> {
    using H = uint64_t;
    static constexpr int BITS = 64;
    __device__ __forceinline__ static H ord(double v) { return ord_f64(v); }
    __device__ __forceinline__ static H lane_value(H v, int l) {
        return ((H)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
    }
    __device__ __forceinline__ static uint64_t lanes_ge(H v, H c) { return __builtin_amdgcn_uicmpl(v, c, 35 /* ICMP_UGE */); }
};

// The same selection as wave_select_single (one wave holds all n <= 64*R candidates; same results, slot for slot) without
// its `rounds` serial max-extractions (each a 64-lane reduction of 64/96-bit keys):
//   1. pivot: every lane takes the largest value it holds; the rounds-th largest of those 64 lane maxima (found by
//      counting, v_readlane broadcasts) is a lower bound P of the rounds-th largest candidate overall;
//   2. the candidates >= P -- at least `rounds`, for scattered data a few more -- are compacted into an LDS list;
//   3. every list entry computes its rank by counting the entries that sort before it and the first `rounds` ranks are
//      written out.  Sort key = (ord(value), ~key) as in Key64 / Key96 (keys are unique).
// More than 64 candidates >= P (the large values crowd into few lanes, or fewer than `rounds` lanes hold anything): a
// bitwise binary search finds the exact rounds-th largest key instead (v_cmp + s_bcnt1 per register and bit; as slow as
// the serial extraction, but rare).  In-kernel stamps, 17 rounds over 992 candidates: serial extraction ~9 us, binary
// search 6.9 + 2.5 us, pivot ~2.5 us.  Needs rounds <= 64.
template <typename V, int R, typename F>
__device__ __forceinline__ void wave_select_thresh(int n, int rounds, F get, V *out_v, int32_t *out_k) {
    using O = OrdOf<V>;
    using H = typename O::H;
    __shared__ V cv[64];
    __shared__ int32_t ck[64];
    const int tid = threadIdx.x;
    if (tid < 64) {
        H hi[R];
        uint32_t lo[R];        // ~key; 0 = empty slot (real keys are < KEY_NONE, so their ~key has the top bit set)
        V val[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int i = tid + j * 64;
            V tv; int32_t tk;
            const bool ok = get(min(i, n - 1), tv, tk) && i < n && tv == tv;     // see wg_select: loads never sit behind a branch
            hi[j] = ok ? O::ord(tv) : 0; lo[j] = ok ? ~(uint32_t)tk : 0; val[j] = ok ? tv : (V)-INFINITY;
        }
        // Empty slots have hi == 0; no real value maps to 0 (ord() of a non-NaN is >= 0x007fffff).
        auto count_ge = [&](H c) {
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < R; ++j) cnt += __popcll(O::lanes_ge(hi[j], c));
            return cnt;
        };
        // 1. pivot
        H lm = hi[0];
#pragma unroll
        for (int j = 1; j < R; ++j) lm = hi[j] > lm ? hi[j] : lm;
        int lrank = 0;                      // lane maxima ranked (ties: lower lane first): a permutation of 0..63
        for (int l = 0; l < 64; ++l) {
            const H o = O::lane_value(lm, l);
            lrank += (o > lm || (o == lm && l < tid)) ? 1 : 0;
        }
        const int pl = __ffsll((long long)__ballot(lrank == rounds - 1)) - 1;
        H P = O::lane_value(lm, pl);
        P = P ? P : 1;                      // fewer than `rounds` lanes hold anything: every candidate is at or above the pivot
        // exact cut, only when the pivot lets too many through
        bool exact = false;
        H T = 0;
        uint32_t TL = 0;
        if (count_ge(P) > 64) {
            exact = true;
            // T = the rounds-th largest high part (0 when fewer than `rounds` candidates exist: then all of them are taken)
            for (int bit = O::BITS - 1; bit >= 0; --bit) {
                const H c = T | ((H)1 << bit);
                if (count_ge(c) >= rounds) T = c;
            }
            // equal values straddling the cut: among hi == T keep the `need` largest low parts (= smallest keys)
            auto count = [&](auto pred) {
                int c = 0;
#pragma unroll
                for (int j = 0; j < R; ++j) c += __popcll(__ballot(pred(j)));
                return c;
            };
            if (count([&](int j) { return hi[j] != 0 && hi[j] >= T; }) > rounds) {
                const int need = rounds - count([&](int j) { return hi[j] != 0 && hi[j] > T; });
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t c = TL | (1u << bit);
                    if (count([&](int j) { return hi[j] == T && lo[j] >= c; }) >= need) TL = c;
                }
            }
        }
        // 2. compact the survivors (at most 64) into the list, in any order
        int m = 0;
        const uint64_t below = ((uint64_t)1 << tid) - 1;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const bool take = exact ? (hi[j] != 0 && (hi[j] > T || (hi[j] == T && lo[j] >= TL))) : hi[j] >= P;
            const uint64_t mask = __ballot(take);
            if (take) {
                const int slot = m + __popcll(mask & below);
                cv[slot] = val[j];
                ck[slot] = (int32_t)~lo[j];
            }
            m += __popcll(mask);
        }
        // 3. one survivor per lane; its rank = number of survivors that sort before it.  (LDS executes a wave's accesses
        // in order; the clobber only keeps the compiler from moving the cross-lane reads above the writes.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        V mv = (V)-INFINITY;
        int32_t mk = KEY_NONE;
        if (tid < m) { mv = cv[tid]; mk = ck[tid]; }
        const H mh = tid < m ? O::ord(mv) : 0;
        const uint32_t ml = tid < m ? ~(uint32_t)mk : 0;
        int rank = 0;
        for (int i = 0; i < m; ++i) {
            const H bh = O::lane_value(mh, i);
            const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)ml, i);
            rank += (bh > mh || (bh == mh && bl > ml)) ? 1 : 0;
        }
        if (tid < m && rank < rounds) { out_v[rank] = mv; out_k[rank] = mk; }
        if (tid >= m && tid < rounds) { out_v[tid] = (V)-INFINITY; out_k[tid] = KEY_NONE; }
    }
    __syncthreads();
}

template <typename V, int R, typename F>
__device__ __forceinline__ void wg_select_regs(int n, int rounds, F get, V *out_v, int32_t *out_k, SelScratch<V> *sc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = FIN_THREADS / 64;
    V v[R];
    int32_t key[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int i = tid + j * FIN_THREADS;
        V tv; int32_t tk;
        const bool ok = get(min(i, n - 1), tv, tk) && i < n && tv == tv;         // see wg_select: loads never sit behind a branch
        v[j] = ok ? tv : (V)-INFINITY; key[j] = ok ? tk : KEY_NONE;
    }
    wave_rounds<V, R>(v, key, rounds, sc->pv[wave], sc->pk[wave], lane);
    __syncthreads();
    if (wave == 0) {
        constexpr int R2 = (NW * (K_MAX + 1) + 63) / 64;
        V v2[R2];
        int32_t k2[R2];
#pragma unroll
        for (int j = 0; j < R2; ++j) {
            const int cnd = lane + j * 64;
            const bool ok = cnd < NW * rounds;
            const int p = ok ? cnd / rounds : 0, rr = ok ? cnd % rounds : 0;
            v2[j] = ok ? sc->pv[p][rr] : (V)-INFINITY;
            k2[j] = ok ? sc->pk[p][rr] : KEY_NONE;
        }
        wave_rounds<V, R2>(v2, k2, rounds, out_v, out_k, lane);
    }
    __syncthreads();
}

// Workgroup selection over n >= 1 candidates.  get(i, v, key) -> bool valid; keys unique, < KEY_NONE.
// get() is called for every register slot with an index clamped into [0, n) and must be BRANCH-FREE (select its
// addresses, load unconditionally, return the validity): a load behind a divergent branch makes hipcc wait for it before
// the next slot's branch, i.e. one exposed memory latency per register slot -- 16 x ~1.1 us in select_kernel's second level
// (measured with in-kernel stamps: 20 of the kernel's 29 us).
// n <= 4096: candidates live in registers, each wave extracts its own `rounds` winners, wave 0
// merges the 4 lists (2 barriers in all).  Larger n: one global sweep per round (slow, rare).
// Results land in out_v/out_k (shared memory) and are visible to every thread on return.
template <typename V, typename F>
__device__ void wg_select(int n, int rounds, F get, V *out_v, int32_t *out_k, SelScratch<V> *sc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = FIN_THREADS / 64;
    if (n <= 64 * SEL_R) {
        // n <= 1024 (the usual case): one wave holds every candidate, no merge stage, one barrier
        // more than a few rounds: pivot + rank-by-counting instead of serial extraction (same results)
        const bool thresh = MMR_SEL_THRESH && rounds <= 64 && rounds >= 6;
        if (n <= 64 * 8) {
            if (thresh) wave_select_thresh<V, 8>(n, rounds, get, out_v, out_k);
            else wave_select_single<V, 8>(n, rounds, get, out_v, out_k);
        } else {
            if (thresh) wave_select_thresh<V, SEL_R>(n, rounds, get, out_v, out_k);
            else wave_select_single<V, SEL_R>(n, rounds, get, out_v, out_k);
        }
        return;
    }
    if (n <= SEL_FAST_MAX) {
        wg_select_regs<V, SEL_R>(n, rounds, get, out_v, out_k, sc);
        return;
    }
    V pv = (V)INFINITY;
    int32_t pk = -1;
    for (int r = 0; r < rounds; ++r) {
        V bv = (V)-INFINITY;
        int32_t bk = KEY_NONE;
        for (int i = tid; i < n; i += FIN_THREADS) {
            V x; int32_t kx;
            if (!get(i, x, kx) || !(x == x)) continue;
            if (r > 0 && !before(pv, pk, x, kx)) continue;  // taken in an earlier round
            if (before(x, kx, bv, bk)) { bv = x; bk = kx; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const V ov = __shfl_xor(bv, off, 64);
            const int32_t ok = __shfl_xor(bk, off, 64);
            if (before(ov, ok, bv, bk)) { bv = ov; bk = ok; }
        }
        if (lane == 0) { sc->pv[wave][0] = bv; sc->pk[wave][0] = bk; }
        __syncthreads();
        if (tid == 0) {
            V fv = sc->pv[0][0];
            int32_t fk = sc->pk[0][0];
            for (int w = 1; w < NW; ++w)
                if (before(sc->pv[w][0], sc->pk[w][0], fv, fk)) { fv = sc->pv[w][0]; fk = sc->pk[w][0]; }
            sc->bv = fv; sc->bk = fk;
            out_v[r] = fv; out_k[r] = fk;
        }
        __syncthreads();
        pv = sc->bv; pk = sc->bk;
        if (pk == KEY_NONE) {  // exhausted: the remaining slots are empty
            if (tid == 0) for (int rr = r + 1; rr < rounds; ++rr) { out_v[rr] = (V)-INFINITY; out_k[rr] = KEY_NONE; }
            __syncthreads();
            return;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// finalize, in three launches so the exact re-score runs wide instead of inside one workgroup:
//   select_kernel   grid Q        picks the KS best tasks, then the KS best tiles inside them
//   rescore_kernel  grid (KS, Q)  exact fp64 dots of the 32 rows of one candidate tile
//   rank_kernel     grid Q        (-dot64, +row) top-k of the KS*32 candidates + the certificate
// ---------------------------------------------------------------------------------------------
struct FinMeta { float bound; float qnorm; };   // per query: best excluded approx max, ||q||_2

__global__ __launch_bounds__(FIN_THREADS) void select_kernel(
    int ks, int ntiles, int tpt, int ntasks, int qpad, const float *__restrict__ bmax,
    const float *__restrict__ tmax, int32_t *__restrict__ sel_tiles /*[Q][KS_MAX]*/, FinMeta *__restrict__ meta,
    const int32_t *__restrict__ gate)
{
    __shared__ SelScratch<float> scf;
    __shared__ float sel_v[KS_MAX + 1];
    __shared__ int32_t sel_task[KS_MAX + 1];
    __shared__ int32_t sel_tile[KS_MAX + 1];
    const int qi = blockIdx.x, tid = threadIdx.x;
    if (gate && gate[qi] == 0) return;          // second tier of the fp32 search: only queries the first tier left open

    // level 1: best ks tasks (+1 to learn the best excluded one)
    wg_select<float>(ntasks, ks + 1, [&](int i, float &v, int32_t &key) {
        v = tmax[(size_t)i * qpad + qi]; key = i; return true; }, sel_v, sel_task, &scf);
    const float bound1 = sel_v[ks];  // -inf when no task was left out
    __syncthreads();
    // level 2: best ks tiles among the selected tasks' tiles, ordered by (-max, +tile)
    // candidate i = (selected task i / 64, tile i % 64 of it): tasks hold at most MAX_TPT = 64 tiles, so the index splits
    // with a shift instead of a division by the run-time tile count (16 divisions per lane were ~1 us of this kernel)
    static_assert(MAX_TPT == 64, "candidate index layout");
    wg_select<float>(ks * MAX_TPT, ks + 1, [&](int i, float &v, int32_t &key) {
        const int slot = i >> 6, t = i & 63;
        const int32_t task = sel_task[slot];
        const int32_t tile = (task == KEY_NONE ? 0 : task) * tpt + t;
        const bool ok = task != KEY_NONE && t < tpt && tile < ntiles;
        v = bmax[(size_t)(ok ? tile : 0) * qpad + qi]; key = tile; return ok; }, sel_v, sel_tile, &scf);
    if (tid < ks) sel_tiles[(size_t)qi * KS_MAX + tid] = sel_tile[tid];
    if (tid == 0) meta[qi].bound = fmaxf(bound1, sel_v[ks]);
}

template <typename T, int PER>
__global__ __launch_bounds__(FIN_THREADS) void rescore_kernel(
    const T *__restrict__ q, const T *__restrict__ gal, int64_t N, int tile_rows,
    const int32_t *__restrict__ sel_tiles, double *__restrict__ cand /*[Q][KS_MAX*32]*/, FinMeta *__restrict__ meta,
    const int32_t *__restrict__ gate)
{
    constexpr int E = PER * 64;
    const int slot = blockIdx.x, qi = blockIdx.y;
    if (gate && gate[qi] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, grp = tid >> 4;   // 16 row groups
    const int32_t tile = sel_tiles[(size_t)qi * KS_MAX + slot];
    QuadQuery<T, PER> qq;
    qq.load(q + (size_t)qi * E, m);
    QuadRow<T, PER> gr[2];
    bool live[2];
    const int passes = tile_rows / 16;      // 2 for 32-row (bf16) tiles, 1 for 16-row (fp32) tiles
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t row = (int64_t)tile * tile_rows + u * 16 + grp;
        live[u] = u < passes && tile != KEY_NONE && row < N;
        gr[u].load(gal + (size_t)(live[u] ? row : 0) * E, m);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const double s = quad_dot<T, PER>(qq, gr[u]);
        if (m == 0 && u < passes) cand[((size_t)qi * KS_MAX + slot) * TILE_ROWS + u * 16 + grp] = live[u] ? s : -INFINITY;
    }
    if (slot == 0 && grp == 0) {   // ||q||: sizes the certificate's margin
        double qn2 = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) qn2 += chunk_partial<PER>(qq.v[i], qq.v[i]);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) qn2 += __shfl_xor(qn2, off, 64);
        if (m == 0) meta[qi].qnorm = (float)sqrt(qn2);
    }
}

__global__ __launch_bounds__(FIN_THREADS) void rank_kernel(
    int64_t N, int k, int ks, int tile_rows, const int32_t *__restrict__ sel_tiles, const double *__restrict__ cand,
    const FinMeta *__restrict__ meta, float scale, float eps_rel, float host_bound,
    const float *__restrict__ dev_bound, int32_t *__restrict__ idx,
    float *__restrict__ score, double *__restrict__ dot64, int32_t *__restrict__ status,
    int32_t *__restrict__ need_exact, const int32_t *__restrict__ gate, const float *__restrict__ qres,
    const float *__restrict__ gres_dev, float gres_rel)
{
    __shared__ SelScratch<double> scd;
    __shared__ double out_v[K_MAX];
    __shared__ int32_t out_k[K_MAX];
    const int qi = blockIdx.x, tid = threadIdx.x;
    if (gate && gate[qi] == 0) return;          // (gate aliases need_exact: this workgroup is its only writer)
    const int32_t *st = sel_tiles + (size_t)qi * KS_MAX;
    const double *cs = cand + (size_t)qi * KS_MAX * TILE_ROWS;
    // candidate i = (slot i / tile_rows, row-in-tile i % tile_rows); cand keeps a 32-entry stride per slot
    const int tr_shift = tile_rows == 32 ? 5 : 4;                     // tile_rows is 32 (bf16) or 16 (fp32)
    wg_select<double>(ks * tile_rows, k, [&](int i, double &v, int32_t &key) {
        const int slot = i >> tr_shift, rr = i & (tile_rows - 1);
        const int32_t tile = st[slot];
        const int64_t row = (int64_t)(tile == KEY_NONE ? 0 : tile) * tile_rows + rr;
        key = (int32_t)row; v = cs[slot * TILE_ROWS + rr];
        return tile != KEY_NONE && row < N; }, out_v, out_k, &scd);
    if (tid < k) {
        const size_t o = (size_t)qi * k + tid;
        const bool has = out_k[tid] != KEY_NONE;
        idx[o] = has ? out_k[tid] : -1;
        score[o] = has ? (float)(out_v[tid] * (double)scale) : -INFINITY;
        if (dot64) dot64[o] = has ? out_v[tid] : -INFINITY;
    }
    if (tid == 0) {
        // Certificate: every excluded tile's max (an fp32 MFMA dot) is <= bound; a row of an excluded
        // tile can only displace the k-th pick if its exact dot reaches kth, i.e. if bound + err >= kth.
        // margin = eps_rel * ||q|| * (largest gallery row norm): the caller's bound and/or the one measured on the
        // device (mmr_gallery_norm_bound); when both are given the larger one wins, so an understated caller bound
        // cannot shrink the margin below what the data needs
        const double bound = (double)meta[qi].bound;
        float gnorm = host_bound;
        if (dev_bound) gnorm = fmaxf(gnorm, *dev_bound);
        double eps = (double)eps_rel * (double)gnorm * (double)meta[qi].qnorm;
        if (qres) {
            // first tier of the split fp32 search: the scan multiplied bf16(q) with hi(g).
            // |q.g - qh.gh| <= |q - qh| |g| + |qh| |g - gh|, with |q - qh| measured per query, |g - gh| <= the measured maximum
            // over the gallery's rows (or gres_rel * the norm bound when the caller has none) and |qh| <= (1 + 2^-8) |q|
            const double gres = gres_dev ? (double)*gres_dev : (double)gres_rel * (double)gnorm;
            eps += (double)qres[qi] * (double)gnorm + (double)meta[qi].qnorm * (1.0 + 0x1p-8) * gres;
        }
        const int kk = (int)(N < k ? N : k);
        const bool ok = (bound == -INFINITY) || (out_k[kk - 1] != KEY_NONE && out_v[kk - 1] > bound + eps);
        need_exact[qi] = ok ? 0 : 1;
        if (status) status[qi] = ok ? 0 : 1;
    }
}

// ---------------------------------------------------------------------------------------------
// exhaustive exact path
// ---------------------------------------------------------------------------------------------
struct ExhEntry { double s; int32_t i; int32_t pad; };

// Insert a wave-uniform (s, id) into the lane-distributed sorted list (lane j = j-th best).
__device__ __forceinline__ void list_insert(double &my_s, int32_t &my_i, double s, int32_t id, int lane) {
    const bool before_me = before(s, id, my_s, my_i);
    const double ps = __shfl_up(my_s, 1, 64);
    const int32_t pi = __shfl_up(my_i, 1, 64);
    if (before_me) {
        const bool before_prev = lane > 0 && before(s, id, ps, pi);
        my_s = before_prev ? ps : s;
        my_i = before_prev ? pi : id;
    }
}

template <typename T, int PER>
__global__ __launch_bounds__(256) void exh_scan_kernel(
    const T *__restrict__ q, const T *__restrict__ gal, int64_t N, int K, int nslab, int64_t rows_per_slab,
    const int32_t *__restrict__ need_exact, ExhEntry *__restrict__ partial)
{
    constexpr int E = PER * 64;
    __shared__ ExhEntry lists[4][K_MAX];
    const int qi = blockIdx.y, slab = blockIdx.x;
    if (need_exact && need_exact[qi] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, sub = lane >> 4;

    QuadQuery<T, PER> qq;
    qq.load(q + (size_t)qi * E, m);

    double my_s = -INFINITY;
    int32_t my_i = KEY_NONE;
    const int64_t r0 = (int64_t)slab * rows_per_slab;
    const int64_t r1 = min(N, r0 + rows_per_slab);
    // each wave pass covers 4 consecutive rows (one per 16-lane group); waves interleave by 4 rows
    for (int64_t rb = r0 + wave * 4; rb < r1; rb += 16) {
        const int64_t r = rb + sub;
        const bool live = r < r1;
        QuadRow<T, PER> gr;
        gr.load(gal + (size_t)(live ? r : rb) * E, m);
        const double s = quad_dot<T, PER>(qq, gr);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const double sg = __shfl(s, 16 * g, 64);
            const int64_t rg = rb + g;
            if (rg < r1 && sg == sg) {
                const double ts = __shfl(my_s, K - 1, 64);
                const int32_t ti = __shfl(my_i, K - 1, 64);
                if (before(sg, (int32_t)rg, ts, ti)) list_insert(my_s, my_i, sg, (int32_t)rg, lane);
            }
        }
    }
    lists[wave][lane] = ExhEntry{my_s, my_i, 0};
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; ++w)
            for (int j = 0; j < K; ++j) {
                const ExhEntry e = lists[w][j];
                if (e.i == KEY_NONE) break;
                const double ts = __shfl(my_s, K - 1, 64);
                const int32_t ti = __shfl(my_i, K - 1, 64);
                if (before(e.s, e.i, ts, ti)) list_insert(my_s, my_i, e.s, e.i, lane);
            }
        if (lane < K) partial[((size_t)qi * nslab + slab) * K + lane] = ExhEntry{my_s, my_i, 0};
    }
}

__global__ __launch_bounds__(FIN_THREADS) void exh_merge_kernel(
    const ExhEntry *__restrict__ partial, int K, int k, int nslab, float scale,
    const int32_t *__restrict__ need_exact, int32_t *__restrict__ idx, float *__restrict__ score,
    double *__restrict__ dot64)
{
    __shared__ SelScratch<double> sc;
    __shared__ double out_v[K_MAX];
    __shared__ int32_t out_k[K_MAX];
    const int qi = blockIdx.x;
    if (need_exact && need_exact[qi] == 0) return;
    const ExhEntry *p = partial + (size_t)qi * nslab * K;
    wg_select<double>(nslab * K, k, [&](int i, double &v, int32_t &key) {
        const ExhEntry e = p[i];
        v = e.s; key = e.i; return e.i != KEY_NONE; }, out_v, out_k, &sc);
    const int tid = threadIdx.x;
    if (tid < k) {
        const size_t o = (size_t)qi * k + tid;
        const bool has = out_k[tid] != KEY_NONE;
        idx[o] = has ? out_k[tid] : -1;
        score[o] = has ? (float)(out_v[tid] * (double)scale) : -INFINITY;
        if (dot64) dot64[o] = has ? out_v[tid] : -INFINITY;
    }
}

// ---------------------------------------------------------------------------------------------
// small dense ops
// ---------------------------------------------------------------------------------------------
template <typename T, int PER>
__global__ __launch_bounds__(256) void similarity_kernel(const T *__restrict__ q, const T *__restrict__ gal,
                                                          int Q, int64_t N, float scale, float *__restrict__ out)
{
    constexpr int E = PER * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    if (n >= N) return;
    float gv[PER];
    load_chunk<T, PER>(gal + (size_t)n * E, lane, gv);
    for (int qi = 0; qi < Q; ++qi) {
        float qv[PER];
        load_chunk<T, PER>(q + (size_t)qi * E, lane, qv);
        const double s = exact_dot<PER>(qv, gv);
        if (lane == 0) out[(size_t)qi * N + n] = (float)(s * (double)scale);
    }
}

// Tip-Adapter logits, fused (reference code/main_custom.py:111,124-127, code/utils.py:182-186):
//   clip_logits  = 100 * F @ W                      F[N,E], W^T given as wt[C,E]
//   affinity     = F @ Kc                           Kc^T given as kt[S,E]
//   cache_logits = exp(-(beta - beta*affinity)) @ V * 10        V[S,C]
//   tip_logits   = clip_logits + alpha * cache_logits
// One wave per feature row: the row stays in registers, every key/class row is a coalesced read,
// the [N,S] affinity matrix is never written.  fp32 accumulate.
template <typename T, int PER>
__global__ __launch_bounds__(256) void tip_logits_kernel(const T *__restrict__ f, const T *__restrict__ wt,
                                                          const T *__restrict__ kt, const float *__restrict__ v,
                                                          int64_t N, int C, int S, float alpha, float beta,
                                                          float *__restrict__ tip, float *__restrict__ clip)
{
    constexpr int E = PER * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    if (n >= N) return;
    float fv[PER];
    load_chunk<T, PER>(f + (size_t)n * E, lane, fv);
    float my_clip = 0.f, my_cache = 0.f;      // lane c < C owns class c
    for (int c = 0; c < C; ++c) {
        float xv[PER];
        load_chunk<T, PER>(wt + (size_t)c * E, lane, xv);
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < PER; ++j) p += fv[j] * xv[j];
        p = wave_sum(p);
        if (lane == c) my_clip = 100.f * p;
    }
    for (int s = 0; s < S; ++s) {
        float xv[PER];
        load_chunk<T, PER>(kt + (size_t)s * E, lane, xv);
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < PER; ++j) p += fv[j] * xv[j];
        p = wave_sum(p);
        const float e = __expf(-(beta - beta * p));
        if (lane < C) my_cache += e * v[(size_t)s * C + lane];
    }
    if (lane < C) {
        tip[(size_t)n * C + lane] = my_clip + alpha * (my_cache * 10.f);
        if (clip) clip[(size_t)n * C + lane] = my_clip;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void l2norm_kernel(T *__restrict__ x, int64_t rows, int E)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= rows) return;
    T *p = x + (size_t)r * E;
    float ss = 0.f;
    for (int j = lane; j < E; j += 64) {
        float v;
        if constexpr (sizeof(T) == 2) v = bf16_to_f32(((const bf16_t *)p)[j]); else v = ((const float *)p)[j];
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss);
    for (int j = lane; j < E; j += 64) {
        if constexpr (sizeof(T) == 2) ((bf16_t *)p)[j] = f32_to_bf16(bf16_to_f32(((const bf16_t *)p)[j]) * inv);
        else ((float *)p)[j] = ((const float *)p)[j] * inv;
    }
}

// Largest row L2 norm of a gallery: sizes the certificate's margin (rank_kernel) from the data instead of from a
// caller's promise.  One wave per row, 16-byte loads, grid-stride; each wave keeps its maximum of sum(x^2) and
// lane 0 publishes sqrt(max) with one integer atomicMax (non-negative floats order like their bit patterns).
// fp32 summation error (~E * 2^-24 relative) is far inside the margin's own headroom.  HBM-bound: one pass.
template <typename T>
__global__ __launch_bounds__(256) void rownorm_max_kernel(const T *__restrict__ gal, int64_t N, int E,
                                                           unsigned int *__restrict__ out_bits)
{
    constexpr int VEC = 16 / sizeof(T);        // elements per 16-byte load
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = E / VEC;
    float mx = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < N; r += (int64_t)gridDim.x * 4) {
        const T *p = gal + (size_t)r * E;
        float ss = 0.f;
        for (int c = lane; c < chunks; c += 64) {
            if constexpr (sizeof(T) == 2) {
                const bf16x8 x = *reinterpret_cast<const bf16x8 *>(p + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float v = bf16_to_f32((bf16_t)x[j]); ss += v * v; }
            } else {
                const float4 x = *reinterpret_cast<const float4 *>(p + c * 4);
                ss += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            }
        }
        ss = wave_sum(ss);
        mx = fmaxf(mx, ss);                    // NaN rows are skipped here like they are by the ranking
    }
    if (lane == 0) atomicMax(out_bits, __float_as_uint(sqrtf(mx) * 1.000001f));
}

// Largest row norm of (gallery - hi): the gallery half of the first-tier margin of the split fp32 search.  Same shape as
// rownorm_max_kernel; the differences are exact fp32 values (residuals of a rounding).
__global__ __launch_bounds__(256) void split_resid_max_kernel(const float *__restrict__ gal, const bf16_t *__restrict__ hi, int64_t N,
                                                              int E, unsigned int *__restrict__ out_bits)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < N; r += (int64_t)gridDim.x * 4) {
        const float *p = gal + (size_t)r * E;
        const bf16_t *ph = hi + (size_t)r * E;
        float ss = 0.f;
        for (int c = lane; c < E / 8; c += 64) {
            const float4 a0 = *reinterpret_cast<const float4 *>(p + c * 8), a1 = *reinterpret_cast<const float4 *>(p + c * 8 + 4);
            const bf16x8 h = *reinterpret_cast<const bf16x8 *>(ph + c * 8);
            const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = x[j] - bf16_to_f32((bf16_t)h[j]); ss += d * d; }
        }
        ss = wave_sum(ss);
        mx = fmaxf(mx, ss);
    }
    if (lane == 0) atomicMax(out_bits, __float_as_uint(sqrtf(mx) * 1.00001f));
}

__global__ void fill_empty_kernel(int32_t *idx, float *score, double *dot64, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        idx[i] = -1;
        score[i] = -INFINITY;
        if (dot64) dot64[i] = -INFINITY;
    }
}

// this rank's message for the all-gather: packed[q][j] = (global id = local id + offset, or -1; fp64 dot bits)
__global__ void pack_kernel(const int32_t *__restrict__ idx, const double *__restrict__ dot, int64_t offset, int n,
                            int64_t *__restrict__ packed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int32_t li = idx[i];
        packed[2 * (size_t)i] = li >= 0 ? (int64_t)li + offset : -1;
        packed[2 * (size_t)i + 1] = __double_as_longlong(dot[i]);
    }
}

// parts*k <= 8*64 candidates with int64 global ids: one wave, candidates in registers.
// PACKED: the lists arrive as the gathered messages themselves, [parts][Q][k][2] int64 = (id, dot bits).
template <bool PACKED>
__global__ __launch_bounds__(64) void merge_kernel(const int64_t *__restrict__ idx_parts,
                                                    const double *__restrict__ dot_parts, int parts, int Q, int k,
                                                    float scale, int64_t *__restrict__ idx, float *__restrict__ score,
                                                    double *__restrict__ dot64)
{
    constexpr int R = 16;  // 64 lanes * 16 = 1024 candidates >= MERGE_MAX_PARTS * K_MAX
    const int qi = blockIdx.x, lane = threadIdx.x;
    const int n = parts * k;
    double v[R];
    int64_t key[R];
    constexpr int64_t NONE = 0x7fffffffffffffffLL;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int i = lane + j * 64;
        v[j] = -INFINITY; key[j] = NONE;
        if (i < n) {
            const size_t o = ((size_t)(i / k) * Q + qi) * k + (i % k);
            const int64_t gi = PACKED ? idx_parts[2 * o] : idx_parts[o];
            const double d = PACKED ? __longlong_as_double(idx_parts[2 * o + 1]) : dot_parts[o];
            if (gi >= 0 && d == d) { v[j] = d; key[j] = gi; }
        }
    }
    for (int r = 0; r < k; ++r) {
        double bv = v[0];
        int64_t bk = key[0];
#pragma unroll
        for (int j = 1; j < R; ++j)
            if (before(v[j], key[j], bv, bk)) { bv = v[j]; bk = key[j]; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64);
            const int64_t ok = __shfl_xor(bk, off, 64);
            if (before(ov, ok, bv, bk)) { bv = ov; bk = ok; }
        }
        if (lane == 0) {
            const size_t o = (size_t)qi * k + r;
            const bool has = bk != NONE;
            idx[o] = has ? bk : -1;
            score[o] = has ? (float)(bv * (double)scale) : -INFINITY;
            if (dot64) dot64[o] = has ? bv : -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (key[j] == bk) { v[j] = -INFINITY; key[j] = NONE; }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct SearchPlan {
    int ntiles, tpt, ntasks, nslab, ks, tile_rows, qmax;
    int64_t rows_per_slab;
    bool fast;  // MFMA scan usable
    size_t off_bmax, off_tmax, off_flags, off_partial, off_seltiles, off_cand, off_meta, off_nb, off_qb, off_qres, total;
};

static bool scan_supports_E(int E) { return E == 128 || E == 256 || E == 512 || E == 768; }
static int scan_qmax(int E, mmr_dtype dt) { return (E <= 512 ? 256 : 128) / (dt == MMR_F32 ? 2 : 1); }
static bool exact_supports_E(int E) { return E == 128 || E == 256 || E == 512 || E == 768 || E == 1024; }

static SearchPlan make_plan(int64_t N, int E, int Q, int k, mmr_dtype dt)
{
    SearchPlan p{};
    p.tile_rows = dt == MMR_F32 ? TILE_ROWS_F32 : TILE_ROWS;
    p.qmax = scan_qmax(E, dt);
    p.ntiles = (int)((N + p.tile_rows - 1) / p.tile_rows);
    if (p.ntiles <= 256) p.tpt = 1;
    else {
        const int m = (p.ntiles + 256 * MAX_TPT - 1) / (256 * MAX_TPT);
        p.tpt = (p.ntiles + 256 * m - 1) / (256 * m);
    }
    // test hook: force the tiles-per-task count (1..64) to reach the large-task-count selection paths
    // with a small gallery
    static const int force_tpt = getenv("MMR_SEARCH_TPT") ? atoi(getenv("MMR_SEARCH_TPT")) : 0;
    if (force_tpt >= 1 && force_tpt <= MAX_TPT) p.tpt = force_tpt;
    p.ntasks = (p.ntiles + p.tpt - 1) / p.tpt;
    p.ks = k + 6 > KS_MAX ? KS_MAX : k + 6;
    p.fast = scan_supports_E(E) && k + 6 <= KS_MAX && N > 0;
    // exhaustive path: 64 row slabs per query when it is THE path; 16 when it only backs up the fast path
    // (it is launched unconditionally there and unflagged queries exit at once: fewer idle workgroups)
    int nslab = (int)((N + 2047) / 2048);
    const int slab_cap = p.fast ? 16 : 64;
    p.nslab = nslab < 1 ? 1 : (nslab > slab_cap ? slab_cap : nslab);
    p.rows_per_slab = (N + p.nslab - 1) / p.nslab;
    const int qc = Q < p.qmax ? (Q + 31) / 32 * 32 : p.qmax;
    size_t off = 0;
    // regions sized by Q and E alone come first: the tiered fp32 search runs a bf16-plan pass and an fp32-plan pass over one
    // workspace, and both must find the flags and the bf16 copy of the queries at the same place
    p.off_flags = off; off += align_up((size_t)(Q > 0 ? Q : 1) * sizeof(int32_t), 256);
    p.off_qb = off; off += align_up((size_t)(Q > 0 ? Q : 1) * E * sizeof(bf16_t), 256);
    p.off_qres = off; off += align_up((size_t)(Q > 0 ? Q : 1) * sizeof(float), 256);
    p.off_bmax = off; off += align_up((size_t)p.ntiles * qc * sizeof(float), 256);
    p.off_tmax = off; off += align_up((size_t)p.ntasks * qc * sizeof(float), 256);
    p.off_partial = off; off += align_up((size_t)(Q > 0 ? Q : 1) * p.nslab * K_MAX * sizeof(ExhEntry), 256);
    p.off_seltiles = off; off += align_up((size_t)qc * KS_MAX * sizeof(int32_t), 256);
    p.off_cand = off; off += align_up((size_t)qc * KS_MAX * TILE_ROWS * sizeof(double), 256);
    p.off_meta = off; off += align_up((size_t)qc * sizeof(FinMeta), 256);
    p.off_nb = off; off += 256;            // measured gallery norm bound (one float) when the caller gives none
    p.total = off;
    return p;
}

template <int E>
static int launch_scan(const bf16_t *q, const bf16_t *gal, int Qc, int64_t N, const SearchPlan &p, int qpad,
                       float *bmax, float *tmax, hipStream_t st)
{
    ProfScope prof(MMR_PROF_SCAN, st);
    using C = ScanCfg<E>;
    const int lds = SCAN_NBUF * C::TILE_BYTES;
    static DeviceOnce once;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_kernel<E>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    const int qwaves = qpad / 32;
    hipLaunchKernelGGL(scan_kernel<E>, dim3(p.ntasks), dim3(C::SCAN_THREADS), lds, st, q, gal, Qc, N, p.ntiles, p.tpt,
                       qwaves, qpad, bmax, tmax);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

template <int E>
static int launch_scan16(const bf16_t *q, const bf16_t *gal, int Qc, int64_t N, const SearchPlan &p, int qpad,
                         float *bmax, float *tmax, hipStream_t st)
{
    ProfScope prof(MMR_PROF_SCAN, st);
    using C = Scan16Cfg<E>;
    const int lds = SCAN_NBUF * C::TILE_BYTES;
    static DeviceOnce once;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan16_kernel<E>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    hipLaunchKernelGGL(scan16_kernel<E>, dim3(p.ntasks), dim3(C::SCAN_THREADS), lds, st, q, gal, Qc, N, p.ntiles, p.tpt,
                       qpad / 16, qpad, bmax, tmax);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

template <int E>
static int launch_scan_f32(const float *q, const float *gal, int Qc, int64_t N, const SearchPlan &p, int qpad,
                           float *bmax, float *tmax, hipStream_t st)
{
    ProfScope prof(MMR_PROF_SCAN, st);
    using C = ScanF32Cfg<E>;
    const int lds = SCAN_NBUF * C::TILE_BYTES;
    static DeviceOnce once;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_f32_kernel<E>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    hipLaunchKernelGGL(scan_f32_kernel<E>, dim3(p.ntasks), dim3(C::SCAN_THREADS), lds, st, q, gal, Qc, N, p.ntiles, p.tpt,
                       qpad / 16, qpad, bmax, tmax);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

template <int E>
static int launch_scan_f32s(const float *q, const float *gal, int Qc, int64_t N, const SearchPlan &p, int qpad,
                            float *bmax, float *tmax, hipStream_t st)
{
    ProfScope prof(MMR_PROF_SCAN, st);
    using C = ScanF32sCfg<E>;
    const int lds = C::LDS;
    static DeviceOnce once;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_f32s_kernel<E>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    hipLaunchKernelGGL(scan_f32s_kernel<E>, dim3(p.ntasks), dim3(C::SCAN_THREADS), lds, st, q, gal, Qc, N, p.ntiles, p.tpt,
                       qpad / 16, qpad, bmax, tmax);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

template <int E>
static int launch_scan_split(const float *q, const bf16_t *ghi, const bf16_t *glo, int Qc, int64_t N, const SearchPlan &p, int qpad,
                             float *bmax, float *tmax, hipStream_t st, const int32_t *gate = nullptr)
{
    ProfScope prof(MMR_PROF_SCAN, st);
    using C = ScanF32sCfg<E>;
    const int lds = C::NBUF * C::TILE_BYTES;
    static DeviceOnce once;
    if (once.first()) {
        MMR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_split_kernel<E>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    hipLaunchKernelGGL(scan_split_kernel<E>, dim3(p.ntasks), dim3(C::SCAN_THREADS), lds, st, q, ghi, glo, Qc, N, p.ntiles, p.tpt,
                       qpad / 16, qpad, bmax, tmax, gate);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

// bf16 scan of one query chunk (the E = 768 form is a run-time choice)
static int launch_scan_bf16(int E, const bf16_t *q, const bf16_t *gal, int Qc, int64_t N, const SearchPlan &p, int qpad,
                            float *bmax, float *tmax, hipStream_t st)
{
    switch (E) {
        case 128: return launch_scan<128>(q, gal, Qc, N, p, qpad, bmax, tmax, st);
        case 256: return launch_scan<256>(q, gal, Qc, N, p, qpad, bmax, tmax, st);
        case 512: return launch_scan<512>(q, gal, Qc, N, p, qpad, bmax, tmax, st);
        default: {
            // MMR_SCAN768=32 keeps the one-wave-per-SIMD 32x32 form for A/B comparisons
            static const int form = getenv("MMR_SCAN768") ? atoi(getenv("MMR_SCAN768")) : 16;
            return form == 32 ? launch_scan<768>(q, gal, Qc, N, p, qpad, bmax, tmax, st)
                              : launch_scan16<768>(q, gal, Qc, N, p, qpad, bmax, tmax, st);
        }
    }
}

template <typename T, int PER>
static int launch_finalize(const T *q, const T *gal, int Qc, int64_t N, int k, const SearchPlan &p, int qpad,
                           const float *bmax, const float *tmax, float scale, float eps_rel, float host_bound,
                           const float *dev_bound, int32_t *idx,
                           float *score, double *dot64, int32_t *status, int32_t *flags, int32_t *sel_tiles,
                           double *cand, FinMeta *meta, hipStream_t st, const int32_t *gate = nullptr,
                           const float *qres = nullptr, const float *gres_dev = nullptr, float gres_rel = 0.f)
{
    ProfScope prof(MMR_PROF_FINALIZE, st);
    hipLaunchKernelGGL(select_kernel, dim3(Qc), dim3(FIN_THREADS), 0, st, p.ks, p.ntiles, p.tpt, p.ntasks, qpad, bmax,
                       tmax, sel_tiles, meta, gate);
    MMR_CHECK_LAUNCH();
    hipLaunchKernelGGL((rescore_kernel<T, PER>), dim3(p.ks, Qc), dim3(FIN_THREADS), 0, st, q, gal, N, p.tile_rows, sel_tiles,
                       cand, meta, gate);
    MMR_CHECK_LAUNCH();
    hipLaunchKernelGGL(rank_kernel, dim3(Qc), dim3(FIN_THREADS), 0, st, N, k, p.ks, p.tile_rows, sel_tiles, cand, meta, scale,
                       eps_rel, host_bound, dev_bound,
                       idx, score, dot64, status, flags, gate, qres, gres_dev, gres_rel);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

template <typename T, int PER>
static int launch_exh(const T *q, const T *gal, int Q, int64_t N, int k, const SearchPlan &p, float scale,
                      const int32_t *flags, ExhEntry *partial, int32_t *idx, float *score, double *dot64,
                      hipStream_t st)
{
    ProfScope prof(MMR_PROF_EXACT, st);
    hipLaunchKernelGGL((exh_scan_kernel<T, PER>), dim3(p.nslab, Q), dim3(256), 0, st, q, gal, N, k, p.nslab,
                       p.rows_per_slab, flags, partial);
    MMR_CHECK_LAUNCH();
    hipLaunchKernelGGL(exh_merge_kernel, dim3(Q), dim3(FIN_THREADS), 0, st, partial, k, k, p.nslab, scale, flags, idx,
                       score, dot64);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

#define MMR_DISPATCH_PER(E, T, ...)                                                    \
    switch (E) {                                                                       \
        case 128: { constexpr int PER = 2; __VA_ARGS__; } break;                              \
        case 256: { constexpr int PER = 4; __VA_ARGS__; } break;                              \
        case 512: { constexpr int PER = 8; __VA_ARGS__; } break;                              \
        case 768: { constexpr int PER = 12; __VA_ARGS__; } break;                             \
        case 1024: { constexpr int PER = 16; __VA_ARGS__; } break;                            \
        default: mmr::set_error("E=%d unsupported (128,256,512,768,1024)", E); return MMR_ENOTSUP; \
    }

}  // namespace mmr

using namespace mmr;

extern "C" size_t mmr_search_workspace_bytes(int64_t N, int E, int Q, int k)
{
    if (N < 0 || Q < 0 || k < 1) return 0;
    const size_t a = make_plan(N, E, Q, k, MMR_BF16).total, b = make_plan(N, E, Q, k, MMR_F32).total;
    return a > b ? a : b;
}

static int launch_norm_bound(const void *gallery, mmr_dtype dtype, int64_t N, int E, float *out, hipStream_t st)
{
    MMR_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), st));
    if (N == 0) return MMR_OK;
    const int64_t want = (N + 3) / 4;
    const dim3 grid((unsigned)(want < 4096 ? want : 4096));
    if (dtype == MMR_BF16) hipLaunchKernelGGL(rownorm_max_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)gallery, N, E, (unsigned int *)out);
    else hipLaunchKernelGGL(rownorm_max_kernel<float>, grid, dim3(256), 0, st, (const float *)gallery, N, E, (unsigned int *)out);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

extern "C" int mmr_gallery_norm_bound(const void *gallery, mmr_dtype dtype, int64_t N, int E, float *bound_out, void *stream)
{
    MMR_CHECK_ARG(dtype == MMR_F32 || dtype == MMR_BF16, "mmr_gallery_norm_bound: dtype %d", (int)dtype);
    MMR_CHECK_ARG(N >= 0 && E >= 8 && E % 8 == 0, "mmr_gallery_norm_bound: bad shape N=%lld E=%d (E must be a multiple of 8)", (long long)N, E);
    MMR_CHECK_ARG(bound_out && (gallery || N == 0), "mmr_gallery_norm_bound: null pointer");
    MMR_CHECK_ARG(((uintptr_t)gallery & 15) == 0, "mmr_gallery_norm_bound: gallery must be 16-byte aligned");
    return launch_norm_bound(gallery, dtype, N, E, bound_out, (hipStream_t)stream);
}

static int cosine_topk_impl(const void *q, const void *gallery, mmr_dtype dtype, int Q, int64_t N, int E, int k,
                            float scale, float gallery_norm_bound, const float *norm_bound_dev, int32_t *idx, float *score,
                            double *dot64, int32_t *status, void *workspace, size_t workspace_bytes, void *stream,
                            const bf16_t *split_hi = nullptr, const bf16_t *split_lo = nullptr,
                            const float *split_resid_dev = nullptr)
{
    MMR_CHECK_ARG(dtype == MMR_F32 || dtype == MMR_BF16, "mmr_cosine_topk: dtype %d", (int)dtype);
    MMR_CHECK_ARG(Q >= 0 && N >= 0, "mmr_cosine_topk: negative size Q=%d N=%lld", Q, (long long)N);
    MMR_CHECK_ARG(N < 0x7fffffff, "mmr_cosine_topk: N=%lld exceeds int32 row ids (shard the gallery)", (long long)N);
    MMR_CHECK_ARG(k >= 1 && k <= K_MAX, "mmr_cosine_topk: k=%d outside [1,%d]", k, K_MAX);
    MMR_CHECK_ARG(scale > 0.f, "mmr_cosine_topk: scale must be > 0 (got %g)", (double)scale);
    MMR_CHECK_ARG(gallery_norm_bound == gallery_norm_bound && gallery_norm_bound < INFINITY, "mmr_cosine_topk: gallery_norm_bound must be finite");
    if (!exact_supports_E(E)) { set_error("mmr_cosine_topk: E=%d unsupported (128,256,512,768,1024)", E); return MMR_ENOTSUP; }
    if (Q == 0) return MMR_OK;
    MMR_CHECK_ARG(q && idx && score && (gallery || N == 0), "mmr_cosine_topk: null pointer");
    MMR_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)gallery & 15) == 0, "mmr_cosine_topk: q/gallery must be 16-byte aligned");
    const SearchPlan p = make_plan(N, E, Q, k, dtype);
    MMR_CHECK_ARG(workspace != nullptr, "mmr_cosine_topk: null workspace");
    if (workspace_bytes < p.total) { set_error("mmr_cosine_topk: workspace %zu < required %zu", workspace_bytes, p.total); return MMR_ENOSPC; }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *bmax = (float *)(ws + p.off_bmax);
    float *tmax = (float *)(ws + p.off_tmax);
    int32_t *flags = (int32_t *)(ws + p.off_flags);
    ExhEntry *partial = (ExhEntry *)(ws + p.off_partial);
    const size_t esz = dtype == MMR_BF16 ? 2 : 4;

    if (N == 0) {  // nothing to rank: every slot is empty (idx -1, score -inf)
        hipLaunchKernelGGL(fill_empty_kernel, dim3((Q * k + 255) / 256), dim3(256), 0, st, idx, score, dot64, Q * k);
        MMR_CHECK_LAUNCH();
        if (status) MMR_CHECK_HIP(hipMemsetAsync(status, 0, (size_t)Q * sizeof(int32_t), st));
        return MMR_OK;
    }

    if (p.fast) {
        // fp32 MFMA accumulation error of a length-E dot is <= ~E*2^-24*|q||g| (bf16 x bf16 products are exact
        // in fp32; fp32 x fp32 products add one rounding each, same order); the margin below is that worst case
        // for E<=1024 with headroom.  It gates the fast path only.
        const float eps_rel = 8e-5f;
        float host_bound = gallery_norm_bound > 0.f ? gallery_norm_bound : 0.f;
        const float *dev_bound = norm_bound_dev;
        if (host_bound == 0.f && !dev_bound) {
            // no bound from the caller: measure it (one extra pass over the gallery; GalleryIndex-style callers
            // measure once with mmr_gallery_norm_bound and pass the device scalar instead)
            float *nb = (float *)(ws + p.off_nb);
            int rcn = launch_norm_bound(gallery, dtype, N, E, nb, st);
            if (rcn != MMR_OK) return rcn;
            dev_bound = nb;
        }
        // MMR_SCAN_F32=exact keeps the fp32-MFMA scan (exact fma chain, 1/16 of the bf16 rate) for A/B comparisons
        static const bool exact_f32 = getenv("MMR_SCAN_F32") && !strcmp(getenv("MMR_SCAN_F32"), "exact");
        // MMR_SPLIT_TIERS=0: split galleries go straight to the three-product scan (A/B and tests of that tier alone)
        static const bool tiers = !(getenv("MMR_SPLIT_TIERS") && atoi(getenv("MMR_SPLIT_TIERS")) == 0);
        const bool split = dtype == MMR_F32 && split_hi && split_lo && !exact_f32;
        const int32_t *gate = nullptr;
        if (split && tiers) {
            // First tier of the split fp32 search: the bf16 scan over the hi array alone (half the bytes, one MFMA product
            // instead of three) with bf16-rounded queries.  Its scores are off by up to |q - qh||g| + |qh||g - gh| (~1e-3 |q||g|
            // against the three-product scan's 1e-5), so its certificate adds exactly that -- measured per query and over the
            // gallery, see rank_kernel -- to the margin and keeps KS_MAX candidate tiles instead of k + 6; the fp64 re-score
            // reads the fp32 rows as always.  A query it cannot certify keeps its flag and goes through the second tier
            // below -- the three-product scan with the tight margin -- and only then to the exhaustive path.  The second
            // tier's launches return at once when no flag is set.
            SearchPlan p1 = make_plan(N, E, Q, k, MMR_BF16);
            // candidate tiles of this tier; MMR_SPLIT_KS1 = 24 measured the same time on random unit rows and certifies less often
            static const int ks1 = getenv("MMR_SPLIT_KS1") ? atoi(getenv("MMR_SPLIT_KS1")) : KS_MAX;
            if (ks1 > p1.ks && ks1 <= KS_MAX) p1.ks = ks1;
            if (workspace_bytes < p1.total) { set_error("mmr_cosine_topk: workspace %zu < required %zu", workspace_bytes, p1.total); return MMR_ENOSPC; }
            bf16_t *qb = (bf16_t *)(ws + p1.off_qb);
            float *qres = (float *)(ws + p1.off_qres);
            hipLaunchKernelGGL(queries_to_bf16_kernel, dim3((Q + 3) / 4), dim3(256), 0, st, (const float *)q, Q, E, qb, qres);
            MMR_CHECK_LAUNCH();
            float *bmax1 = (float *)(ws + p1.off_bmax), *tmax1 = (float *)(ws + p1.off_tmax);
            for (int q0 = 0; q0 < Q; q0 += p1.qmax) {
                const int Qc = (Q - q0) < p1.qmax ? (Q - q0) : p1.qmax;
                const int qpad = (Qc + 31) / 32 * 32;
                int rc = launch_scan_bf16(E, qb + (size_t)q0 * E, split_hi, Qc, N, p1, qpad, bmax1, tmax1, st);
                if (rc != MMR_OK) return rc;
                MMR_DISPATCH_PER(E, float, {
                    rc = launch_finalize<float, PER>((const float *)q + (size_t)q0 * E, (const float *)gallery, Qc, N, k, p1, qpad,
                                                     bmax1, tmax1, scale, eps_rel, host_bound, dev_bound, idx + (size_t)q0 * k,
                                                     score + (size_t)q0 * k, dot64 ? dot64 + (size_t)q0 * k : nullptr,
                                                     status ? status + q0 : nullptr, flags + q0,
                                                     (int32_t *)(ws + p1.off_seltiles), (double *)(ws + p1.off_cand),
                                                     (FinMeta *)(ws + p1.off_meta), st, nullptr, qres + q0, split_resid_dev,
                                                     0x1p-8f);
                });
                if (rc != MMR_OK) return rc;
            }
            gate = flags;
        }
        const int qmax = p.qmax;
        for (int q0 = 0; q0 < Q; q0 += qmax) {
            const int Qc = (Q - q0) < qmax ? (Q - q0) : qmax;
            const int qpad = (Qc + 31) / 32 * 32;
            const char *qc = (const char *)q + (size_t)q0 * E * esz;
            const int32_t *gq = gate ? gate + q0 : nullptr;
            int rc;
            if (dtype == MMR_BF16) {
                rc = launch_scan_bf16(E, (const bf16_t *)qc, (const bf16_t *)gallery, Qc, N, p, qpad, bmax, tmax, st);
            } else {
                if (split) {        // the caller holds the gallery's hi / lo split (mmr_gallery_split_bf16)
                    switch (E) {
                        case 128: rc = launch_scan_split<128>((const float *)qc, split_hi, split_lo, Qc, N, p, qpad, bmax, tmax, st, gq); break;
                        case 256: rc = launch_scan_split<256>((const float *)qc, split_hi, split_lo, Qc, N, p, qpad, bmax, tmax, st, gq); break;
                        case 512: rc = launch_scan_split<512>((const float *)qc, split_hi, split_lo, Qc, N, p, qpad, bmax, tmax, st, gq); break;
                        default: rc = launch_scan_split<768>((const float *)qc, split_hi, split_lo, Qc, N, p, qpad, bmax, tmax, st, gq); break;
                    }
                } else if (exact_f32) {
                    switch (E) {
                        case 128: rc = launch_scan_f32<128>((const float *)qc, (const float *)gallery, Qc, N, p, qpad, bmax, tmax, st); break;
                        case 256: rc = launch_scan_f32<256>((const float *)qc, (const float *)gallery, Qc, N, p, qpad, bmax, tmax, st); break;
                        case 512: rc = launch_scan_f32<512>((const float *)qc, (const float *)gallery, Qc, N, p, qpad, bmax, tmax, st); break;
                        default: rc = launch_scan_f32<768>((const float *)qc, (const float *)gallery, Qc, N, p, qpad, bmax, tmax, st); break;
                    }
                } else {
                    switch (E) {
                        case 128: rc = launch_scan_f32s<128>((const float *)qc, (const float *)gallery, Qc, N, p, qpad, bmax, tmax, st); break;
                        case 256: rc = launch_scan_f32s<256>((const float *)qc, (const float *)gallery, Qc, N, p, qpad, bmax, tmax, st); break;
                        case 512: rc = launch_scan_f32s<512>((const float *)qc, (const float *)gallery, Qc, N, p, qpad, bmax, tmax, st); break;
                        default: rc = launch_scan_f32s<768>((const float *)qc, (const float *)gallery, Qc, N, p, qpad, bmax, tmax, st); break;
                    }
                }
            }
            if (rc != MMR_OK) return rc;
            int32_t *o_idx = idx + (size_t)q0 * k;
            float *o_score = score + (size_t)q0 * k;
            double *o_dot = dot64 ? dot64 + (size_t)q0 * k : nullptr;
            int32_t *o_status = status ? status + q0 : nullptr;
            if (dtype == MMR_BF16) {
                MMR_DISPATCH_PER(E, bf16_t, {
                    rc = launch_finalize<bf16_t, PER>((const bf16_t *)qc, (const bf16_t *)gallery, Qc, N, k, p, qpad, bmax,
                                                      tmax, scale, eps_rel, host_bound, dev_bound, o_idx, o_score, o_dot, o_status, flags + q0,
                                                      (int32_t *)(ws + p.off_seltiles), (double *)(ws + p.off_cand),
                                                      (FinMeta *)(ws + p.off_meta), st);
                });
            } else {
                MMR_DISPATCH_PER(E, float, {
                    rc = launch_finalize<float, PER>((const float *)qc, (const float *)gallery, Qc, N, k, p, qpad, bmax, tmax,
                                                     scale, eps_rel, host_bound, dev_bound, o_idx, o_score, o_dot, o_status, flags + q0,
                                                     (int32_t *)(ws + p.off_seltiles), (double *)(ws + p.off_cand),
                                                     (FinMeta *)(ws + p.off_meta), st, gq);
                });
            }
            if (rc != MMR_OK) return rc;
        }
        int rc;
        if (dtype == MMR_BF16) {
            MMR_DISPATCH_PER(E, bf16_t, {
                rc = launch_exh<bf16_t, PER>((const bf16_t *)q, (const bf16_t *)gallery, Q, N, k, p, scale, flags, partial,
                                             idx, score, dot64, st);
            });
        } else {
            MMR_DISPATCH_PER(E, float, {
                rc = launch_exh<float, PER>((const float *)q, (const float *)gallery, Q, N, k, p, scale, flags, partial,
                                            idx, score, dot64, st);
            });
        }
        return rc;
    }

    if (status) {
        // 1 = exhaustive path for every query
        MMR_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)status, 1, (size_t)Q, st));
    }
    int rc;
    if (esz == 2) {
        MMR_DISPATCH_PER(E, bf16_t, {
            rc = launch_exh<bf16_t, PER>((const bf16_t *)q, (const bf16_t *)gallery, Q, N, k, p, scale, nullptr, partial,
                                         idx, score, dot64, st);
        });
    } else {
        MMR_DISPATCH_PER(E, float, {
            rc = launch_exh<float, PER>((const float *)q, (const float *)gallery, Q, N, k, p, scale, nullptr, partial,
                                        idx, score, dot64, st);
        });
    }
    return rc;
}

extern "C" int mmr_cosine_topk(const void *q, const void *gallery, mmr_dtype dtype, int Q, int64_t N, int E, int k,
                               float scale, float gallery_norm_bound, int32_t *idx, float *score, double *dot64,
                               int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    return cosine_topk_impl(q, gallery, dtype, Q, N, E, k, scale, gallery_norm_bound, nullptr, idx, score, dot64, status,
                            workspace, workspace_bytes, stream);
}

extern "C" int mmr_cosine_topk_ex(const void *q, const void *gallery, mmr_dtype dtype, int Q, int64_t N, int E, int k,
                                  float scale, float gallery_norm_bound, const float *gallery_norm_bound_dev, int32_t *idx,
                                  float *score, double *dot64, int32_t *status, void *workspace, size_t workspace_bytes,
                                  void *stream)
{
    return cosine_topk_impl(q, gallery, dtype, Q, N, E, k, scale, gallery_norm_bound, gallery_norm_bound_dev, idx, score,
                            dot64, status, workspace, workspace_bytes, stream);
}

extern "C" int mmr_gallery_split_bf16(const float *gallery, int64_t N, int E, void *hi, void *lo, float *resid_bound_out,
                                      void *stream)
{
    MMR_CHECK_ARG(N >= 0 && E >= 8 && E % 8 == 0, "mmr_gallery_split_bf16: bad shape N=%lld E=%d", (long long)N, E);
    if (N == 0) return MMR_OK;
    MMR_CHECK_ARG(gallery && hi && lo, "mmr_gallery_split_bf16: null pointer");
    MMR_CHECK_ARG((((uintptr_t)gallery | (uintptr_t)hi | (uintptr_t)lo) & 15) == 0, "mmr_gallery_split_bf16: pointers must be 16-byte aligned");
    const int64_t n8 = N * (E / 8);
    MMR_CHECK_ARG((n8 + 255) / 256 < 0x7fffffff, "mmr_gallery_split_bf16: gallery too large for one launch");
    hipLaunchKernelGGL(split_gallery_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gallery, n8,
                       (bf16_t *)hi, (bf16_t *)lo);
    MMR_CHECK_LAUNCH();
    if (resid_bound_out) {
        MMR_CHECK_HIP(hipMemsetAsync(resid_bound_out, 0, sizeof(float), (hipStream_t)stream));
        const int64_t want = (N + 3) / 4;
        hipLaunchKernelGGL(split_resid_max_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream,
                           gallery, (const bf16_t *)hi, N, E, (unsigned int *)resid_bound_out);
        MMR_CHECK_LAUNCH();
    }
    return MMR_OK;
}

extern "C" int mmr_cosine_topk_split(const void *q, const void *gallery, const void *gallery_hi, const void *gallery_lo,
                                     const float *split_resid_bound_dev, int Q, int64_t N, int E, int k, float scale,
                                     float gallery_norm_bound, const float *gallery_norm_bound_dev, int32_t *idx, float *score,
                                     double *dot64, int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    MMR_CHECK_ARG((gallery_hi && gallery_lo) || N == 0, "mmr_cosine_topk_split: null split arrays");
    MMR_CHECK_ARG((((uintptr_t)gallery_hi | (uintptr_t)gallery_lo) & 15) == 0, "mmr_cosine_topk_split: split arrays must be 16-byte aligned");
    return cosine_topk_impl(q, gallery, MMR_F32, Q, N, E, k, scale, gallery_norm_bound, gallery_norm_bound_dev, idx, score, dot64,
                            status, workspace, workspace_bytes, stream, (const bf16_t *)gallery_hi, (const bf16_t *)gallery_lo,
                            split_resid_bound_dev);
}

extern "C" int mmr_similarity(const void *q, const void *gallery, mmr_dtype dtype, int Q, int64_t N, int E, float scale,
                              float *out, void *stream)
{
    MMR_CHECK_ARG(dtype == MMR_F32 || dtype == MMR_BF16, "mmr_similarity: dtype %d", (int)dtype);
    MMR_CHECK_ARG(Q >= 0 && N >= 0, "mmr_similarity: negative size");
    if (Q == 0 || N == 0) return MMR_OK;
    MMR_CHECK_ARG(q && gallery && out, "mmr_similarity: null pointer");
    MMR_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)gallery & 15) == 0, "mmr_similarity: q/gallery must be 16-byte aligned");
    MMR_CHECK_ARG((N + 3) / 4 < 0x7fffffff, "mmr_similarity: N too large");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((N + 3) / 4));
    if (dtype == MMR_BF16) {
        MMR_DISPATCH_PER(E, bf16_t, {
            hipLaunchKernelGGL((similarity_kernel<bf16_t, PER>), grid, dim3(256), 0, st, (const bf16_t *)q,
                               (const bf16_t *)gallery, Q, N, scale, out);
        });
    } else {
        MMR_DISPATCH_PER(E, float, {
            hipLaunchKernelGGL((similarity_kernel<float, PER>), grid, dim3(256), 0, st, (const float *)q,
                               (const float *)gallery, Q, N, scale, out);
        });
    }
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

extern "C" int mmr_l2norm_rows(void *x, mmr_dtype dtype, int64_t rows, int E, void *stream)
{
    MMR_CHECK_ARG(dtype == MMR_F32 || dtype == MMR_BF16, "mmr_l2norm_rows: dtype %d", (int)dtype);
    MMR_CHECK_ARG(rows >= 0 && E >= 1, "mmr_l2norm_rows: bad shape rows=%lld E=%d", (long long)rows, E);
    if (rows == 0) return MMR_OK;
    MMR_CHECK_ARG(x != nullptr, "mmr_l2norm_rows: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == MMR_BF16) hipLaunchKernelGGL(l2norm_kernel<bf16_t>, grid, dim3(256), 0, st, (bf16_t *)x, rows, E);
    else hipLaunchKernelGGL(l2norm_kernel<float>, grid, dim3(256), 0, st, (float *)x, rows, E);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

extern "C" int mmr_topk_merge(const int64_t *idx_parts, const double *dot_parts, int parts, int Q, int k, float scale,
                              int64_t *idx, float *score, double *dot64, void *stream)
{
    MMR_CHECK_ARG(parts >= 1 && Q >= 0 && k >= 1 && k <= K_MAX, "mmr_topk_merge: bad shape parts=%d Q=%d k=%d", parts, Q, k);
    MMR_CHECK_ARG(scale > 0.f, "mmr_topk_merge: scale must be > 0");
    if (Q == 0) return MMR_OK;
    MMR_CHECK_ARG(idx_parts && dot_parts && idx && score, "mmr_topk_merge: null pointer");
    MMR_CHECK_ARG(parts * k <= 1024, "mmr_topk_merge: parts*k=%d exceeds 1024", parts * k);
    hipLaunchKernelGGL(merge_kernel<false>, dim3(Q), dim3(64), 0, (hipStream_t)stream, idx_parts, dot_parts, parts, Q, k,
                       scale, idx, score, dot64);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

extern "C" int mmr_topk_pack(const int32_t *idx, const double *dot64, int Q, int k, int64_t row_offset, int64_t *packed,
                             void *stream)
{
    MMR_CHECK_ARG(Q >= 0 && k >= 1 && k <= K_MAX, "mmr_topk_pack: bad shape Q=%d k=%d", Q, k);
    if (Q == 0) return MMR_OK;
    MMR_CHECK_ARG(idx && dot64 && packed, "mmr_topk_pack: null pointer");
    const int n = Q * k;
    hipLaunchKernelGGL(pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, dot64, row_offset, n, packed);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

extern "C" int mmr_topk_merge_packed(const int64_t *packed_parts, int parts, int Q, int k, float scale, int64_t *idx,
                                     float *score, double *dot64, void *stream)
{
    MMR_CHECK_ARG(parts >= 1 && Q >= 0 && k >= 1 && k <= K_MAX, "mmr_topk_merge_packed: bad shape parts=%d Q=%d k=%d", parts, Q, k);
    MMR_CHECK_ARG(scale > 0.f, "mmr_topk_merge_packed: scale must be > 0");
    if (Q == 0) return MMR_OK;
    MMR_CHECK_ARG(packed_parts && idx && score, "mmr_topk_merge_packed: null pointer");
    MMR_CHECK_ARG(parts * k <= 1024, "mmr_topk_merge_packed: parts*k=%d exceeds 1024", parts * k);
    hipLaunchKernelGGL(merge_kernel<true>, dim3(Q), dim3(64), 0, (hipStream_t)stream, packed_parts, (const double *)nullptr,
                       parts, Q, k, scale, idx, score, dot64);
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}

extern "C" int mmr_tip_adapter_logits(const void *features, const void *clip_weights_t, const void *cache_keys_t,
                                      const float *cache_values, mmr_dtype dtype, int64_t N, int E, int C, int S,
                                      float alpha, float beta, float *tip_logits, float *clip_logits, void *stream)
{
    MMR_CHECK_ARG(dtype == MMR_F32 || dtype == MMR_BF16, "mmr_tip_adapter_logits: dtype %d", (int)dtype);
    MMR_CHECK_ARG(N >= 0 && C >= 1 && C <= 64 && S >= 0, "mmr_tip_adapter_logits: bad shape N=%lld C=%d S=%d (C <= 64)", (long long)N, C, S);
    if (N == 0) return MMR_OK;
    MMR_CHECK_ARG(features && clip_weights_t && tip_logits && (S == 0 || (cache_keys_t && cache_values)), "mmr_tip_adapter_logits: null pointer");
    MMR_CHECK_ARG((((uintptr_t)features | (uintptr_t)clip_weights_t | (uintptr_t)cache_keys_t) & 15) == 0, "mmr_tip_adapter_logits: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((N + 3) / 4));
    if (dtype == MMR_BF16) {
        MMR_DISPATCH_PER(E, bf16_t, {
            hipLaunchKernelGGL((tip_logits_kernel<bf16_t, PER>), grid, dim3(256), 0, st, (const bf16_t *)features,
                               (const bf16_t *)clip_weights_t, (const bf16_t *)cache_keys_t, cache_values, N, C, S, alpha,
                               beta, tip_logits, clip_logits);
        });
    } else {
        MMR_DISPATCH_PER(E, float, {
            hipLaunchKernelGGL((tip_logits_kernel<float, PER>), grid, dim3(256), 0, st, (const float *)features,
                               (const float *)clip_weights_t, (const float *)cache_keys_t, cache_values, N, C, S, alpha,
                               beta, tip_logits, clip_logits);
        });
    }
    MMR_CHECK_LAUNCH();
    return MMR_OK;
}
