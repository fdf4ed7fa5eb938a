/* The C ABI without Python or torch: a plain C program that owns its device buffers (HIP runtime), fills a gallery,
 * and asks libmmr_hip.so for the top-k -- the call a non-Python host of the reference's search step
 * (`100. * features @ ref.t()` + `topk`, reference code/search_image.py:107, code/utils.py:17) would make.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_abi_search.c -o c_abi_search \
 *       -L<pkg>/csrc -l:libmmr_hip.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,<pkg>/csrc -Wl,-rpath,/opt/rocm/lib -lm
 *
 * It checks the result against a double-precision loop on the host and exits non-zero on any mismatch.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mmr.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_MMR(x) do { int r_ = (x); if (r_ != MMR_OK) { fprintf(stderr, "%s: %s\n", #x, mmr_last_error()); return 3; } } while (0)

int main(void)
{
    const int64_t N = 20000;
    const int E = 512, Q = 5, K = 10;
    float *gal = (float *)malloc((size_t)N * E * sizeof(float)), *qry = (float *)malloc((size_t)Q * E * sizeof(float));
    uint32_t s = 12345u;
    for (int64_t i = 0; i < N * E; ++i) { s = s * 1664525u + 1013904223u; gal[i] = (float)((int)(s >> 9) % 2001 - 1000) / 1000.0f; }
    for (int i = 0; i < Q * E; ++i) { s = s * 1664525u + 1013904223u; qry[i] = (float)((int)(s >> 9) % 2001 - 1000) / 1000.0f; }

    float *d_gal, *d_q, *d_score;
    int32_t *d_idx;
    void *d_ws;
    const size_t ws_bytes = mmr_search_workspace_bytes(N, E, Q, K);
    CHECK_HIP(hipMalloc((void **)&d_gal, (size_t)N * E * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_q, (size_t)Q * E * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_score, (size_t)Q * K * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_idx, (size_t)Q * K * sizeof(int32_t)));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpy(d_gal, gal, (size_t)N * E * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_q, qry, (size_t)Q * E * sizeof(float), hipMemcpyHostToDevice));

    /* rows are not normalised here: pass an upper bound on their norm (sqrt(E) for entries in [-1,1]) */
    CHECK_MMR(mmr_cosine_topk(d_q, d_gal, MMR_F32, Q, N, E, K, 100.0f, sqrtf((float)E), d_idx, d_score, NULL, NULL, d_ws,
                              ws_bytes, NULL /* default stream */));
    CHECK_HIP(hipDeviceSynchronize());
    int32_t idx[5 * 10];
    float score[5 * 10];
    CHECK_HIP(hipMemcpy(idx, d_idx, sizeof(idx), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(score, d_score, sizeof(score), hipMemcpyDeviceToHost));

    int bad = 0;
    double *dots = (double *)malloc((size_t)N * sizeof(double));
    for (int qi = 0; qi < Q; ++qi) {
        for (int64_t r = 0; r < N; ++r) {
            double acc = 0.0;
            for (int e = 0; e < E; ++e) acc += (double)qry[qi * E + e] * (double)gal[r * E + e];
            dots[r] = acc;
        }
        for (int j = 0; j < K; ++j) {          /* selection sort of the top K by (-dot, +row) */
            int64_t best = -1;
            for (int64_t r = 0; r < N; ++r)
                if (!isnan(dots[r]) && (best < 0 || dots[r] > dots[best])) best = r;
            if (idx[qi * K + j] != (int32_t)best || fabs((double)score[qi * K + j] - 100.0 * dots[best]) > 1e-3 * fabs(100.0 * dots[best]) + 1e-3) {
                fprintf(stderr, "query %d rank %d: got row %d score %f, expected row %lld score %f\n", qi, j, idx[qi * K + j],
                        score[qi * K + j], (long long)best, 100.0 * dots[best]);
                ++bad;
            }
            dots[best] = NAN;
        }
    }
    printf("mmr v%d: top-%d of %d queries over %lld x %d rows through the C ABI: %s\n", mmr_version(), K, Q, (long long)N, E,
           bad ? "MISMATCH" : "matches the host loop");
    return bad ? 1 : 0;
}
